for i in 1 2; do for f in "" "--no-prefetch" "--no-defer-final" "--no-prefetch --no-defer-final"; do
python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --strong-episodes 0 --no-standalone --validate-episodes 0 $f 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-36s %.2f episodes/s  %.1f ms/step' % ('$f' or 'default', d['value'], d['ms_per_step']))"
done; done
