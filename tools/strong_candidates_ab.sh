mkdir -p gpurun_out/r4j
for i in 1 2; do for k in 8 0 5; do
  MFT_SLAB_CANDIDATES=$k MFT_SLAB_HINTS=0 python3 bench.py --strong-only --gpus 1 --strong-episodes 600 --episodes-per-batch 128 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('K=$k wall %.3f s  %.2f episodes/s  engine ready %.3f s' % (d['wall_s'], d['episodes_per_s'], d['engine_ready_after_s']))"
done; done > gpurun_out/r4j/strong_candidates.txt 2>&1
cat gpurun_out/r4j/strong_candidates.txt
