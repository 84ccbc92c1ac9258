#!/bin/bash
# the driver's command: python bench.py (default flags), timed end to end
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_e}; rm -rf $O; mkdir -p $O
T0=$(date +%s.%N)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "python bench.py wall: $(echo "$(date +%s.%N) - $T0" | bc) s"
tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'])
print('strong',json.dumps(d['strong_scaling'])[:600])
print('validation',d['validation'])
for k,v in (d.get('other_configs') or {}).items(): print(k, json.dumps(v)[:700])
"
