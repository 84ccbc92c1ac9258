#!/bin/bash
# full -m gpu suite, smoke, then the driver's default bench line.  gpurun --timeout 2400 -- bash tools/r06_full.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_f}; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -8 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
T0=$(date +%s)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "python bench.py wall: $(( $(date +%s) - T0 )) s"
tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'])
print('strong',json.dumps(d['strong_scaling'])[:700])
print('ranks',d.get('rccl_ranks'))
for k,v in (d.get('other_configs') or {}).items(): print(k, json.dumps(v)[:500])
"
