#!/bin/bash
# round-6 closing run: full -m gpu suite, smoke, the driver's default bench line, kernel-by-kernel timelines of the replayed
# meta-training step at k = 1 and k = 4.   gpurun --timeout 3000 -- bash tools/r06_final.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_z}; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
T0=$(date +%s)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
echo "python bench.py wall: $(( $(date +%s) - T0 )) s"
tail -1 $O/bench_default.json | python3 -c "
import json,sys
d=json.loads(sys.stdin.read())
print('value',d['value'],'ms',d['ms_per_step'],'frac',d['roofline']['frac'],'traffic',d['roofline']['traffic'])
print('validation',d['validation']['episodes'],d['validation']['abs_diff'],d['validation']['bar'],d['validation']['ok'])
print('strong',d['strong_scaling']['episodes_per_s'], d['strong_scaling'].get('emulated_world_8',{}).get('projected_speedup_vs_this_1gpu_leg'))
for k,v in (d.get('other_configs') or {}).items(): print(k, v.get('value'), v.get('ms_per_step'), (v.get('dominant') or {}).get('frac'), v.get('error'))
"
for K in 1 4; do
  rocprofv3 --kernel-trace -d $O/tr$K --output-format csv -- python3 bench.py --workload metatrain --episodes-per-rank $K --steps 30 --warmup 5 --no-cpu-baseline > $O/run$K.log 2>&1
  f=$(find $O/tr$K -name "*kernel_trace.csv" | head -1)
  python3 tools/metatrain_graph_timeline.py "$f" > $O/metatrain_graph_timeline_k$K.txt
  head -3 $O/metatrain_graph_timeline_k$K.txt | cut -c1-200
done
find $O -name "*.csv" -size +1M -delete
python3 tools/accuracy_g19.py > $O/accuracy_g19.txt 2>&1; cat $O/accuracy_g19.txt | tail -12
