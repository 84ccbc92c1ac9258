#!/bin/bash
# quick loop of round 6: head / meta-training tests, the k = 1 and k = 4 bench lines, the replayed step in launch order.
#   gpurun --timeout 1200 -- bash tools/r06_quick.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r06_q}; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_metatrain_gpu.py tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -x -q -k "not trajectory and not accuracy" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -3 $O/pytest.log
for k in 1 2 4; do
  echo -n "k=$k: " | tee -a $O/ab.txt
  python3 bench.py --workload metatrain --episodes-per-rank $k --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['last_loss'])" | tee -a $O/ab.txt
done
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/metatrain_graph_timeline.py "$f" --sequence > $O/seq.txt
head -1 $O/seq.txt | cut -c1-120
find $O -name "*.csv" -size +1M -delete
