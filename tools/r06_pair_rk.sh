#!/bin/bash
# Register-K pair-MLP layer (mft_pair_mlp_layer_rk) A/B on the meta-training step: parity tests first, then k = 1, 2, 4, 8 with the
# form on (default) and off (MFT_PAIR_RK_ROWS=0), then the timeline of the default step.
#   gpurun --timeout 1500 -- bash tools/r06_pair_rk.sh <tag>
TAG=${1:-r06_rk}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_engine_gpu.py tests/test_metatrain_gpu.py -m gpu -x -q -k "register_k or pair or set_forward_loss or lockstep or graph" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
for k in 1 2 4; do for rows in 65536 16384 0; do
  echo -n "k=$k pair_rk_rows=$rows: " | tee -a $O/ab.txt
  MFT_PAIR_RK_ROWS=$rows python3 bench.py --workload metatrain --episodes-per-rank $k --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['last_loss'])" | tee -a $O/ab.txt
done; done
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/metatrain_graph_timeline.py "$f" > $O/metatrain_graph_timeline.txt
head -30 $O/metatrain_graph_timeline.txt
find $O -name "*.csv" -size +1M -delete
