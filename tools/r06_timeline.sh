#!/bin/bash
# kernel-by-kernel timeline of one replayed meta-training step (+ the metatrain tests first): gpurun -- bash tools/r06_timeline.sh <tag> [k]
TAG=${1:-r06_d}; K=${2:-1}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_metatrain_gpu.py tests/test_modules_gpu.py -m gpu -q -x > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -6 $O/pytest.log
python3 bench.py --workload metatrain --episodes-per-rank $K --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_metatrain.json 2> $O/bench_metatrain.err
tail -1 $O/bench_metatrain.json | cut -c1-330
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --episodes-per-rank $K --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/metatrain_graph_timeline.py "$f" > $O/metatrain_graph_timeline.txt
head -70 $O/metatrain_graph_timeline.txt
find $O -name "*.csv" -size +1M -delete
