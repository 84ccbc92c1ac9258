#!/bin/bash
# Round-6 GPU check: the whole -m gpu suite, then the replayed meta-training step's kernel-by-kernel timeline and bench line.
#   gpurun --timeout 1500 -- bash tools/r06_check.sh <tag> [pytest args]
TAG=${1:-r06_a}; shift
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 -m pytest tests -m gpu -x -q "$@" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -5 $O/pytest.log
python3 bench.py --workload metatrain --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_metatrain.json 2> $O/bench_metatrain.err
tail -1 $O/bench_metatrain.json | cut -c1-400
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/metatrain_graph_timeline.py "$f" > $O/metatrain_graph_timeline.txt
head -75 $O/metatrain_graph_timeline.txt
find $O -name "*.csv" -size +1M -delete
