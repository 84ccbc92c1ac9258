#!/bin/bash
# Round-6: the meta-training tests + the lockstep (k episodes per step) bench lines.  gpurun --timeout 1200 -- bash tools/r06_lockstep.sh <tag>
TAG=${1:-r06_b}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/$TAG; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_metatrain_gpu.py tests/test_modules_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -25 $O/pytest.log
for k in 1 2 4 8; do
  python3 bench.py --workload metatrain --episodes-per-rank $k --steps 200 --warmup 10 --no-cpu-baseline > $O/bench_metatrain_k$k.json 2> $O/bench_metatrain_k$k.err
  tail -1 $O/bench_metatrain_k$k.json | cut -c1-330
  tail -2 $O/bench_metatrain_k$k.err
done
