#!/bin/bash
# kernel timeline of two inner steps of the default bench: gpurun -- bash tools/timeline.sh [out]
OUT=${1:-gpurun_out/timeline.txt}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/tl; mkdir -p $O
rocprofv3 --kernel-trace -d $O/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --validate-episodes 0 --strong-episodes 0 > $O/kt_bench.log 2>&1
python3 tools/rocpd_timeline.py $(find $O/kt -name "*.db" | head -1) 2 > $OUT
find $O/kt -name "*.db" -delete
cat $OUT
