#!/bin/bash
# Only the kernel-trace part of tools/final_profiles.sh (+ a default bench line right after it): gpurun -- bash tools/kernel_trace_only.sh <tag>
TAG=${1:-r04_f}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --validate-episodes 0 --strong-episodes 0 > /dev/null 2>&1     # (warm the box: clocks, page cache)
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --validate-episodes 0 --strong-episodes 0 > $O/kt_bench.log 2>&1
python3 tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/${TAG}_kernel_stats_E128_pipelined.txt
python3 tools/rocpd_timeline.py $(find $O/kt -name "*.db" | head -1) 2 > $O/${TAG}_timeline.txt
find $O/kt -name "*.db" -delete
head -6 $O/${TAG}_kernel_stats_E128_pipelined.txt | cut -c1-180
python3 bench.py --no-other-configs > $O/${TAG}_bench_E128.json 2> $O/bench.err
tail -1 $O/${TAG}_bench_E128.json | cut -c1-200
