#!/bin/bash
# Per-dispatch timeline of ONE replayed meta-training step: rocprofv3 --kernel-trace of bench.py --workload metatrain, then
# tools/metatrain_graph_timeline.py on the kernel_trace csv (durations, gaps, by-kernel totals of the last replay).
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/mtl; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
tail -1 $O/run.log | cut -c1-200
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
python3 tools/metatrain_graph_timeline.py "$f" > $O/timeline.txt
head -70 $O/timeline.txt
find $O -name "*.csv" -size +1M -delete
