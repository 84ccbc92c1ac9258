cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_seq; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace -d $O/tr --output-format csv -- python3 bench.py --workload metatrain --steps 30 --warmup 5 --no-cpu-baseline > $O/run.log 2>&1
f=$(find $O/tr -name "*kernel_trace.csv" | head -1)
head -1 $f > $O/header.txt
python3 tools/metatrain_graph_timeline.py "$f" --sequence > $O/seq.txt
find $O -name "*.csv" -size +1M -delete
