# kernel-trace summaries of one bench step with and without the fused next-step forward (same box, back to back)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/trace_ab
mkdir -p $O
ARGS="--steps 1 --warmup 1 --no-cpu-baseline --no-standalone --validate-episodes 0 --strong-episodes 0"
rocprofv3 --kernel-trace --stats -d $O/fused -o kt -- python3 bench.py $ARGS > $O/fused.log 2>&1
python3 tools/rocpd_stats.py $(find $O/fused -name "*.db" | head -1) 40 > $O/fused_stats.txt
find $O/fused -name "*.db" -delete
MFT_FUSE_NEXT=0 rocprofv3 --kernel-trace --stats -d $O/unfused -o kt -- python3 bench.py $ARGS > $O/unfused.log 2>&1
python3 tools/rocpd_stats.py $(find $O/unfused -name "*.db" | head -1) 40 > $O/unfused_stats.txt
find $O/unfused -name "*.db" -delete
tail -1 $O/fused.log | cut -c1-200; tail -1 $O/unfused.log | cut -c1-200
