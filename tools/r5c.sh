#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r5c; mkdir -p $O
bash tools/forward_mfma_profile.sh > $O/forward_mfma.log 2>&1
COMMON="--no-cpu-baseline --strong-episodes 0 --no-standalone --validate-episodes 0"
for A in "--n-shot 20 --episodes-per-batch 96 --steps 2 --warmup 1" "--n-shot 20 --episodes-per-batch 128 --steps 2 --warmup 1" "--n-shot 50 --episodes-per-batch 64 --steps 1 --warmup 1" "--n-shot 50 --episodes-per-batch 128 --steps 1 --warmup 1"; do
  python3 bench.py $A $COMMON 2>$O/last.err | tail -1 > $O/line.json
  python3 - "$A" $O/line.json >> $O/other_configs.txt <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[2]))
    print("%-62s -> %8.3f episodes/s %10.2f ms/batch  fused=%s" % (sys.argv[1], d["value"], d["ms_per_step"], d["whole_path_hbm"]["fused_next_forward"]))
except Exception as e:
    print("%-62s FAILED %r" % (sys.argv[1], e))
PY
done
cat $O/other_configs.txt
rocprofv3 --kernel-trace --stats -d $O/mt_trace --output-format csv -- python3 bench.py --workload metatrain --steps 50 --warmup 5 --no-cpu-baseline > $O/mt_trace.log 2>&1
f=$(find $O/mt_trace -name "*kernel_stats.csv" | head -1)
{ echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --workload metatrain --steps 50 --warmup 5 --no-cpu-baseline (graphed step + 3 eager timing steps)"; tail -1 $O/mt_trace.log | cut -c1-400; head -40 "$f" | cut -c1-220; } > $O/metatrain_kernel_trace.txt
head -25 $O/metatrain_kernel_trace.txt | cut -c1-200
find $O -name "*.csv" -size +1M -delete
