#!/bin/bash
# Smaller lockstep batches with each round-5 kernel change switched off in turn, and the fused next-step forward on / off: gpurun -- bash tools/small_batches_ab.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/e64; mkdir -p $O; : > $O/out.txt
COMMON="--no-cpu-baseline --strong-episodes 0 --no-standalone --validate-episodes 0"
run() {
  env $1 python3 bench.py $2 $COMMON 2>/dev/null | tail -1 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('%-44s %-34s -> %8.3f episodes/s %9.2f ms/batch fused=%s' % ('$1', '$2', d['value'], d['ms_per_step'], d['whole_path_hbm']['fused_next_forward']))" >> $O/out.txt
}
for r in 1 2; do
run "A=1" "--episodes-per-batch 64 --steps 6 --warmup 2"
run "MFT_STEM_FUSED_FILL=0" "--episodes-per-batch 64 --steps 6 --warmup 2"
run "MFT_X3_KNOBS=30" "--episodes-per-batch 64 --steps 6 --warmup 2"
run "MFT_FUSE_NEXT=0" "--episodes-per-batch 64 --steps 6 --warmup 2"
run "A=1" "--episodes-per-batch 32 --steps 8 --warmup 2"
run "MFT_FUSE_NEXT=0" "--episodes-per-batch 32 --steps 8 --warmup 2"
done
cat $O/out.txt
