#!/bin/bash
# The non-headline configurations on the current tree (one line each): gpurun -- bash tools/other_configs.sh OUT
OUT=${1:-gpurun_out/other_configs.txt}
mkdir -p $(dirname $OUT); : > $OUT
COMMON="--no-cpu-baseline --strong-episodes 0 --no-standalone --validate-episodes 0"
run() {
  python3 bench.py $1 $COMMON 2>/dev/null | tail -1 > /tmp/oc_line.json
  python3 - "$1" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/oc_line.json'))
    print("%-70s -> %8.3f episodes/s %10.2f ms/step" % (sys.argv[1], d["value"], d["ms_per_step"]))
except Exception as e:
    print("%-70s FAILED %r" % (sys.argv[1], e))
PY
}
run "--n-shot 20 --episodes-per-batch 96 --steps 2 --warmup 1"
run "--n-shot 20 --episodes-per-batch 128 --steps 2 --warmup 1"
run "--n-shot 50 --episodes-per-batch 64 --steps 1 --warmup 1"
run "--image-size 224 --episodes-per-batch 32 --steps 2 --warmup 1"
run "--episodes-per-batch 120 --steps 4 --warmup 1"
run "--episodes-per-batch 64 --steps 6 --warmup 2"
run "--episodes-per-batch 32 --steps 8 --warmup 2"
run "--episodes-per-batch 16 --steps 8 --warmup 2"
run "--workload metatrain --steps 100 --warmup 5"
run "--workload metafinetune --steps 20 --warmup 5"
cat $OUT
