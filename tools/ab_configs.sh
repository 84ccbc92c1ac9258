#!/bin/bash
# Other configurations with / without an environment switch, same lease: gpurun -- bash tools/ab_configs.sh OUT VAR A B
OUT=$1; VAR=$2; A=$3; B=$4
mkdir -p $(dirname $OUT); : > $OUT
COMMON="--no-cpu-baseline --strong-episodes 0 --no-standalone --validate-episodes 0"
for args in "--n-shot 20 --episodes-per-batch 96 --steps 2 --warmup 1" "--n-shot 50 --episodes-per-batch 128 --steps 1 --warmup 1" "--image-size 224 --episodes-per-batch 32 --steps 2 --warmup 1" "--episodes-per-batch 120 --steps 4 --warmup 1" "--episodes-per-batch 32 --steps 8 --warmup 2"; do
  for v in $A $B; do
    env $VAR=$v python3 bench.py $args $COMMON 2>/dev/null | tail -1 > /tmp/ab_line.json
    python3 - "$VAR=$v $args" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab_line.json'))
    print("%-90s -> %8.3f episodes/s %10.2f ms/step  acc %.2f" % (sys.argv[1], d["value"], d["ms_per_step"], d.get("mean_acc", 0)))
except Exception as e:
    print("%-90s FAILED %r" % (sys.argv[1], e))
PY
  done
done
cat $OUT
