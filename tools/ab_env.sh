#!/bin/bash
# Same-lease A/B of an environment switch on the headline bench: gpurun -- bash tools/ab_env.sh VAR A B [rounds] [outfile]
# alternates  VAR=A / VAR=B  bench.py runs (short: no CPU baseline, no strong-scaling leg) and prints episodes/s + roofline fractions.
VAR=$1; A=$2; B=$3; N=${4:-2}; OUT=${5:-gpurun_out/ab_env.txt}
mkdir -p $(dirname $OUT)
: > $OUT
for i in $(seq 1 $N); do
  for v in $A $B; do
    env $VAR=$v python3 bench.py --steps 8 --warmup 2 --no-cpu-baseline --strong-episodes 0 2>/dev/null | tail -1 > /tmp/ab_line.json
    python3 - "$VAR=$v" >> $OUT <<'PY'
import json, sys
d = json.load(open('/tmp/ab_line.json'))
r, x = d["roofline"], d.get("roofline_mfma_x3", {})
print("%-22s %7.2f episodes/s  ms/step %8.2f  wgrad+adam in situ %6.1f GB/s (%.3f)  trunk conv %6.1f us (frac %.3f; alone %s us)  acc %.2f  val %s" % (
    sys.argv[1], d["value"], d["ms_per_step"], r["achieved"], r["frac"], x.get("avg_launch_us", 0), x.get("frac", 0),
    x.get("standalone", {}).get("avg_launch_us"), d.get("mean_acc", 0), d.get("validation", {}).get("mean_acc")))
PY
  done
done
cat $OUT
