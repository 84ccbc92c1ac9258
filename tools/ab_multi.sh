#!/bin/bash
# Same-lease comparison of several environment settings on the headline bench:
#   gpurun -- bash tools/ab_multi.sh OUT "A=1 B=2" "A=3" ...      (each argument = one setting, "-" = defaults; two rounds)
OUT=$1; shift
mkdir -p $(dirname $OUT); : > $OUT
for i in 1 2; do
  for cfg in "$@"; do
    if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
    env $envs python3 bench.py --steps 6 --warmup 2 --no-cpu-baseline --strong-episodes 0 --no-standalone 2>/dev/null | tail -1 > /tmp/ab_line.json
    python3 - "$cfg" >> $OUT <<'PY'
import json, sys
try:
    d = json.load(open('/tmp/ab_line.json'))
    r, x, pw = d["roofline"], d.get("roofline_mfma_x3", {}), d.get("power", {})
    print("%-44s %7.2f episodes/s  ms/step %8.2f  dominant in situ %6.1f GB/s (%.3f)  trunk conv %6.1f us  %s W %s MHz  acc %.2f" % (
        sys.argv[1], d["value"], d["ms_per_step"], r["achieved"], r["frac"], x.get("avg_launch_us", 0), pw.get("socket_w_median"),
        pw.get("shader_mhz_median"), d.get("mean_acc", 0)))
except Exception as e:
    print("%-44s FAILED %r" % (sys.argv[1], e))
PY
  done
done
cat $OUT
