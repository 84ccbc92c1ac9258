#!/bin/bash
# Episodes per lockstep batch, default path (fused next-step forward where E % 32 == 0): gpurun -- bash tools/e_sweep.sh "128 160 192 256"
cd $GRAFT_REPO_ROOT
O=gpurun_out/e_sweep; mkdir -p $O
for E in ${1:-128 160 192 256}; do
  python bench.py --episodes-per-batch $E --steps ${2:-4} --warmup 1 --no-cpu-baseline --no-standalone --strong-episodes 0 --validate-episodes 0 > $O/E$E.json 2> $O/E$E.err
  python - $O/E$E.json $E <<'PY'
import json, sys
l = [x for x in open(sys.argv[1]) if x.startswith("{")]
if not l:
    print("E=%s: no line" % sys.argv[2]); sys.exit(0)
d = json.loads(l[-1])
print("E=%4s  %8.2f episodes/s  %9.2f ms/batch  %7.3f ms/step  dominant frac %.3f  fused %s" % (
    sys.argv[2], d["value"], d["ms_per_step"], d["ms_per_step"] / 500.0, d["roofline"]["frac"], d["whole_path_hbm"].get("fused_next_forward")))
PY
done
