python tools/x3_knob_ab.py 0 3 0 3 2>&1 | grep -v amdgpu | grep -E "knob"
for i in 1 2; do for k in 0 3; do MFT_X3_KNOBS=$k python bench.py --no-cpu-baseline --validate-episodes 8 --no-standalone --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 knob $k', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma_x3']['achieved'], d['validation']['mean_acc'], d['validation']['golden_mean_acc'])"; done; done
