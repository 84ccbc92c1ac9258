python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -k "x3 or bf16x3 or stem_cache or two_stream or presplit or golden" 2>&1 | tail -3
python tools/x3_knob_ab.py 41 42 41 42 2>&1 | grep -v amdgpu | grep -E "knob|identical"
for i in 1 2 3; do for k in 41 42; do MFT_X3_KNOBS=$k python bench.py --no-cpu-baseline --validate-episodes 8 --no-standalone --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('x3 knob $k', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline_mfma_x3']['achieved'], d['validation']['mean_acc'])"; done; done
