#!/bin/bash
# Round 5: f16x2 shared-tap trunk kernels with TWO weight-tile buffers (one barrier per tap; knob 31, default) vs one (knob 30, round 4's form).
#   gpurun -- bash tools/ab_x3_bdb.sh
cd $GRAFT_REPO_ROOT
O=gpurun_out/x3bdb; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py tests/test_engine_gpu.py -m gpu -q -x -k "x3 or f16x2 or bf16x3 or fold or trunk or golden" > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python tools/x3_knob_ab.py 30 31 30 31 2>&1 | grep -v amdgpu | tail -12 | tee $O/knob_ab.txt
for i in 1 2; do for k in 31 30; do MFT_X3_KNOBS=$k python bench.py --no-cpu-baseline --validate-episodes 8 --strong-episodes 0 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); x=d['roofline_mfma_x3']; print('x3 knob $k  %.2f episodes/s  %.2f ms/batch  dominant %.3f  trunk conv in situ %.1f us, alone %.1f us  val %s' % (d['value'], d['ms_per_step'], d['roofline']['frac'], x['avg_launch_us'], x['standalone']['avg_launch_us'], d['validation']['mean_acc']))"; done; done | tee $O/bench_ab.txt
