cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcx2
python -m pytest tests/test_kernels_gpu.py -x -q -k "x3" 2>&1 | tail -2
rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES -d gpurun_out/pmcx2/b --output-format csv -- python3 tools/x3_tune.py 128 trunk.6.C2 > gpurun_out/pmcx2/b.log 2>&1
python tools/pmc_summary.py gpurun_out/pmcx2/b 60 | grep -E "conv_x3_kernel<128, 64" | cut -c1-60,100-170
find gpurun_out/pmcx2 -name "*.csv" -size +2M -delete
python tools/step_breakdown.py 128 20 2>&1 | grep -E "x3_bnstats|inner step"
for k in 40 41 40 41; do MFT_X3_KNOBS=$k python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-120; done
