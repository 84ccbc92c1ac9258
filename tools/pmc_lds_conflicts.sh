O=gpurun_out/r4z
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/b --output-format csv -- python3 tools/x3_accuracy.py > $O/b.log 2>&1
python3 tools/pmc_summary.py $O/b 80 | grep -E "kernel  |conv_x3_s1_kernel<128, 64, 2>|conv_x3_s1_kernel<128, 64, 3>|conv_x3_s1_kernel<128, 64>" | cut -c1-60,100-175
find $O -name "*.csv" -size +2M -delete
