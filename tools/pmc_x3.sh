cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pmcx
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d gpurun_out/pmcx/a --output-format csv -- python3 tools/x3_tune.py 128 trunk.6.C2 > gpurun_out/pmcx/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d gpurun_out/pmcx/b --output-format csv -- python3 tools/x3_tune.py 128 trunk.6.C2 > gpurun_out/pmcx/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_FLAT -d gpurun_out/pmcx/c --output-format csv -- python3 tools/x3_tune.py 128 trunk.6.C2 > gpurun_out/pmcx/c.log 2>&1
for d in a b c; do tail -2 gpurun_out/pmcx/$d.log | cut -c1-300; python tools/pmc_summary.py gpurun_out/pmcx/$d 60 | grep -E "conv_x3_kernel<128, 64" | cut -c1-60,100-170; done
find gpurun_out/pmcx -name "*.csv" -size +2M -delete
