# PMC passes (own runs, counters only) over the dominant launch alone: rows kernel vs the walking fused kernel (tools/wgrad_fwd_time.py)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/pmcwf
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/a --output-format csv -- python3 tools/wgrad_fwd_time.py 128 > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD -d $O/b --output-format csv -- python3 tools/wgrad_fwd_time.py 128 > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_INSTS_VMEM_WR -d $O/c --output-format csv -- python3 tools/wgrad_fwd_time.py 128 > $O/c.log 2>&1
for d in a b c; do python tools/pmc_summary.py $O/$d 200 | grep -E "wgrad_adam" | sed -E "s/void \(anonymous namespace\):://" | cut -c1-64,100-180; done > $O/summary.txt
cat $O/summary.txt
find $O -name "*.csv" -size +1M -delete
