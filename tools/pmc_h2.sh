#!/bin/bash
# Stall / unit counters of the f16x2 trunk convolutions (tools/x3_accuracy.py as the workload): gpurun -- bash tools/pmc_h2.sh <outdir>
O=${1:-gpurun_out/pmch2}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
mkdir -p $O
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $O/a --output-format csv -- python3 tools/x3_accuracy.py > $O/a.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD --kernel-trace -d $O/b --output-format csv -- python3 tools/x3_accuracy.py > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_WAVES SQ_ACTIVE_INST_FLAT --kernel-trace -d $O/c --output-format csv -- python3 tools/x3_accuracy.py > $O/c.log 2>&1
for d in a b c; do python3 tools/pmc_summary.py $O/$d 80 | grep -E "kernel  |conv_x3_s1_kernel<128, 64, 2>|conv_x3_kernel<128, 64, false, false, 0, 0, false, 2>" | cut -c1-52,100-175; done > $O/summary.txt
find $O -name "*.csv" -size +2M -delete
cat $O/summary.txt
