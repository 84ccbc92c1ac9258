#!/bin/bash
# north_star "MFMA utilisation on ResNet10 forward, rocprof-reported" (round-4 verdict row ns1): the whole ResNet10 forward (train-mode
# BatchNorm, groups of 100 images) over 12,800 images of 84x84, (a) timed, (b) rocprofv3 kernel trace, (c) rocprofv3 --pmc
# SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (its own pass).   gpurun -- bash tools/forward_mfma_profile.sh [fp32|x3]
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/fwd; mkdir -p $O
for V in fp32 x3; do
  python3 tools/forward_tflops.py 12800 84 $V > $O/tflops_$V.txt 2>&1
  rocprofv3 --kernel-trace --stats -d $O/trace_$V --output-format csv -- python3 tools/forward_tflops.py 12800 84 $V > $O/trace_$V.log 2>&1
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_$V --output-format csv -- python3 tools/forward_tflops.py 12800 84 $V > $O/pmc_$V.log 2>&1
  { echo "== ResNet10 forward, trunk.4-6 on $V kernels =="; grep "ResNet10 forward" $O/tflops_$V.txt
    echo "-- rocprofv3 --kernel-trace --stats (top kernels)"; f=$(find $O/trace_$V -name "*kernel_stats.csv" | head -1); head -14 "$f" | cut -c1-200
    echo "-- rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; python3 tools/pmc_mfma_util.py $O/pmc_$V 14; } > $O/forward_mfma_$V.txt
  cat $O/forward_mfma_$V.txt | cut -c1-220
done
find $O -name "*.csv" -size +1M -delete
