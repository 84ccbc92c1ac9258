#!/bin/bash
# MFMA utilisation of the meta-training step from PMC counters (their own pass; no trace domains beside --pmc):
#   gpurun --timeout 900 -- bash tools/r06_metatrain_pmc.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_pmc; rm -rf $O; mkdir -p $O
for K in 1 4; do
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -d $O/pmc_k$K --output-format csv -- python3 bench.py --workload metatrain --episodes-per-rank $K --steps 40 --warmup 5 --no-cpu-baseline > $O/pmc_k$K.log 2>&1
  { echo "== bench.py --workload metatrain --episodes-per-rank $K --steps 40: rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE (own pass) =="; python3 tools/pmc_mfma_util.py $O/pmc_k$K 60; } > $O/r06_metatrain_mfma_util_k$K.txt
  head -30 $O/r06_metatrain_mfma_util_k$K.txt | cut -c1-200
done
find $O -name "*.csv" -size +1M -delete
