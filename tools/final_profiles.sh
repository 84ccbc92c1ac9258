#!/bin/bash
# Regenerates the committed round profiles on the GPU box: gpurun -- bash tools/final_profiles.sh <tag>
# (kernel-trace stats, PMC traffic / MFMA passes in their own runs, single-stream step breakdown, one default bench line)
TAG=${1:-r01_f}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
python3 bench.py > $O/bench.json 2> $O/bench.err
tail -1 $O/bench.json | cut -c1-400
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-standalone > $O/kt_bench.log 2>&1
tail -1 $O/kt_bench.log | cut -c1-300
python3 tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/${TAG}_kernel_stats.txt
head -8 $O/${TAG}_kernel_stats.txt | cut -c1-180
find $O/kt -name "*.db" -delete
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-standalone --epochs 1 --gen-examples 2 --no-pipeline --no-defer-final"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py $PMCARGS > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py $PMCARGS > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -- python3 bench.py $PMCARGS > $O/pmc_m.log 2>&1
python3 tools/pmc_summary.py $O/pmc_f 14 > $O/${TAG}_pmc_fetch.txt
python3 tools/pmc_summary.py $O/pmc_w 14 > $O/${TAG}_pmc_write.txt
python3 tools/pmc_mfma_util.py $O/pmc_m 16 > $O/${TAG}_pmc_mfma_util.txt
head -4 $O/${TAG}_pmc_fetch.txt | cut -c1-60,100-170; head -4 $O/${TAG}_pmc_write.txt | cut -c1-60,100-170
find $O -name "*.csv" -size +1M -delete
python3 tools/step_breakdown.py 128 20 > $O/${TAG}_step_breakdown.txt 2>&1
grep "inner step" $O/${TAG}_step_breakdown.txt
