#!/bin/bash
# Regenerates the committed round profiles on the GPU box: gpurun -- bash tools/final_profiles.sh <tag> <round> <head>
# (kernel-trace stats, PMC traffic / MFMA passes in their own runs, single-stream step breakdown, one default bench line,
#  GNN-phase traces at the 20-/50-shot graph sizes)
TAG=${1:-r02_a}
RND=${2:-2}
HEAD=${3:-unknown}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --validate-episodes 0 --strong-episodes 0 > $O/kt_bench.log 2>&1
tail -1 $O/kt_bench.log | cut -c1-300
python3 tools/rocpd_stats.py $(find $O/kt -name "*.db" | head -1) > $O/${TAG}_kernel_stats_E128_pipelined.txt
head -8 $O/${TAG}_kernel_stats_E128_pipelined.txt | cut -c1-180
find $O/kt -name "*.db" -delete
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-standalone --epochs 1 --gen-examples 2 --no-pipeline --no-defer-final --validate-episodes 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py $PMCARGS > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py $PMCARGS > $O/pmc_w.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_m -- python3 bench.py $PMCARGS > $O/pmc_m.log 2>&1
( python3 tools/pmc_summary.py $O/pmc_f 14; python3 tools/pmc_summary.py $O/pmc_w 14 ) > $O/${TAG}_pmc_bench_E128.txt
python3 tools/pmc_mfma_util.py $O/pmc_m 16 > $O/${TAG}_pmc_mfma_util.txt
python3 tools/pmc_traffic_json.py $O/pmc_f $O/pmc_w 128 $RND $HEAD > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
head -4 $O/${TAG}_pmc_bench_E128.txt | cut -c1-60,100-170
find $O -name "*.csv" -size +1M -delete
python3 tools/step_breakdown.py 128 20 > $O/${TAG}_step_breakdown_E128.txt 2>&1
grep "inner step" $O/${TAG}_step_breakdown_E128.txt
# GNN phase (fused pair-MLP kernels) at the 20-/50-shot graph sizes: kernel trace + MFMA-busy counters
for NS in 20 50; do
  rocprofv3 --kernel-trace --stats -d $O/kt_gnn$NS -o kt -- python3 tools/gnn_time.py $NS 128 fused 3 > $O/gnn$NS.log 2>&1
  ( grep fused $O/gnn$NS.log; python3 tools/rocpd_stats.py $(find $O/kt_gnn$NS -name "*.db" | head -1) 12 ) > $O/${TAG}_gnn_kernel_stats_${NS}shot_E128.txt
  find $O/kt_gnn$NS -name "*.db" -delete
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_gnn$NS -- python3 tools/gnn_time.py $NS 128 fused 1 > $O/pmc_gnn$NS.log 2>&1
  python3 tools/pmc_mfma_util.py $O/pmc_gnn$NS 8 > $O/${TAG}_gnn_pmc_mfma_util_${NS}shot.txt
done
find $O -name "*.csv" -size +1M -delete
# the default bench line LAST: profiles/pmc_traffic.json now belongs to this kernel source, so roofline.traffic is filled in
python3 bench.py > $O/${TAG}_bench_E128.json 2> $O/bench.err
tail -1 $O/${TAG}_bench_E128.json | cut -c1-400
# power / clock: the step runs at the package power limit (hwmon sampling; no privileges needed)
python3 tools/power_probe.py 128 5 2>&1 | grep -v amdgpu.ids > $O/${TAG}_power_probe.txt
python3 tools/power_breakdown.py 128 3 2>&1 | grep -v amdgpu.ids > $O/${TAG}_power_breakdown.txt
tail -7 $O/${TAG}_power_probe.txt
ls $O
