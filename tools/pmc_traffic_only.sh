#!/bin/bash
# Only the PMC traffic record of tools/final_profiles.sh (two separate --pmc passes + the provenance JSON): gpurun -- bash tools/pmc_traffic_only.sh <round> <head> [tag]
RND=${1:-4}; HEAD=${2:-unknown}; TAG=${3:-r04_c}
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/final
mkdir -p $O
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-standalone --epochs 1 --gen-examples 2 --no-pipeline --no-defer-final --validate-episodes 0 --strong-episodes 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py $PMCARGS > $O/pmc_f.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_w -- python3 bench.py $PMCARGS > $O/pmc_w.log 2>&1
( python3 tools/pmc_summary.py $O/pmc_f 14; python3 tools/pmc_summary.py $O/pmc_w 14 ) > $O/${TAG}_pmc_bench_E128.txt
python3 tools/pmc_traffic_json.py $O/pmc_f $O/pmc_w 128 $RND $HEAD > $O/pmc_traffic.log 2>&1
cp profiles/pmc_traffic.json $O/pmc_traffic.json
cat $O/pmc_traffic.log | tail -2 | cut -c1-600
find $O -name "*.csv" -size +1M -delete
