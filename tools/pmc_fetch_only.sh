cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r4p; mkdir -p $O
PMCARGS="--steps 1 --warmup 0 --no-cpu-baseline --no-standalone --epochs 1 --gen-examples 2 --no-pipeline --no-defer-final --validate-episodes 0 --strong-episodes 0"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_f -- python3 bench.py $PMCARGS > $O/pmc_f.log 2>&1
python3 tools/pmc_summary.py $O/pmc_f 6 > $O/pmc_fetch_xcd1.txt
find $O -name "*.csv" -size +1M -delete
cat $O/pmc_fetch_xcd1.txt | cut -c1-170
