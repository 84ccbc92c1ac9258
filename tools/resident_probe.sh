#!/bin/bash
# Round-5 go/no-go for the step-blocked last-block loop: the resident 3R+3W stream microbench
# and its FETCH_SIZE / WRITE_SIZE per pass (separate --pmc passes).   gpurun -- bash tools/resident_probe.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/resident; mkdir -p $O
hipcc --offload-arch=gfx950 -O3 -w -o /tmp/resident_stream.bin tools/microbench/resident_stream.hip || exit 1
/tmp/resident_stream.bin > $O/resident_stream.txt 2>&1
cat $O/resident_stream.txt
for MB in 96 132 176 512; do
  for C in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $C --kernel-trace --output-format csv -d $O/pmc_${MB}_$C -- /tmp/resident_stream.bin pmc $MB > $O/pmc_${MB}_$C.log 2>&1
    python3 tools/pmc_summary.py $O/pmc_${MB}_$C 4 2>&1 | sed "s/^/[$MB MB $C] /" >> $O/resident_pmc.txt
  done
done
cat $O/resident_pmc.txt
find $O -name "*.csv" -size +1M -delete
