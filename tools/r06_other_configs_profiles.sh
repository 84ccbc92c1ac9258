#!/bin/bash
# rocprofv3 --kernel-trace --stats of the four `other_configs` child commands of bench.py (BASELINE configs[2], [3], [4] and the
# reference's 224x224), one untruncated per-kernel summary each:  gpurun --timeout 1500 -- bash tools/r06_other_configs_profiles.sh
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_oc; rm -rf $O; mkdir -p $O
COMMON="--no-cpu-baseline --no-standalone --strong-episodes 0 --validate-episodes 0 --no-other-configs"
prof() {   # key, args...
  key=$1; shift
  rocprofv3 --kernel-trace --stats -d $O/$key -o kt -- python3 bench.py "$@" $COMMON > $O/${key}_bench.json 2> $O/${key}.err
  db=$(find $O/$key -name "*.db" | head -1)
  { echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py $* $COMMON   (MI355X, round-6 tree; every kernel, nothing cut)";
    echo "# bench line: $(tail -1 $O/${key}_bench.json | cut -c1-260)";
    python3 tools/rocpd_stats.py "$db"; } > $O/r06_other_configs_${key}_kernel_stats.txt
  find $O/$key -name "*.db" -delete
  head -8 $O/r06_other_configs_${key}_kernel_stats.txt | cut -c1-200
}
prof configs2_20shot --n-shot 20 --steps 1 --warmup 1
prof configs4_50shot --n-shot 50 --steps 1 --warmup 1
prof configs3_metatrain --workload metatrain --steps 300 --warmup 10
prof configs3_metatrain_lockstep4 --workload metatrain --episodes-per-rank 4 --steps 200 --warmup 10
prof reference_224 --image-size 224 --episodes-per-batch 32 --steps 2 --warmup 1
