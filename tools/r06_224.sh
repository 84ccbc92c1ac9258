cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for E in 32 64 128; do
  echo -n "224x224 E=$E: "; python3 bench.py --image-size 224 --episodes-per-batch $E --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --strong-episodes 0 --validate-episodes 0 --no-other-configs 2>/tmp/err_$E.txt | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['launch_time_share'], d['whole_path_hbm']['fused_next_forward'])" || tail -3 /tmp/err_$E.txt
done
