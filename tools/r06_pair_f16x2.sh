cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_engine_gpu.py -m gpu -q -x -k "fused_pair_mlp" 2>&1 | tail -4
for k in 1 4; do for h in 0 1 0 1; do
  echo -n "k=$k pair_f16x2=$h: "; MFT_PAIR_F16X2=$h python3 bench.py --workload metatrain --episodes-per-rank $k --steps 300 --warmup 10 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['last_loss'])"
done; done
for h in 0 1; do echo -n "20-shot pair_f16x2=$h: "; MFT_PAIR_F16X2=$h python3 bench.py --n-shot 20 --steps 1 --warmup 1 --no-cpu-baseline --no-standalone --strong-episodes 0 --validate-episodes 0 --no-other-configs 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['value'], d['mean_acc'])"; done
MFT_PAIR_F16X2=1 python3 -m pytest tests/test_metatrain_gpu.py tests/test_engine_gpu.py -m gpu -q -x 2>&1 | tail -4
