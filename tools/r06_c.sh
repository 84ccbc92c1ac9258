cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r06_c; rm -rf $O; mkdir -p $O
python3 -m pytest tests/test_metatrain_gpu.py tests/test_modules_gpu.py tests/test_kernels_gpu.py tests/test_drivers_gpu.py -m gpu -q -x -k "not f16x2 and not x3" > $O/pytest.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest.log
tail -12 $O/pytest.log
for k in 1 4; do
  python3 bench.py --workload metatrain --episodes-per-rank $k --steps 300 --warmup 10 --no-cpu-baseline > $O/bench_metatrain_k$k.json 2> $O/bench_metatrain_k$k.err
  tail -1 $O/bench_metatrain_k$k.json | cut -c1-330
done
