cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
python3 -m pytest tests/test_accuracy_gpu.py -m gpu -q -x -k "g19" -s 2>&1 | grep -v Warning | tail -15
