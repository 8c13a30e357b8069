cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
timeout 900 python -m pytest -m gpu -x -q tests/test_gpu_training.py -k hipgraph > gpurun_out/r05l_test.log 2>&1; echo "rc $?" >> gpurun_out/r05l_test.log
