#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_ops.py -m gpu -x -q -s > gpurun_out/r02/c13_tests.log 2>&1; echo "tests rc $?"; grep -E "losses|passed|failed|FAILED|Error" gpurun_out/r02/c13_tests.log | tail -8
