#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_training.py tests/test_gpu_preprocess.py -m gpu -x -q -s > gpurun_out/r02/c14_tests.log 2>&1; echo "tests rc $?"; grep -E "losses|passed|failed|FAILED|Error" gpurun_out/r02/c14_tests.log | tail -6
bash tools/r02_profiles.sh 2>&1 | tail -12
