#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_preprocess.py tests/test_gpu_predict_parity.py -x -q > gpurun_out/r02/q7_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|Error" gpurun_out/r02/q7_tests.log | tail -3
python tools/time_interp.py 2>&1 | grep -v amdgpu | tail -4
python tools/time_predict.py 2>&1 | grep -v amdgpu | grep -E "predict \(all\)|_extract"
