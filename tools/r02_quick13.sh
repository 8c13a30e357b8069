#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_predict_vs_reference.py tests/test_gpu_predict_parity.py tests/test_gpu_reference_cuda_text.py -x -q -m gpu > gpurun_out/r02/q13_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|Error|Mismatch|Max " gpurun_out/r02/q13_tests.log | tail -8
