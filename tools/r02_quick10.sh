#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_reference_cuda_text.py tests/test_gpu_preprocess.py -x -q > gpurun_out/r02/q10_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|Error|assert" gpurun_out/r02/q10_tests.log | tail -8
WHAT=predict REPS=6 CUDNN_BENCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q10 -o p -- python tools/prof_icc.py > gpurun_out/prof_q10.log 2>&1
grep -E "k_valid" gpurun_out/prof_q10/p_kernel_stats.csv | cut -c1-110
rm -rf gpurun_out/prof_q10
timeout 300 python examples/singleview_3d_train.py --steps 6 --global-batch 2 2>&1 | grep -v amdgpu | tail -3
