#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_preprocess.py tests/test_gpu_predict_parity.py -x -q > gpurun_out/r02/q8_tests.log 2>&1; echo "tests rc $?"; grep -E "passed|failed|Error" gpurun_out/r02/q8_tests.log | tail -3
WHAT=predict REPS=6 CUDNN_BENCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q8 -o p -- python tools/prof_icc.py > gpurun_out/prof_q8.log 2>&1
grep -E "k_valid|k_interp|k_sc_|k_avgvox" gpurun_out/prof_q8/p_kernel_stats.csv | cut -c1-110
rm -rf gpurun_out/prof_q8
