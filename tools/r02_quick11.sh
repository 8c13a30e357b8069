#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/q11_tests.log 2>&1; echo "gpu tests rc $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/r02/q11_tests.log | tail -5
WHAT=predict REPS=6 CUDNN_BENCH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_q11 -o p -- python tools/prof_icc.py > gpurun_out/prof_q11.log 2>&1
grep -E "k_sc_reduce|k_sc_gemm|k_valid" gpurun_out/prof_q11/p_kernel_stats.csv | cut -c1-130
rm -rf gpurun_out/prof_q11
