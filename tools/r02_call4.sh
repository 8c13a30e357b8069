#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02/c6_tests.log 2>&1; echo "gpu tests rc $?"; tail -3 gpurun_out/r02/c6_tests.log
bash tools/prof_k.sh r02v3  2>&1 | tail -4
grep -E "k_icc" gpurun_out/prof_r02v3/icc_kernel_stats.csv | cut -c1-150 | head -5
timeout 120 python tools/stamps_icc.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r02/c6_stamps.log
