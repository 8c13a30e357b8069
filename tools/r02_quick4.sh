#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_icc.py tests/test_gpu_accuracy_population.py -x -q -s > gpurun_out/r02/${TAG}_tests.log 2>&1; echo "icc tests rc $?"; grep -E "population|passed|failed|Error" gpurun_out/r02/${TAG}_tests.log | tail -3
python tools/stamps_fused.py 2>&1 | grep -v amdgpu | tail -14 | head -9
python tools/stamps_bin.py 2>&1 | grep -v amdgpu | tail -7
VARIANTS=0 python tools/time_icc_stage.py 2>&1 | grep -v amdgpu | tail -1
python tools/time_icc_quick.py 2>&1 | grep -v amdgpu | tail -1
