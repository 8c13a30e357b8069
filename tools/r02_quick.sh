#!/bin/bash
# quick ICC loop: ICC tests, per-kernel averages, stage stamps
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_icc.py tests/test_gpu_accuracy_population.py -x -q -s > gpurun_out/r02/${TAG}_tests.log 2>&1; echo "icc tests rc $?"; grep -E "population|passed|failed" gpurun_out/r02/${TAG}_tests.log | tail -3
bash tools/prof_k.sh $TAG 2>&1 | tail -5
python tools/time_icc_quick.py 2>&1 | grep -v amdgpu | tail -3
timeout 120 python tools/stamps_icc.py 2>&1 | grep -v amdgpu.ids > gpurun_out/r02/${TAG}_stamps.log; grep -E "total|pass|load|reduce|voxels" gpurun_out/r02/${TAG}_stamps.log
STAMP_REFINE=1 timeout 120 python tools/stamps_icc.py 2>&1 | grep -E "^bin " 
