#!/bin/bash
# ICC A/B loop: tests, then per-variant kernel averages + loop time.  usage: r02_quick2.sh <tag> [dbgmask...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
TAG=${1:-q}; shift
mkdir -p gpurun_out/r02
timeout 900 python -m pytest tests/test_gpu_icc.py -x -q > gpurun_out/r02/${TAG}_tests.log 2>&1; echo "icc tests rc $?"; grep -E "passed|failed|Error" gpurun_out/r02/${TAG}_tests.log | tail -3
python tools/time_icp.py 2>&1 | grep -v amdgpu | tail -1
for V in 0 "$@"; do
  echo "--- MF_ICC_DEBUG=$V"
  MF_ICC_DEBUG=$V bash tools/prof_k.sh ${TAG}_$V MF_ICC_DEBUG=$V 2>&1 | tail -3
  MF_ICC_DEBUG=$V python tools/time_icc_quick.py 2>&1 | grep -v amdgpu | tail -1
  rm -rf gpurun_out/prof_${TAG}_$V
done
