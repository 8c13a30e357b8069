#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/c15_tests.log 2>&1; echo "gpu tests rc $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/r02/c15_tests.log | tail -5
bash tools/r02_profiles.sh 2>&1 | tail -14
timeout 600 python bench.py > gpurun_out/r02/c15_bench.json 2> gpurun_out/r02/c15_bench.err; echo "bench rc $?"; cut -c1-1500 gpurun_out/r02/c15_bench.json
