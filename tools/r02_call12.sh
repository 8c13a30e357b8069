#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
timeout 1500 python -m pytest tests -m gpu -x -q -s > gpurun_out/r02/c12_tests.log 2>&1; echo "gpu tests rc $?"; grep -E "population|passed|failed|FAILED" gpurun_out/r02/c12_tests.log | tail -5
