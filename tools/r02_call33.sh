#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/c33_tests.log 2>&1; echo "gpu tests rc $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/r02/c33_tests.log | tail -5
timeout 600 python bench.py > gpurun_out/r02/c33_bench.json 2> gpurun_out/r02/c33_bench.err; echo "bench rc $?"; cut -c1-250 gpurun_out/r02/c33_bench.json
