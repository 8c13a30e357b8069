#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r02/c32_tests.log 2>&1; echo "gpu tests rc $?"; grep -E "passed|failed|FAILED|Error" gpurun_out/r02/c32_tests.log | tail -5
bash tools/r02_profiles.sh 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/r02/c32_bench.json 2> gpurun_out/r02/c32_bench.err; echo "bench rc $?"; cut -c1-900 gpurun_out/r02/c32_bench.json
timeout 300 python bench.py --scenes-per-gpu 8 --steps 10 --no-cpu-baseline > gpurun_out/r02/c32_bench_s8.json 2>/dev/null; cut -c1-200 gpurun_out/r02/c32_bench_s8.json
timeout 300 python bench.py --dtype bf16 --steps 10 --no-cpu-baseline > gpurun_out/r02/c32_bench_bf16.json 2>/dev/null; cut -c1-200 gpurun_out/r02/c32_bench_bf16.json
