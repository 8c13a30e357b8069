#!/bin/bash
# Round profile collection (run on the GPU box through gpurun); outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
R=${1:-r01}
# 1. per-kernel time of the default bench command
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/$R/bench -o bench -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/$R/bench.log 2>&1
# 2. HBM counters of the voxelize op, separate passes (TCC slots: FETCH_SIZE 3, WRITE_SIZE 2)
REPS=20 timeout 200 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/$R/vox_fetch -o v -- python tools/prof_voxelize.py > gpurun_out/$R/vox_fetch.log 2>&1
REPS=20 timeout 200 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/$R/vox_write -o v -- python tools/prof_voxelize.py > gpurun_out/$R/vox_write.log 2>&1
ls gpurun_out/$R/*
tail -1 gpurun_out/$R/bench.log | cut -c1-400
