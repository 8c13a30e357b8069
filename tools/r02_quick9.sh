#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r02; python tools/torch_prof_predict.py 2>&1 | grep -v amdgpu > gpurun_out/r02/q9_torchprof.txt; tail -70 gpurun_out/r02/q9_torchprof.txt | cut -c1-230
