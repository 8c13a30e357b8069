#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/stamps_bin.py 2>&1 | grep -v amdgpu | tail -7
python tools/time_icc_quick.py 2>&1 | grep -v amdgpu | tail -1
