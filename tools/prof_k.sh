#!/bin/bash
# usage: tools/prof_k.sh <tag> [env...]  -> prints k_icc kernel averages (us)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
env "$@" WHAT=icc REPS=2 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o icc -- python tools/prof_icc.py > gpurun_out/prof_$tag.log 2>&1
echo "== $tag $@"
grep -E "k_icc_(bin|tile|accum|step|fused)" gpurun_out/prof_$tag/icc_kernel_stats.csv | sed 's/(anonymous namespace):://g' | awk -F'",' '{split($2,a,","); printf "%-32s avg %.1f us\n", substr($1,2,30), a[3]/1000}'
