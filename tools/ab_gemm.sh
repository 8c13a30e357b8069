#!/bin/bash
# A/B of NT-engine variants on the GPU box: tools/ab_gemm.sh <tag> "<ENV=..,ENV=..>" ...   (logs: gpurun_out/<tag>/)
# every variant = one run of tools/time_gemm_bf16.py 16 --no-stock --quick under that environment (comma-separated)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p "$O"; shift
n=0
for v in "$@"; do
  n=$((n+1))
  env ${v//,/ } timeout 600 python tools/time_gemm_bf16.py 16 --no-stock --quick > "$O/gemm_$n.log" 2>&1
  echo "[$v] rc $?"; tail -n 1 "$O/gemm_$n.log" | python -c "
import sys, json
for l in sys.stdin:
    l=l.strip()
    if not l.startswith('{'): continue
    d=json.loads(l)
    print({k:{kk.replace('_tflops',''):vv for kk,vv in v.items() if 'tflops' in kk} for k,v in d.items() if isinstance(v,dict)})
"
done
