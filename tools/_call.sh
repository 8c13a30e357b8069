cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r04w "py=tools/time_gemm_bf16.py+16+--no-stock" > gpurun_out/r04w_0.log 2>&1
