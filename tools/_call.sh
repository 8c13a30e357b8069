cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05m "py=tools/time_gemm_bf16.py+16+--no-stock" "t=test_gpu_bf16_kernels.py" > gpurun_out/r05m_0.log 2>&1
