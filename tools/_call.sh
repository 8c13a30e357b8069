cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05h "py=tools/time_gemm_bf16.py+16+--no-stock" "t=test_gpu_bf16_kernels.py" > gpurun_out/r05h_0.log 2>&1
python examples/singleview_3d_train.py --global-batch 16 --steps 10 --json gpurun_out/r05h/train_rate.json > gpurun_out/r05h/train_rate.log 2>&1
