cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05p "t=test_gpu_bf16_kernels.py" > gpurun_out/r05p_0.log 2>&1
timeout 300 python examples/singleview_3d_train.py --global-batch 16 --steps 12 --graph --json gpurun_out/r05p/train_graph.json > gpurun_out/r05p/train_graph.log 2>&1
timeout 300 python examples/singleview_3d_train.py --global-batch 16 --steps 10 --json gpurun_out/r05p/train_eager.json > gpurun_out/r05p/train_eager.log 2>&1
