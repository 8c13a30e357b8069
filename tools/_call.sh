cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MF_MARK=erfinv
bash tools/gpu_call.sh r04z "t=test_gpu_bf16_kernels.py" "prof=train=MF_TRAIN_MARK=1+python+examples/singleview_3d_train.py+--global-batch+16+--steps+8+--json+gpurun_out/r04z/train.json" > gpurun_out/r04z_0.log 2>&1
