cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export MF_MARK=erfinv
timeout 900 python -m pytest -m gpu -x -q -s tests/test_gpu_checkpoint.py -k round_trip > gpurun_out/r05a_ckpt.log 2>&1; echo "rc $?" >> gpurun_out/r05a_ckpt.log
bash tools/gpu_call.sh r05a "k=average_distance+or+add_loss+or+training+or+loss" "prof=train=MF_TRAIN_MARK=1+python+examples/singleview_3d_train.py+--global-batch+16+--steps+8+--json+gpurun_out/r05a/train.json" > gpurun_out/r05a_0.log 2>&1
