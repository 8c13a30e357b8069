cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04f; mkdir -p $O
echo "=== bf16 tests"; timeout 900 python -m pytest tests/test_gpu_bf16_kernels.py tests/test_gpu_training.py -m gpu -x -q > $O/tests.log 2>&1; echo "rc $?"; tail -12 $O/tests.log | cut -c1-300
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
