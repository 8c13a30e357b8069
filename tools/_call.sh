cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04l; mkdir -p $O
echo "=== time bf16"; timeout 600 python tools/time_gemm_bf16.py 16 --no-stock > $O/time_gemm_bf16.log 2>&1; echo "rc $?"; tail -1 $O/time_gemm_bf16.log | cut -c1-1800
echo "=== bf16 tests"; timeout 900 python -m pytest tests/test_gpu_bf16_kernels.py -m gpu -q > $O/tests.log 2>&1; echo "rc $?"; tail -3 $O/tests.log | cut -c1-200
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
