cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04g; mkdir -p $O
echo "=== bf16 tests"; timeout 900 python -m pytest tests/test_gpu_bf16_kernels.py tests/test_gpu_training.py -m gpu -q > $O/tests.log 2>&1; echo "rc $?"; tail -12 $O/tests.log | cut -c1-300
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
echo "=== train profile, one steady step"
MF_TRAIN_MARK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python examples/singleview_3d_train.py --steps 6 --global-batch 16 > $O/prof_train.log 2>&1; echo "rc $?"
MF_MARK=erfinv python tools/kernel_stats.py $O/prof_train > $O/train_bf16_steady_step_kernel_stats.csv; head -36 $O/train_bf16_steady_step_kernel_stats.csv | cut -c1-150; rm -rf $O/prof_train
