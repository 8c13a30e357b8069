cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04k; mkdir -p $O
echo "=== train rate"; timeout 600 python examples/singleview_3d_train.py --steps 8 --global-batch 16 --json $O/train_bf16.json > $O/train.log 2>&1; echo "rc $?"; tail -2 $O/train.log
echo "=== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "rc $?"; tail -4 $O/tests.log | cut -c1-300
bash tools/gpu_call.sh r04k "pmc=vox_fetch=FETCH_SIZE=WHAT=vox+REPS=5+python+tools/prof_kernels.py" "pmc=vox_write=WRITE_SIZE=WHAT=vox+REPS=5+python+tools/prof_kernels.py" "pmc=icc_fetch=FETCH_SIZE=WHAT=icc+REPS=3+python+tools/prof_kernels.py" "pmc=icc_write=WRITE_SIZE=WHAT=icc+REPS=3+python+tools/prof_kernels.py" > $O/pmc.log 2>&1; tail -2 $O/pmc.log | cut -c1-200
echo "=== train profile"; MF_TRAIN_MARK=1 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_train -o p -- python examples/singleview_3d_train.py --steps 6 --global-batch 16 > $O/prof_train.log 2>&1; echo "rc $?"
MF_MARK=erfinv python tools/kernel_stats.py $O/prof_train > $O/train_bf16_steady_step_kernel_stats.csv; head -24 $O/train_bf16_steady_step_kernel_stats.csv | cut -c1-150; rm -rf $O/prof_train
