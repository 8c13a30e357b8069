cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04j; mkdir -p $O
bash tools/gpu_call.sh r04j "prof=tdf=WHAT=tdf+REPS=10+python+tools/prof_kernels.py" 2>&1 | grep -E "k_tdf|k_pocc|rc " | cut -c1-160
echo "=== full gpu suite"; timeout 1500 python -m pytest tests -m gpu -q -x > $O/tests.log 2>&1; echo "rc $?"; tail -6 $O/tests.log | cut -c1-300
