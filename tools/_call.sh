# The GPU-box command of the round's last validation call (gpurun -- 'bash tools/_call.sh'):
# smoke + the whole -m gpu suite, the bench line, the captured training step.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh final smoke tests > gpurun_out/final_0.log 2>&1
timeout 900 python bench.py > gpurun_out/final/bench_default.json 2> gpurun_out/final/bench_default.err; echo "bench rc $?" >> gpurun_out/final_0.log
timeout 300 python examples/singleview_3d_train.py --global-batch 16 --steps 12 --graph --json gpurun_out/final/train_graph.json > gpurun_out/final/train_graph.log 2>&1; echo "graph train rc $?" >> gpurun_out/final_0.log
