cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05n smoke tests > gpurun_out/r05n_0.log 2>&1
timeout 900 python bench.py > gpurun_out/r05n/bench_default.json 2> gpurun_out/r05n/bench_default.err; echo "bench rc $?" >> gpurun_out/r05n_0.log
for i in 1 2; do timeout 300 python examples/singleview_3d_train.py --global-batch 16 --steps 12 --graph --json gpurun_out/r05n/train_graph_$i.json > gpurun_out/r05n/train_graph_$i.log 2>&1; echo "graph train $i rc $?" >> gpurun_out/r05n_0.log; done
