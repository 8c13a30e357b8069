cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05k
timeout 600 python -X faulthandler examples/singleview_3d_train.py --global-batch 16 --steps 14 --graph --json gpurun_out/r05k/train_graph.json > gpurun_out/r05k/train_graph.log 2>&1; echo "rc $?" >> gpurun_out/r05k/train_graph.log
