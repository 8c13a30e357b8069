#!/bin/bash
# 20 fresh processes of the captured data-parallel training step over RCCL at world size 1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p "$O"
ok=0
for i in $(seq 1 20); do
  timeout 300 python examples/singleview_3d_train.py --ddp --graph --global-batch 16 --steps 6 --json "$O/run_$i.json" > "$O/run_$i.log" 2>&1
  rc=$?
  line=$(python -c "import json; d=json.load(open('$O/run_$i.json')); print(d['hipgraph_step'], d['hipgraph_capture_error'], d['objects_per_s_steady_mean'], d['exchange_chunks'])" 2>/dev/null)
  echo "process $i: rc $rc  captured / error / objects_per_s / chunks: $line"
  [ $rc -eq 0 ] && ok=$((ok+1))
done
echo "clean processes: $ok / 20"
