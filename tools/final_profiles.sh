#!/bin/bash
# Round-end records on the GPU box (gpurun_out/final6/): the bench line, the steady-state kernel tables of the bench's
# timed steps and of one training step, the training rates (eager / captured / captured under DDP over RCCL)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/final6; mkdir -p "$O"
timeout 900 python bench.py > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "bench rc $?"
MF_MARK=k_icc_scene_setup MF_BENCH_MARK=1 bash tools/gpu_call.sh final6 "prof=bench_steady=MF_BENCH_MARK=1+python+bench.py+--no-extras+--no-cpu-baseline+--no-latency-probe" | tail -12 | cut -c1-150
MF_MARK=erfinv MF_SEQ=$O/train_seq.csv bash tools/gpu_call.sh final6 "prof=train_step=MF_TRAIN_MARK=1+python+examples/singleview_3d_train.py+--global-batch+16+--steps+8" | tail -12 | cut -c1-150
for v in "eager:" "graph:--graph" "ddp_eager:--ddp" "ddp_graph:--ddp+--graph"; do
  n=${v%%:*}; f=${v#*:}
  timeout 300 python examples/singleview_3d_train.py --global-batch 16 --steps 12 ${f//+/ } --json "$O/train_$n.json" > "$O/train_$n.log" 2>&1; echo "train $n rc $?"
  python -c "import json; d=json.load(open('$O/train_$n.json')); print({k:d[k] for k in ('objects_per_s_steady_mean','hipgraph_step','ddp')})"
done
