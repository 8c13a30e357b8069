#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
bash tools/r02_quick.sh q11
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r02/q11_bench.json 2> gpurun_out/r02/q11_bench.err; echo "bench rc $?"; tail -3 gpurun_out/r02/q11_bench.err
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r02/q11_bench.json").read().strip().splitlines()[-1])
for k in ("value","value_serial","value_handwritten_path","handwritten_path_ms","latency_batch1_ms","stage_ms","ms_per_step"): print(k, d.get(k))
print(json.dumps(d["roofline"])[:900])
PY
