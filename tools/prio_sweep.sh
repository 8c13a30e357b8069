#!/bin/bash
# gpurun helper: two-stream step under the four stream-priority settings + 8 scenes/GPU.
mkdir -p gpurun_out
for p in none net-high icc-low; do
  timeout 150 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --priority $p > gpurun_out/prio_$p.json 2> gpurun_out/prio_$p.err
  echo "$p: $(python -c "import json;d=json.load(open('gpurun_out/prio_$p.json'));print(d['value'],d['ms_per_step'])" 2>&1 | tail -1)"
done
timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --scenes-per-gpu 8 > gpurun_out/s8.json 2> gpurun_out/s8.err
python -c "import json;d=json.load(open('gpurun_out/s8.json'));print('s8',d['value'],d['ms_per_step'],d['stage_ms'])"
