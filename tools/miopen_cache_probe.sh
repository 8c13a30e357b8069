#!/bin/bash
# Populate an MIOpen user find-db by one bench run, then time a second run against it (gpurun_out/<tag>/)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/$1; mkdir -p "$O/db"
export MIOPEN_USER_DB_PATH=$PWD/$O/db
for i in 1 2; do
  t0=$(date +%s); timeout 900 python bench.py > "$O/bench_$i.json" 2> "$O/bench_$i.err"; echo "rc $? wall $(( $(date +%s) - t0 )) s"
  tail -n 1 "$O/bench_$i.err"
  python -c "
import json
d=json.loads(open('$O/bench_$i.json').read().strip().splitlines()[-1])
print({k:d.get(k) for k in ('value','ms_per_step','value_scenes8','train_objects_per_s','train_objects_per_s_hipgraph','bench_wall_s')}); print(d.get('bench_wall_sections_s')); print({k:(v['achieved'],v['frac']) for k,v in d.get('roofline_bf16_kernels',{}).items()})
"
done
ls -la "$O/db"; du -sh "$O/db"
