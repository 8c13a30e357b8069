#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=predict REPS=6 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_predict -o p -- python tools/prof_icc.py > gpurun_out/prof_predict.log 2>&1
python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_predict/p_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total ms', tot/1e6)
for r in rows[:28]:
    print(f"{float(r['TotalDurationNs'])/1e6:9.3f} ms  calls {r['Calls']:>5}  avg {float(r['AverageNs'])/1e3:9.1f} us  {r['Name'][:110]}")
PY
