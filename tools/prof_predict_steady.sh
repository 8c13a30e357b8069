#!/bin/bash
# steady-state kernel list of ONE Model.predict call (after MIOpen find has settled)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
WHAT=predict REPS=8 CUDNN_BENCH=1 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_ps -o p -- python tools/prof_icc.py > gpurun_out/prof_ps.log 2>&1
python - <<'PY'
import csv, collections
rows=list(csv.DictReader(open('gpurun_out/prof_ps/p_kernel_trace.csv')))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last step = kernels after the last k_icc_step-free gap: take the tail window: find last 'k_avgvox_link' occurrence
idx=[i for i,r in enumerate(rows) if 'k_avgvox_link' in r['Kernel_Name']]
# the final wl.step() contains one predict: its voxelize is the last occurrence; predict started ~ at the previous 'SubTensorOpWithScalar' chain; take window between the last two avgvox_link +- context
last=idx[-1]; prev=idx[-2]
# predict window: from just after the previous predict's end. approximate: kernels between prev and last belong to (rest of prev predict) + (start of last predict); use last full predict = from first kernel after prev predict's heads... simpler: take period = rows[prev:last]
win=rows[prev:last]
tot=sum(int(r['End_Timestamp'])-int(r['Start_Timestamp']) for r in win)
span=int(win[-1]['End_Timestamp'])-int(win[0]['Start_Timestamp'])
print(f"kernels in one predict period: {len(win)}  sum of durations {tot/1e6:.3f} ms  wall span {span/1e6:.3f} ms")
agg=collections.OrderedDict()
for r in win:
    n=r['Kernel_Name'][:90]; d=int(r['End_Timestamp'])-int(r['Start_Timestamp'])
    a=agg.setdefault(n,[0,0]); a[0]+=d; a[1]+=1
for n,(d,c) in sorted(agg.items(), key=lambda kv:-kv[1][0])[:30]:
    print(f"{d/1e3:9.1f} us  x{c:3d}  {n}")
PY
