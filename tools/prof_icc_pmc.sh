#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
WHAT=icc REPS=2 timeout 150 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d gpurun_out/r01/icc_fetch -o i -- python tools/prof_icc.py > gpurun_out/r01/icc_fetch.log 2>&1
WHAT=icc REPS=2 timeout 150 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d gpurun_out/r01/icc_write -o i -- python tools/prof_icc.py > gpurun_out/r01/icc_write.log 2>&1
python - <<'PY'
import csv, collections
def per_kernel(path, counter):
    agg=collections.defaultdict(lambda:[0.0,0])
    for r in csv.DictReader(open(path)):
        if r['Counter_Name']==counter and 'k_icc_' in r['Kernel_Name']:
            agg[r['Kernel_Name'][:60]][0]+=float(r['Counter_Value']); agg[r['Kernel_Name'][:60]][1]+=1
    return agg
f=per_kernel('gpurun_out/r01/icc_fetch/i_counter_collection.csv','FETCH_SIZE')
w=per_kernel('gpurun_out/r01/icc_write/i_counter_collection.csv','WRITE_SIZE')
for n in f: print(n, 'FETCH_KB avg', round(f[n][0]/f[n][1],1), 'WRITE_KB avg', round(w[n][0]/max(w[n][1],1),1), 'calls', f[n][1])
PY
