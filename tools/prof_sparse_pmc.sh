#!/bin/bash
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r01
timeout 150 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/r01/sc_pmc -o s -- python tools/time_sparse_conv.py > gpurun_out/r01/sc_pmc.log 2>&1
python - <<'PY'
import csv, collections
agg=collections.defaultdict(lambda: collections.defaultdict(lambda:[0.0,0]))
dur=collections.defaultdict(lambda:[0,0])
try:
    for r in csv.DictReader(open('gpurun_out/r01/sc_pmc/s_counter_collection.csv')):
        n=r['Kernel_Name']
        if 'k_sc_' in n:
            agg[n][r['Counter_Name']][0]+=float(r['Counter_Value']); agg[n][r['Counter_Name']][1]+=1
    for n,d in agg.items():
        print(n[:50], {k:round(v[0]/v[1],1) for k,v in d.items()})
except Exception as e:
    print("ERR", e)
    import subprocess; print(subprocess.run("tail -5 gpurun_out/r01/sc_pmc.log", shell=True, capture_output=True, text=True).stdout)
PY
