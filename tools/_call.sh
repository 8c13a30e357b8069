cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
hipcc --offload-arch=gfx950 -O2 -o /tmp/probe_tr16 tools/probe_tr16.hip 2>/dev/null && /tmp/probe_tr16 > gpurun_out/r04x_probe.log 2>&1; echo "probe rc $?" >> gpurun_out/r04x_probe.log
bash tools/gpu_call.sh r04x "py=tools/time_gemm_bf16.py+16+--no-stock" "t=test_gpu_bf16_kernels.py" > gpurun_out/r04x_0.log 2>&1
bash tools/gpu_call.sh r04x "pmc=g1=SQ_INSTS_VALU,SQ_INSTS_MFMA,SQ_WAVE_CYCLES,SQ_WAIT_INST_ANY=python+tools/time_gemm_bf16.py+16+--no-stock" > gpurun_out/r04x_1.log 2>&1
bash tools/gpu_call.sh r04x "pmc=g2=SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE,SQ_WAIT_INST_LDS,SQ_WAVES=python+tools/time_gemm_bf16.py+16+--no-stock" > gpurun_out/r04x_2.log 2>&1
python - <<'P'
import json
out={}
for i in (1,2):
    d=json.load(open(f"gpurun_out/r04x/pmc_g{i}.json"))
    for k,v in d.items():
        if "gemm" in k: out.setdefault(k,{}).update(v)
for k,v in out.items():
    w=v.get("SQ_WAVES",1)
    print(k[:36], {a[3:]:round(b/w,1) for a,b in v.items() if a.startswith("SQ_")}, v.get("avg_duration_us_under_pmc"))
json.dump(out, open("gpurun_out/r04x/gemm_issue_pmc.json","w"), indent=1)
P
