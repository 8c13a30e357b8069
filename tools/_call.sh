cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
bash tools/gpu_call.sh r05o "pmc=train_fetch=FETCH_SIZE=python+examples/singleview_3d_train.py+--global-batch+16+--steps+5" > gpurun_out/r05o_0.log 2>&1
bash tools/gpu_call.sh r05o "pmc=train_write=WRITE_SIZE=python+examples/singleview_3d_train.py+--global-batch+16+--steps+5" > gpurun_out/r05o_1.log 2>&1
python - <<'P'
import json
f=json.load(open("gpurun_out/r05o/pmc_train_fetch.json")); w=json.load(open("gpurun_out/r05o/pmc_train_write.json"))
out={}
for k,v in f.items():
    r={"launches":v["launches"],"us_under_pmc":v.get("avg_duration_us_under_pmc"),"fetch_bytes":v.get("fetch_bytes_x2_gfx950")}
    if k in w: r["write_bytes"]=w[k].get("write_bytes")
    if r["us_under_pmc"] and r.get("fetch_bytes") is not None and r.get("write_bytes") is not None:
        r["tb_per_s"]=round((r["fetch_bytes"]+r["write_bytes"])/r["us_under_pmc"]/1e6,3)
    out[k]=r
json.dump(out,open("gpurun_out/r05o/train_kernels_hbm_pmc.json","w"),indent=1)
for k,v in sorted(out.items(), key=lambda kv:-(kv[1]["us_under_pmc"] or 0)*kv[1]["launches"])[:40]:
    print(k[:60], v)
P
