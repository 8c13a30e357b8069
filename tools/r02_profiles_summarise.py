"""Condense the raw rocprofv3 outputs of tools/r02_profiles.sh into the small files that are
committed under profiles/ (run on the GPU box right after collection)."""
import collections
import csv
import json
import os
import sys

P = sys.argv[1]
OUT = os.path.join(P, "summary")
os.makedirs(OUT, exist_ok=True)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0][:90]


# 1. steady state: kernels between the 2nd and 3rd k_icc_scene_setup (= the timed steps)
rows = list(csv.DictReader(open(os.path.join(P, "bench", "bench_kernel_trace.csv"))))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
marks = [i for i, r in enumerate(rows) if "k_icc_scene_setup" in r["Kernel_Name"]]
assert len(marks) >= 3, marks
win = rows[marks[1] + 1: marks[2]]
agg = collections.OrderedDict()
for r in win:
    d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    a = agg.setdefault(short(r["Kernel_Name"]), [0, 0, 1 << 62, 0])
    a[0] += 1; a[1] += d; a[2] = min(a[2], d); a[3] = max(a[3], d)
total = sum(a[1] for a in agg.values())
span = int(win[-1]["End_Timestamp"]) - int(win[0]["Start_Timestamp"])
with open(os.path.join(OUT, "r02_bench_steady_kernel_stats.csv"), "w") as f:
    f.write("# rocprofv3 --kernel-trace of `MF_BENCH_MARK=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline`; "
            "ONLY the 10 timed steps (between the marker launches; MIOpen find mode ran in the warm-up); "
            f"sum of kernel durations {total / 1e6:.3f} ms, wall span {span / 1e6:.3f} ms (two streams overlap)\n")
    f.write("Kernel,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percent\n")
    for n, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        f.write(f'"{n}",{a[0]},{a[1]},{a[1] / a[0]:.1f},{a[2]},{a[3]},{100 * a[1] / total:.2f}\n')


def counters(path, names, pick):
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if r["Counter_Name"] in names and pick(n):
            c = out[n][r["Counter_Name"]]
            c[0] += float(r["Counter_Value"]); c[1] += 1
    return {n: {k: v[0] / v[1] for k, v in d.items()} | {"calls": max(v[1] for v in d.values())} for n, d in out.items()}


# 2. HBM traffic per launch.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE reports half
# the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section): doubled as the guide says.
def hbm(prefix, pick):
    f = counters(os.path.join(P, prefix + "_fetch", prefix[0] + "_counter_collection.csv"), {"FETCH_SIZE"}, pick)
    w = counters(os.path.join(P, prefix + "_write", prefix[0] + "_counter_collection.csv"), {"WRITE_SIZE"}, pick)
    res = {}
    for n in f:
        fb = f[n]["FETCH_SIZE"] * 1024 * 2
        wb = w.get(n, {}).get("WRITE_SIZE", 0.0) * 1024
        res[n] = dict(fetch_bytes_corrected=round(fb), write_bytes=round(wb), traffic_bytes=round(fb + wb),
                      calls=f[n]["calls"])
    return res


icc = hbm("icc", lambda n: n.startswith("k_icc_"))
json.dump(dict(command="rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (separate passes) -- WHAT=icc python "
                       "tools/prof_icc.py (tools/r02_profiles.sh); 1 scene x 8 objects, per launch averages; "
                       "FETCH_SIZE x2 (gfx950 correction of the guide)", kernels=icc),
          open(os.path.join(OUT, "r02_icc_pmc_raw.json"), "w"), indent=1)
pred = hbm("pred", lambda n: n.startswith(("k_interp", "k_avgvox", "k_sc_")))
json.dump(dict(command="same, WHAT=predict (B = 8 objects)", kernels=pred),
          open(os.path.join(OUT, "r02_predict_pmc_raw.json"), "w"), indent=1)

# 3. MFMA pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)
m = counters(os.path.join(P, "pred_mfma", "p_counter_collection.csv"),
             {"SQ_VALU_MFMA_BUSY_CYCLES", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"}, lambda n: True)
rowsm = []
for n, d in m.items():
    if d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) <= 0 or d.get("GRBM_GUI_ACTIVE", 0) <= 0:
        continue
    util = d["SQ_VALU_MFMA_BUSY_CYCLES"] / (d["GRBM_GUI_ACTIVE"] / 8 * 1024)
    rowsm.append(dict(kernel=n, calls_profiled=d["calls"], mfma_busy_cycles=round(d["SQ_VALU_MFMA_BUSY_CYCLES"]),
                      gui_active_cycles_per_xcd=round(d["GRBM_GUI_ACTIVE"] / 8), mfma_pipe_util=round(util, 4)))
rowsm.sort(key=lambda r: -r["mfma_busy_cycles"] * r["calls_profiled"])
json.dump(dict(command="rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -- "
                       "WHAT=predict python tools/prof_icc.py; util = MFMA_BUSY / (GUI_ACTIVE/8 * 1024 SIMDs)",
               kernels=rowsm[:40]), open(os.path.join(OUT, "r02_predict_mfma_util.json"), "w"), indent=1)
print("summaries:", os.listdir(OUT))
