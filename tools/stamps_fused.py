"""Per-workgroup phase stamps of one k_icc_fused launch (MF_ICC_DEBUG=32)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ICC_DEBUG"] = "32"
os.environ.setdefault("MF_LIBMFHIP", "libmfhip_dbg.so")
import morefusion_amd as mf  # noqa: E402
from bench import Workload, parse  # noqa: E402
args = parse(); wl = Workload(args, 0, torch.device("cuda", 0))
lib = mf._lib.lib(); icc = wl.icc; icc.prepare()
st_ = torch.cuda.current_stream().cuda_stream
def launch(sg, q=None, t=None):
    mf._lib.check(lib.mf_icc_launch_stage(ctypes.byref(icc.desc), q, t, icc.ws.data_ptr(), sg, st_), "stage")
launch(0, wl.q0.data_ptr(), wl.t0.data_ptr())
for _ in range(3): launch(2)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
lib.mf_icc_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
st = buf.reshape(4096, 8)[:512].astype(np.int64)
ran = st[:, 4] > 0          # reached the end (tiles with an own winner)
started = st[:, 0] > 0
t0 = st[started, 0].min()
us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
names = (("records", 0, 1), ("pass1", 1, 2), ("pass2", 2, 3), ("gather+compact", 3, 5), ("math+rowsums", 5, 7), ("reduce", 7, 4), ("total", 0, 4))
print("WGs started", started.sum(), "with own winners", ran.sum(), " span", (st[ran, 4].max() - t0) / 100.0, "us; start skew", (st[started, 0].max() - t0) / 100.0)
for n, a, b in names:
    x = us(a, b)[ran]; print(f"{n:15s} mean {x.mean():6.2f} max {x.max():6.2f}")
tot = us(0, 4); tot[~ran] = -1
for w in np.argsort(-tot)[:8]:
    print(f"wg {w:3d} grid {w // 64} plane {(w % 64) // 2:2d} half {w % 2} recs {st[w, 6]:5d} start {(st[w,0]-t0)/100.0:5.2f} " +
          " ".join(f"{n} {us(a, b)[w]:.2f}" for n, a, b in names))
# placement: HW_REG_HW_ID bits: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...; XCC_ID low bits
pl = buf.reshape(4096, 8)[2048:2048 + 512].astype(np.int64)
hw, xcc = pl[:, 0], pl[:, 1] & 0xf
cu, sh, se = (hw >> 8) & 0xf, (hw >> 12) & 1, (hw >> 13) & 7
key = xcc * 1000 + se * 100 + sh * 20 + cu
import collections
groups = collections.defaultdict(list)
for w in range(512):
    if started[w]: groups[int(key[w])].append(w)
print("distinct (xcc,se,sh,cu):", len(groups), " workgroups per CU histogram:", collections.Counter(len(v) for v in groups.values()))
print("xcc of wg 0..15:", list(xcc[:16]), " first CU groups:", [v for v in list(groups.values())[:12]])
dur = tot.copy()
pair = [sum(max(dur[w], 0) for w in v) for v in groups.values()]
print("sum of workgroup durations per CU: mean %.2f max %.2f min %.2f" % (np.mean(pair), np.max(pair), np.min(pair)))
