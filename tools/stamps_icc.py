import os, sys, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ICC_DEBUG"] = "32"
os.environ.setdefault("MF_LIBMFHIP", "libmfhip_dbg.so")
import morefusion_amd as mf
from bench import Workload, parse
args = parse(); wl = Workload(args, 0, torch.device("cuda", 0))
if os.environ.get("STAMP_REFINE"):   # two loop iterations: the second k_icc_bin carries the folded step
    wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
    wl.icc.refine(wl.q, wl.t, wl.m, wl.v, 2)
else:
    loss, gq, gt = wl.icc.loss_grad(wl.q0, wl.t0)   # one iteration (eager launches)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
mf._lib.lib().mf_icc_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
st = buf.reshape(4096, 8)[:1024].astype(np.int64)
t0 = st[:, 0].min()
d = lambda a, b: (st[:, b] - st[:, a]) / 100.0   # wall_clock64 = 100 MHz -> us
print("tile WGs", (st[:, 0] > 0).sum())
for name, a, b in (("init+load", 0, 1), ("pass1", 1, 2), ("pass2", 2, 3), ("epilogue", 3, 4), ("total", 0, 4)):
    x = d(a, b); print(f"tile {name:9s} mean {x.mean():7.2f} us  max {x.max():7.2f} us  (argmax wg {x.argmax()})")
print("tile start skew (us): max", (st[:, 0] - t0).max() / 100.0, " end max", (st[:, 4] - t0).max() / 100.0)
ns = st[:, 6]
print("records per tile: max", ns.max(), "mean", ns.mean())
worst = np.argsort(-d(0, 4))[:6]
for w in worst: print("wg", w, "grid", w // 64, "plane", (w % 64) // 2, "half", w % 2, "T", ns[w], "load", d(0,1)[w], "p1", d(1,2)[w], "p2", d(2,3)[w], "epi", d(3,4)[w])
sb = buf.reshape(4096, 8)[3072:3072+512].astype(np.int64)
live = sb[:, 3] > 0
db = lambda a, b: (sb[live, b] - sb[live, a]) / 100.0
print("bin WGs", (sb[:, 0] > 0).sum(), "live", live.sum())
steps = (("loads+gather", 0, 4), ("step math", 4, 5), ("count", 5, 1)) if os.environ.get("STAMP_REFINE") else (("loads+count", 0, 1),)
for name, a, b in steps + (("global atomic", 1, 2), ("stores", 2, 3), ("total", 0, 3)):
    x = db(a, b); print(f"bin {name:13s} mean {x.mean():7.2f} us  max {x.max():7.2f} us")
print("bin start skew max", (sb[sb[:,0]>0, 0] - sb[sb[:,0]>0, 0].min()).max() / 100.0, "end max", (sb[live, 3] - sb[sb[:,0]>0, 0].min()).max() / 100.0)

st2 = buf.reshape(4096, 8)[2048:2048+256].astype(np.int64)
d2 = lambda a, b: (st2[:, b] - st2[:, a]) / 100.0
print("K2 WGs", (st2[:, 0] > 0).sum())
for name, a, b in (("loads", 0, 1), ("voxels", 1, 2), ("reduce", 2, 3), ("total", 0, 3)):
    x = d2(a, b); print(f"K2 {name:9s} mean {x.mean():7.2f} us  max {x.max():7.2f} us  (argmax wg {x.argmax()})")
print("K2 start skew max", (st2[:, 0] - st2[:, 0].min()).max() / 100.0, "end max", (st2[:, 3] - st2[:, 0].min()).max() / 100.0)
