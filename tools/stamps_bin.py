"""Per-workgroup phase stamps of the last k_icc_bin launch of a 20-iteration refinement (MF_ICC_DEBUG=32)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ICC_DEBUG"] = "32"
os.environ.setdefault("MF_LIBMFHIP", "libmfhip_dbg.so")
import morefusion_amd as mf  # noqa: E402
from bench import Workload, parse  # noqa: E402
args = parse(); wl = Workload(args, 0, torch.device("cuda", 0))
for _ in range(2):
    wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
    wl.icc.refine(wl.q, wl.t, wl.m, wl.v, 20)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
mf._lib.lib().mf_icc_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
sb = buf.reshape(4096, 8)[3072:3072 + 512].astype(np.int64)
started = sb[:, 0] > 0
live = sb[:, 3] > sb[:, 0]
t0 = sb[started, 0].min()
us = lambda a, b: (sb[:, b] - sb[:, a]) / 100.0
names = (("loads+gather", 0, 4), ("step math", 4, 5), ("count", 5, 1), ("global atomic", 1, 2), ("stores", 2, 3), ("total", 0, 3))
print("bin WGs", started.sum(), "live", live.sum(), "span", (sb[live, 3].max() - t0) / 100.0, "start skew", (sb[started, 0].max() - t0) / 100.0)
for n, a, b in names:
    x = us(a, b)[live]; print(f"{n:14s} mean {x.mean():6.2f} max {x.max():6.2f}")
