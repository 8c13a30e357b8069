"""Per-workgroup phase stamps of the LAST k_icc_iter launch of a short refinement (MF_ICC_DEBUG=32; needs the
library built with `make ICC_DEBUG=1`)."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["MF_ICC_DEBUG"] = "32"
os.environ.setdefault("MF_LIBMFHIP", "libmfhip_dbg.so")
import morefusion_amd as mf  # noqa: E402
from bench import Workload, parse  # noqa: E402
args = parse(); wl = Workload(args, 0, torch.device("cuda", 0))
lib = mf._lib.lib()
n_iter = int(os.environ.get("STAMP_ITERS", "12"))
for _ in range(3):
    wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
    wl.icc.refine(wl.q, wl.t, wl.m, wl.v, n_iter, step0=0, alpha_q=0.01, alpha_t=0.001)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, np.uint64)
lib.mf_icc_debug_stamps(buf.ctypes.data_as(ctypes.c_void_p), buf.size)
nwg = 64 * wl.icc.desc.n_objects
st = buf.reshape(4096, 8)[:nwg].astype(np.int64)
info = buf.reshape(4096, 8)[2048:2048 + nwg].astype(np.int64)
t0 = st[:, 0].min()
us = lambda a, b: (st[:, b] - st[:, a]) / 100.0
names = (("trip1+table(A)", 0, 1), ("sums(B)", 1, 2), ("step(C)", 2, 3), ("classify(D)", 3, 4), ("pass1(E)", 4, 5),
         ("pass2(F)", 5, 6), ("voxel", 6, 7), ("total", 0, 7))
print("WGs", nwg, "span", (st[:, 7].max() - t0) / 100.0, "us; start skew", (st[:, 0].max() - t0) / 100.0,
      "slow tiles", int(info[:, 2].sum()), "mean blocks", info[:, 0].mean(), "mean survivors", info[:, 1].mean(),
      "max survivors", info[:, 1].max(), "mean touched bins", info[:, 3].mean(), "max", info[:, 3].max())
for n, a, b in names:
    x = us(a, b); print(f"{n:16s} mean {x.mean():6.2f} p90 {np.percentile(x, 90):6.2f} max {x.max():6.2f}")
print("end time after kernel start: mean %.2f p90 %.2f max %.2f" % (((st[:, 7] - t0) / 100.0).mean(), np.percentile((st[:, 7] - t0) / 100.0, 90), ((st[:, 7] - t0) / 100.0).max()))
sb = buf.reshape(4096, 8)[1024:1024 + nwg].astype(np.int64)
rel = lambda col: (sb[:, col] - st[:, 0]) / 100.0
for nm, col in (("gather done", 0), ("state issued", 1), ("table done (last wave)", 2), ("zero fill done", 3)):
    print(f"  since start: {nm:24s} mean {rel(col).mean():6.2f} max {rel(col).max():6.2f}")
relA = lambda col: (sb[:, col] - st[:, 1]) / 100.0
for nm, col in (("record loads issued", 4), ("sums done (wave 0)", 5)):
    print(f"  since barrier A: {nm:20s} mean {relA(col).mean():6.2f} max {relA(col).max():6.2f}")
tot = us(0, 7)
for w in np.argsort(-tot)[:6]:
    print(f"wg {w:3d} obj {w // 64} plane {(w % 64) // 2:2d} half {w % 2} blocks {info[w,0]} surv {info[w,1]} slow {info[w,2]} bins {info[w,3]} start {(st[w,0]-t0)/100.0:5.2f} " +
          " ".join(f"{n} {us(a, b)[w]:.2f}" for n, a, b in names))
