"""One k_icc_fused launch on a prepared scene, timed per MF_ICC_DEBUG variant (valid inputs every time)."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
from bench import Workload, parse  # noqa: E402
args = parse()
wl = Workload(args, 0, torch.device("cuda", 0))
lib = mf._lib.lib()
icc = wl.icc
icc.prepare()
st = torch.cuda.current_stream().cuda_stream
stage = 2 if icc.desc.grid_ne_binary else 1
def launch(sg, q=None, t=None):
    mf._lib.check(lib.mf_icc_launch_stage(ctypes.byref(icc.desc), q, t, icc.ws.data_ptr(), sg, st), "stage")
for v in [int(x) for x in os.environ.get("VARIANTS", "0").split(",")]:
    os.environ["MF_ICC_DEBUG"] = str(v)
    launch(0, wl.q0.data_ptr(), wl.t0.data_ptr())
    for _ in range(5): launch(stage)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e9
    for rep in range(5):
        s.record()
        for _ in range(50): launch(stage)
        e.record(); torch.cuda.synchronize()
        best = min(best, s.elapsed_time(e) / 50 * 1e3)
    print(f"dbg {v:4d}: stage {stage} launch {best:.2f} us (back-to-back, includes ~1.5 us boundary)")
