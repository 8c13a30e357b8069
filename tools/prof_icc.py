"""Profile driver: ICC refine (8 objects, 100 iterations) x REPS, for rocprofv3."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
from bench import Workload, parse  # noqa: E402

sys.argv = [sys.argv[0]] + sys.argv[1:]
args = parse()
if os.environ.get("CUDNN_BENCH"): torch.backends.cudnn.benchmark = True
wl = Workload(args, 0, torch.device("cuda", 0))
reps = int(os.environ.get("REPS", "5"))
what = os.environ.get("WHAT", "icc")
for _ in range(reps):
    if what in ("icc", "all"):
        wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
        wl.icc.refine(wl.q, wl.t, wl.m, wl.v, args.icc_iters, step0=0, alpha_q=0.01, alpha_t=0.001)
    if what in ("predict", "all"):
        with torch.no_grad():
            wl.model.predict(**wl.inputs)
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
wl.step()
e.record()
torch.cuda.synchronize()
print("step ms", s.elapsed_time(e))
