"""ICC refine timing only (us per iteration), for MF_ICC_* tuning sweeps."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse  # noqa: E402
args = parse()
wl = Workload(args, 0, torch.device("cuda", 0))
def run():
    wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
    wl.icc.refine(wl.q, wl.t, wl.m, wl.v, args.icc_iters, step0=0, alpha_q=0.01, alpha_t=0.001)
for _ in range(3): run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): run()
e.record(); torch.cuda.synchronize()
print(f"U={os.environ.get('MF_ICC_U')} SX={os.environ.get('MF_ICC_SX')} us/iter = {s.elapsed_time(e)/20/args.icc_iters*1e3:.2f}  pose checksum {float(wl.q.double().sum()+wl.t.double().sum()):.9f}")
