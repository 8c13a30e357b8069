"""ICC refine timing (us per iteration from the 100- vs 20-iteration difference), default path, for env-variable A/B runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse  # noqa: E402
args = parse()
wl = Workload(args, 0, torch.device("cuda", 0))
def run(n):
    wl.q.copy_(wl.q0); wl.t.copy_(wl.t0); wl.m.zero_(); wl.v.zero_()
    wl.icc.refine(wl.q, wl.t, wl.m, wl.v, n, step0=0, alpha_q=0.01, alpha_t=0.001)
def timed(n, reps=20):
    for _ in range(3): run(n)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): run(n)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / reps
t100, t20 = timed(100), timed(20)
print(f"MF_ICC_DEBUG={os.environ.get('MF_ICC_DEBUG')} scenes={args.scenes_per_gpu} us/iter = {(t100 - t20) / 80 * 1e3:.3f}  checksum {float(wl.q.double().sum() + wl.t.double().sum()):.12f}")
