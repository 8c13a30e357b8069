"""Time ICC refine (no profiler): ms per 100 iterations, for tuning."""
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
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print(f"MF_ICC_SX={os.environ.get('MF_ICC_SX')} scenes={args.scenes_per_gpu} icc ms/refine = {s.elapsed_time(e)/10:.3f}  us/iter = {s.elapsed_time(e)/10/args.icc_iters*1e3:.2f}")
# clock check: after a heavy GEMM warm-up
import subprocess
x = torch.randn(8192, 8192, device="cuda")
torch.cuda.synchronize()
import time
t0 = time.time()
while time.time() - t0 < 3.0:
    y = x @ x
torch.cuda.synchronize()
s.record()
for _ in range(10): run()
e.record(); torch.cuda.synchronize()
print(f"after GEMM warm-up: icc us/iter = {s.elapsed_time(e)/10/args.icc_iters*1e3:.2f}")
print(subprocess.run("rocm-smi --showclocks 2>&1 | grep -E 'sclk|mclk|fclk' | head -6", shell=True, capture_output=True, text=True).stdout)
# interleave: is one refine faster when repeated back-to-back 100x?
s.record()
for _ in range(100): run()
e.record(); torch.cuda.synchronize()
print(f"100 back-to-back refines: us/iter = {s.elapsed_time(e)/100/args.icc_iters*1e3:.2f}")
print(subprocess.run("rocm-smi --showclocks 2>&1 | grep -E 'sclk|mclk|fclk' | head -6", shell=True, capture_output=True, text=True).stdout)
