"""Time the pre-processing row (mf_instance_stats + mf_instance_crops) on one RGB-D frame."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morefusion_amd import geometry, synthetic  # noqa: E402
from oracle import oracle_np as O  # noqa: E402  (host timing beside it)
f = synthetic.make_rgbd_frame(0)
rgb, depth, label = (torch.as_tensor(f[k]).cuda() for k in ("rgb", "depth", "label"))
ids = torch.as_tensor(f["instance_ids"]).cuda()
run = lambda: geometry.instance_crops(rgb, depth, f["K"], label, ids)
for _ in range(5): out = run()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): out = run()
e.record(); torch.cuda.synchronize()
us = s.elapsed_time(e) / 200 * 1e3
H, W = depth.shape; n = ids.numel(); S = 256
alg = H * W * 11 + n * S * S * 15
print(f"instance_crops: {us:.1f} us/frame ({n} instances, {H}x{W}); algorithmic {alg/1e6:.2f} MB -> {alg/us/1e3:.1f} GB/s")
t0 = time.perf_counter(); O.instance_crops(f["rgb"], f["depth"], f["K"], f["label"], f["instance_ids"]); t1 = time.perf_counter()
print(f"host restatement (NumPy): {(t1-t0)*1e3:.1f} ms/frame")
pitch = torch.full((n,), 0.008, device="cuda")
for _ in range(3): geometry.grid_origin(out["pcd"], pitch)
torch.cuda.synchronize(); s.record()
for _ in range(50): geometry.grid_origin(out["pcd"], pitch)
e.record(); torch.cuda.synchronize()
print(f"grid_origin (torch sort): {s.elapsed_time(e)/50*1e3:.1f} us")
