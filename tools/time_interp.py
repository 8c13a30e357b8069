"""interpolate_voxel_grid forward at the pose network's two shapes (B = 8 objects, 1000 points each):
HIP-event time per call, with and without the row ranges."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf
B, P = 8, 1000
for C, X in ((256, 16), (512, 8)):
    vox = torch.randn(B, C, X, X, X, device="cuda")
    pts = torch.rand(B * P, 3, device="cuda") * (X - 1)
    bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(P)
    start = torch.arange(B + 1, dtype=torch.int32, device="cuda") * P
    for name, bs in (("scan", None), ("ranges", start)):
        f = lambda: mf.functions.interpolate_voxel_grid(vox, pts, bi, channels_first=True, batch_start=bs)
        for _ in range(5): f()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(100): f()
        e.record(); torch.cuda.synchronize()
        us = s.elapsed_time(e) * 10
        alg = B * (C * X ** 3 * 4 + P * 16 + P * C * 4)
        print(f"C={C} X={X} {name:7s} {us:7.1f} us/call  {alg / us / 1e6:.2f} TB/s = {alg / us / 1e6 / 8:.3f} of 8 TB/s")
