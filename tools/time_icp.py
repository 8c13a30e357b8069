"""ICP driver loop: fused mf_icp_refine vs the autograd + host-Adam loop (ms per 100 iterations)."""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
f = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fixture_pose_refinement_00000002.npz"))
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()
tgt = dev((np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32))
src = dev(f["pcd_cad"].astype(np.float32))
def fused(L):
    links = [mf.contrib.IterativeClosestPointLink(f["transform_init"]).to_gpu() for _ in range(L)]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    mf.contrib.icp_refine(links, [src] * L, [tgt] * L, n_iter=100)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
def loop():
    link = mf.contrib.IterativeClosestPointLink(f["transform_init"]).to_gpu()
    opt = mf.optimizers.Adam(alpha=0.01).setup(link)
    link.translation.update_rule.hyperparam.alpha *= 0.1
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(100):
        link.zerograds(); loss = link(src, tgt); loss.backward(); opt.update()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e3
fused(1); loop()
print(f"icp 100 iterations (S={len(src)}, T={len(tgt)}): fused L=1 {min(fused(1) for _ in range(3)):.2f} ms, "
      f"fused L=8 {min(fused(8) for _ in range(3)):.2f} ms, autograd loop {min(loop() for _ in range(2)):.2f} ms")
