import os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf
from morefusion_amd.contrib.singleview_3d.models.sparse_conv import SparseVoxelConv3d
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
B, Cs, Cd, Cout, n, D = 8, 144, 16, 256, 1000, 32
conv = torch.nn.Conv3d(Cs + Cd, Cout, 4, 2, padding=1).cuda()
# clustered surface-like points (sphere cap), as in the model
u = torch.rand(B * n, 2, device="cuda") * 2 - 1
pts = torch.stack([16 + 9 * u[:, 0], 16 + 9 * u[:, 1], 16 - 8 * torch.sqrt((1 - (u ** 2).sum(1) / 2).clamp(min=0))], 1)
vals = torch.randn(B * n, Cs, device="cuda")
bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(n)
vox, counts = mf.functions.average_voxelization_3d(vals, pts, bi, batch_size=B, origin=(0, 0, 0), pitch=1.0, dimensions=(D,) * 3, return_counts=True)
print("occupied voxels per object", float((counts > 0).sum()) / B)
h_occ = torch.randn(B, Cd, D, D, D, device="cuda")
op = SparseVoxelConv3d(conv)
def timeit(name, fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:40s} {s.elapsed_time(e)/reps*1e3:9.1f} us")
with torch.no_grad():
    timeit("dense conv3 (cat + MIOpen + relu)", lambda: torch.relu(conv(torch.cat([vox, h_occ], 1))))
    timeit("sparse conv3 (incl. dense 16ch part)", lambda: op(vox, counts, h_occ, max_rows=B * n))
    timeit("sparse conv3 (sparse part only)", lambda: op(vox, counts, None, max_rows=B * n))
    timeit("dense 16ch conv only", lambda: F.conv3d(h_occ, op.Wd, None, stride=2, padding=1))
