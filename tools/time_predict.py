"""Stage timing of Model.predict on the MI355X (events; steady state)."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf
from bench import Workload, parse
args = parse()
if os.environ.get("CUDNN_BENCH"): torch.backends.cudnn.benchmark = True
wl = Workload(args, 0, torch.device("cuda", 0))
m, inp = wl.model, wl.inputs
B = wl.B

def timeit(name, fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): out = fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:34s} {s.elapsed_time(e)/reps*1e3:9.1f} us")
    return out

with torch.no_grad():
    rgb = inp["rgb"].float().permute(0, 3, 1, 2).contiguous()
    timeit("predict (all)", lambda: m.predict(**inp))
    h512 = timeit("resnet18", lambda: m.resnet_extractor(rgb))
    hrgb = timeit("pspnet", lambda: m.pspnet_extractor(h512))
    for cl in (False, True):
        x = rgb.contiguous(memory_format=torch.channels_last) if cl else rgb
        mm = m.resnet_extractor.to(memory_format=torch.channels_last) if cl else m.resnet_extractor
        timeit(f"resnet18 channels_last={cl}", lambda: mm(x))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        timeit("resnet18 bf16 autocast", lambda: m.resnet_extractor(rgb))
        timeit("pspnet bf16 autocast", lambda: m.pspnet_extractor(h512))
    P = 1000
    values = torch.randn(B, 32, P, device="cuda"); points = torch.rand(B, 3, P, device="cuda") * 20 + 6
    g = inp["grid_nontarget_empty"]
    timeit("_extract (3D part)", lambda: m._extract(values, points, g))
    feat2 = torch.randn(B, P, 144, device="cuda")
    vox = timeit("voxelize", lambda: m._voxelize(feat2, points.transpose(1, 2)))
    gg = g.float()[:, None]
    hocc = timeit("conv1_occ+conv2_occ", lambda: F.relu(m.conv2_occ(F.relu(m.conv1_occ(gg)))))
    v160 = torch.cat([vox, hocc], 1)
    h3 = timeit("conv3 (160->256 k4s2)", lambda: F.relu(m.conv3(v160)))
    h4 = timeit("conv4 (256->512 k4s2)", lambda: F.relu(m.conv4(h3)))
    v160cl = v160.contiguous(memory_format=torch.channels_last_3d)
    m3 = m.conv3.to(memory_format=torch.channels_last_3d)
    timeit("conv3 channels_last_3d", lambda: F.relu(m3(v160cl)))
    with torch.autocast("cuda", dtype=torch.bfloat16):
        timeit("conv3 bf16", lambda: F.relu(m.conv3(v160)))
        timeit("conv4 bf16", lambda: F.relu(m.conv4(h3)))
        timeit("conv3 bf16 cl3d", lambda: F.relu(m3(v160cl)))
    bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(P)
    idx = points.transpose(1, 2).reshape(B * P, 3).contiguous()
    timeit("interp feat3", lambda: mf.functions.interpolate_voxel_grid(h3, idx / 2, bi, channels_first=True))
    timeit("interp feat4", lambda: mf.functions.interpolate_voxel_grid(h4, idx / 4, bi, channels_first=True))
    feat = torch.randn(B, 984, P, device="cuda")
    def heads():
        for n in ("rot", "trans", "conf"):
            x = F.relu(getattr(m, f"conv1_{n}")(feat)); x = F.relu(getattr(m, f"conv2_{n}")(x))
            x = F.relu(getattr(m, f"conv3_{n}")(x)); x = getattr(m, f"conv4_{n}")(x)
        return x
    timeit("heads (3 x 4 conv1d)", heads)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        timeit("heads bf16", heads)
