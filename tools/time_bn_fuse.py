"""A/B of the fused BatchNorm(eval) + add + ReLU launch (ops2d.bn_act) inside Model.predict at B = 8 on the MI355X."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse
args = parse()
wl = Workload(args, 0, torch.device("cuda", 0))
m, inp = wl.model, wl.inputs


def timeit(name, fn, reps=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps): fn()
    e.record(); torch.cuda.synchronize()
    print(f"{name:44s} {s.elapsed_time(e)/reps*1e3:9.1f} us", flush=True)


from morefusion_amd.models import ops2d
import torch.nn.functional as F
with torch.no_grad():
    for shape in ((8, 64, 64, 64), (8, 128, 32, 32), (8, 512, 32, 32)):
        bn = torch.nn.BatchNorm2d(shape[1]).cuda().eval()
        x = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
        idn = torch.randn(shape, device="cuda").contiguous(memory_format=torch.channels_last)
        timeit(f"{shape} k_bn_act bn+relu", lambda: ops2d.bn_act(x, bn, relu=True), 100)
        timeit(f"{shape} torch    bn+relu", lambda: F.relu(bn(x)), 100)
        timeit(f"{shape} k_bn_act bn+add+relu", lambda: ops2d.bn_act(x, bn, identity=idn, relu=True), 100)
        timeit(f"{shape} torch    bn+add+relu", lambda: F.relu(bn(x) + idn), 100)
    rgb = inp["rgb"].permute(0, 3, 1, 2)
    for rnd in range(2):
        for knob in ("0", "1"):
            os.environ["MF_TORCH_BN"] = knob
            tag = "torch BN + add + ReLU" if knob == "1" else "fused k_bn_act"
            timeit(f"predict, {tag}", lambda: m.predict(**inp))
            timeit(f"resnet18 alone, {tag}", lambda: m.resnet_extractor(rgb))
