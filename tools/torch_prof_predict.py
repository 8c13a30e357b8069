"""Which aten ops own the elementwise glue kernels of one Model.predict (torch profiler, shapes recorded)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse  # noqa: E402
from torch.profiler import profile, ProfilerActivity
args = parse()
torch.backends.cudnn.benchmark = True
wl = Workload(args, 0, torch.device("cuda", 0))
m, inp = wl.model, wl.inputs
with torch.no_grad():
    for _ in range(4): m.predict(**inp)
    torch.cuda.synchronize()
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
        m.predict(**inp)
        torch.cuda.synchronize()
rows = prof.key_averages(group_by_input_shape=True)
def ct(e):
    return getattr(e, "self_device_time_total", None) or getattr(e, "self_cuda_time_total", 0)
conv = sum(ct(e) for e in rows if "convolution" in e.key)
print(f"self device time in aten::*convolution*: {conv / 1e3:.3f} ms")
other = [e for e in rows if e.key.startswith("aten::") and "convolution" not in e.key and ct(e) > 0]
other.sort(key=ct, reverse=True)
print(f"all other aten ops: {sum(ct(e) for e in other) / 1e3:.3f} ms")
for e in other[:60]:
    print(f"{ct(e):9.1f} us x{e.count:3d}  {e.key:34s} {str(e.input_shapes)[:150]}")
