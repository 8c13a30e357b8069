"""Model.predict vs Model.predict_graphed (everything after the point selection replayed from one hipGraph) at the
bench's batch (B = 8 objects) on the MI355X: is the eager step launch-bound?"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import Workload, parse
args = parse()
torch.backends.cudnn.benchmark = True
wl = Workload(args, 0, torch.device("cuda", 0))
m, inp = wl.model, wl.inputs


def per_call(name, fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    print(f"{name:40s} {(time.perf_counter() - t0) / reps * 1e3:8.3f} ms", flush=True)


with torch.no_grad():
    for rnd in range(2):
        per_call("predict (eager)", lambda: m.predict(**inp))
        per_call("predict_graphed (hipGraph replay)", lambda: m.predict_graphed(**inp, clone=False))
    a = m.predict(**inp)
    b = m.predict_graphed(**inp)
    for x, y in zip(a, b):
        print("max |eager - graphed|", float((x - y).abs().max()))
