"""Driver for PMC passes: mf_average_voxelization_3d_fwd on the exact arguments
Model.predict passes it (B=8, P=1000, C=144, 32^3), REPS launches."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf
from bench import Workload, parse
import morefusion_amd.contrib.singleview_3d.models.model as model_mod
args = parse()
wl = Workload(args, 0, torch.device("cuda", 0))
cap = {}
real = model_mod.functions_module.average_voxelization_3d
def spy(values, points, bi, **kw):
    cap["a"] = (values.clone(), points.clone(), bi.clone()); cap["kw"] = kw
    return real(values, points, bi, **kw)
model_mod.functions_module.average_voxelization_3d = spy
with torch.no_grad():
    wl.model.predict(**wl.inputs)
model_mod.functions_module.average_voxelization_3d = real
torch.cuda.synchronize()
for _ in range(int(os.environ.get("REPS", "20"))):
    real(*cap["a"], **cap["kw"])
torch.cuda.synchronize()
print("done")
