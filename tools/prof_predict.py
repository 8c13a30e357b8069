"""Profile driver for rocprofv3: REPS x Model.predict at batch B (env B, default 8), eager or through the
hipGraph path (GRAPH=1), after a warm-up that is bracketed out with marker launches (k_to_channels_last on a
tiny buffer: kernel_stats.py keeps only what lies between the 2nd and 3rd marker, MF_MARK=k_to_channels_last).
PART=volumetric profiles Model._pose_from_features alone (features of one predict call kept fixed)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
from morefusion_amd import _lib  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402

B = int(os.environ.get("B", "8"))
reps = int(os.environ.get("REPS", "5"))
graph = os.environ.get("GRAPH") == "1"
part = os.environ.get("PART", "predict")
torch.backends.cudnn.benchmark = True
torch.manual_seed(0)
model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
if os.environ.get("CF") == "1":
    model.channels_last_3d = False
b = mf.synthetic.make_singleview_batch(B, seed=0)
inp = {k: torch.as_tensor(b[k]).cuda() for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
mk_src = torch.zeros(64, device="cuda")
mk_dst = torch.zeros(64, device="cuda")


def marker():
    _lib.check(_lib.lib().mf_to_channels_last(mk_src.data_ptr(), mk_dst.data_ptr(), 1, 8, 8, _lib.stream_ptr()), "marker")


with torch.no_grad():
    if part == "volumetric":
        pix = model._select_points(inp["pcd"])
        values, points = model._backbone_features(inp["rgb"], inp["pcd"], pix)
        args = (inp["class_id"], values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
        fn = lambda: model._pose_from_features(*args)  # noqa: E731
    elif graph:
        fn = lambda: model.predict_graphed(**inp, clone=False)  # noqa: E731
    else:
        fn = lambda: model.predict(**inp)  # noqa: E731
    marker()
    for _ in range(4):
        fn()
    torch.cuda.synchronize()
    marker()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    marker()
    torch.cuda.synchronize()
print("done", B, reps, graph, part)
