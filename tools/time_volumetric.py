"""Stage timings of the volumetric part of the pose network on one MI355X: the channels-last hand-written
path (round 3) beside round 2's channels-first path.  Prints a JSON record (also written to $OUT if set)."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
from morefusion_amd import _lib  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models.volumetric_cl import ChannelsLastVolumetric  # noqa: E402


def t_ms(fn, reps=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(reps):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / reps


def main():
    torch.backends.cudnn.benchmark = True
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    vol = ChannelsLastVolumetric(model)
    rec = {}
    L = _lib.lib()
    with torch.no_grad():
        for B in (8, 1):
            r = {}
            h3 = torch.relu(torch.randn(B, 16 ** 3, 256, device="cuda"))
            flop = 2.0 * B * 512 * 512 * 64 * 256
            for split in sorted({1, 2, 4, 8, 16, 32, 64}):
                ms = t_ms(lambda: vol.conv_k4s2("conv4", model.conv4, h3, B, 16, cin=256, split=split))
                r[f"conv4_split{split}_ms"] = round(ms, 4)
                r[f"conv4_split{split}_tflops"] = round(flop / ms / 1e9, 1)
            r["conv4_default_split"] = L.mf_conv3d_k4s2_default_split(B, 256, 512, 16)
            h3cf = h3.transpose(1, 2).reshape(B, 256, 16, 16, 16).contiguous()
            ms = t_ms(lambda: F.relu(model.conv4(h3cf)))
            r["conv4_miopen_ms"] = round(ms, 4)
            r["conv4_miopen_tflops"] = round(flop / ms / 1e9, 1)
            b = mf.synthetic.make_singleview_batch(B, seed=1)
            grid = torch.as_tensor(b["grid_nontarget_empty"]).cuda()
            r["occ_convs_ms"] = round(t_ms(lambda: vol.occupancy(grid)), 4)
            g = grid.float()[:, None]
            r["occ_convs_miopen_ms"] = round(t_ms(lambda: F.relu(model.conv2_occ(F.relu(model.conv1_occ(g))))), 4)
            h_occ = vol.occupancy(grid)
            flop3 = 2.0 * B * 4096 * 256 * 64 * 16
            for split in (1, 2, 4):
                ms = t_ms(lambda: vol.conv_k4s2("conv3_occ", model.conv3, h_occ, B, 32, cin=16, c_off=144, relu=False,
                                                bias=False, split=split))
                r[f"conv3_dense_split{split}_ms"] = round(ms, 4)
                r[f"conv3_dense_split{split}_tflops"] = round(flop3 / ms / 1e9, 1)
            h_occ_cf = h_occ.transpose(1, 2).reshape(B, 16, 32, 32, 32).contiguous()
            wd = model.conv3.weight[:, 144:].contiguous()
            r["conv3_dense_miopen_ms"] = round(t_ms(lambda: F.conv3d(h_occ_cf, wd, None, stride=2, padding=1)), 4)

            inp = {k: torch.as_tensor(b[k]).cuda() for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
            pix = model._select_points(inp["pcd"])
            values, points = model._backbone_features(inp["rgb"], inp["pcd"], pix)
            args = (inp["class_id"], values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
            pv = (points - inp["origin"].float()[:, :, None]) / inp["pitch"].float()[:, None, None]
            P = values.shape[2]
            feat, _ = vol.features(values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
            pts = pv.transpose(1, 2).reshape(B * P, 3).contiguous()
            bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(P)
            dense = vol.conv_k4s2("conv3_occ", model.conv3, h_occ, B, 32, cin=16, c_off=144, relu=False, bias=False)
            r["sparse_conv3_cl_ms"] = round(t_ms(lambda: vol._sparse.from_points_cl(feat[:, 72:216], feat.stride(0), pts, bi, B, dense, 32)), 4)
            f2 = feat[:, 72:216].contiguous()
            dense_cf = dense.transpose(1, 2).reshape(B, 256, 16, 16, 16).contiguous()
            # channels-first kernel with a precomputed dense addend (h_dense=None path adds nothing; time the kernels only)
            r["sparse_conv3_cf_ms"] = round(t_ms(lambda: vol._sparse.from_points(f2, pts, bi, batch_size=B, h_dense=None, dim=32)), 4)
            h3r = vol._sparse.from_points_cl(feat[:, 72:216], feat.stride(0), pts, bi, B, dense, 32)
            h4r = vol.conv_k4s2("conv4", model.conv4, h3r, B, 16, cin=256)
            r["sample_feat3_cl_ms"] = round(t_ms(lambda: vol.sample(h3r, 16, pts / 2.0, bi, feat[:, 216:472], feat.stride(0))), 4)
            r["sample_feat4_cl_ms"] = round(t_ms(lambda: vol.sample(h4r, 8, pts / 4.0, bi, feat[:, 472:984], feat.stride(0))), 4)
            r["heads_ms"] = round(t_ms(lambda: vol.heads(feat, B, P)), 4)
            r["features_ms"] = round(t_ms(lambda: vol.features(values, points, inp["pitch"].float(), inp["origin"].float(),
                                                               inp["grid_nontarget_empty"])), 4)
            model.channels_last_3d = True
            r["pose_from_features_cl_ms"] = round(t_ms(lambda: model._pose_from_features(*args)), 4)
            model.channels_last_3d = False
            r["pose_from_features_cf_ms"] = round(t_ms(lambda: model._pose_from_features(*args)), 4)
            model.channels_last_3d = True
            r["backbone_features_ms"] = round(t_ms(lambda: model._backbone_features(inp["rgb"], inp["pcd"], pix)), 4)
            r["predict_cl_ms"] = round(t_ms(lambda: model.predict(**inp)), 4)
            model.channels_last_3d = False
            r["predict_cf_ms"] = round(t_ms(lambda: model.predict(**inp)), 4)
            model.channels_last_3d = True
            rec[f"B{B}"] = r
    s = json.dumps(rec, indent=1)
    print(s)
    if os.environ.get("OUT"):
        open(os.environ["OUT"], "w").write(s)


if __name__ == "__main__":
    main()
