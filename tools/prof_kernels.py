"""Profile driver for rocprofv3 (kernel trace / PMC passes): every hand-written kernel of the path at the
shapes the pose network and the refinement drivers give it, REPS launches each.  tools/pmc_summary.py groups
the launches by (kernel, grid size), so one kernel at two shapes gives two rows."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as mf  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models.volumetric_cl import ChannelsLastVolumetric  # noqa: E402

REPS = int(os.environ.get("REPS", "5"))
WHAT = set(os.environ.get("WHAT", "vox,interp,tdf,icp,add,valid,net,icc").split(","))
B, P, D = 8, 1000, 32
Fn = mf.functions
torch.manual_seed(0)
rs = np.random.RandomState(0)
dev = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()  # noqa: E731
fx = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "fixture_pose_refinement_00000002.npz"))
bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(P)

if "vox" in WHAT:   # training path: dense average voxelization fwd + bwd (B=8, P=1000, C=144, 32^3)
    pts = dev(np.clip(rs.normal(16, 4, (B * P, 3)), 0.6, 30.4).astype(np.float32))
    vals = torch.randn(B * P, 144, device="cuda", requires_grad=True)
    for _ in range(REPS):
        y = Fn.average_voxelization_3d(vals, pts, bi, batch_size=B, origin=(0, 0, 0), pitch=1.0, dimensions=(D, D, D))
        y.backward(torch.ones_like(y))
        vals.grad = None
if "interp" in WHAT:  # channels-first sampler fwd + bwd (training), channels-last sampler (inference)
    for C, X in ((256, 16), (512, 8)):
        vox = torch.randn(B, C, X, X, X, device="cuda", requires_grad=True)
        p = torch.rand(B * P, 3, device="cuda") * (X - 1)
        start = torch.arange(B + 1, dtype=torch.int32, device="cuda") * P
        vcl = vox.detach().reshape(B, C, -1).transpose(1, 2).contiguous()
        out = torch.empty(B * P, 984, device="cuda")
        for _ in range(REPS):
            v = Fn.interpolate_voxel_grid(vox, p, bi, channels_first=True, batch_start=start)
            v.backward(torch.ones_like(v))
            vox.grad = None
            mf._lib.check(mf._lib.lib().mf_interpolate_voxel_grid_cl_fwd(
                vcl.data_ptr(), p.data_ptr(), bi.data_ptr(), B * P, B, C, X, X, X, out.data_ptr(), 984,
                mf._lib.stream_ptr()), "cl")
if "tdf" in WHAT:   # truncated distance function fwd + bwd, pseudo occupancy (one 32^3 grid, 3000 points)
    pts = (dev(fx["pcd_cad"].astype(np.float32)[:3000])).requires_grad_(True)
    pitch = float(np.ptp(fx["pcd_cad"], axis=0).max() / 24)
    origin = tuple((fx["pcd_cad"].min(0) - 4 * pitch).tolist())
    for _ in range(REPS):
        m = Fn.truncated_distance_function(pts, pitch=pitch, origin=origin, dims=(D, D, D), truncation=2 * pitch)
        m.sum().backward()
        pts.grad = None
if "icp" in WHAT:   # fused ICP loop on the fixture (k_icp + k_icp_step)
    tgt = dev((np.argwhere(fx["grid_target"] >= 0.5) * fx["pitch"] + fx["origin"]).astype(np.float32))
    src = dev(fx["pcd_cad"].astype(np.float32))
    for _ in range(max(1, REPS // 3)):
        link = mf.contrib.IterativeClosestPointLink(fx["transform_init"]).to_gpu()
        mf.contrib.icp_refine([link], [src], [tgt], n_iter=20)
if "add" in WHAT:   # fused ADD / ADD-S loss fwd + bwd (B=8 objects, P=1000 predicted poses, 500 model points)
    cad = torch.rand(B, 500, 3, device="cuda") * 0.1
    T_true = torch.eye(4, device="cuda").repeat(B, 1, 1)
    q = torch.randn(B * P, 4, device="cuda")
    q = (q / q.norm(dim=1, keepdim=True)).requires_grad_(True)
    t = (torch.randn(B * P, 3, device="cuda") * 0.01).requires_grad_(True)
    sym = torch.tensor([i % 2 == 0 for i in range(B)], device="cuda")
    for _ in range(REPS):
        T_pred = Fn.transformation_matrix(q, t).reshape(B, P, 4, 4)
        add = Fn.average_distance_batch(cad, T_true, T_pred, sym)
        add.sum().backward()
        q.grad = t.grad = None
if "net" in WHAT or "valid" in WHAT:
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    b = mf.synthetic.make_singleview_batch(B, seed=0)
    inp = {k: torch.as_tensor(b[k]).cuda() for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        for _ in range(REPS):
            pix = model._select_points(inp["pcd"])                      # k_valid_order
        if "net" in WHAT:   # the whole channels-last volumetric part at B = 8 and conv4 at B = 1
            values, points = model._backbone_features(inp["rgb"], inp["pcd"], pix)
            args = (inp["class_id"], values, points, inp["pitch"].float(), inp["origin"].float(), inp["grid_nontarget_empty"])
            for _ in range(REPS):
                model._pose_from_features(*args)
            for _ in range(REPS):
                model.predict(**inp)                                    # incl. k_psp_tail, rows hand-over
            vol = model._volumetric_cl
            h3 = torch.relu(torch.randn(1, 16 ** 3, 256, device="cuda"))
            for _ in range(REPS):
                vol.conv_k4s2("conv4", model.conv4, h3, 1, 16, cin=256)
if WHAT & {"conv4", "conv3d", "conv4b1", "heads1"}:   # one shape of the MFMA kernels per process (same grid size otherwise)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    vol = ChannelsLastVolumetric(model)
    with torch.no_grad():
        if "conv4" in WHAT:
            h3 = torch.relu(torch.randn(B, 16 ** 3, 256, device="cuda"))
            for _ in range(REPS):
                vol.conv_k4s2("conv4", model.conv4, h3, B, 16, cin=256)
        if "conv4b1" in WHAT:
            h3 = torch.relu(torch.randn(1, 16 ** 3, 256, device="cuda"))
            for _ in range(REPS):
                vol.conv_k4s2("conv4", model.conv4, h3, 1, 16, cin=256)
        if "conv3d" in WHAT:
            h_occ = torch.relu(torch.randn(B, 32 ** 3, 16, device="cuda"))
            for _ in range(REPS):
                vol.conv_k4s2("conv3_occ", model.conv3, h_occ, B, 32, cin=16, c_off=144, relu=False, bias=False)
        if "heads1" in WHAT:
            feat = torch.zeros(B * P, 992, device="cuda"); feat[:, :984].normal_()
            for _ in range(REPS):
                vol.heads(feat, B, P)
if "icc" in WHAT:   # pose_refinement ICC: 1 scene x 8 objects, 100 iterations
    sc = mf.synthetic.make_icc_scene(8, seed=0, fixtures=[dict(np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", f"fixture_pose_refinement_0000000{i}.npz"))) for i in range(3)])
    d = dict(points=sc["points"], sdf=sc["sdf"], pitch=sc["pitch"], origin=sc["origin"], grid_target=sc["grid_target"],
             grid_nontarget_empty=sc["grid_nontarget_empty"])
    icc = mf.contrib.IccScenes([d], sdf_offset=0.02, device="cuda")
    from morefusion_amd.geometry import quaternion_from_matrix
    q0 = dev(np.stack([quaternion_from_matrix(T) for T in sc["transform_init"]]).astype(np.float32))
    t0 = dev(sc["transform_init"][:, :3, 3].astype(np.float32))
    for _ in range(max(1, REPS // 3)):
        q, t = q0.clone(), t0.clone()
        m, v = torch.zeros(8, 7, device="cuda"), torch.zeros(8, 7, device="cuda")
        icc.refine(q, t, m, v, 100, step0=0, alpha_q=0.01, alpha_t=0.001)
torch.cuda.synchronize()
print("done", sorted(WHAT))
