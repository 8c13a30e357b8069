"""Parity of the HIP ops (called through the C-ABI via morefusion_amd) against the
oracle and the committed golden vectors.  Needs a real MI355X: run with -m gpu."""
import numpy as np
import pytest
import torch

from conftest import golden
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
F = mf.functions


def dev(x, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(x))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda()


def _dense(shape, idx, val, dtype):
    a = np.zeros(int(np.prod(shape)), dtype=dtype)
    a[idx] = val
    return a.reshape(shape)


def test_library_is_loaded_and_native():
    assert mf._lib.lib().mf_version() >= 100


# ---- A1/A2 ------------------------------------------------------------------------
def test_average_voxelization_3d_vs_reference_golden():
    g = golden("ref_average_voxelization_3d.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    values = dev(g["values"]).requires_grad_(True)
    y, counts = F.average_voxelization_3d(
        values, dev(g["points"]), dev(g["batch_indices"]), batch_size=B,
        origin=g["origin"], pitch=float(g["pitch"]), dimensions=(D, D, D), return_counts=True)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    c_ref = _dense((B, D, D, D), g["counts_nonzero_index"], g["counts_nonzero_value"], np.int32)
    assert y.dtype == torch.float32 and counts.dtype == torch.int32
    np.testing.assert_array_equal(counts.cpu().numpy(), c_ref)  # voxel indices bit-exact
    np.testing.assert_array_equal(y.detach().cpu().numpy(), y_ref)  # same sum order: bit-exact
    gy = np.random.RandomState(1).uniform(-1, 1, y.shape).astype(np.float32)
    y.backward(dev(gy))
    np.testing.assert_array_equal(values.grad.cpu().numpy(), g["gvalues"])


def test_average_voxelization_3d_model_shape_collisions_and_out_of_range():
    g = golden("ref_average_voxelization_3d_model.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    y, counts = F.average_voxelization_3d(
        dev(g["values"]), dev(g["points"]), dev(g["batch_indices"]), batch_size=B,
        origin=(0, 0, 0), pitch=1.0, dimensions=(D, D, D), return_counts=True)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    c_ref = _dense((B, D, D, D), g["counts_nonzero_index"], g["counts_nonzero_value"], np.int32)
    np.testing.assert_array_equal(counts.cpu().numpy(), c_ref)
    np.testing.assert_array_equal(y.cpu().numpy(), y_ref)


@pytest.mark.parametrize("dims,C,B,n", [((5, 5, 5), 3, 2, 77), ((7, 6, 5), 1, 1, 300),
                                        ((32, 32, 32), 144, 2, 2000), ((16, 16, 16), 5, 3, 0)])
def test_average_voxelization_3d_vs_oracle_shapes(dims, C, B, n):
    rs = np.random.RandomState(n + C)
    pts = rs.uniform(-1.5, max(dims) + 0.5, (n, 3)).astype(np.float32)
    if n:
        pts[: n // 4] = np.floor(pts[: n // 4]) + 0.5  # exact .5 ties -> half-away rule
    vals = rs.uniform(-1, 1, (n, C)).astype(np.float32)
    bi = rs.randint(0, B, n).astype(np.int32)
    kw = dict(batch_size=B, origin=(0.0, 0.0, 0.0), pitch=1.0, dimensions=dims)
    y, c = F.average_voxelization_3d(dev(vals), dev(pts), dev(bi), return_counts=True, **kw)
    y_o, c_o = O.average_voxelization_3d(vals, pts, bi, mode="gpu", **kw)
    np.testing.assert_array_equal(c.cpu().numpy(), c_o)
    np.testing.assert_array_equal(y.cpu().numpy(), y_o)


def test_average_voxelization_3d_errors():
    v, p, b = dev(np.zeros((4, 2), np.float32)), dev(np.zeros((4, 3), np.float32)), dev(np.zeros(4, np.int32))
    kw = dict(batch_size=1, origin=(0, 0, 0), pitch=1.0)
    with pytest.raises(ValueError, match="dimensions must be a tuple of 4 integers"):
        F.average_voxelization_3d(v, p, b, dimensions=[4, 4, 4], **kw)
    pn = p.clone()
    pn[1, 1] = float("nan")
    with pytest.raises(ValueError, match="points include nan"):
        F.average_voxelization_3d(v, pn, b, dimensions=(4, 4, 4), **kw)
    with pytest.raises(TypeError):
        F.average_voxelization_3d(v, p, b.long(), dimensions=(4, 4, 4), **kw)
    # CPU tensors take the product's CPU path (round 5, like the reference's get_array_module): same values as the GPU
    # path here (no coordinate sits on a rounding boundary: the forks differ only at exact .5)
    y_cpu = F.average_voxelization_3d(v.cpu(), p.cpu(), b.cpu(), dimensions=(4, 4, 4), **kw)
    assert not y_cpu.is_cuda
    np.testing.assert_allclose(y_cpu.numpy(), F.average_voxelization_3d(v, p, b, dimensions=(4, 4, 4), **kw).cpu().numpy(),
                               rtol=1e-6, atol=1e-7)
    with pytest.raises(RuntimeError, match="no CPU fallback"):  # mixed devices are still an error
        F.average_voxelization_3d(v, p.cpu(), b, dimensions=(4, 4, 4), **kw)


# ---- A3 ------------------------------------------------------------------------------
def test_max_voxelization_3d_vs_reference_golden():
    g = golden("ref_max_voxelization_3d.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    values = dev(g["values"]).requires_grad_(True)
    y, ind = F.max_voxelization_3d(
        values, dev(g["points"]), dev(g["batch_indices"]), dev(g["intensities"]), batch_size=B,
        origin=g["origin"], pitch=float(g["pitch"]), dimensions=(D, D, D), return_indices=True)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    i_ref = np.full(B * D ** 3, -1, np.int32)
    i_ref[g["indices_valid_index"]] = g["indices_valid_value"]
    np.testing.assert_array_equal(ind.cpu().numpy().reshape(-1), i_ref)
    np.testing.assert_array_equal(y.detach().cpu().numpy(), y_ref)
    gy = np.random.RandomState(1).uniform(-1, 1, y.shape).astype(np.float32)
    y.backward(dev(gy))
    np.testing.assert_array_equal(values.grad.cpu().numpy(), g["gvalues"])


def test_max_voxelization_3d_ties_lowest_index():
    pts = np.zeros((6, 3), np.float32) + 1.2
    vals = np.arange(12, dtype=np.float32).reshape(6, 2)
    inten = np.array([1, 3, 3, 2, 3, -1], np.float32)
    bi = np.zeros(6, np.int32)
    kw = dict(batch_size=1, origin=(0, 0, 0), pitch=1.0, dimensions=(4, 4, 4))
    y, ind = F.max_voxelization_3d(dev(vals), dev(pts), dev(bi), dev(inten), return_indices=True, **kw)
    y_o, ind_o = O.max_voxelization_3d(vals, pts, bi, inten, **kw)
    assert ind_o[0, 1, 1, 1] == 1
    np.testing.assert_array_equal(ind.cpu().numpy(), ind_o)
    np.testing.assert_array_equal(y.cpu().numpy(), y_o)


# ---- A4 ------------------------------------------------------------------------------
def test_interpolate_voxel_grid_vs_reference_golden():
    g = golden("ref_interpolate_voxel_grid.npz")
    vox = np.random.RandomState(int(g["vox_seed"])).uniform(-1, 1, tuple(g["vox_shape"])).astype(np.float32)
    v = F.interpolate_voxel_grid(dev(vox), dev(g["points"]), dev(g["batch_indices"])).cpu().numpy()
    # bit-exact against the gpu-fork oracle; within 1e-6 of the reference's CPU twin for
    # non-negative coordinates (the twin builds its weights in float64)
    np.testing.assert_array_equal(v, O.interpolate_voxel_grid(vox, g["points"], g["batch_indices"], mode="gpu"))
    nonneg = (g["points"] >= 0).all(axis=1)
    np.testing.assert_allclose(v[nonneg], g["values"][nonneg], rtol=0, atol=1e-6)


@pytest.mark.parametrize("shape,n", [((2, 256, 16, 16, 16), 2000), ((2, 512, 8, 8, 8), 2000),
                                     ((1, 3, 5, 6, 7), 50), ((1, 2, 40, 40, 40), 100)])
@pytest.mark.parametrize("channels_first", [False, True])
def test_interpolate_voxel_grid_fwd_bwd_vs_oracle(shape, n, channels_first):
    rs = np.random.RandomState(sum(shape) + n)
    B = shape[0]
    vox = rs.uniform(-1, 1, shape).astype(np.float32)
    pts = (rs.uniform(-1, 1, (n, 3)) * 0.6 + 0.5).astype(np.float32) * np.array(shape[2:], np.float32)
    bi = rs.randint(0, B, n).astype(np.int32)
    vt = dev(vox).requires_grad_(True)
    out = F.interpolate_voxel_grid(vt, dev(pts), dev(bi), channels_first=channels_first)
    ref = O.interpolate_voxel_grid(vox, pts, bi, mode="gpu")
    got = out.detach().cpu().numpy()
    np.testing.assert_array_equal(got.T if channels_first else got, ref)
    gy = rs.uniform(-1, 1, ref.shape).astype(np.float32)
    out.backward(dev(gy.T.copy() if channels_first else gy))
    gref = O.interpolate_voxel_grid_backward(gy, pts, bi, shape, mode="gpu")
    np.testing.assert_allclose(vt.grad.cpu().numpy(), gref, rtol=1e-5, atol=1e-5)  # float atomics order


@pytest.mark.parametrize("channels_first", [False, True])
def test_interpolate_voxel_grid_row_ranges_and_orphan_rows(channels_first):
    """``batch_start`` (rows of item b = [start[b], start[b+1])): same bits as the batch_indices
    scan, forward and backward, with ragged items and one empty item; rows whose batch index is
    outside [0, B) -- or outside every range -- come back as zeros on both paths."""
    rs = np.random.RandomState(5)
    shape = (4, 256, 16, 16, 16)
    counts = [700, 0, 1300, 500]
    n_orphan = 37
    n = sum(counts) + n_orphan
    bi = np.concatenate([np.full(c, b, np.int32) for b, c in enumerate(counts)] + [np.full(n_orphan, 9, np.int32)])
    start = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    vox = rs.uniform(-1, 1, shape).astype(np.float32)
    pts = (rs.uniform(-0.1, 1.1, (n, 3)) * 16).astype(np.float32)
    gy = rs.uniform(-1, 1, (n, shape[1])).astype(np.float32)
    outs, grads = [], []
    for bs in (None, dev(start)):
        vt = dev(vox).requires_grad_(True)
        out = F.interpolate_voxel_grid(vt, dev(pts), dev(bi), channels_first=channels_first, batch_start=bs)
        out.backward(dev(gy.T.copy() if channels_first else gy))
        got = out.detach().cpu().numpy()
        outs.append(got.T if channels_first else got)
        grads.append(vt.grad.cpu().numpy())
    valid = bi < 4
    ref = O.interpolate_voxel_grid(vox, pts[valid], bi[valid], mode="gpu")
    for got in outs:
        np.testing.assert_array_equal(got[valid], ref)
        np.testing.assert_array_equal(got[~valid], 0.0)
    gref = O.interpolate_voxel_grid_backward(gy[valid], pts[valid], bi[valid], shape, mode="gpu")
    for g in grads:
        np.testing.assert_allclose(g, gref, rtol=1e-5, atol=1e-5)
    with pytest.raises(TypeError):
        F.interpolate_voxel_grid(dev(vox), dev(pts), dev(bi), batch_start=dev(start[:-1]))


# ---- A5 ------------------------------------------------------------------------------
def test_occupancy_grid_3d_known_answer_and_config1():
    g = golden("ref_occupancy_grid_3d.npz")
    m = F.occupancy_grid_3d(dev(g["known_points"]), pitch=1, origin=(0, 0, 0), dims=(5, 5, 5))
    nonzero = [[0, 0, 0], [0, 1, 0], [0, 0, 1], [4, 3, 4], [3, 4, 4], [4, 4, 4]]
    expect = np.zeros((5, 5, 5), bool)
    expect[tuple(zip(*nonzero))] = True
    np.testing.assert_array_equal(m.cpu().numpy() > 0, expect)
    np.testing.assert_array_equal(m.cpu().numpy(), g["known_grid"])
    p = float(g["c1_pitch"])
    m1 = F.occupancy_grid_3d(dev(g["c1_points"]), pitch=p, origin=(-16 * p,) * 3, dims=(32,) * 3)
    np.testing.assert_array_equal(m1.cpu().numpy(), g["c1_grid"])  # BASELINE config 1, bit-exact
    m2 = F.occupancy_grid_3d(dev(g["c1_points"][:200]), pitch=p, origin=(-16 * p,) * 3,
                             dims=(32,) * 3, threshold=2)
    np.testing.assert_array_equal(m2.cpu().numpy(), g["c1_grid_thr2"])


def test_occupancy_grid_3d_backward_vs_oracle():
    rs = np.random.RandomState(0)
    pts = rs.uniform(0.5, 5.5, (40, 3)).astype(np.float32)
    gm = rs.uniform(-1, 1, (8, 7, 6)).astype(np.float32)
    kw = dict(pitch=0.9, origin=(0.1, 0.0, -0.1), dims=(8, 7, 6), threshold=1.5)
    pt = dev(pts).requires_grad_(True)
    F.occupancy_grid_3d(pt, **kw).backward(dev(gm))
    ref = O.occupancy_grid_3d_backward(gm, pts, **kw)
    np.testing.assert_allclose(pt.grad.cpu().numpy(), ref, rtol=1e-4, atol=1e-5)


# ---- A6/A7 ---------------------------------------------------------------------------
@pytest.mark.parametrize("case", ["selftest", "fixture", "big_ksize", "noncubic"])
def test_truncated_distance_function_vs_oracle(case, fixtures3):
    if case == "selftest":  # truncated_distance_function.py:216-230 (__main__ self check)
        pts = np.array([[0.5, 0.5, 0.5], [1.48, 1.48, 1.48]], np.float32)
        kw = dict(pitch=0.5, origin=(0, 0, 0), dims=(5, 5, 5), truncation=1.2)
    elif case == "fixture":
        f = fixtures3[1]
        T = f["transform_init"]
        pts = (f["pcd_cad"] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
        kw = dict(pitch=float(np.float32(f["pitch"])), origin=tuple(f["origin"]), dims=(32,) * 3,
                  truncation=float(np.float32(2) * np.float32(f["pitch"])))
    elif case == "big_ksize":
        pts = np.random.RandomState(3).uniform(-1, 9, (500, 3)).astype(np.float32)
        kw = dict(pitch=0.5, origin=(0, 0, 0), dims=(16, 16, 16), truncation=2.3)  # ksize 5
    else:
        pts = np.random.RandomState(4).uniform(-1, 6, (800, 3)).astype(np.float32)
        kw = dict(pitch=0.25, origin=(0.1, 0.2, 0.3), dims=(20, 9, 13), truncation=0.5)
    pt = dev(pts).requires_grad_(True)
    tdf, idx = F.truncated_distance_function(pt, return_indices=True, **kw)
    tdf_o, flat_o, ks = O.truncated_distance_function(pts, **kw)
    np.testing.assert_array_equal(tdf.detach().cpu().numpy(), tdf_o)
    np.testing.assert_array_equal(idx.cpu().numpy(), np.where(flat_o >= 0, flat_o // ks ** 3, -1))
    gm = np.random.RandomState(5).uniform(-1, 1, tdf_o.shape).astype(np.float32)
    tdf.backward(dev(gm))
    gp = O.truncated_distance_function_backward(gm, pts, flat_o, ks, pitch=kw["pitch"], origin=kw["origin"])
    np.testing.assert_allclose(pt.grad.cpu().numpy(), gp, rtol=1e-5, atol=1e-6)


def test_pseudo_occupancy_voxelization_vs_oracle(fixtures3):
    f = fixtures3[0]
    T = f["transform_init"]
    pts = (f["pcd_cad"] @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
    sdf = mf.synthetic.synthetic_sdf(f["pcd_cad"])
    kw = dict(pitch=float(np.float32(f["pitch"])), origin=tuple(f["origin"]), dims=(32,) * 3,
              threshold=2, sdf_offset=0.02)
    outs = F.pseudo_occupancy_voxelization(dev(pts), dev(sdf), **kw)
    refs = O.pseudo_occupancy_voxelization(pts, sdf, **kw)
    for a, b in zip(outs, refs):
        np.testing.assert_array_equal(a.cpu().numpy(), b)
    # all-outside sdf -> max weight 0 -> NaN grids, like the reference (0/0)
    outs = F.pseudo_occupancy_voxelization(dev(pts), dev(-np.ones_like(sdf)), **{**kw, "sdf_offset": 0})
    assert torch.isnan(outs[2]).all()


# ---- A8 (torch ops, device-agnostic) -----------------------------------------------
def test_transforms_vs_reference_golden():
    g = golden("ref_transforms.npz")
    q, t = dev(g["q"]), dev(g["t"])
    np.testing.assert_allclose(F.quaternion_matrix(q).cpu().numpy(), g["quaternion_matrix"], atol=1e-6)
    np.testing.assert_allclose(F.transformation_matrix(q, t).cpu().numpy(), g["transformation_matrix"], atol=1e-6)
    np.testing.assert_allclose(F.transform_points(dev(g["points"]), dev(g["transformation_matrix"])).cpu().numpy(),
                               g["transform_points"], atol=1e-6)


# ---- A11 / A12 -----------------------------------------------------------------------
@pytest.mark.parametrize("R,Q", [(500, 50000), (1, 10), (1500, 3000), (37, 1)])
def test_nn_vs_bruteforce(R, Q):
    rs = np.random.RandomState(R + Q)
    ref = rs.uniform(size=(R, 3)).astype(np.float32)
    query = rs.uniform(size=(Q, 3)).astype(np.float32)
    if R > 2:
        ref[2] = ref[1]  # exact tie -> lowest index
    idx = mf.geometry.nn(dev(ref), dev(query))
    assert idx.dtype == torch.int64
    np.testing.assert_array_equal(idx.cpu().numpy(), O.nn(ref, query))


def test_average_distance_add_and_add_s():
    g = golden("ref_average_distance.npz")
    pts, Tt, Tp = dev(g["points"]), dev(g["transform_true"]), dev(g["transforms_pred"])
    add = F.average_distance(pts, Tt, Tp)
    np.testing.assert_allclose(add.cpu().numpy(), g["add"], rtol=1e-6, atol=1e-7)
    add_s = F.average_distance(pts, Tt, Tp, symmetric=True)
    ref = O.average_distance(g["points"], g["transform_true"], g["transforms_pred"], symmetric=True)
    np.testing.assert_allclose(add_s.cpu().numpy(), ref, rtol=1e-6, atol=1e-7)
    assert (add_s <= add + 1e-7).all()


# ---- A13/A14 network ------------------------------------------------------------------
def test_model_predict_shapes_and_3d_part_vs_oracle():
    from morefusion_amd.contrib.singleview_3d.models import Model
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    b = mf.synthetic.make_singleview_batch(2, seed=3)
    inputs = {k: torch.as_tensor(b[k]).cuda() for k in
              ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        rot, trans, conf = model.predict(**inputs)
        assert rot.shape == (2, 1000, 4) and trans.shape == (2, 1000, 3) and conf.shape == (2, 1000)
        # chainer's F.normalize is x / (|x| + 1e-5): unit up to 1e-5 / |x| (random-init heads: |x| ~ 0.05)
        np.testing.assert_allclose(rot.norm(dim=2).cpu().numpy(), 1.0, atol=2e-3)
        assert ((conf > 0) & (conf < 1)).all()
        # origin=None path: median - 15.5*pitch (model.py:202-205) == the synthetic origin
        r2, t2, c2 = model.predict(**{**inputs, "origin": None})
        np.testing.assert_allclose(t2.cpu().numpy(), trans.cpu().numpy(), atol=1e-4)
        # the voxel ops inside _extract against the oracle, on the network's own tensors
        P = 1000
        values = torch.randn(2, 32, P, device="cuda")
        points = torch.rand(2, 3, P, device="cuda") * 24 + 4
        feat = model._extract(values, points, inputs["grid_nontarget_empty"])
        assert feat.shape == (2, 72 + 144 + 256 + 512, P)
        feat2 = torch.cat((F_relu(model.conv2_rgb(F_relu(model.conv1_rgb(values)))),
                           F_relu(model.conv2_pcd(F_relu(model.conv1_pcd(15.5 - points))))), 1)
        vox = model._voxelize(feat2.transpose(1, 2).contiguous(), points.transpose(1, 2))
        bi = np.arange(2, dtype=np.int32).repeat(P)
        vox_o, _ = O.average_voxelization_3d(
            feat2.transpose(1, 2).reshape(2 * P, -1).cpu().numpy(),
            points.transpose(1, 2).reshape(2 * P, 3).cpu().numpy(), bi, batch_size=2,
            origin=(0, 0, 0), pitch=1.0, dimensions=(32, 32, 32))
        np.testing.assert_array_equal(vox.cpu().numpy(), vox_o)


def F_relu(x):
    return torch.nn.functional.relu(x)


def test_model_training_step_backward_through_hip_ops():
    """Model.forward (= predict + ADD/ADD-S confidence loss, model.py:277-481) and its
    backward: gradients flow through average_voxelization_3d / interpolate_voxel_grid
    (HIP backward kernels) into the 2-D backbone; one SGD step lowers the loss."""
    from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels
    torch.manual_seed(0)
    np.random.seed(0)
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (800, 3)).astype(np.float32) for c in mf.synthetic.CLASS_PITCH}
    model = Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).cuda().train()
    b = mf.synthetic.make_singleview_batch(2, seed=20)  # classes include a symmetric one -> ADD-S
    inputs = {k: torch.as_tensor(b[k]).cuda() for k in
              ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty",
               "quaternion_true", "translation_true")}
    opt = torch.optim.SGD(model.parameters(), lr=1e-5)
    losses = []
    for _ in range(3):
        np.random.seed(1)  # same point subsample / CAD subsample every step
        torch.manual_seed(1)  # same dropout mask
        opt.zero_grad()
        loss = model(**inputs)
        loss.backward()
        losses.append(float(loss.detach()))
        opt.step()
    assert np.isfinite(losses).all()
    for name in ("conv1_rgb.weight", "conv3.weight", "conv4.weight", "resnet_extractor.conv1.weight",
                 "conv4_rot.weight", "conv1_occ.weight"):
        g = dict(model.named_parameters())[name].grad
        assert g is not None and torch.isfinite(g).all() and float(g.abs().sum()) > 0, name
    assert losses[-1] < losses[0], losses
    ev = model.evaluate(class_id=inputs["class_id"], quaternion_true=inputs["quaternion_true"],
                        translation_true=inputs["translation_true"],
                        quaternion_pred=inputs["quaternion_true"].float(),
                        translation_pred=inputs["translation_true"].float())
    assert ev["add"] < 1e-6 and ev["add_s"] < 1e-6  # identical poses -> zero ADD / ADD-S


# ---- edge cases: empty / ragged / degenerate inputs ---------------------------------
def test_empty_and_degenerate_inputs():
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device="cuda")  # noqa: E731
    # no points at all: dense zero output, zero counts
    y, c = F.average_voxelization_3d(z(0, 3), z(0, 3), z(0, dt=torch.int32), batch_size=2, origin=(0, 0, 0),
                                     pitch=1.0, dimensions=(4, 4, 4), return_counts=True)
    assert y.shape == (2, 3, 4, 4, 4) and float(y.abs().sum()) == 0 and int(c.sum()) == 0
    # every point outside the grid / invalid batch index
    pts = torch.tensor([[9.0, 9, 9], [-5, 0, 0], [1, 1, 1]], device="cuda")
    bi = torch.tensor([0, 0, 7], dtype=torch.int32, device="cuda")
    y, c = F.average_voxelization_3d(torch.ones(3, 2, device="cuda"), pts, bi, batch_size=1, origin=(0, 0, 0),
                                     pitch=1.0, dimensions=(4, 4, 4), return_counts=True)
    assert float(y.abs().sum()) == 0 and int(c.sum()) == 0
    # interpolation with zero points; TDF with zero points -> all truncation, no winners
    v = F.interpolate_voxel_grid(torch.rand(1, 2, 4, 4, 4, device="cuda"), z(0, 3), z(0, dt=torch.int32))
    assert v.shape == (0, 2)
    tdf, idx = F.truncated_distance_function(z(0, 3), pitch=0.5, origin=(0, 0, 0), dims=(4, 4, 4),
                                             truncation=1.0, return_indices=True)
    assert (tdf == 1.0).all() and (idx == -1).all()
    # a pile-up of > 64 points in one voxel takes the chain fallback path, still exact
    n = 200
    pts = torch.full((n, 3), 1.2, device="cuda")
    vals = torch.arange(n * 2, dtype=torch.float32, device="cuda").reshape(n, 2) * 0.37
    y, c = F.average_voxelization_3d(vals, pts, z(n, dt=torch.int32), batch_size=1, origin=(0, 0, 0),
                                     pitch=1.0, dimensions=(4, 4, 4), return_counts=True)
    y_o, c_o = O.average_voxelization_3d(vals.cpu().numpy(), pts.cpu().numpy(), np.zeros(n, np.int32), batch_size=1,
                                         origin=(0, 0, 0), pitch=1.0, dimensions=(4, 4, 4))
    assert int(c[0, 1, 1, 1]) == n
    np.testing.assert_array_equal(y.cpu().numpy(), y_o)
    # nn with a single reference point
    idx = mf.geometry.nn(torch.rand(1, 3, device="cuda"), torch.rand(5, 3, device="cuda"))
    assert (idx == 0).all()


# ---- A13 sparse conv3 (fp32 MFMA) --------------------------------------------------------
@pytest.mark.parametrize("B,Cs,Cd,Cout,n", [(2, 144, 16, 256, 1000), (1, 8, 0, 64, 40), (3, 12, 4, 128, 300)])
def test_sparse_conv3_matches_dense_conv3d(B, Cs, Cd, Cout, n):
    from morefusion_amd.contrib.singleview_3d.models.sparse_conv import SparseVoxelConv3d
    torch.manual_seed(B + Cs)
    D = 32
    conv = torch.nn.Conv3d(Cs + Cd, Cout, 4, 2, padding=1).cuda()
    pts = torch.rand(B * n, 3, device="cuda") * 36 - 2  # some outside, borders included
    pts[:10] = torch.tensor([0.2, 31.4, 15.0], device="cuda")  # corner/border pile-up
    vals = torch.randn(B * n, Cs, device="cuda")
    bi = torch.arange(B, dtype=torch.int32, device="cuda").repeat_interleave(n)
    vox, counts = F.average_voxelization_3d(vals, pts, bi, batch_size=B, origin=(0, 0, 0), pitch=1.0,
                                            dimensions=(D, D, D), return_counts=True)
    h_occ = torch.randn(B, Cd, D, D, D, device="cuda") if Cd else None
    op = SparseVoxelConv3d(conv)
    got = op(vox, counts, h_occ, max_rows=B * n)
    with torch.no_grad():
        full = torch.cat([vox, h_occ], 1) if Cd else vox
        ref = torch.relu(conv(full))
    assert got.shape == ref.shape == (B, Cout, 16, 16, 16)
    np.testing.assert_allclose(got.cpu().numpy(), ref.cpu().numpy(), rtol=2e-4, atol=2e-4)
    got2 = op(vox, counts, h_occ, max_rows=B * n)  # deterministic: bitwise equal run to run
    assert torch.equal(got, got2)
    pre = op(vox, counts, h_occ, max_rows=B * n, relu=False)
    assert float(pre.min()) < 0  # relu flag honoured
    # fed by the points (no dense voxelized tensor): the same bits
    got3 = op.from_points(vals, pts, bi, batch_size=B, h_dense=h_occ, dim=D)
    assert torch.equal(got3, got)
    assert torch.equal(op.from_points(vals, pts, bi, batch_size=B, h_dense=h_occ, dim=D, relu=False), pre)


def test_model_predict_under_bf16_autocast_matches_fp32_roughly():
    """bf16 autocast for the stock convolutions (BASELINE config 5's precision); the HIP ops
    keep float32 and must be handed float32 tensors whatever the autocast state."""
    from morefusion_amd.contrib.singleview_3d.models import Model
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    b = mf.synthetic.make_singleview_batch(2, seed=4)
    inputs = {k: torch.as_tensor(b[k]).cuda() for k in
              ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        rot, trans, conf = model.predict(**inputs)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            rot_b, trans_b, conf_b = model.predict(**inputs)
    for x in (rot_b, trans_b, conf_b):
        assert torch.isfinite(x.float()).all()
    assert float((trans_b.float() - trans).abs().max()) < 0.05  # metres; same pose field
    assert float((conf_b.float() - conf).abs().max()) < 0.1
