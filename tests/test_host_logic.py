"""Host-side logic of the path on CPU: rigid-transform ops (device-agnostic torch),
chainer-style Adam, quaternion_from_matrix, metrics, pre-processing helpers, synthetic
inputs, point selection of Model.predict, shard arithmetic."""
import os

import numpy as np
import pytest
import torch

import morefusion_amd as mf
from conftest import golden
from oracle import oracle_np as O

F = mf.functions


def test_transform_functions_match_reference_golden_and_squeeze_rules():
    g = golden("ref_transforms.npz")
    q, t = torch.from_numpy(g["q"]), torch.from_numpy(g["t"])
    np.testing.assert_allclose(F.quaternion_matrix(q).numpy(), g["quaternion_matrix"], atol=1e-6)
    np.testing.assert_allclose(F.quaternion_matrix(q[0]).numpy(), g["quaternion_matrix_1d"], atol=1e-6)
    np.testing.assert_allclose(F.transformation_matrix(q, t).numpy(), g["transformation_matrix"], atol=1e-6)
    np.testing.assert_allclose(F.transformation_matrix(q[0], t[0]).numpy(), g["transformation_matrix_1d"], atol=1e-6)
    np.testing.assert_array_equal(F.translation_matrix(t).numpy(), g["translation_matrix"])
    assert F.translation_matrix(t[0]).shape == (4, 4)
    np.testing.assert_array_equal(
        F.compose_transform(torch.from_numpy(g["quaternion_matrix"][:, :3, :3]), t).numpy(), g["compose_transform"])
    pts = torch.from_numpy(g["points"])
    T = torch.from_numpy(g["transformation_matrix"])
    np.testing.assert_allclose(F.transform_points(pts, T).numpy(), g["transform_points"], atol=1e-6)
    np.testing.assert_allclose(F.transform_points(pts, T[0]).numpy(), g["transform_points_1"], atol=1e-6)
    # bit-equal to the oracle's fixed evaluation order (what the HIP kernels use)
    np.testing.assert_array_equal(F.transform_points(pts, T).numpy(), O.transform_points(g["points"], g["transformation_matrix"]))


def test_quaternion_matrix_autograd_equals_reference_backward_rule():
    g = golden("ref_transforms.npz")
    q = torch.from_numpy(g["q"]).double().requires_grad_(True)
    gR = torch.from_numpy(g["gR"]).double()
    (F.quaternion_matrix(q) * gR).sum().backward()
    np.testing.assert_allclose(q.grad.numpy(), O.quaternion_matrix_backward(g["q"].astype(np.float64), g["gR"].astype(np.float64)),
                               rtol=1e-9, atol=1e-12)
    # the hand-written dR/dQ of the reference (quaternion_matrix.py:36-51), via its golden
    qs = g["q"].astype(np.float64)
    qs = qs * np.sqrt(2.0 / (qs ** 2).sum(1, keepdims=True))
    Q = torch.from_numpy(qs[:, :, None] * qs[:, None, :]).requires_grad_(True)
    R = torch.eye(4, dtype=torch.float64).repeat(5, 1, 1).clone()
    R[:, 0, 0] = 1 - Q[:, 2, 2] - Q[:, 3, 3]; R[:, 0, 1] = Q[:, 1, 2] - Q[:, 3, 0]; R[:, 0, 2] = Q[:, 1, 3] + Q[:, 2, 0]
    R[:, 1, 0] = Q[:, 1, 2] + Q[:, 3, 0]; R[:, 1, 1] = 1 - Q[:, 1, 1] - Q[:, 3, 3]; R[:, 1, 2] = Q[:, 2, 3] - Q[:, 1, 0]
    R[:, 2, 0] = Q[:, 1, 3] - Q[:, 2, 0]; R[:, 2, 1] = Q[:, 2, 3] + Q[:, 1, 0]; R[:, 2, 2] = 1 - Q[:, 1, 1] - Q[:, 2, 2]
    (R * gR).sum().backward()
    np.testing.assert_allclose(Q.grad.numpy(), g["gq_outer"], rtol=1e-6, atol=1e-6)


def test_average_distance_add_on_cpu_tensors():
    g = golden("ref_average_distance.npz")
    add = F.average_distance(torch.from_numpy(g["points"]), torch.from_numpy(g["transform_true"]),
                             torch.from_numpy(g["transforms_pred"]))
    np.testing.assert_allclose(add.numpy(), g["add"], rtol=1e-6, atol=1e-7)


def test_chainer_adam_matches_restatement_including_per_param_alpha():
    rs = np.random.RandomState(0)
    q0, t0 = rs.normal(size=(3, 4)).astype(np.float32), rs.normal(size=(3, 3)).astype(np.float32)

    class Link(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.quaternion = torch.nn.Parameter(torch.from_numpy(q0.copy()))
            self.translation = torch.nn.Parameter(torch.from_numpy(t0.copy()))

    link = Link()
    opt = mf.optimizers.Adam(alpha=0.01).setup(link)
    link.translation.update_rule.hyperparam.alpha *= 0.1
    qo, to = q0.copy(), t0.copy()
    ref = O.ChainerAdam([qo, to], [0.01, 0.001])
    for k in range(25):
        gq, gt = rs.normal(size=(3, 4)).astype(np.float32), rs.normal(size=(3, 3)).astype(np.float32) * 10
        link.quaternion.grad, link.translation.grad = torch.from_numpy(gq), torch.from_numpy(gt)
        opt.update()
        ref.update([gq, gt])
    np.testing.assert_allclose(link.quaternion.detach().numpy(), qo, rtol=0, atol=2e-7)
    np.testing.assert_allclose(link.translation.detach().numpy(), to, rtol=0, atol=2e-7)
    # first step of Adam moves every coordinate by alpha (bias-corrected, sign-like)
    assert abs(abs(qo - q0).max() - 0.01 * 25) < 0.25


def test_quaternion_from_matrix_roundtrip_and_sign(fixtures3):
    for f in fixtures3:
        q = mf.geometry.quaternion_from_matrix(f["transform_init"])
        np.testing.assert_allclose(q, O.quaternion_from_matrix(f["transform_init"]), atol=1e-12)
        assert q[0] >= 0 and abs(np.linalg.norm(q) - 1) < 1e-6
        R = O.quaternion_matrix(q)[:3, :3]
        np.testing.assert_allclose(R, f["transform_init"][:3, :3], atol=1e-5)


def test_metrics_match_reference_golden():
    g = golden("ref_metrics.npz")
    auc, x, y = mf.metrics.ycb_video_add_auc(g["errors"], return_xy=True)
    np.testing.assert_allclose(auc, g["add_auc"], rtol=1e-12)
    np.testing.assert_allclose(x, g["add_auc_x"])
    np.testing.assert_allclose(y, g["add_auc_y"])
    np.testing.assert_allclose(mf.metrics.ycb_video_add_auc(g["errors"] * 10), g["add_auc_x10"], rtol=1e-12)
    assert mf.metrics.ycb_video_add_auc(np.full(5, 1.0)) == 0
    np.testing.assert_allclose(mf.metrics.auc_for_errors(g["errors"], 0.1), g["auc_for_errors"], rtol=1e-9)
    a = golden("ref_average_distance.npz")
    adds, add_ss = mf.metrics.average_distance([a["points"]] * 2, [a["transform_true"]] * 2,
                                               list(a["transforms_pred"][:2]))
    np.testing.assert_allclose(adds, a["add"][:2], rtol=1e-5)
    assert (add_ss <= adds + 1e-12).all()


def test_preprocessing_helpers_match_reference_golden():
    g = golden("ref_preprocess.npz")
    np.testing.assert_array_equal(
        mf.geometry.pointcloud_from_depth(g["depth"], fx=30.0, fy=31.0, cx=15.5, cy=11.5), g["pc_z"])
    np.testing.assert_array_equal(
        mf.geometry.pointcloud_from_depth(g["depth"], fx=30.0, fy=31.0, cx=15.5, cy=11.5, depth_type="euclidean"),
        g["pc_euclid"])
    np.testing.assert_array_equal(mf.geometry.masks_to_bboxes(g["masks"]), g["bboxes"])  # incl. empty mask
    assert mf.geometry.masks_to_bboxes(g["masks"][0]).shape == (4,)
    x = torch.from_numpy(g["median_in"])
    np.testing.assert_array_equal(mf.extra.median(x, axis=0).numpy(), g["median_even"])  # mean of middles
    np.testing.assert_array_equal(mf.extra.median(x[:9], axis=0).numpy(), g["median_odd"])
    np.testing.assert_array_equal(mf.extra.median(x).numpy(), g["median_flat"])
    T = mf.geometry.compose_transform(R=np.eye(3) * 2, t=np.array([1.0, 2, 3]))
    assert T[3, 3] == 1 and T[0, 0] == 2 and T[2, 3] == 3


def test_synthetic_scene_shapes(fixtures3):
    sc = mf.synthetic.make_icc_scene(8, seed=0, fixtures=fixtures3)
    assert len(sc["points"]) == len(sc["sdf"]) == 8
    assert sc["grid_target"].shape == sc["grid_nontarget_empty"].shape == (8, 32, 32, 32)
    assert sc["pitch"].dtype == np.float32 and sc["origin"].shape == (8, 3)
    for p, s in zip(sc["points"], sc["sdf"]):
        assert p.dtype == np.float32 and p.shape == (len(s), 3) and 2500 < len(s) < 6000
    assert sc["transform_gt"][0] is None and sc["transform_gt"][3] is not None
    b = mf.synthetic.make_singleview_batch(2, seed=0)
    assert b["rgb"].dtype == np.uint8 and b["rgb"].shape == (2, 256, 256, 3)
    assert b["pcd"].shape == (2, 256, 256, 3) and np.isnan(b["pcd"]).any()
    assert b["grid_nontarget_empty"].dtype == bool


def test_model_point_selection_follows_reference_rng():
    from morefusion_amd.contrib.singleview_3d.models import Model
    m = Model(n_fg_class=21, with_occupancy=True).eval()
    mask = torch.zeros(2, 64, 64, dtype=torch.bool)
    mask[0, 10:50, 5:60] = True  # 2200 > 1000 points: permutation[:1000]
    mask[1, 3:13, 4:24] = True  # 200 < 1000 points: arange + randint padding
    # the valid-pixel list comes from mf_valid_pixel_order on the GPU (tests/test_gpu_preprocess.py);
    # here: the host half, on a list built by NumPy
    order = torch.zeros(2, 64 * 64, dtype=torch.int32)
    for b in range(2):
        idx = np.flatnonzero(mask[b].numpy().ravel())
        order[b, :len(idx)] = torch.from_numpy(idx.astype(np.int32))
    pix = m._subsample(order, mask.reshape(2, -1).sum(1).numpy()).numpy()
    iy, ix = np.where(mask[0].numpy())
    keep = np.random.RandomState(1234).permutation(len(iy))[:1000]  # model.py:211-213
    np.testing.assert_array_equal(pix[0], iy[keep] * 64 + ix[keep])
    iy, ix = np.where(mask[1].numpy())
    keep = np.r_[np.arange(200), np.random.RandomState(1234).randint(0, 200, 800)]  # :214-218
    np.testing.assert_array_equal(pix[1], iy[keep] * 64 + ix[keep])
    assert sum(p.numel() for p in m.parameters()) > 30e6


def test_shard_range_partitions_everything_once():
    from morefusion_amd.parallel import shard_range
    for n in (0, 1, 7, 8, 64, 65):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


def test_pspnet_sampled_tail_equals_dense_forward():
    """PSPNetExtractor.forward_sampled == forward + gather (incl. image-border pixels)."""
    from morefusion_amd.models import PSPNetExtractor
    torch.manual_seed(0)
    net = PSPNetExtractor().eval()
    x = torch.randn(2, 512, 8, 8)  # -> 64x64 output
    dense = net(x)
    Ho = Wo = 64
    pix = torch.randint(0, Ho * Wo, (2, 50))
    pix[0, :6] = torch.tensor([0, Wo - 1, (Ho - 1) * Wo, Ho * Wo - 1, 5, 7 * Wo])  # corners/edges
    ref = torch.gather(dense.reshape(2, 32, -1), 2, pix[:, None, :].expand(2, 32, -1))
    got = net.forward_sampled(x, pix)
    assert got.shape == (2, 32, 50)
    np.testing.assert_allclose(got.detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=2e-5)


def test_voxel_grid_wire_format_roundtrip(fixtures3):
    from morefusion_amd import data_formats as DF
    g = fixtures3[0]["grid_target"]  # float32 32^3 with 341 occupied voxels
    idx, val, dims = DF.encode_voxel_grid(g)
    assert idx.numel() == int((g != 0).sum()) == 341 and dims == (32, 32, 32)
    # flat index convention i*Y*Z + j*Z + k (collision_based_pose_refinement.py:91-94)
    i, j, k = np.nonzero(g)
    np.testing.assert_array_equal(idx.numpy(), i * 32 * 32 + j * 32 + k)
    np.testing.assert_array_equal(DF.decode_voxel_grid(idx, val, dims).numpy(), g)
    np.testing.assert_array_equal(DF.decode_voxel_grid(idx, None, dims).numpy(), g != 0)
    with pytest.raises(ValueError):
        DF.decode_voxel_grid(torch.tensor([32 ** 3]), None, dims)


def test_grid_algebra_eval_and_train_cases():
    from morefusion_amd import data_formats as DF
    rs = np.random.RandomState(0)
    shape = (8, 8, 8)
    gt, gn, ge = rs.uniform(size=shape), rs.uniform(size=shape), rs.uniform(size=shape)
    target, nte = DF.grids_for_network(gt, gn, ge)
    assert target.dtype == bool and nte.dtype == bool
    np.testing.assert_array_equal(target, gt > 0.5)
    np.testing.assert_array_equal(nte, ((gn > 0.5) ^ target) | ((ge > 0.5) ^ target))
    full = (rs.uniform(size=shape) > 0.7).astype(np.int32)
    ids = rs.randint(0, 4, shape).astype(np.int32)
    seen = set()
    for seed in range(40):
        t2, n2 = DF.grids_for_network(gt, gn, ge, full, ids, train=True, random_state=np.random.RandomState(seed))
        np.testing.assert_array_equal(t2, target)
        seen.add(int(n2.sum()))
    assert len(seen) > 5  # the nine cases really produce different no-entry grids
    # reference call sequence: choice(ids, size=randint(1, n+1)) then choice(cases)
    r = np.random.RandomState(3)
    pick = r.choice(np.array([1, 2, 3]), size=r.randint(1, 4), replace=False)
    case = r.choice(DF.GRID_CASES)
    _, n3 = DF.grids_for_network(gt, gn, ge, full, ids, train=True, random_state=np.random.RandomState(3))
    nf = np.isin(ids, pick) ^ full.astype(bool)
    if case == "nontarget_full":
        np.testing.assert_array_equal(n3, nf)
    _, n4 = DF.grids_for_network(gt, gn, ge, full, ids, train=False)
    np.testing.assert_array_equal(n4, nte)  # evaluation ignores the *_full grids


def test_chainer_key_convention_matches_the_reference_link_tree():
    """tests/golden/ref_chainer_param_paths.json lists every parameter path + shape of the
    reference network, obtained by EXECUTING its link definitions (oracle/gen_golden_params.py:
    models/dense_fusion/resnet.py, pspnet.py, contrib/singleview_3d/models/model.py:48-91).
    serializers.chainer_key must map the torch model onto exactly that set."""
    import json
    from morefusion_amd import serializers as S
    from morefusion_amd.contrib.singleview_3d.models import Model
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ref_chainer_param_paths.json")))["params"]
    model = Model(n_fg_class=21, with_occupancy=True)
    mine = {}
    for name, key, tensor in S._entries(model):
        mine[key] = [] if name.endswith("prelu.weight") else list(tensor.shape)  # PReLU slope: shape ()
    assert set(mine) == set(ref), (sorted(set(mine) - set(ref)), sorted(set(ref) - set(mine)))
    assert mine == ref
    assert len(ref) == 77


def test_chainer_npz_roundtrip_and_key_convention(tmp_path):
    from morefusion_amd import serializers as S
    from morefusion_amd.contrib.singleview_3d.models import Model
    assert S.chainer_key("resnet_extractor.res2.0.conv1.weight") == "resnet_extractor/res2/a/conv1/W"
    assert S.chainer_key("resnet_extractor.res5.1.conv2.weight") == "resnet_extractor/res5/b1/conv2/W"
    assert S.chainer_key("resnet_extractor.res3.0.residual_conv.weight") == "resnet_extractor/res3/a/residual_conv/W"
    assert S.chainer_key("pspnet_extractor.psp.convs.3.weight") == "pspnet_extractor/psp/conv4/W"
    assert S.chainer_key("pspnet_extractor.up2.prelu.weight") == "pspnet_extractor/up2/prelu/W"
    assert S.chainer_key("conv4_conf.bias") == "conv4_conf/b"
    torch.manual_seed(0)
    a = Model(n_fg_class=21, with_occupancy=True)
    f = tmp_path / "snapshot_model_best_auc.npz"
    S.save_npz(f, a)
    with np.load(f) as z:
        keys = set(z.files)
        assert z["pspnet_extractor/up1/prelu/W"].shape == ()
        assert z["conv3/W"].shape == (256, 160, 4, 4, 4) and z["conv1_rgb/W"].shape == (64, 32, 1)
    assert "resnet_extractor/mean" not in keys and "conv2_occ/b" in keys
    # 1 stem + 8 blocks*2 + 3 residual convs, no biases (nobias=True in the reference)
    assert sum(k.startswith("resnet_extractor/") for k in keys) == 20
    torch.manual_seed(1)
    b = Model(n_fg_class=21, with_occupancy=True)
    assert S.load_npz(f, b) == []
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb)
    # trainer-snapshot prefix + strictness
    with np.load(f) as z:
        d = {"updater/model:main/" + k: z[k] for k in z.files if k != "conv4_rot/b"}
    g = tmp_path / "trainer.npz"
    np.savez(g, **d)
    with pytest.raises(KeyError):
        S.load_npz(g, b, path="updater/model:main/")
    S.load_npz(g, b, path="updater/model:main/", strict=False)
    c = Model(n_fg_class=20, with_occupancy=True)
    with pytest.raises(ValueError):
        S.load_npz(f, c)


def test_chainer_compat_facade_host_side():
    from morefusion_amd import chainer_compat as chainer
    from morefusion_amd.chainer_compat import cuda
    # concat_examples: dicts / tuples / padding, NumPy out when no device is given
    ex = [dict(class_id=np.int32(2), pcd=np.zeros((4, 4, 3), np.float32), rgb=np.ones((4, 4, 3), np.uint8)),
          dict(class_id=np.int32(5), pcd=np.ones((4, 4, 3), np.float32), rgb=np.zeros((4, 4, 3), np.uint8))]
    out = chainer.dataset.concat_examples(ex, device=-1)
    assert out["class_id"].tolist() == [2, 5] and out["pcd"].shape == (2, 4, 4, 3) and out["rgb"].dtype == np.uint8
    a, b = chainer.dataset.concat_examples([(np.arange(3), np.float32(1)), (np.arange(2), np.float32(2))],
                                           padding=(-1, None))
    np.testing.assert_array_equal(a, [[0, 1, 2], [0, 1, -1]])
    np.testing.assert_array_equal(b, [1, 2])
    # Variable.array, to_cpu, config switches
    x = torch.ones(3, requires_grad=True) * 2
    assert not x.array.requires_grad and isinstance(cuda.to_cpu(x), np.ndarray)
    assert cuda.get_array_module(x) is torch and cuda.get_array_module(np.zeros(1)) is np
    with chainer.using_config("train", False):
        assert chainer.config.train is False
    assert chainer.config.train is True
    with chainer.using_config("enable_backprop", False), chainer.no_backprop_mode():
        assert not torch.is_grad_enabled()
    assert torch.is_grad_enabled()
    if not torch.cuda.is_available():
        with pytest.raises(RuntimeError, match="no CPU fallback"):
            cuda.to_gpu(np.zeros(3))
    assert chainer.optimizers.Adam(alpha=0.01).hyperparam.alpha == 0.01
    assert callable(chainer.serializers.load_npz)


def test_occupancy_grid_1d_2d_vs_reference_golden():
    """The toy siblings of occupancy_grid_3d against outputs of the
    reference's own code (oracle/gen_golden.py), values and gradients."""
    from conftest import golden
    from morefusion_amd.functions.geometry import occupancy_grid_1d, occupancy_grid_2d
    g = golden("ref_occupancy_grid_12d.npz")
    p1 = torch.from_numpy(g["p1"]).requires_grad_(True)
    m1 = occupancy_grid_1d(p1, pitch=1, origin=0, dimension=5)
    np.testing.assert_allclose(m1.detach().numpy(), g["m1"], rtol=0, atol=1e-7)
    np.testing.assert_allclose(
        occupancy_grid_1d(p1.detach() * 0.1, pitch=0.1, origin=-0.05, dimension=8).numpy(), g["m1b"], atol=1e-6)
    # the Function's backward: d_IJ = i - (x - o)/pitch  =>  dL/dx = sum_i -g/pitch
    cells = torch.arange(5.0)
    d = cells[None, :] - (p1[:, None] - 0) / 1
    (d * torch.from_numpy(g["g1"])).sum().backward()
    np.testing.assert_allclose(p1.grad.numpy(), g["gp1"], rtol=1e-6, atol=1e-6)
    p2 = torch.from_numpy(g["p2"]).requires_grad_(True)
    m2 = occupancy_grid_2d(p2, pitch=1, origin=(0, 0), dimension=(5, 6))
    assert m2.shape == (6, 5)  # the reference's [dimension[1], dimension[0]] layout
    np.testing.assert_allclose(m2.detach().numpy(), g["m2"], rtol=0, atol=1e-6)
    m2b = occupancy_grid_2d(p2, pitch=0.5, origin=(-1.0, 0.5), dimension=(9, 7), threshold=2)
    np.testing.assert_allclose(m2b.detach().numpy(), g["m2b"], rtol=0, atol=1e-6)
    m2b.sum().backward()
    assert torch.isfinite(p2.grad).all() and float(p2.grad.abs().sum()) > 0
    with pytest.raises(TypeError):
        occupancy_grid_2d(p2.detach().double(), pitch=1, origin=(0, 0), dimension=(5, 6))


def test_predict_host_logic_decoder_modes_agree(monkeypatch):
    """Model.predict end to end on the CPU with the two HIP ops replaced by the C oracle (test
    stand-ins only): the default path (sampled PSPNet tail) and the fully dense decoder give
    the same poses -- the host logic around the kernels is mode-independent."""
    from oracle import oracle_c as OC
    import morefusion_amd as mf
    from morefusion_amd.contrib.singleview_3d.models import Model
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod

    def avg_cpu(values, points, batch_indices, *, batch_size, origin, pitch, dimensions, return_counts=False, **kw):
        m, c = OC.average_voxelization_3d(values.numpy(), points.numpy(), batch_indices.numpy(),
                                          batch_size=batch_size, origin=origin, pitch=pitch, dimensions=dimensions)
        return (torch.from_numpy(m), torch.from_numpy(c)) if return_counts else torch.from_numpy(m)

    def interp_cpu(vox, points, batch_indices, channels_first=False, batch_start=None):
        out = torch.from_numpy(OC.interpolate_voxel_grid(vox.numpy(), points.numpy(), batch_indices.numpy()))
        return out.t().contiguous() if channels_first else out

    def select_cpu(self, pcd):  # stand-in for mf_valid_pixel_order: np.where, then the product's host half
        from oracle import oracle_np as O_
        order, counts = O_.valid_pixel_order(pcd.numpy())
        return self._subsample(torch.from_numpy(order), counts)

    monkeypatch.setattr(model_mod.functions_module, "average_voxelization_3d", avg_cpu)
    monkeypatch.setattr(model_mod.functions_module, "interpolate_voxel_grid", interp_cpu)
    monkeypatch.setattr(Model, "_select_points", select_cpu)
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).eval()
    b = mf.synthetic.make_singleview_batch(1, seed=3)
    inp = {k: torch.as_tensor(b[k]) for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    outs = []
    with torch.no_grad():
        for tail in (True, False):
            model.sparse_pspnet_tail = tail
            outs.append(model.predict(**inp))
    q, t, c = outs[0]
    assert q.shape == (1, 1000, 4) and t.shape == (1, 1000, 3) and c.shape == (1, 1000)
    # chainer's F.normalize is x / (|x| + 1e-5): unit up to 1e-5 / |x| (random-init heads: |x| ~ 0.05)
    np.testing.assert_allclose(q.norm(dim=2).numpy(), 1.0, atol=2e-3)
    for other in outs[1:]:
        for a, b_ in zip(other, outs[0]):
            np.testing.assert_allclose(a.numpy(), b_.numpy(), rtol=0, atol=5e-6)


def test_raw_example_schema_and_transform():
    """raw dataset-schema examples (probability grids, *_full grids) -> the reference's
    Transform (train.py:27-140) -> concat_examples -> exactly the keys / dtypes the caller passes
    to Model.predict (demo.py:80-93); the training variant draws its grid case from the RNG."""
    from morefusion_amd import synthetic
    from morefusion_amd.chainer_compat import dataset
    ex = synthetic.make_singleview_examples(2, seed=3)
    assert sorted(ex[0]) == ["class_id", "grid_empty", "grid_nontarget", "grid_nontarget_full", "grid_target",
                             "grid_target_full", "origin", "pcd", "pitch", "quaternion_true", "rgb",
                             "translation_true"]
    for k in ("grid_target", "grid_nontarget", "grid_empty"):
        g = ex[0][k]
        assert g.dtype == np.float32 and g.shape == (32, 32, 32) and 0 <= g.min() and g.max() <= 1
        assert 0.01 < (g > 0.5).mean() < 0.9  # probabilities, on both sides of the 0.5 threshold
    assert set(np.unique(ex[0]["grid_target_full"])) == {0, 1} and ex[0]["grid_nontarget_full"].dtype == np.int32
    t = [synthetic.transform_example(e) for e in ex]
    assert "grid_empty" not in t[0] and "grid_target_full" not in t[0]
    tgt, ne = t[0]["grid_target"], t[0]["grid_nontarget_empty"]
    assert tgt.dtype == bool and ne.dtype == bool and not (tgt & ne).any()
    # evaluation = the "empty+nontarget" case of the grid algebra on the thresholded grids
    target = ex[0]["grid_target"] > 0.5
    want = ((ex[0]["grid_nontarget"] > 0.5) ^ target) | ((ex[0]["grid_empty"] > 0.5) ^ target)
    np.testing.assert_array_equal(ne, want)
    batch = dataset.concat_examples(t)
    assert batch["rgb"].shape == (2, 256, 256, 3) and batch["pcd"].dtype == np.float32
    assert batch["grid_nontarget_empty"].shape == (2, 32, 32, 32)
    cases = {synthetic.transform_example(ex[0], train=True, random_state=np.random.RandomState(s))
             ["grid_nontarget_empty"].sum() for s in range(12)}
    assert len(cases) > 3  # different GRID_CASES / instance subsets are drawn
    no_occ = synthetic.transform_example(ex[0], with_occupancy=False)
    assert "pitch" not in no_occ and "origin" not in no_occ and "grid_target" not in no_occ


def test_resnet18_extractor_architecture_and_checkpoint_keys(tmp_path):
    """pretrained_resnet18=True (models/resnet.py:7-52): stride-8 output like the DenseFusion
    ResNet18, BatchNorm frozen even in train(), no gradient below res2, chainercv2 child names
    in the checkpoint keys, npz round trip."""
    from morefusion_amd import serializers as S
    from morefusion_amd.contrib.singleview_3d.models import Model
    from morefusion_amd.models import ResNet18Extractor
    torch.manual_seed(0)
    net = ResNet18Extractor().train()
    x = torch.randint(0, 256, (2, 3, 64, 64)).float()
    before = net.res3.unit1.body.conv1.bn.running_mean.clone()
    y = net(x)
    assert y.shape == (2, 512, 8, 8)
    y.sum().backward()
    assert torch.equal(net.res3.unit1.body.conv1.bn.running_mean, before)          # statistics frozen
    assert net.init_block.conv.conv.weight.grad is None and net.res2.unit2.body.conv2.conv.weight.grad is None
    assert net.res3.unit1.body.conv1.conv.weight.grad is not None
    assert S.chainer_key("resnet_extractor.res4.unit1.identity_conv.conv.weight") == "resnet_extractor/res4/unit1/identity_conv/conv/W"
    assert S.chainer_key("resnet_extractor.init_block.conv.bn.running_var") == "resnet_extractor/init_block/conv/bn/avg_var"
    assert S.chainer_key("resnet_extractor.res5.unit2.body.conv2.bn.weight") == "resnet_extractor/res5/unit2/body/conv2/bn/gamma"
    a = Model(n_fg_class=21, pretrained_resnet18=True, with_occupancy=True)
    b = Model(n_fg_class=21, pretrained_resnet18=True, with_occupancy=True)
    S.save_npz(str(tmp_path / "m.npz"), a)
    assert S.load_npz(str(tmp_path / "m.npz"), b) == []
    for (ka, va), (kb, vb) in zip(a.state_dict().items(), b.state_dict().items()):
        assert ka == kb and torch.equal(va, vb), ka
    with pytest.raises(NotImplementedError, match="occupancy term"):
        Model(n_fg_class=21, loss="add+occupancy")


def test_reference_package_name_and_link_xp():
    """Boundary conveniences: ``import morefusion`` resolves to this package (same module objects),
    and links / the model expose ``xp`` like a Chainer link on the GPU (``link.xp.asarray(points)``,
    ``model.xp.arange``: examples/ycb_video/singleview_3d/demo.py, contrib/occupancy_registration.py:86-88)."""
    import importlib
    import morefusion
    import morefusion_amd
    assert morefusion is morefusion_amd
    assert importlib.import_module("morefusion.contrib.singleview_3d.models").Model is \
        morefusion_amd.contrib.singleview_3d.models.Model
    from morefusion.functions import average_voxelization_3d, truncated_distance_function  # noqa: F401
    from morefusion.geometry import nn  # noqa: F401
    link = morefusion.contrib.IterativeClosestPointLink(np.eye(4, dtype=np.float32))
    x = link.xp.asarray(np.arange(6).reshape(2, 3), dtype=np.float32)
    assert isinstance(x, torch.Tensor) and x.dtype == torch.float32 and x.device == link.quaternion.device
    assert link.xp.concatenate([x, x], axis=0).shape == (4, 3)
    assert link.xp.arange(3).tolist() == [0, 1, 2] and link.xp.zeros((2,), np.int32).dtype == torch.int32
    assert not bool(link.xp.isnan(x).any())


def test_grids_for_network_vs_the_executed_reference_transform():
    """data_formats.grids_for_network against tests/golden/ref_transform.npz: the reference's own ``Transform``
    (examples/ycb_video/singleview_3d/train.py:27-140) executed by oracle/gen_golden_transform.py -- evaluation mode
    and training mode over all nine grid cases, with none / one / several non-target instance ids (the RNG call
    sequence is part of the contract: same seeded RandomState -> same case and id subset)."""
    from conftest import golden
    from morefusion_amd.data_formats import grids_for_network
    g = golden("ref_transform.npz")
    tags = sorted({k.split("__")[0] for k in g})
    assert len(tags) >= 40
    cases = set()
    for tag in tags:
        c = {k.split("__", 1)[1]: g[k] for k in g if k.startswith(tag + "__")}
        train, seed = bool(c["train"]), int(c["seed"])
        target, nte = grids_for_network(c["grid_target"], c["grid_nontarget"], c["grid_empty"], c["grid_target_full"],
                                        c["grid_nontarget_full"], train=train, random_state=np.random.RandomState(seed))
        assert target.dtype == bool and nte.dtype == bool
        np.testing.assert_array_equal(target, c["out_grid_target"], err_msg=tag)
        np.testing.assert_array_equal(nte, c["out_grid_nontarget_empty"], err_msg=f"{tag} case {c['case']}")
        cases.add(str(c["case"]))
    assert len(cases) == 9


def test_graphed_predict_sees_a_replaced_parameter_on_the_next_call():
    """contrib/singleview_3d/models/graphed.py: the replay cache is keyed by the identity, address and in-place version
    of whatever tensor currently sits in every parameter / buffer slot of the module tree -- a replaced Parameter
    object (``module.weight = nn.Parameter(..)``, ``load_state_dict(assign=True)``) or an in-place update drops every
    captured entry on the very next call (a cached tensor list kept replaying the old weights)."""
    import torch
    from morefusion_amd.contrib.singleview_3d.models.graphed import GraphedPredict
    m = torch.nn.Sequential(torch.nn.Linear(3, 3), torch.nn.BatchNorm1d(3))
    g = GraphedPredict(m)
    args = (torch.zeros(2, 3),)
    k = g._key(args)
    g.entries[k] = object()
    assert g._key(args) == k and g.entries                      # nothing changed: the entry survives
    m[0].weight = torch.nn.Parameter(m[0].weight.detach().clone())  # a new object, same values
    g._key(args)
    assert not g.entries
    g.entries[k] = object()
    with torch.no_grad():
        m[1].running_mean.add_(1.0)                              # a buffer updated in place
    g._key(args)
    assert not g.entries
    g.entries[k] = object()
    m.load_state_dict({n: v.clone() for n, v in m.state_dict().items()}, assign=True)
    g._key(args)
    assert not g.entries
