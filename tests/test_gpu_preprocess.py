"""Pre-processing row on the MI355X through the C-ABI: per-instance statistics,
masked crops and centerize geometry, bit-exact against the oracle (uint8 rgb, float32 points,
NaN pattern), then frame -> crops -> grid placement -> Model.predict end to end."""
import numpy as np
import pytest
import torch

from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
from morefusion_amd import geometry, synthetic  # noqa: E402


def _run(frame, **kw):
    return geometry.instance_crops(
        torch.as_tensor(frame["rgb"]).cuda(), torch.as_tensor(frame["depth"]).cuda(), frame["K"],
        torch.as_tensor(frame["label"]).cuda(), frame["instance_ids"], **kw)


@pytest.mark.parametrize("seed,hw", [(0, (480, 640)), (1, (480, 640)), (2, (300, 420))])
def test_instance_crops_bit_exact_vs_oracle(seed, hw):
    f = synthetic.make_rgbd_frame(seed, *hw)
    out = _run(f)
    rgb, pcd, keep, bbox = O.instance_crops(f["rgb"], f["depth"], f["K"], f["label"], f["instance_ids"])
    np.testing.assert_array_equal(out["keep"].cpu().numpy(), keep)
    np.testing.assert_array_equal(out["bbox"].cpu().numpy(), bbox)
    valid = ~np.isnan(f["depth"])
    np.testing.assert_array_equal(out["n_valid"].cpu().numpy(),
                                  [((f["label"] == i) & valid).sum() for i in f["instance_ids"]])
    np.testing.assert_array_equal(out["rgb"].cpu().numpy(), rgb)
    got = out["pcd"].cpu().numpy()
    assert got.dtype == np.float32
    np.testing.assert_array_equal(got, pcd.astype(np.float32))  # NaN == NaN positions included


def test_instance_crops_other_size_threshold_and_errors():
    f = synthetic.make_rgbd_frame(3)
    out = _run(f, image_size=64, min_valid=20)
    rgb, pcd, keep, _ = O.instance_crops(f["rgb"], f["depth"], f["K"], f["label"], f["instance_ids"],
                                         image_size=64, min_valid=20)
    assert keep[4]  # the 36-pixel instance passes a threshold of 20 valid points
    np.testing.assert_array_equal(out["keep"].cpu().numpy(), keep)
    np.testing.assert_array_equal(out["rgb"].cpu().numpy(), rgb)
    np.testing.assert_array_equal(out["pcd"].cpu().numpy(), pcd.astype(np.float32))
    none = geometry.instance_crops(torch.as_tensor(f["rgb"]).cuda(), torch.as_tensor(f["depth"]).cuda(),
                                   f["K"], torch.as_tensor(f["label"]).cuda(), np.zeros(0, np.int32))
    assert none["rgb"].shape == (0, 256, 256, 3) and none["keep"].numel() == 0
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        geometry.instance_crops(torch.as_tensor(f["rgb"]), torch.as_tensor(f["depth"]), f["K"],
                                torch.as_tensor(f["label"]), f["instance_ids"])
    with pytest.raises(TypeError):
        geometry.instance_crops(torch.as_tensor(f["rgb"]).cuda().float(), torch.as_tensor(f["depth"]).cuda(),
                                f["K"], torch.as_tensor(f["label"]).cuda(), f["instance_ids"])


def test_frame_to_poses_end_to_end():
    from morefusion_amd.contrib.singleview_3d.models import Model
    f = synthetic.make_rgbd_frame(0)
    out = _run(f)
    keep = out["keep"]
    class_id = torch.tensor([2, 5, 9, 12, 15, 16, 19, 21], dtype=torch.int32)[keep.cpu()]
    rgb, pcd = out["rgb"][keep], out["pcd"][keep]
    pitch = torch.tensor([synthetic.CLASS_PITCH[int(c)] for c in class_id], dtype=torch.float32).cuda()
    origin = geometry.grid_origin(pcd, pitch, dim=32)
    ref = np.stack([np.nanmedian(p, axis=(0, 1)) for p in pcd.cpu().numpy()]) - 15.5 * pitch.cpu().numpy()[:, None]
    np.testing.assert_allclose(origin.cpu().numpy(), ref, rtol=0, atol=1e-6)
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    grid = torch.zeros((int(keep.sum()), 32, 32, 32), dtype=torch.bool, device="cuda")
    with torch.no_grad():
        # first call of these shapes: MIOpen settles its solver choice (may differ from later calls)
        model.predict(class_id=class_id.cuda(), rgb=rgb, pcd=pcd, pitch=pitch, origin=origin,
                      grid_nontarget_empty=grid)
        q, t, c = model.predict(class_id=class_id.cuda(), rgb=rgb, pcd=pcd, pitch=pitch, origin=origin,
                                grid_nontarget_empty=grid)
        q2, t2, c2 = model.predict(class_id=class_id.cuda(), rgb=rgb, pcd=pcd, grid_nontarget_empty=grid)
    assert q.shape[0] == int(keep.sum()) and torch.isfinite(q).all() and torch.isfinite(t).all()
    # origin=None places the grids with the same batched median.  The two calls may run
    # different MIOpen solutions for the stock 2-D convolutions (first call of a shape vs the
    # cached choice), so this is a float32-convolution tolerance, not bit equality.
    for a, b in ((q, q2), (t, t2), (c, c2)):
        torch.testing.assert_close(a, b, rtol=2e-3, atol=2e-3)


def test_wire_format_grids_decode_on_device_and_feed_icc(fixtures3):
    """the three recorded instances as the ROS pipeline ships them -- sparse
    VoxelGrid.msg (flat indices + values), decoded ON THE DEVICE (data_formats.decode_voxel_grid)
    -- go straight into the ICC refinement; same loss / gradients as the dense fixtures, and the
    oracle's."""
    from morefusion_amd import data_formats as DF
    from oracle import oracle_c as OC
    sc = mf.synthetic.make_icc_scene(3, seed=0, fixtures=fixtures3)
    grids = {}
    for key in ("grid_target", "grid_nontarget_empty"):
        dec = []
        for g in sc[key]:
            idx, val, dims = DF.encode_voxel_grid(g)            # what the mapping node publishes
            d = DF.decode_voxel_grid(idx.cuda(), val.cuda(), dims)  # collision_based_pose_refinement.py:86-98
            assert d.is_cuda and d.shape == (32, 32, 32)
            np.testing.assert_array_equal(d.cpu().numpy(), g)
            dec.append(d)
        grids[key] = torch.stack(dec)
    dev = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()  # noqa: E731
    link = mf.contrib.IterativeCollisionCheckLink(sc["transform_init"], sdf_offset=0.02).to_gpu()
    loss = link([dev(p) for p in sc["points"]], [dev(s) for s in sc["sdf"]], dev(sc["pitch"]), dev(sc["origin"]),
                grids["grid_target"], grids["grid_nontarget_empty"])
    loss.backward()
    q0, t0 = link.quaternion.detach().cpu().numpy(), link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(sc["points"], sc["sdf"], sc["pitch"], sc["origin"], sc["grid_target"],
                                          sc["grid_nontarget_empty"], q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(float(loss.detach()), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(link.quaternion.grad.cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)


def test_raw_examples_through_transform_into_predict():
    """dataset-schema examples -> Transform -> concat_examples -> to_gpu ->
    Model.predict with the caller's keyword interface (demo.py:80-100); same poses as feeding the
    pre-transformed batch."""
    from morefusion_amd.chainer_compat import cuda, dataset
    from morefusion_amd.contrib.singleview_3d.models import Model
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).cuda().eval()
    examples = [mf.synthetic.transform_example(e) for e in mf.synthetic.make_singleview_examples(2, seed=5)]
    batch = dataset.concat_examples(examples)
    inputs = {k: cuda.to_gpu(batch[k]) for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        model.predict(**inputs)
        q, t, c = model.predict(**inputs)
    assert q.shape == (2, 1000, 4) and torch.isfinite(q).all() and torch.isfinite(t).all()
    idx = c.argmax(dim=1)
    pose = torch.cat([q[torch.arange(2), idx], t[torch.arange(2), idx]], 1)
    assert pose.shape == (2, 7)
    # the no-entry grid matters: a different grid changes the prediction
    with torch.no_grad():
        q2, _, _ = model.predict(**{**inputs, "grid_nontarget_empty": torch.zeros_like(inputs["grid_nontarget_empty"])})
    assert float((q2 - q).abs().max()) > 0


def test_point_selection_kernel_and_reference_rng():
    """Model._select_points = mf_valid_pixel_order + the reference's NumPy-RNG subsample
    (contrib/singleview_3d/models/model.py:195-220): bit-exact against np.where + RandomState(1234)
    on the synthetic batch (float32 and float64 input), a crafted ragged case, and the raw kernel
    on an unaligned view."""
    import ctypes
    from morefusion_amd.contrib.singleview_3d.models import Model
    m = Model(n_fg_class=21, with_occupancy=True).eval()
    b = synthetic.make_singleview_batch(3, seed=4)
    pcd = b["pcd"].copy()
    pcd[2] = np.nan                     # third crop: 400 valid pixels < n_point (arange + randint padding)
    pcd[2, 100:110, 90:130] = 1.0
    for dtype in (np.float32, np.float64):
        pix = m._select_points(torch.as_tensor(pcd.astype(dtype)).cuda()).cpu().numpy()
        for i in range(3):
            iy, ix = np.where(~np.isnan(pcd[i]).any(axis=2))
            n = len(iy)
            rs = np.random.RandomState(1234)
            keep = rs.permutation(n)[:1000] if n >= 1000 else np.r_[np.arange(n), rs.randint(0, n, 1000 - n)]
            np.testing.assert_array_equal(pix[i], iy[keep] * pcd.shape[2] + ix[keep])
    # raw kernel, image rows not 16-byte aligned and H*W not a multiple of 4
    HW = 70 * 71
    buf = torch.full((2 * HW * 3 + 1,), float("nan")).cuda()
    view = buf[1:].reshape(2, HW, 3)
    g = torch.Generator().manual_seed(0)
    vals = torch.randn(2, HW, 3, generator=g)
    vals[torch.rand(2, HW, generator=g) < 0.4] = float("nan")
    view.copy_(vals)
    order = torch.full((2, HW), -1, dtype=torch.int32).cuda()
    counts = torch.zeros(2, dtype=torch.int32).cuda()
    mf._lib.check(mf._lib.lib().mf_valid_pixel_order(view.data_ptr(), 2, HW, order.data_ptr(), counts.data_ptr(),
                                                     mf._lib.stream_ptr()), "mf_valid_pixel_order")
    for i in range(2):
        want = np.flatnonzero(~np.isnan(vals[i].numpy()).any(axis=1))
        assert int(counts[i]) == len(want)
        np.testing.assert_array_equal(order[i, :len(want)].cpu().numpy(), want)
    with pytest.raises(ValueError):
        m._select_points(torch.full((1, 8, 8, 3), float("nan")).cuda())  # "an example has no valid point"
