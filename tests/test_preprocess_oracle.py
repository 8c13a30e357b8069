"""CPU checks of the pre-processing row's oracle and of the host-side grid
placement.  imgviz / OpenCV are absent from this image, so the resize restatements are pinned by
properties (and against torch's half-pixel bilinear), not by reference outputs: parity unpinned."""
import os

import numpy as np
import torch

from morefusion_amd import geometry, synthetic
from oracle import oracle_np as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_oracle_backprojection_and_bbox_match_reference_golden():
    g = np.load(os.path.join(GOLD, "ref_preprocess.npz"))
    pc = O.pointcloud_from_depth(g["depth"], 30.0, 31.0, 15.5, 11.5)
    np.testing.assert_array_equal(pc, g["pc_z"])
    for m, bb in zip(g["masks"], g["bboxes"]):
        np.testing.assert_array_equal(O.mask_to_bbox(m), bb.astype(np.int64))


def test_nearest_resize_rules():
    a = np.arange(6 * 8, dtype=np.float64).reshape(6, 8)
    np.testing.assert_array_equal(O.cv2_resize_nearest(a, 6, 8), a)
    np.testing.assert_array_equal(O.cv2_resize_nearest(a, 12, 16), a.repeat(2, 0).repeat(2, 1))
    np.testing.assert_array_equal(O.cv2_resize_nearest(a, 3, 4), a[::2, ::2])
    # 8 -> 5 columns: floor(x * 1.6) = 0 1 3 4 6
    np.testing.assert_array_equal(O.cv2_resize_nearest(a, 6, 5)[0], a[0, [0, 1, 3, 4, 6]])


def test_linear_resize_u8_properties():
    rs = np.random.RandomState(0)
    const = np.full((17, 23, 3), 201, np.uint8)
    assert (O.cv2_resize_linear_u8(const, 40, 31) == 201).all()
    assert (O.cv2_resize_linear_u8(const, 9, 11) == 201).all()
    img = rs.randint(0, 256, (40, 64, 3)).astype(np.uint8)
    box = O.cv2_resize_linear_u8(img, 20, 32)  # exact 2:1 -> rounded 2x2 mean
    ref = (img.astype(int).reshape(20, 2, 32, 2, 3).sum(axis=(1, 3)) + 2) >> 2
    np.testing.assert_array_equal(box, ref)
    for (h, w) in [(57, 91), (23, 40), (80, 128), (40, 100)]:
        out = O.cv2_resize_linear_u8(img, h, w).astype(np.float64)
        t = torch.from_numpy(img.astype(np.float64)).permute(2, 0, 1)[None]
        ref = torch.nn.functional.interpolate(t, (h, w), mode="bilinear", align_corners=False)
        ref = ref[0].permute(1, 2, 0).numpy()
        assert np.abs(out - ref).max() <= 1.0, (h, w, np.abs(out - ref).max())


def test_centerize_geometry_and_instance_crops():
    f = synthetic.make_rgbd_frame(0)
    rgb, pcd, keep, bbox = O.instance_crops(f["rgb"], f["depth"], f["K"], f["label"], f["instance_ids"])
    np.testing.assert_array_equal(keep, [1, 1, 1, 1, 0, 0, 1, 0])
    np.testing.assert_array_equal(bbox[0], [10, 20, 110, 330])
    np.testing.assert_array_equal(bbox[-1], [0, 0, 0, 0])
    # wide box 100 x 310 -> 83 x 256 centred vertically: rows [86, 169) carry data
    rows = np.flatnonzero((~np.isnan(pcd[0, :, :, 2])).any(axis=1))
    assert rows.min() >= 86 and rows.max() < 169
    assert (rgb[0, :86] == 0).all() and (rgb[0, 169:] == 0).all()
    # the 256 x 256 instance is copied verbatim (centerize returns the source)
    y1, x1, y2, x2 = bbox[2]
    np.testing.assert_array_equal(rgb[2], f["rgb"][y1:y2, x1:x2])
    full = O.pointcloud_from_depth(f["depth"], f["K"][0, 0], f["K"][1, 1], f["K"][0, 2], f["K"][1, 2])
    np.testing.assert_array_equal(pcd[2], full[y1:y2, x1:x2])
    # skipped instances are pure padding
    assert (rgb[4] == 0).all() and np.isnan(pcd[4]).all() and np.isnan(pcd[5]).all()
    # every valid crop point is one of the frame's points of that instance
    pts = pcd[1][~np.isnan(pcd[1]).any(axis=2)]
    src = full[(f["label"] == 5) & ~np.isnan(f["depth"])]
    assert len(pts) and {tuple(p) for p in pts} <= {tuple(p) for p in src}


def test_grid_origin_matches_numpy_nanmedian_and_nanmean():
    f = synthetic.make_rgbd_frame(1)
    _, pcd, keep, _ = O.instance_crops(f["rgb"], f["depth"], f["K"], f["label"], f["instance_ids"])
    pitch = np.linspace(0.004, 0.011, len(keep)).astype(np.float32)
    t = torch.from_numpy(pcd.astype(np.float32))
    got_med = geometry.grid_origin(t, pitch, dim=32).numpy()
    got_mean = geometry.grid_origin(t, pitch, dim=32, center="mean").numpy()
    for i in range(len(keep)):
        if not keep[i]:
            assert np.isnan(got_med[i]).all()
            continue
        p = pcd[i].astype(np.float32)
        np.testing.assert_allclose(got_med[i], np.nanmedian(p, axis=(0, 1)) - 15.5 * pitch[i], rtol=0, atol=1e-7)
        np.testing.assert_allclose(got_mean[i], np.nanmean(p.astype(np.float64), axis=(0, 1)) - 15.5 * pitch[i],
                                   rtol=0, atol=2e-6)


def test_crop_kernel_source_on_the_host_matches_oracle(tmp_path):
    """csrc/preprocess.hip's k_pre_crops has no barriers: its text is compiled with g++ behind a
    shim (tests/host_emul/mf_common.h) and run thread by thread -- the kernel source itself is
    checked against the oracle without a GPU (the -m gpu test repeats this on the device)."""
    import ctypes
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if shutil.which("g++") is None:
        import pytest
        pytest.skip("g++ not available")
    src = tmp_path / "preprocess.cpp"
    shutil.copy(os.path.join(root, "morefusion_amd", "csrc", "preprocess.hip"), src)
    shutil.copy(os.path.join(root, "tests", "host_emul", "mf_common.h"), tmp_path / "mf_common.h")
    so = tmp_path / "libpre_host.so"
    subprocess.run(["g++", "-O1", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
                    "-I", os.path.join(root, "include"), "-o", str(so), str(src)], check=True)
    L = ctypes.CDLL(str(so))
    p, i, d = ctypes.c_void_p, ctypes.c_int, ctypes.c_double
    L.mf_instance_crops.argtypes = [p, p, p, i, i, d, d, d, d, p, p, i, i, i, p, p, p, p]
    for seed, (H, W), S in [(0, (480, 640), 256), (2, (300, 420), 256), (3, (480, 640), 64)]:
        f = synthetic.make_rgbd_frame(seed, H, W)
        ids = f["instance_ids"]
        n = len(ids)
        stats = np.zeros((n, 6), np.int32)  # what mf_instance_stats produces (LDS kernel: device only)
        for k, a in enumerate(ids):
            m = f["label"] == a
            stats[k] = ([*O.mask_to_bbox(m), m.sum(), (m & ~np.isnan(f["depth"])).sum()] if m.any()
                        else [2 ** 31 - 1, 2 ** 31 - 1, 0, 0, 0, 0])
        rgb_out = np.full((n, S, S, 3), 77, np.uint8)
        pcd_out = np.zeros((n, S, S, 3), np.float32)
        keep = np.zeros(n, np.uint8)
        K = f["K"]
        rgb, depth, label = (np.ascontiguousarray(f[k]) for k in ("rgb", "depth", "label"))
        L.mf_instance_crops(rgb.ctypes.data, depth.ctypes.data, label.ctypes.data, H, W, K[0, 0], K[1, 1],
                            K[0, 2], K[1, 2], ids.ctypes.data, stats.ctypes.data, n, S, 50,
                            rgb_out.ctypes.data, pcd_out.ctypes.data, keep.ctypes.data, None)
        r, pc, kp, _ = O.instance_crops(f["rgb"], f["depth"], K, f["label"], ids, image_size=S)
        np.testing.assert_array_equal(keep.astype(bool), kp)
        np.testing.assert_array_equal(rgb_out, r)
        np.testing.assert_array_equal(pcd_out, pc.astype(np.float32))
