"""TEST INFRASTRUCTURE: one guarded emulator run of the network's hand-written kernels, in a process of its own
(an out-of-bounds access is a SIGSEGV; ``python -X faulthandler`` names the call).

    python guard_case.py volumetric after|before     the whole channels-last volumetric path
                                                     (volumetric_cl.py: point prep, point MLP, occupancy convs,
                                                     conv3 dense + sparse, conv4, samplers, heads, pose epilogue)
                                                     vs the dense channels-first formulation of model.py:93-164,232-275
    python guard_case.py frontend after|before       valid-pixel order, the fused PSPNet tail, instance crops
    python guard_case.py training after|before       the bf16 training operators (bf16_ops.py over gemm_bf16.hip,
                                                     voxelize.hip, interp.hip): conv forward / data / weight gradient
                                                     (k4 s2 with a channel offset, k3 with dilation), 1x1 convolutions
                                                     with ragged N / K / row counts, channels-last voxelization and
                                                     sampling, forward + backward -- the bodies of
                                                     tests/test_emul_bf16_ops.py.  These kernels mask with
                                                     out-of-range BUFFER loads over a 2 GB span (no hardware bound at
                                                     the tensor's end): the guard pages are the bound here.

Every tensor any of these kernels is handed -- inputs, packed weights, scratch, workspaces, outputs -- sits
against an inaccessible page on the named side (emul.GuardedTensors)."""
import ctypes
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from host_emul import emul  # noqa: E402


def _bind(L):
    from morefusion_amd import _lib
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype


def _patch_lib(G):
    from morefusion_amd import _lib
    _lib.lib = lambda: G
    _lib.require_gpu = lambda *a: None
    _lib.stream_ptr = lambda: None

    def check(code, what):
        if code:
            raise RuntimeError(what)
    _lib.check = check


def volumetric(side):
    from oracle import oracle_c as OC
    import morefusion_amd as mf
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod
    from morefusion_amd.contrib.singleview_3d.models import Model

    L = emul.build(["conv3d.hip", "sparseconv.hip", "interp.hip", "linear.hip", "pointops.hip"])
    _bind(L)

    def avg_cpu(values, points, batch_indices, *, batch_size, origin, pitch, dimensions, return_counts=False, **kw):
        m, c = OC.average_voxelization_3d(values.numpy(), points.numpy(), batch_indices.numpy(),
                                          batch_size=batch_size, origin=origin, pitch=pitch, dimensions=dimensions)
        return (torch.from_numpy(m), torch.from_numpy(c)) if return_counts else torch.from_numpy(m)

    def interp_cpu(vox, points, batch_indices, channels_first=False, batch_start=None):
        out = torch.from_numpy(OC.interpolate_voxel_grid(vox.numpy(), points.numpy(), batch_indices.numpy()))
        return out.t().contiguous() if channels_first else out

    model_mod.functions_module.average_voxelization_3d = avg_cpu
    model_mod.functions_module.interpolate_voxel_grid = interp_cpu

    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).eval()
    P = 160
    model._n_point = P
    b = mf.synthetic.make_singleview_batch(1, seed=3)
    rs = np.random.RandomState(0)
    # image features at P sampled pixels (the 2-D backbone is not under test) + their voxel-frame points;
    # some points sit ON and OUTSIDE the grid faces (negative coordinates, >= D - 1): the corner / tap clamps
    values = torch.from_numpy(rs.uniform(-1, 1, (1, 32, P)).astype(np.float32))
    centre = rs.uniform(8, 24, (1, 3, 1))
    pv = centre + rs.normal(0, 3.0, (1, 3, P))
    pv[0, :, 0] = (0.0, 0.0, 0.0)
    pv[0, :, 1] = (31.0, 31.0, 31.0)
    pv[0, :, 2] = (-0.4, 31.4, 15.0)
    pv[0, :, 3] = (31.49, -0.49, 0.2)
    pv[0, :, 4] = (-3.0, 12.0, 40.0)      # outside the grid: dropped by the voxelization, zero samples
    pv[0, :, 5] = (30.6, 30.7, 30.9)
    points_vox = torch.from_numpy(pv.astype(np.float32))
    pitch = torch.as_tensor(b["pitch"], dtype=torch.float32)
    origin = torch.as_tensor(b["origin"], dtype=torch.float32)
    points_cam = points_vox * pitch[:, None, None] + origin[:, :, None]
    grid = torch.as_tensor(b["grid_nontarget_empty"])
    class_id = torch.as_tensor(b["class_id"])

    with torch.no_grad():
        model.sparse_conv3 = False          # dense channels-first formulation (CPU torch convs)
        want = model._pose_from_features(class_id, values, points_cam, pitch, origin, grid)
        model.sparse_conv3 = True
        log = open(os.environ.get("MF_GUARD_LOG", os.devnull), "w")
        with emul.GuardedTensors(L, side, log) as G:
            _patch_lib(G)
            got = model._pose_from_features_cl(class_id, values, points_cam, pitch, origin, grid)
            ncalls = G.calls
    assert ncalls >= 15, ncalls
    for g, w, tol in zip(got, want, (2e-4, 2e-6, 2e-4)):
        assert g.shape == w.shape
        np.testing.assert_allclose(g.numpy(), w.numpy(), rtol=0, atol=tol)
    print(f"GUARD_OK volumetric {side} calls={ncalls}")


def frontend(side):
    from morefusion_amd.models.backbone2d import PSPNetExtractor
    from morefusion_amd.contrib.singleview_3d.models import Model
    L = emul.build(["psp_tail.hip", "preprocess.hip"])
    _bind(L)
    torch.manual_seed(0)
    log = open(os.environ.get("MF_GUARD_LOG", os.devnull), "w")
    net = PSPNetExtractor().eval()
    B, H, W, P = 2, 12, 10, 37
    u2 = torch.randn(B, 64, H, W)
    Ho, Wo = 2 * H, 2 * W
    rs = np.random.RandomState(1)
    pix = rs.randint(0, Ho * Wo, (B, P))
    pix[0, :6] = [0, Wo - 1, (Ho - 1) * Wo, Ho * Wo - 1, Wo, 2 * Wo - 1]   # corners and edges
    pix = torch.from_numpy(pix)
    with torch.no_grad():
        ref = net._tail(u2, net._tail_taps(pix, H, W)).transpose(1, 2).reshape(B * P, 32)
        # the extractor's own host code (weight packs, pixel list) under the guard: stub the stock layers away
        net.psp = net.up1 = torch.nn.Identity()
        for fmt in (torch.contiguous_format, torch.channels_last):
            x = u2.contiguous(memory_format=fmt)
            net.up2 = torch.nn.Identity()
            with emul.GuardedTensors(L, side, log) as G:
                _patch_lib(G)
                out = net.forward_sampled_rows(x, pix)
            np.testing.assert_allclose(out.numpy(), ref.numpy(), rtol=0, atol=2e-5)
    # valid-pixel order (Model._select_points): ragged counts, an image without NaNs, one with a single valid pixel
    m = Model(n_fg_class=21, with_occupancy=True).eval()
    m._n_point = 50
    for (Hh, Ww) in ((64, 64), (70, 59)):   # 4096 = exactly one chunk; 4130: a 34-pixel tail chunk
        pcd = rs.uniform(-1, 1, (3, Hh, Ww, 3)).astype(np.float32)
        pcd[0][rs.uniform(size=(Hh, Ww)) < 0.4] = np.nan
        pcd[2][...] = np.nan
        pcd[2, Hh - 1, Ww - 1] = 0.5
        t = torch.from_numpy(pcd)
        with emul.GuardedTensors(L, side, log) as G:
            _patch_lib(G)
            got = m._select_points(t)
        for i in range(3):
            valid = np.flatnonzero(~np.isnan(pcd[i]).any(-1))
            keep = m._keep_indices(len(valid))
            np.testing.assert_array_equal(got[i].numpy(), valid[keep])
    # instance crops (mf_instance_stats + mf_instance_crops): masks touching the image border, an empty instance, an
    # instance with too few valid depths, vs the oracle's restatement of the reference's host loop
    from oracle import oracle_np as O
    from morefusion_amd.geometry import instance_crops
    Hh, Ww, S = 60, 72, 32
    rgb = rs.randint(0, 255, (Hh, Ww, 3)).astype(np.uint8)
    depth = rs.uniform(0.4, 1.2, (Hh, Ww)).astype(np.float32)
    depth[rs.uniform(size=(Hh, Ww)) < 0.1] = np.nan
    label = np.zeros((Hh, Ww), np.int32)
    label[0:20, 0:25] = 1            # touches two borders
    label[30:60, 50:72] = 2          # touches the far corner
    label[25:28, 30:33] = 3          # 9 pixels: fewer than min_valid
    Kmat = np.array([[80.0, 0, 36.0], [0, 80.0, 30.0], [0, 0, 1]])
    ids = np.array([1, 2, 3, 7], np.int32)   # 7: no pixel at all
    with emul.GuardedTensors(L, side, log) as G:
        _patch_lib(G)
        out = instance_crops(torch.from_numpy(rgb), torch.from_numpy(depth), Kmat, torch.from_numpy(label), ids,
                             image_size=S, min_valid=20)
    w_rgb, w_pcd, w_keep, _ = O.instance_crops(rgb, depth, Kmat, label, ids, image_size=S, min_valid=20)
    np.testing.assert_array_equal(out["keep"].numpy(), w_keep)
    np.testing.assert_array_equal(out["rgb"].numpy()[w_keep], w_rgb[w_keep])
    np.testing.assert_array_equal(out["pcd"].numpy()[w_keep], w_pcd[w_keep].astype(np.float32))
    assert list(out["keep"].numpy()) == [True, True, False, False]
    print(f"GUARD_OK frontend {side}")


class _MonkeyPatch:
    """The two pytest.MonkeyPatch calls the test bodies make, without pytest (no undo: the process ends with the case)."""

    def setenv(self, k, v):
        os.environ[k] = v

    def setattr(self, obj, name, value, raising=True):
        setattr(obj, name, value)


def training(side):
    from morefusion_amd.contrib.singleview_3d.models import bf16_ops
    import test_emul_bf16_ops as T   # the test bodies: operators vs torch float32 autograd / the oracle
    L = emul.build(["gemm_bf16.hip", "voxelize.hip", "interp.hip"])
    _bind(L)
    log = open(os.environ["MF_GUARD_LOG"], "w") if os.environ.get("MF_GUARD_LOG") else None
    with emul.GuardedTensors(L, side, log) as G:
        _patch_lib(G)
        T.test_conv3d_operator_forward_and_gradients(bf16_ops)
        # both forms of the occupancy branch behind the guard pages: the narrow voxels-as-columns kernel (its padding taps
        # are masked buffer loads) and the implicit-GEMM engine
        for narrow in (True, False):
            T.test_occupancy_branch_operator_chain(bf16_ops, _MonkeyPatch(), narrow)
        for n, Kin, N, relu in ((150, 3, 8, True), (130, 64, 63, False), (70, 200, 136, True)):
            T.test_linear_operator_forward_and_gradients.__wrapped__(bf16_ops, n, Kin, N, relu) if hasattr(
                T.test_linear_operator_forward_and_gradients, "__wrapped__") else \
                T.test_linear_operator_forward_and_gradients(bf16_ops, n, Kin, N, relu)
        T.test_channels_last_bf16_voxelization_and_sampling_vs_oracle(bf16_ops)
    assert G.calls > 40, G.calls
    print(f"GUARD_OK training {side}")


if __name__ == "__main__":
    {"volumetric": volumetric, "frontend": frontend, "training": training}[sys.argv[1]](sys.argv[2])
