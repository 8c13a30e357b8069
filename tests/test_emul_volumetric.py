"""The channels-last inference path of the pose network (volumetric_cl.py: occupancy convs, conv3 dense +
sparse, conv4 implicit GEMM, channels-last samplers, points-major heads) END TO END ON THE CPU: the kernel
text of conv3d.hip / sparseconv.hip / interp.hip runs behind the fiber emulator with torch CPU tensors as
"device" memory, and its (rot, trans, conf) are compared with the channels-first dense formulation of
contrib/singleview_3d/models/model.py:93-164,232-275 (``Model._extract`` with dense Conv3d, oracle voxel ops).
One object: conv4 alone is 2 M emulated MFMA wave-instructions (~1 minute)."""
import ctypes

import numpy as np
import pytest
import torch

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


def test_channels_last_volumetric_path_matches_dense_formulation(monkeypatch):
    from oracle import oracle_c as OC
    import morefusion_amd as mf
    from morefusion_amd import _lib
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod
    from morefusion_amd.contrib.singleview_3d.models import Model

    L = emul.build(["conv3d.hip", "sparseconv.hip", "interp.hip", "linear.hip", "pointops.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(L, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    monkeypatch.setattr(_lib, "lib", lambda: L)
    monkeypatch.setattr(_lib, "require_gpu", lambda *a: None)
    monkeypatch.setattr(_lib, "stream_ptr", lambda: None)
    monkeypatch.setattr(_lib, "check", lambda code, what: (_ for _ in ()).throw(RuntimeError(what)) if code else None)

    def avg_cpu(values, points, batch_indices, *, batch_size, origin, pitch, dimensions, return_counts=False, **kw):
        m, c = OC.average_voxelization_3d(values.numpy(), points.numpy(), batch_indices.numpy(),
                                          batch_size=batch_size, origin=origin, pitch=pitch, dimensions=dimensions)
        return (torch.from_numpy(m), torch.from_numpy(c)) if return_counts else torch.from_numpy(m)

    def interp_cpu(vox, points, batch_indices, channels_first=False, batch_start=None):
        out = torch.from_numpy(OC.interpolate_voxel_grid(vox.numpy(), points.numpy(), batch_indices.numpy()))
        return out.t().contiguous() if channels_first else out

    monkeypatch.setattr(model_mod.functions_module, "average_voxelization_3d", avg_cpu)
    monkeypatch.setattr(model_mod.functions_module, "interpolate_voxel_grid", interp_cpu)

    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True).eval()
    P = 160
    model._n_point = P
    b = mf.synthetic.make_singleview_batch(1, seed=3)
    rs = np.random.RandomState(0)
    # image features at P sampled pixels (the 2-D backbone is not under test) + their voxel-frame points
    values = torch.from_numpy(rs.uniform(-1, 1, (1, 32, P)).astype(np.float32))
    centre = rs.uniform(8, 24, (1, 3, 1))
    points_vox = torch.from_numpy((centre + rs.normal(0, 3.0, (1, 3, P))).astype(np.float32))
    pitch = torch.as_tensor(b["pitch"], dtype=torch.float32)
    origin = torch.as_tensor(b["origin"], dtype=torch.float32)
    points_cam = points_vox * pitch[:, None, None] + origin[:, :, None]
    grid = torch.as_tensor(b["grid_nontarget_empty"])
    class_id = torch.as_tensor(b["class_id"])

    with torch.no_grad():
        model.sparse_conv3 = False          # dense channels-first formulation (CPU torch convs)
        want = model._pose_from_features(class_id, values, points_cam, pitch, origin, grid)
        model.sparse_conv3 = True
        got = model._pose_from_features_cl(class_id, values, points_cam, pitch, origin, grid)
    for g, w, tol in zip(got, want, (2e-4, 2e-6, 2e-4)):
        assert g.shape == w.shape
        np.testing.assert_allclose(g.numpy(), w.numpy(), rtol=0, atol=tol)
