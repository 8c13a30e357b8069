"""``Model.predict`` against golden outputs produced by EXECUTING THE REFERENCE'S OWN NETWORK CODE
(``oracle/gen_golden_predict.py``: models/dense_fusion/resnet.py, pspnet.py and
contrib/singleview_3d/models/model.py run on a torch-CPU stand-in for Chainer's links, the voxel ops
as CUDA text; weights = this package's model under ``torch.manual_seed(0)``, injected through the
pinned parameter paths).  CPU: the host logic with oracle stand-ins for the HIP ops (dense data
flow).  GPU: the shipped path (sampled PSPNet tail, points-fed sparse conv3, fused kernels)."""
import numpy as np
import pytest
import torch

from conftest import golden


def _batch():
    import morefusion_amd as mf
    g = golden("ref_predict.npz")
    b = mf.synthetic.make_singleview_batch(int(g["batch_size"]), seed=int(g["seed"]))
    return g, {k: torch.as_tensor(b[k]) for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}


def _model(g):
    from morefusion_amd.contrib.singleview_3d.models import Model
    torch.manual_seed(int(g["weight_seed"]))
    return Model(n_fg_class=21, with_occupancy=True).eval()


def _check(outs, g, tag, tol):
    q, t, c = (x.detach().cpu().numpy() for x in outs)
    np.testing.assert_allclose(q, g[f"{tag}__quaternion"], rtol=0, atol=tol)
    np.testing.assert_allclose(c, g[f"{tag}__confidence"], rtol=0, atol=tol)
    np.testing.assert_allclose(t, g[f"{tag}__translation"], rtol=0, atol=tol * 0.01)  # metres (1e-3 voxel ~ 1e-5 m)


def test_predict_host_logic_vs_reference_network_code(monkeypatch):
    from oracle import oracle_c as OC
    from oracle import oracle_np as O
    import morefusion_amd.contrib.singleview_3d.models.model as model_mod
    from morefusion_amd.contrib.singleview_3d.models import Model

    def avg_cpu(values, points, batch_indices, *, batch_size, origin, pitch, dimensions, return_counts=False, **kw):
        m, c = OC.average_voxelization_3d(values.numpy(), points.numpy(), batch_indices.numpy(),
                                          batch_size=batch_size, origin=origin, pitch=pitch, dimensions=dimensions)
        return (torch.from_numpy(m), torch.from_numpy(c)) if return_counts else torch.from_numpy(m)

    def interp_cpu(vox, points, batch_indices, channels_first=False, batch_start=None):
        out = torch.from_numpy(OC.interpolate_voxel_grid(vox.numpy(), points.numpy(), batch_indices.numpy()))
        return out.t().contiguous() if channels_first else out

    def select_cpu(self, pcd):
        order, counts = O.valid_pixel_order(pcd.numpy())
        return self._subsample(torch.from_numpy(order), counts)

    monkeypatch.setattr(model_mod.functions_module, "average_voxelization_3d", avg_cpu)
    monkeypatch.setattr(model_mod.functions_module, "interpolate_voxel_grid", interp_cpu)
    monkeypatch.setattr(Model, "_select_points", select_cpu)
    g, inp = _batch()
    model = _model(g)
    model.sparse_pspnet_tail = False   # dense decoder + gather (model.py:181-222)
    model.sparse_conv3 = False         # dense Conv3d (model.py:118-128)
    with torch.no_grad():
        _check(model.predict(**inp), g, "given", 2e-4)
        inp2 = dict(inp)
        inp2["origin"] = None          # the median rule of model.py:202-207
        _check(model.predict(**inp2), g, "median", 2e-4)


@pytest.mark.gpu
def test_predict_gpu_path_vs_reference_network_code():
    g, inp = _batch()
    model = _model(g).cuda()
    inp = {k: v.cuda() for k, v in inp.items()}
    with torch.no_grad():
        model.predict(**inp)  # MIOpen solver choice settles on the first call of a shape
        _check(model.predict(**inp), g, "given", 1e-3)
        inp["origin"] = None
        _check(model.predict(**inp), g, "median", 1e-3)
