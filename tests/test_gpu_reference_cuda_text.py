"""The HIP kernels against golden vectors produced by EXECUTING THE REFERENCE'S OWN CUDA TEXT
(``oracle/gen_golden_cuda.py`` -> tests/golden/ref_cuda_*.npz): no oracle in between.
K7/K8 truncated_distance_function, F2 pseudo_occupancy_voxelization, K5/K6 interpolate_voxel_grid
(GPU forms), K9 geometry.nn, F3/F4 the ICC / ICP links' loss on the fixture scenes."""
import numpy as np
import pytest
import torch

from conftest import golden

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
import morefusion_amd.functions as F  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _close(got, want, rel=2e-4):
    """Gradient vectors: float32 sums of ~10^4 terms in different orders."""
    want = np.asarray(want, np.float64)
    np.testing.assert_allclose(np.asarray(got, np.float64), want, rtol=1e-3, atol=rel * float(np.abs(want).max()))


@pytest.mark.parametrize("tag", ["d32_t1", "d32_t2", "d32_t3", "d8x12x10_t2"])
def test_tdf_hip_vs_reference_cuda_text(tag):
    g = golden("ref_cuda_tdf.npz")
    c = {k.split("__", 1)[1]: g[k] for k in g if k.startswith(tag + "__")}
    K = int(c["ksize"]) ** 3
    pt = dev(c["points"]).requires_grad_(True)
    tdf, idx = F.truncated_distance_function(pt, pitch=float(c["pitch"]), origin=tuple(c["origin"]),
                                             dims=tuple(int(v) for v in c["dims"]),
                                             truncation=float(c["truncation"]), return_indices=True)
    np.testing.assert_array_equal(tdf.detach().cpu().numpy(), c["matrix"])
    # the reference returns `indices // K` (point id of the winner; lowest flat index on ties)
    np.testing.assert_array_equal(idx.cpu().numpy(), np.where(c["indices"] >= 0, c["indices"] // K, -1))
    tdf.backward(dev(c["gmatrix"]))
    np.testing.assert_allclose(pt.grad.cpu().numpy(), c["gpoints"], rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("tag,thr,off", [("t2_off0", 2, 0.0), ("t2_off002", 2, 0.02), ("t1_off0", 1, 0.0)])
def test_pseudo_occupancy_hip_vs_reference(tag, thr, off):
    g = golden("ref_cuda_pseudo_occupancy.npz")
    outs = F.pseudo_occupancy_voxelization(dev(g["points"]), dev(g["sdf"]), pitch=float(g["pitch"]),
                                           origin=tuple(g["origin"]), dims=(32,) * 3, threshold=thr, sdf_offset=off)
    for o, k in zip(outs, ("uniform", "surface", "inside")):
        np.testing.assert_array_equal(o.cpu().numpy(), g[f"{tag}__{k}"])


def test_interpolate_hip_vs_reference_cuda_text():
    g = golden("ref_cuda_interpolate.npz")
    vox = dev(g["voxelized"]).requires_grad_(True)
    out = F.interpolate_voxel_grid(vox, dev(g["points"]), dev(g["batch_indices"]))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["values"])
    out.backward(dev(g["gvalues"]))
    np.testing.assert_allclose(vox.grad.cpu().numpy(), g["gvoxelized"], rtol=1e-5, atol=1e-6)


def test_nn_hip_vs_reference_raw_kernel():
    g = golden("ref_cuda_nn.npz")
    np.testing.assert_array_equal(mf.geometry.nn(dev(g["ref"]), dev(g["query"])).cpu().numpy(), g["indices"])


@pytest.mark.parametrize("n,off", [(1, 0.0), (3, 0.0), (3, 0.02), (8, 0.0), (8, 0.02)])
def test_icc_link_loss_hip_vs_reference(n, off, fixtures3):
    """IterativeCollisionCheckLink.forward of the reference, executed (K7 text underneath), vs the
    fused HIP path on the same scene and poses (both iteration layouts)."""
    g = golden("ref_cuda_links.npz")
    sc = mf.synthetic.make_icc_scene(n, seed=0, fixtures=fixtures3)
    link = mf.contrib.IterativeCollisionCheckLink(sc["transform_init"], sdf_offset=off).to_gpu()
    with torch.no_grad():
        link.quaternion.copy_(dev(g[f"icc_q_n{n}"]))
        link.translation.copy_(dev(g[f"icc_t_n{n}"]))
    args = ([dev(p) for p in sc["points"]], [dev(s) for s in sc["sdf"]], dev(np.asarray(sc["pitch"], np.float32)),
            dev(np.stack(sc["origin"]).astype(np.float32)), dev(np.stack(sc["grid_target"]).astype(np.float32)),
            dev(np.stack(sc["grid_nontarget_empty"]).astype(np.float32)))
    loss_t = link(*args)
    np.testing.assert_allclose(float(loss_t.detach()), float(g[f"icc_loss_n{n}_off{off}"]), rtol=2e-5, atol=2e-6)
    # gradients: the reference's own backward methods, run (oracle/chainer_tape.py + cuda_text.py)
    gg = golden("ref_cuda_link_gradients.npz")
    loss_t.backward()
    _close(link.quaternion.grad.cpu().numpy(), gg[f"icc_gq_n{n}_off{off}"])
    _close(link.translation.grad.cpu().numpy(), gg[f"icc_gt_n{n}_off{off}"])


def test_icp_link_loss_hip_vs_reference(fixtures3):
    g = golden("ref_cuda_links.npz")
    f = fixtures3[2]
    target = dev((np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32))
    link = mf.contrib.IterativeClosestPointLink(np.eye(4, dtype=np.float32)).to_gpu()
    with torch.no_grad():
        link.quaternion.copy_(dev(g["icp_q"]))
        link.translation.copy_(dev(g["icp_t"]))
    loss_t = link(dev(f["pcd_cad"].astype(np.float32)), target)
    np.testing.assert_allclose(float(loss_t.detach()), float(g["icp_loss"]), rtol=2e-5)
    gg = golden("ref_cuda_link_gradients.npz")
    loss_t.backward()
    _close(link.quaternion.grad.cpu().numpy(), gg["icp_gq"])
    _close(link.translation.grad.cpu().numpy(), gg["icp_gt"])


def test_voxelization_hip_vs_reference_cuda_text():
    g = golden("ref_cuda_voxelization.npz")
    D, B = int(g["dim"]), int(g["batch_size"])
    kw = dict(batch_size=B, origin=tuple(g["origin"]), pitch=float(g["pitch"]), dimensions=(D, D, D))
    values = dev(g["values"]).requires_grad_(True)
    y, counts = F.average_voxelization_3d(values, dev(g["points"]), dev(g["batch_indices"]), return_counts=True, **kw)
    np.testing.assert_array_equal(counts.cpu().numpy(), g["avg_counts"])
    np.testing.assert_array_equal(y.detach().cpu().numpy(), g["avg_matrix"])
    y.backward(dev(g["gy"]))
    np.testing.assert_array_equal(values.grad.cpu().numpy(), g["avg_gvalues"])
    values = dev(g["values"]).requires_grad_(True)
    ym, ind = F.max_voxelization_3d(values, dev(g["points"]), dev(g["batch_indices"]), dev(g["intensities"]),
                                    return_indices=True, **kw)
    np.testing.assert_array_equal(ind.cpu().numpy(), g["max_indices"])
    np.testing.assert_array_equal(ym.detach().cpu().numpy(), g["max_matrix"])
    ym.backward(dev(g["gy"]))
    np.testing.assert_allclose(values.grad.cpu().numpy(), g["max_gvalues"], rtol=1e-6, atol=1e-7)


@pytest.mark.parametrize("sym", [False, True])
def test_average_distance_hip_vs_reference(sym):
    """ADD / ADD-S loss and its gradient to the predicted transforms against the reference's own
    average_distance run on the CPU (nn = the RawKernel text; backward through oracle/chainer_tape.py)."""
    g = golden("ref_cuda_average_distance.npz")
    tag = "adds" if sym else "add"
    Tp = dev(g["transforms_pred"]).requires_grad_(True)
    out = F.average_distance(dev(g["points"]), dev(g["transform_true"]), Tp, symmetric=sym)
    np.testing.assert_allclose(out.detach().cpu().numpy(), g[f"{tag}_value"], rtol=5e-6, atol=1e-8)
    (out * dev(g[f"{tag}_gout"])).sum().backward()
    want = g[f"{tag}_gT"][:, :3, :]            # the bottom row of a rigid transform is constant
    got = Tp.grad.cpu().numpy()[:, :3, :]
    np.testing.assert_allclose(got, want, rtol=2e-3, atol=2e-4 * float(np.abs(want).max()))


def test_occupancy_registration_link_hip_vs_reference():
    """(f3) contrib/occupancy_registration.py:21-60 executed from the reference (its OccupancyGrid3D
    forward / backward, QuaternionMatrix / ComposeTransform backward, under oracle/chainer_tape.py):
    loss and gradients of the HIP-backed link at the same pose."""
    g = golden("ref_cuda_link_gradients.npz")
    link = mf.contrib.OccupancyRegistrationLink(g["occreg_q"], g["occreg_t"]).to_gpu()
    loss = link(dev(g["occreg_model"]), dev(g["occreg_grid_target"]), pitch=float(g["occreg_pitch"]),
                origin=tuple(g["occreg_origin"]), threshold=1.5)
    np.testing.assert_allclose(float(loss.detach()), float(g["occreg_loss"]), rtol=2e-5, atol=2e-6)
    loss.backward()
    _close(link.quaternion.grad.cpu().numpy(), g["occreg_gq"])
    _close(link.translation.grad.cpu().numpy(), g["occreg_gt"])


@pytest.mark.parametrize("mode", ["add/add_s", "add"])
def test_model_loss_hip_vs_reference(mode):
    """A14: contrib/singleview_3d/models/model.py:377-434 (Model.loss) executed from the reference
    (per-object loop over its average_distance, nn = the RawKernel text) vs the batched fused loss of
    this build: value and gradients to the per-point quaternions, translations and confidences."""
    from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels
    g = golden("ref_cuda_model_loss.npz")
    model = Model(n_fg_class=21, with_occupancy=True, loss=mode,
                  models=PitchTableModels({2: g["cad_2"], 13: g["cad_13"]}))
    q = dev(g["quaternion_pred"]).requires_grad_(True)
    t = dev(g["translation_pred"]).requires_grad_(True)
    c = dev(g["confidence_pred"]).requires_grad_(True)
    np.random.seed(int(g["seed"]))
    loss = model.loss(class_id=torch.as_tensor(g["class_id"]), quaternion_true=dev(g["quaternion_true"]),
                      translation_true=dev(g["translation_true"]), quaternion_pred=q, translation_pred=t,
                      confidence_pred=c)
    tag = mode.replace("/", "_")
    np.testing.assert_allclose(float(loss.detach()), float(g[f"{tag}__loss"]), rtol=1e-5)
    loss.backward()
    _close(q.grad.cpu().numpy(), g[f"{tag}__gq"], rel=5e-4)
    _close(t.grad.cpu().numpy(), g[f"{tag}__gt"], rel=5e-4)
    _close(c.grad.cpu().numpy(), g[f"{tag}__gc"], rel=5e-4)
