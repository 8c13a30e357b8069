"""The oracle against golden vectors produced by EXECUTING THE REFERENCE'S OWN CUDA TEXT and the
Python around it (``oracle/gen_golden_cuda.py``: the strings of ``cuda.elementwise(...)`` /
``cupy.RawKernel`` compiled by g++ and run sequentially, float32, no FMA contraction; the links'
``forward`` run on NumPy).  This pins the half of the path that has no CPU implementation in the
reference: truncated_distance_function fwd/bwd (K7/K8), pseudo_occupancy_voxelization (F2),
interpolate_voxel_grid GPU fwd + bwd (K5/K6), geometry.nn (K9), the ICC / ICP links' loss
(F3/F4).  (CPU; the same files are checked against the HIP kernels in tests/test_gpu_*.py.)"""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle_c as OC
from oracle import oracle_np as O

TDF_CASES = ["d32_t1", "d32_t2", "d32_t3", "d8x12x10_t2"]


def _tdf_case(g, tag):
    return {k.split("__", 1)[1]: g[k] for k in g if k.startswith(tag + "__")}


@pytest.mark.parametrize("tag", TDF_CASES)
def test_tdf_forward_backward_vs_reference_cuda_text(tag):
    c = _tdf_case(golden("ref_cuda_tdf.npz"), tag)
    dims = tuple(int(v) for v in c["dims"])
    tdf, flat, ksize = O.truncated_distance_function(c["points"], pitch=c["pitch"], origin=c["origin"], dims=dims,
                                                     truncation=c["truncation"])
    assert ksize == int(c["ksize"])
    np.testing.assert_array_equal(tdf, c["matrix"])          # bit-exact distances
    np.testing.assert_array_equal(flat, c["indices"])        # the reference's `i` = p * K + k (lowest i on ties)
    assert (c["indices"] >= 0).sum() > 100
    gp = O.truncated_distance_function_backward(c["gmatrix"], c["points"], flat, ksize, pitch=c["pitch"],
                                                origin=c["origin"])
    # per-point float sums in a different order than the sequential atomicAdd: not bit-exact
    np.testing.assert_allclose(gp, c["gpoints"], rtol=2e-5, atol=2e-6)
    if dims == (32, 32, 32):  # the C port (cubic grids)
        tdf_c, idx_c = OC.truncated_distance_function(c["points"], pitch=float(c["pitch"]), origin=c["origin"],
                                                      dims=dims, truncation=float(c["truncation"]))
        np.testing.assert_array_equal(tdf_c, c["matrix"])
        np.testing.assert_array_equal(idx_c, c["indices"])


@pytest.mark.parametrize("tag,thr,off", [("t2_off0", 2, 0.0), ("t2_off002", 2, 0.02), ("t1_off0", 1, 0.0)])
def test_pseudo_occupancy_voxelization_vs_reference(tag, thr, off):
    g = golden("ref_cuda_pseudo_occupancy.npz")
    gu, gs, gi = O.pseudo_occupancy_voxelization(g["points"], g["sdf"], pitch=g["pitch"], origin=g["origin"],
                                                 dims=(32,) * 3, threshold=thr, sdf_offset=off)
    np.testing.assert_array_equal(gu, g[f"{tag}__uniform"])
    np.testing.assert_array_equal(gs, g[f"{tag}__surface"])
    np.testing.assert_array_equal(gi, g[f"{tag}__inside"])
    assert (g[f"{tag}__inside"] > 0).sum() > 50


def test_interpolate_gpu_forms_vs_reference_cuda_text():
    g = golden("ref_cuda_interpolate.npz")
    out = O.interpolate_voxel_grid(g["voxelized"], g["points"], g["batch_indices"], mode="gpu")
    np.testing.assert_array_equal(out, g["values"])           # same j order per (point, channel)
    gv = O.interpolate_voxel_grid_backward(g["gvalues"], g["points"], g["batch_indices"], g["voxelized"].shape,
                                           mode="gpu")
    np.testing.assert_allclose(gv, g["gvoxelized"], rtol=1e-5, atol=1e-6)  # float atomics: order differs
    out_c = OC.interpolate_voxel_grid(g["voxelized"], g["points"], g["batch_indices"])
    np.testing.assert_array_equal(out_c, g["values"])


def test_nn_vs_reference_raw_kernel():
    g = golden("ref_cuda_nn.npz")
    np.testing.assert_array_equal(O.nn(g["ref"], g["query"]), g["indices"])
    np.testing.assert_array_equal(OC.nn(g["ref"], g["query"]), g["indices"])
    assert g["indices"][3] == 5   # exact tie between ref rows 5 and 17: argmin keeps the first


@pytest.mark.parametrize("n,off", [(1, 0.0), (3, 0.0), (3, 0.02), (8, 0.0), (8, 0.02)])
def test_icc_link_forward_vs_reference(n, off, fixtures3):
    """contrib/iterative_collision_check_link.py:31-99 executed from the reference (K7 text underneath)
    on the fixture scenes; the oracle's loss (and the C port's) must agree to float32 rounding of
    the final reductions (NumPy pairwise sums vs the ports' orders)."""
    import morefusion_amd.synthetic as synthetic
    g = golden("ref_cuda_links.npz")
    sc = synthetic.make_icc_scene(n, seed=0, fixtures=fixtures3)
    q, t = g[f"icc_q_n{n}"], g[f"icc_t_n{n}"]
    want = float(g[f"icc_loss_n{n}_off{off}"])
    args = (sc["points"], sc["sdf"], sc["pitch"], sc["origin"], np.stack(sc["grid_target"]),
            np.stack(sc["grid_nontarget_empty"]).astype(np.float32))
    loss = O.icc_loss(*args, q, t, sdf_offset=off, grad=False)
    loss = loss[0] if isinstance(loss, tuple) else loss
    np.testing.assert_allclose(float(loss), want, rtol=2e-6, atol=2e-7)
    loss_c = OC.icc_loss_grad(*args, q, t, sdf_offset=off)[0]
    np.testing.assert_allclose(float(loss_c), want, rtol=2e-5, atol=2e-6)


def test_icp_link_forward_vs_reference(fixtures3):
    g = golden("ref_cuda_links.npz")
    f = fixtures3[2]
    target = (np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32)
    source = f["pcd_cad"].astype(np.float32)
    np.testing.assert_allclose(O.transformation_matrix(g["icp_q"][None], g["icp_t"][None])[0], g["icp_T"], rtol=0, atol=1e-7)
    loss = O.icp_loss(source, target, g["icp_q"], g["icp_t"], grad=False)
    loss = loss[0] if isinstance(loss, tuple) else loss
    np.testing.assert_allclose(float(loss), float(g["icp_loss"]), rtol=2e-6)
    np.testing.assert_allclose(OC.icp_loss_grad(source, target, g["icp_q"], g["icp_t"])[0], float(g["icp_loss"]), rtol=2e-5)


def test_voxelization_gpu_forms_vs_reference_cuda_text():
    """K1-K4 (the forward_gpu / backward_gpu kernels, round-half-away index rule) executed from the
    reference: exact .5 coordinates, shared voxels, exact intensity ties."""
    g = golden("ref_cuda_voxelization.npz")
    D, B = int(g["dim"]), int(g["batch_size"])
    kw = dict(batch_size=B, origin=g["origin"], pitch=g["pitch"], dimensions=(D, D, D))
    m, c = O.average_voxelization_3d(g["values"], g["points"], g["batch_indices"], mode="gpu", **kw)
    np.testing.assert_array_equal(c, g["avg_counts"])
    np.testing.assert_array_equal(m, g["avg_matrix"])
    gv = O.average_voxelization_3d_backward(g["gy"], g["points"], g["batch_indices"], c, origin=g["origin"],
                                            pitch=g["pitch"], dimensions=(D, D, D), mode="gpu")
    np.testing.assert_array_equal(gv, g["avg_gvalues"])
    mm, ind = O.max_voxelization_3d(g["values"], g["points"], g["batch_indices"], g["intensities"], mode="gpu", **kw)
    np.testing.assert_array_equal(ind, g["max_indices"])
    np.testing.assert_array_equal(mm, g["max_matrix"])
    np.testing.assert_allclose(O.max_voxelization_3d_backward(g["gy"], ind, len(g["points"])), g["max_gvalues"],
                               rtol=1e-6, atol=1e-7)
    # the CPU fork (round-half-even) differs on the exact halves: the fork is real
    _, c_cpu = O.average_voxelization_3d(g["values"], g["points"], g["batch_indices"], mode="cpu", **kw)
    assert (c_cpu != g["avg_counts"]).any()


def _close(got, want, rel=2e-4):
    """Gradient vectors: float32 sums of ~10^4 terms in different orders."""
    want = np.asarray(want, np.float64)
    np.testing.assert_allclose(np.asarray(got, np.float64), want, rtol=1e-3, atol=rel * float(np.abs(want).max()))


@pytest.mark.parametrize("n,off", [(1, 0.0), (3, 0.0), (3, 0.02), (8, 0.0), (8, 0.02)])
def test_icc_link_gradients_vs_reference_backward(n, off, fixtures3):
    """d loss / d (quaternion, translation) of the ICC link obtained by RUNNING the reference: its own
    Function.backward methods (QuaternionMatrix, ComposeTransform, the CUDA text of the TDF backward)
    composed by oracle/chainer_tape.py (chainer's elementary rules; maximum ties -> first argument)."""
    import morefusion_amd.synthetic as synthetic
    g, gg = golden("ref_cuda_links.npz"), golden("ref_cuda_link_gradients.npz")
    sc = synthetic.make_icc_scene(n, seed=0, fixtures=fixtures3)
    args = (sc["points"], sc["sdf"], sc["pitch"], sc["origin"], np.stack(sc["grid_target"]),
            np.stack(sc["grid_nontarget_empty"]).astype(np.float32))
    q, t = g[f"icc_q_n{n}"], g[f"icc_t_n{n}"]
    want_q, want_t = gg[f"icc_gq_n{n}_off{off}"], gg[f"icc_gt_n{n}_off{off}"]
    assert np.abs(want_t).max() > 0.1
    _, (gq, gt, _aux) = O.icc_loss(*args, q, t, sdf_offset=off, grad=True)
    _close(gq, want_q)
    _close(gt, want_t)
    out_c = OC.icc_loss_grad(*args, q, t, sdf_offset=off)
    _close(out_c[1], want_q)
    _close(out_c[2], want_t)


def test_icp_link_gradient_vs_reference_backward(fixtures3):
    g, gg = golden("ref_cuda_links.npz"), golden("ref_cuda_link_gradients.npz")
    f = fixtures3[2]
    target = (np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32)
    source = f["pcd_cad"].astype(np.float32)
    _, (gq, gt) = O.icp_loss(source, target, g["icp_q"], g["icp_t"], grad=True)
    _close(gq, gg["icp_gq"])
    _close(gt, gg["icp_gt"])
    out_c = OC.icp_loss_grad(source, target, g["icp_q"], g["icp_t"])
    _close(out_c[1], gg["icp_gq"])
    _close(out_c[2], gg["icp_gt"])


@pytest.mark.parametrize("sym", [False, True])
def test_average_distance_vs_reference(sym):
    """functions/loss/average_distance.py:40-85 executed (ADD-S through the RawKernel nn text)."""
    g = golden("ref_cuda_average_distance.npz")
    out = O.average_distance(g["points"], g["transform_true"], g["transforms_pred"], symmetric=sym)
    np.testing.assert_allclose(out, g["adds_value" if sym else "add_value"], rtol=2e-6, atol=1e-8)


def test_occupancy_registration_link_vs_reference():
    """(f3) the oracle's occupancy_grid_3d forward / backward chained like
    contrib/occupancy_registration.py:21-60 against the reference link, executed."""
    g = golden("ref_cuda_link_gradients.npz")
    model, gt = g["occreg_model"], g["occreg_grid_target"]
    pitch, origin = float(g["occreg_pitch"]), tuple(g["occreg_origin"])
    T = O.transformation_matrix(g["occreg_q"], g["occreg_t"])
    pw = O.transform_points(model, T)
    grid = O.occupancy_grid_3d(pw, pitch=pitch, origin=origin, dims=gt.shape[1:], threshold=1.5)
    occd, unocc = gt[0], gt[1]
    loss = (unocc * grid).sum() / grid.sum() - (occd * grid).sum() / occd.sum()
    np.testing.assert_allclose(loss, float(g["occreg_loss"]), rtol=2e-5, atol=2e-6)
    gg = unocc / grid.sum() - (unocc * grid).sum() / grid.sum() ** 2 - occd / occd.sum()
    gp = O.occupancy_grid_3d_backward(gg.astype(np.float32), pw, pitch=pitch, origin=origin, dims=gt.shape[1:],
                                      threshold=1.5)
    _close(gp.sum(axis=0), g["occreg_gt"])  # d loss / d translation = sum of the point gradients


def test_model_loss_add_host_logic_vs_reference():
    """A14 on the CPU: Model.loss in 'add' mode (the plain-torch composite of the ADD loss, no HIP op
    involved) against the reference's Model.loss executed under the tape -- value and gradients."""
    import torch
    from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels
    g = golden("ref_cuda_model_loss.npz")
    model = Model(n_fg_class=21, with_occupancy=True, loss="add",
                  models=PitchTableModels({2: g["cad_2"], 13: g["cad_13"]}))
    q = torch.tensor(g["quaternion_pred"], requires_grad=True)
    t = torch.tensor(g["translation_pred"], requires_grad=True)
    c = torch.tensor(g["confidence_pred"], requires_grad=True)
    np.random.seed(int(g["seed"]))
    loss = model.loss(class_id=torch.as_tensor(g["class_id"]), quaternion_true=torch.tensor(g["quaternion_true"]),
                      translation_true=torch.tensor(g["translation_true"]), quaternion_pred=q, translation_pred=t,
                      confidence_pred=c)
    np.testing.assert_allclose(float(loss.detach()), float(g["add__loss"]), rtol=1e-5)
    loss.backward()
    _close(q.grad.numpy(), g["add__gq"], rel=5e-4)
    _close(t.grad.numpy(), g["add__gt"], rel=5e-4)
    _close(c.grad.numpy(), g["add__gc"], rel=5e-4)
