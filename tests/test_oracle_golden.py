"""Pin the NumPy oracle (oracle/oracle_np.py) against golden vectors produced by
executing the reference's own CPU code (oracle/gen_golden.py) and against the
reference tests' known-answer vectors.  CPU only."""
import numpy as np
import pytest

from conftest import golden
from oracle import oracle_np as O


def _dense(shape, idx, val, dtype):
    a = np.zeros(int(np.prod(shape)), dtype=dtype)
    a[idx] = val
    return a.reshape(shape)


@pytest.mark.parametrize("mode", ["cpu", "gpu"])
def test_average_voxelization_3d_reference_test_setup(mode):
    g = golden("ref_average_voxelization_3d.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    y, counts = O.average_voxelization_3d(
        g["values"], g["points"], g["batch_indices"], batch_size=B,
        origin=g["origin"], pitch=g["pitch"], dimensions=(D, D, D), mode=mode)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    c_ref = _dense((B, D, D, D), g["counts_nonzero_index"], g["counts_nonzero_value"], np.int32)
    np.testing.assert_array_equal(counts, c_ref)  # bit-exact voxel indices
    np.testing.assert_array_equal(y, y_ref)  # same summation order -> bit-exact
    gy = np.random.RandomState(1).uniform(-1, 1, y.shape).astype(np.float32)
    gv = O.average_voxelization_3d_backward(
        gy, g["points"], g["batch_indices"], counts, origin=g["origin"],
        pitch=g["pitch"], dimensions=(D, D, D), mode=mode)
    np.testing.assert_array_equal(gv, g["gvalues"])


@pytest.mark.parametrize("mode", ["cpu", "gpu"])
def test_average_voxelization_3d_model_shape(mode):
    g = golden("ref_average_voxelization_3d_model.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    y, counts = O.average_voxelization_3d(
        g["values"], g["points"], g["batch_indices"], batch_size=B,
        origin=(0, 0, 0), pitch=1.0, dimensions=(D, D, D), mode=mode)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    c_ref = _dense((B, D, D, D), g["counts_nonzero_index"], g["counts_nonzero_value"], np.int32)
    assert c_ref.max() >= 3  # the case really has collisions
    np.testing.assert_array_equal(counts, c_ref)
    np.testing.assert_array_equal(y, y_ref)


def test_round_half_fork():
    """cpu fork = half-to-even, gpu fork = half-away."""
    pts = np.array([[0.5, 1.5, 2.5], [-0.5, 3.5, 0.49999997]], dtype=np.float32)
    i_cpu = O.voxel_index(pts, np.zeros(3, np.float32), np.float32(1), "cpu")
    i_gpu = O.voxel_index(pts, np.zeros(3, np.float32), np.float32(1), "gpu")
    np.testing.assert_array_equal(i_cpu, [[0, 2, 2], [0, 4, 0]])
    np.testing.assert_array_equal(i_gpu, [[1, 2, 3], [-1, 4, 0]])


def test_max_voxelization_3d():
    g = golden("ref_max_voxelization_3d.npz")
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    y, ind = O.max_voxelization_3d(
        g["values"], g["points"], g["batch_indices"], g["intensities"], batch_size=B,
        origin=g["origin"], pitch=g["pitch"], dimensions=(D, D, D))
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    i_ref = np.full(B * D ** 3, -1, np.int32)
    i_ref[g["indices_valid_index"]] = g["indices_valid_value"]
    np.testing.assert_array_equal(ind.reshape(-1), i_ref)
    np.testing.assert_array_equal(y, y_ref)
    gy = np.random.RandomState(1).uniform(-1, 1, y.shape).astype(np.float32)
    gv = O.max_voxelization_3d_backward(gy, ind, g["points"].shape[0])
    np.testing.assert_allclose(gv, g["gvalues"], rtol=0, atol=0)


def test_interpolate_voxel_grid_cpu_fork_bitexact_gpu_fork_close():
    g = golden("ref_interpolate_voxel_grid.npz")
    vox = np.random.RandomState(int(g["vox_seed"])).uniform(-1, 1, tuple(g["vox_shape"])).astype(np.float32)
    v_cpu = O.interpolate_voxel_grid(vox, g["points"], g["batch_indices"], mode="cpu")
    np.testing.assert_array_equal(v_cpu, g["values"])
    # gpu fork differs from the CPU twin only for negative coordinates (trunc vs floor)
    v_gpu = O.interpolate_voxel_grid(vox, g["points"], g["batch_indices"], mode="gpu")
    nonneg = (g["points"] >= 0).all(axis=1)
    np.testing.assert_allclose(v_gpu[nonneg], g["values"][nonneg], rtol=0, atol=1e-6)
    assert (~nonneg).any()


def test_occupancy_grid_3d_known_answer_and_random():
    g = golden("ref_occupancy_grid_3d.npz")
    # reference golden: tests/functions_tests/geometry_tests/test_occupancy_grid_3d.py:28-38
    m = O.occupancy_grid_3d(g["known_points"], pitch=1, origin=(0, 0, 0), dims=(5, 5, 5))
    nonzero = [[0, 0, 0], [0, 1, 0], [0, 0, 1], [4, 3, 4], [3, 4, 4], [4, 4, 4]]
    expect = np.zeros((5, 5, 5), bool)
    expect[tuple(zip(*nonzero))] = True
    np.testing.assert_array_equal(m > 0, expect)
    np.testing.assert_array_equal(m, g["known_grid"])
    p = float(g["c1_pitch"])
    m1 = O.occupancy_grid_3d(g["c1_points"], pitch=p, origin=(-16 * p,) * 3, dims=(32,) * 3)
    np.testing.assert_array_equal(m1, g["c1_grid"])
    m2 = O.occupancy_grid_3d(g["c1_points"][:200], pitch=p, origin=(-16 * p,) * 3,
                             dims=(32,) * 3, threshold=2)
    np.testing.assert_array_equal(m2, g["c1_grid_thr2"])


def test_occupancy_grid_3d_backward_finite_difference():
    rs = np.random.RandomState(0)
    pts = rs.uniform(0.5, 4.5, (6, 3))
    gm = rs.uniform(-1, 1, (6, 6, 6))
    kw = dict(pitch=0.9, origin=(0.1, 0.0, -0.1), dims=(6, 6, 6), threshold=1.5)
    ga = O.occupancy_grid_3d_backward(gm, pts, **kw)
    eps = 1e-6
    gn = np.zeros_like(pts)
    for i in range(pts.shape[0]):
        for d in range(3):
            pp, pm = pts.copy(), pts.copy()
            pp[i, d] += eps
            pm[i, d] -= eps
            gn[i, d] = ((O.occupancy_grid_3d(pp, **kw) - O.occupancy_grid_3d(pm, **kw)) * gm).sum() / (2 * eps)
    np.testing.assert_allclose(ga, gn, rtol=1e-5, atol=1e-6)


def test_transforms():
    g = golden("ref_transforms.npz")
    q, t = g["q"], g["t"]
    np.testing.assert_array_equal(O.quaternion_matrix(q), g["quaternion_matrix"])
    np.testing.assert_array_equal(O.quaternion_matrix(q[0]), g["quaternion_matrix_1d"])
    np.testing.assert_array_equal(O.transformation_matrix(q, t), g["transformation_matrix"])
    np.testing.assert_array_equal(O.transformation_matrix(q[0], t[0]), g["transformation_matrix_1d"])
    np.testing.assert_array_equal(O.translation_matrix(t), g["translation_matrix"])
    np.testing.assert_array_equal(
        O.compose_transform(g["quaternion_matrix"][:, :3, :3], t), g["compose_transform"])
    # transform_points: the reference's matmul order is BLAS-defined -> 1e-6
    np.testing.assert_allclose(O.transform_points(g["points"], g["transformation_matrix"]),
                               g["transform_points"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(O.transform_points(g["points"], g["transformation_matrix"][0]),
                               g["transform_points_1"], rtol=0, atol=1e-6)
    # unit quaternion <-> matrix round trip (reference test: == trimesh quaternion_matrix)
    qn = q / np.linalg.norm(q, axis=1, keepdims=True)
    for i in range(5):
        q_back = O.quaternion_from_matrix(O.quaternion_matrix(qn[i].astype(np.float64)))
        s = 1.0 if qn[i, 0] >= 0 else -1.0
        np.testing.assert_allclose(q_back, s * qn[i], atol=1e-6)


def test_quaternion_matrix_backward():
    g = golden("ref_transforms.npz")
    # hand-written dR/dQ of the reference (quaternion_matrix.py:36-51), via gR -> gQ
    q = g["q"].astype(np.float64)
    gR = g["gR"].astype(np.float64)
    ga = O.quaternion_matrix_backward(q, gR)
    eps = 1e-6
    gn = np.zeros_like(q)
    for i in range(q.shape[0]):
        for d in range(4):
            qp, qm = q.copy(), q.copy()
            qp[i, d] += eps
            qm[i, d] -= eps
            gn[i, d] = ((O.quaternion_matrix(qp) - O.quaternion_matrix(qm)) * gR).sum() / (2 * eps)
    np.testing.assert_allclose(ga, gn, rtol=1e-6, atol=1e-7)


def test_average_distance_add():
    g = golden("ref_average_distance.npz")
    add = O.average_distance(g["points"], g["transform_true"], g["transforms_pred"])
    np.testing.assert_allclose(add, g["add"], rtol=1e-6, atol=1e-7)
    # loss == metric (tests/functions_tests/loss_tests/test_average_distance.py:24-31)
    for i in range(g["transforms_pred"].shape[0]):
        a, _ = O.metrics_average_distance(g["points"], g["transform_true"], g["transforms_pred"][i])
        np.testing.assert_allclose(add[i], a, rtol=1e-5)


def test_metrics_and_median():
    g = golden("ref_metrics.npz")
    np.testing.assert_allclose(O.ycb_video_add_auc(g["errors"]), g["add_auc"], rtol=1e-12)
    np.testing.assert_allclose(O.ycb_video_add_auc(g["errors"] * 10), g["add_auc_x10"], rtol=1e-12)
    assert O.ycb_video_add_auc(np.full(5, 1.0)) == 0 == g["add_auc_none"]
    p = golden("ref_preprocess.npz")
    np.testing.assert_array_equal(O.median(p["median_in"], axis=0), p["median_even"])
    np.testing.assert_array_equal(O.median(p["median_in"][:9], axis=0), p["median_odd"])
    np.testing.assert_array_equal(O.median(p["median_in"]), p["median_flat"])


def test_cuda_text_goldens_do_not_depend_on_fma_contraction_where_it_matters():
    """``oracle/gen_golden_cuda_fma.py`` re-ran the reference's CUDA kernel text compiled WITH fused multiply-add
    contraction (what nvcc does by default) and compared it with the committed un-contracted goldens.  Every
    integer output -- voxel indices / counts, max-voxel winners, TDF arg-min ids, nearest-neighbour indices -- is
    identical in the two forks; floats move by at most a few ulp.  ("bit-exact voxel indices" therefore does not
    hinge on which fork a CUDA compiler picks; the HIP kernels are built -ffp-contract=off.)"""
    import json
    import os
    from conftest import GOLDEN as GOLDEN_DIR
    rep = json.load(open(os.path.join(GOLDEN_DIR, "ref_cuda_fma_fork.json")))
    assert rep["integer_outputs_identical"] is True and rep["integer_elements_that_differ"] == 0
    assert set(rep["files"]) == {"ref_cuda_tdf.npz", "ref_cuda_pseudo_occupancy.npz", "ref_cuda_interpolate.npz",
                                 "ref_cuda_nn.npz", "ref_cuda_voxelization.npz", "ref_cuda_links.npz"}
    moved = 0
    for f, rec in rep["files"].items():
        if rec == "identical":
            continue
        for k, v in rec.items():
            assert v["kind"] == "float", (f, k)               # no integer array appears among the differences
            assert v["max_abs_diff_over_max_abs"] < 5e-6, (f, k, v)
            moved += 1
    assert moved > 0   # the contracted fork really is another compilation (float results do move)
