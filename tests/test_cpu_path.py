"""BASELINE config 1 -- "single object, 1k-point cloud -> 32^3 occupancy_grid_3d on CPU/NumPy path": the PRODUCT's
CPU dispatch (morefusion_amd/functions/geometry/_cpu.py, chosen by the array type like the reference's
get_array_module) against golden vectors produced by executing the reference's NumPy code
(oracle/gen_golden.py -> tests/golden/ref_occupancy_grid_3d.npz, ref_average_voxelization_3d*.npz)."""
import os

import numpy as np
import pytest
import torch

import morefusion  # the alias package: reference call sites import `morefusion.functions`
import morefusion_amd as mf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden(name):
    return np.load(os.path.join(GOLDEN, name))


def _dense(shape, index, value, dtype):
    out = np.zeros(int(np.prod(shape)), dtype)
    out[index] = value
    return out.reshape(shape)


def test_reference_known_answer_through_the_numpy_path():
    """tests/functions_tests/geometry_tests/test_occupancy_grid_3d.py:12-38 (setUp + check_forward), NumPy input."""
    points = np.array([[0, 0.05, 0.1], [3.9, 3.95, 4]], dtype=np.float32)
    matrix = morefusion.functions.occupancy_grid_3d(points, pitch=1, origin=(0, 0, 0), dims=(5, 5, 5))
    nonzero = [[0, 0, 0], [0, 1, 0], [0, 0, 1], [4, 3, 4], [3, 4, 4], [4, 4, 4]]
    matrix_bool = np.zeros(tuple(matrix.shape), dtype=bool)
    matrix_bool[tuple(zip(*nonzero))] = True
    np.testing.assert_array_equal(np.asarray(matrix) > 0, matrix_bool)
    g = golden("ref_occupancy_grid_3d.npz")
    np.testing.assert_array_equal(np.asarray(matrix), g["known_grid"])  # the reference's own output, bit for bit


def test_config1_1k_points_into_32_cubed_equals_the_reference_numpy_output():
    g = golden("ref_occupancy_grid_3d.npz")
    pts, pitch = g["c1_points"], float(g["c1_pitch"])
    origin = np.full(3, -16 * pitch, np.float32)
    for thr, key, n in ((1, "c1_grid", 1000), (2, "c1_grid_thr2", 200)):
        for x in (pts[:n], torch.from_numpy(pts[:n])):  # ndarray and CPU tensor both take the CPU path
            grid = mf.functions.occupancy_grid_3d(x, pitch=pitch, origin=origin, dims=(32, 32, 32), threshold=thr)
            assert isinstance(grid, torch.Tensor) and not grid.is_cuda and grid.dtype == torch.float32
            np.testing.assert_array_equal(grid.numpy(), g[key])
    assert (g["c1_grid"] > 0).sum() > 100


def test_occupancy_grid_3d_cpu_gradient_reaches_the_arg_min_points():
    g = golden("ref_occupancy_grid_3d.npz")
    pts = torch.from_numpy(g["fn_points"].copy()).requires_grad_(True)
    grid = mf.functions.occupancy_grid_3d(pts, pitch=1.0, origin=(0, 0, 0), dims=(8, 8, 8), threshold=2)
    w = torch.from_numpy(np.random.RandomState(0).uniform(-1, 1, (8, 8, 8)).astype(np.float32))
    (grid * w).sum().backward()
    eps, num = 1e-3, np.zeros((16, 3))
    base = g["fn_points"].astype(np.float64)
    for i in range(16):
        for d in range(3):
            vals = []
            for sgn in (1, -1):
                p = base.copy(); p[i, d] += sgn * eps
                y = mf.functions.occupancy_grid_3d(p.astype(np.float32), pitch=1.0, origin=(0, 0, 0), dims=(8, 8, 8), threshold=2)
                vals.append(float((y.double() * w.double()).sum()))
            num[i, d] = (vals[0] - vals[1]) / (2 * eps)
    np.testing.assert_allclose(pts.grad.numpy(), num, atol=5e-2, rtol=5e-2)  # (the reference's own tolerance class)


def test_wrong_dtype_raises_like_check_type_forward():
    with pytest.raises(TypeError):
        mf.functions.occupancy_grid_3d(np.zeros((4, 3), np.float64), pitch=1, origin=(0, 0, 0), dims=(4, 4, 4))


@pytest.mark.parametrize("name", ["ref_average_voxelization_3d.npz", "ref_average_voxelization_3d_model.npz"])
def test_average_voxelization_3d_numpy_path_equals_forward_cpu(name):
    g = golden(name)
    B, D, C = int(g["batch_size"]), 32, g["values"].shape[1]
    origin = g["origin"] if "origin" in g.files else (0, 0, 0)
    pitch = float(g["pitch"]) if "pitch" in g.files else 1.0
    values = torch.from_numpy(g["values"].copy()).requires_grad_(True)
    y, counts = mf.functions.average_voxelization_3d(
        values, g["points"], g["batch_indices"], batch_size=B, origin=origin, pitch=pitch, dimensions=(D, D, D),
        return_counts=True)
    y_ref = _dense((B, C, D, D, D), g["y_nonzero_index"], g["y_nonzero_value"], np.float32)
    c_ref = _dense((B, D, D, D), g["counts_nonzero_index"], g["counts_nonzero_value"], np.int32)
    np.testing.assert_array_equal(counts.numpy(), c_ref)  # bit-exact voxel indices (half-to-even fork)
    np.testing.assert_array_equal(y.detach().numpy(), y_ref)  # the loop's summation order -> bit-exact
    if "gvalues" in g.files:  # backward_cpu (:120-145) on the generator's seeded upstream gradient
        gy = np.random.RandomState(int(g["gy_seed"])).uniform(-1, 1, y.shape).astype(np.float32)
        y.backward(torch.from_numpy(gy))
        np.testing.assert_array_equal(values.grad.numpy(), g["gvalues"])


def test_average_voxelization_3d_numpy_path_errors():
    v, p, b = np.zeros((4, 2), np.float32), np.zeros((4, 3), np.float32), np.zeros(4, np.int32)
    with pytest.raises(ValueError, match="dimensions must be a tuple of 4 integers"):
        mf.functions.average_voxelization_3d(v, p, b, batch_size=1, origin=(0, 0, 0), pitch=1.0, dimensions=[4, 4, 4])
    p2 = p.copy(); p2[1, 1] = np.nan
    with pytest.raises(ValueError, match="points include nan"):
        mf.functions.average_voxelization_3d(v, p2, b, batch_size=1, origin=(0, 0, 0), pitch=1.0, dimensions=(4, 4, 4))
    with pytest.raises(TypeError):
        mf.functions.average_voxelization_3d(v, p, b.astype(np.int64), batch_size=1, origin=(0, 0, 0), pitch=1.0,
                                             dimensions=(4, 4, 4))


def test_cpu_path_does_not_touch_the_oracle():
    import sys
    src = open(os.path.join(os.path.dirname(mf.__file__), "functions", "geometry", "_cpu.py")).read()
    assert "oracle" not in src
    assert not any(m == "oracle" or m.startswith("oracle.") for m in sys.modules
                   if getattr(sys.modules[m], "__file__", None) and "morefusion_amd" in (sys.modules[m].__file__ or ""))
