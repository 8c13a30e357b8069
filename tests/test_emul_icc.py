"""The ICC kernel SOURCE (csrc/icc.hip: k_icc_bin, k_icc_tile, k_icc_accum, k_icc_step and the
hipGraph loop of mf_icc_refine) compiled for the host behind tests/host_emul (fiber emulator)
and checked against the oracle WITHOUT a GPU.  The -m gpu tests (tests/test_gpu_icc.py) repeat
these comparisons on the MI355X at full size; here scenes are kept small enough for the
emulator (one workgroup at a time, every GPU thread a fiber)."""
import os
import sys

import numpy as np
import pytest

from oracle import oracle_c as OC
from oracle import oracle_np as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "host_emul"))
import emul  # noqa: E402

import morefusion_amd.synthetic as synthetic  # noqa: E402

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


@pytest.fixture(scope="module")
def lib():
    return emul.build(["icc.hip"])


@pytest.fixture(params=[1, 0], ids=["single_pass", "two_kernel"])
def sp(request):
    """Both iteration layouts: k_icc_bin -> k_icc_fused (the {0,1} no-entry grids every caller
    passes) and k_icc_bin -> k_icc_tile -> k_icc_accum (any grid values)."""
    return request.param


def _dict(sc):
    return dict(points=sc["points"], sdf=sc["sdf"], pitch=sc["pitch"], origin=sc["origin"],
                grid_target=sc["grid_target"], grid_nontarget_empty=sc["grid_nontarget_empty"])


def _args(sc):
    return (sc["points"], sc["sdf"], sc["pitch"], sc["origin"], sc["grid_target"], sc["grid_nontarget_empty"])


def _pose0(sc):
    q = np.stack([O.quaternion_from_matrix(T) for T in sc["transform_init"]]).astype(np.float32)
    t = sc["transform_init"][:, :3, 3].astype(np.float32).copy()
    return q, t


def test_icc_kernel_source_loss_and_grad_vs_oracle(lib, fixtures3, sp):
    sc = synthetic.make_icc_scene(4, seed=0, fixtures=fixtures3)
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=sp)
    q0, t0 = _pose0(sc)
    loss, gq, gt = S.loss_grad(q0, t0)
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(loss[0], l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq, gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt, gt_o, rtol=2e-3, atol=2e-4)
    # the bins are left empty for the next iteration: a second call gives the same bits
    loss2, gq2, gt2 = S.loss_grad(q0, t0)
    np.testing.assert_array_equal(loss, loss2)
    np.testing.assert_array_equal(gq, gq2)
    np.testing.assert_array_equal(gt, gt2)


def test_icc_kernel_source_refine_teacher_forced_vs_oracle(lib, fixtures3, sp=1):
    """One fused step (bin -> tile -> accum -> step, captured in a graph) from the oracle's state
    at iteration k lands on the oracle's iterate k+1 (same pin as the GPU test)."""
    n, iters = 3, 4
    sc = synthetic.make_icc_scene(n, seed=0, fixtures=fixtures3)
    q0, t0 = _pose0(sc)
    _, _, losses_o, traj_o, hist_o = OC.icc_refine(*_args(sc), q0, t0, n_iter=iters, sdf_offset=0.02,
                                                  return_adam=True)
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=sp)
    for k in range(iters - 1):
        q, t = traj_o[k, :, :4].copy(), traj_o[k, :, 4:].copy()
        m, v = hist_o[k, 0].copy(), hist_o[k, 1].copy()
        loss = np.zeros((1, 1), np.float32)
        S.refine(q, t, m, v, 1, step0=k, losses=loss)
        np.testing.assert_allclose(loss[0, 0], losses_o[k], rtol=2e-5, atol=2e-6)
        assert np.abs(np.concatenate([q, t], 1) - traj_o[k + 1]).max() < 1e-5
    # free-running graph replay: same first losses
    q, t = q0.copy(), t0.copy()
    m, v = np.zeros((n, 7), np.float32), np.zeros((n, 7), np.float32)
    losses = np.zeros((3, 1), np.float32)
    S.refine(q, t, m, v, 3, losses=losses)
    np.testing.assert_allclose(losses[:, 0], losses_o[:3], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("thr,sp", [(3, 0), (4, 1)])
def test_icc_kernel_source_ragged_scenes_and_wider_kernel(lib, thr, sp):
    """Scenes of 1 and 3 objects in one batch (ragged tables, a scene without any 'other' grid)
    and wider TDF kernels (truncated_distance_function.py:36-38, ceil(truncation / pitch) in
    float32 made odd): threshold 4 -> kernel size 5 for every grid; threshold 3 -> 3 or 5
    depending on how 3 * pitch / pitch rounds for each grid's pitch, as in the reference."""
    scenes = [synthetic.make_icc_scene(n, seed=20 + n) for n in (1, 3)]
    S = emul.EmulIccScenes(lib, [_dict(s) for s in scenes], voxel_threshold=thr, sdf_offset=0.02, single_pass=sp)
    q0 = np.concatenate([_pose0(s)[0] for s in scenes])
    t0 = np.concatenate([_pose0(s)[1] for s in scenes])
    loss, gq, gt = S.loss_grad(q0, t0)
    lo = 0
    for k, s in enumerate(scenes):
        n = len(s["points"])
        l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(s), q0[lo:lo + n], t0[lo:lo + n], voxel_threshold=thr,
                                              sdf_offset=0.02)
        np.testing.assert_allclose(loss[k], l_o, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(gq[lo:lo + n], gq_o, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(gt[lo:lo + n], gt_o, rtol=2e-3, atol=2e-4)
        lo += n


@pytest.mark.parametrize("n_iter,sp", [(2, 0), (3, 1)])
def test_icc_kernel_source_loop_equals_single_steps(lib, n_iter, sp):
    """The n_iter loop (optimiser step folded into the next iteration's binning kernel, state
    ping-ponging between the caller's arrays and the workspace copy, last step as its own
    kernel) walks exactly the iterates of n_iter one-iteration calls, for even and odd n_iter;
    losses[k] and traj[k] (pose BEFORE step k) land in the right rows."""
    scenes = [synthetic.make_icc_scene(n, seed=30 + n) for n in (2, 1)]
    S = emul.EmulIccScenes(lib, [_dict(s) for s in scenes], sdf_offset=0.02, single_pass=sp)
    q0 = np.concatenate([_pose0(s)[0] for s in scenes])
    t0 = np.concatenate([_pose0(s)[1] for s in scenes])
    q, t = q0.copy(), t0.copy()
    m, v = np.zeros((3, 7), np.float32), np.zeros((3, 7), np.float32)
    losses = np.zeros((n_iter, 2), np.float32)
    traj = np.zeros((n_iter, 3, 7), np.float32)
    S.refine(q, t, m, v, n_iter, losses=losses, traj=traj)
    q1, t1 = q0.copy(), t0.copy()
    m1, v1 = np.zeros((3, 7), np.float32), np.zeros((3, 7), np.float32)
    for k in range(n_iter):
        np.testing.assert_array_equal(traj[k], np.concatenate([q1, t1], 1))
        l1 = np.zeros((1, 2), np.float32)
        S.refine(q1, t1, m1, v1, 1, step0=k, losses=l1)
        np.testing.assert_array_equal(losses[k], l1[0])
    for a, b in ((q, q1), (t, t1), (m, m1), (v, v1)):
        np.testing.assert_array_equal(a, b)
    assert np.abs(q - q0).max() > 1e-4  # it did move


def _lattice_tie_scene():
    """Two objects whose points sit on a half-voxel lattice exactly symmetric about the voxel
    centres of a power-of-two grid, identity poses: every voxel sees many candidates at EXACTLY
    the same distance -> the winner is decided by the lowest-candidate-id rule alone (random sdf
    makes the loss depend on it)."""
    rs = np.random.RandomState(3)
    pitch = np.float32(1.0 / 128)
    dim = 32
    objs = []
    for k, lo in enumerate([(6, 6, 6), (11, 9, 8)]):
        ax = [(np.arange(2 * 9) * 0.5 + 0.25 + l) for l in lo]
        g = np.stack(np.meshgrid(*ax, indexing="ij"), -1).reshape(-1, 3) * float(pitch)
        rs.shuffle(g)
        objs.append(g.astype(np.float32))
    n = len(objs)
    T = np.tile(np.eye(4, dtype=np.float32), (n, 1, 1))
    return dict(points=objs, sdf=[rs.uniform(-0.01, 0.03, len(p)).astype(np.float32) for p in objs],
                pitch=np.full(n, pitch, np.float32), origin=np.zeros((n, 3), np.float32),
                grid_target=(rs.uniform(size=(n, dim, dim, dim)) < 0.3).astype(np.float32),
                grid_nontarget_empty=(rs.uniform(size=(n, dim, dim, dim)) < 0.5).astype(np.float32),
                transform_init=T)


def test_icc_kernel_source_exact_distance_ties(lib, fixtures3, sp):
    """Exact distance ties in (almost) every voxel: the winner is decided by the lowest-candidate-id
    rule of pass 2 alone; loss and gradients depend on it through the random sdf."""
    sc = _lattice_tie_scene()
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=sp)
    q0, t0 = _pose0(sc)
    loss, gq, gt = S.loss_grad(q0, t0)
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(loss[0], l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq, gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt, gt_o, rtol=2e-3, atol=2e-4)


def test_icc_kernel_source_fractional_no_entry_grid_takes_the_two_kernel_path(lib, fixtures3):
    """A no-entry grid with values strictly between 0 and 1 (maximum(no-entry, other) is then a
    genuine comparison against the normalised other-grid, iterative_collision_check_link.py:83-85):
    the wrapper detects it and the two-kernel path reproduces the oracle."""
    sc = synthetic.make_icc_scene(3, seed=0, fixtures=fixtures3)
    rs = np.random.RandomState(0)
    sc = dict(sc)
    sc["grid_nontarget_empty"] = (sc["grid_nontarget_empty"] * rs.uniform(0.05, 1.0, sc["grid_nontarget_empty"].shape)).astype(np.float32)
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02)
    assert S.desc.grid_ne_binary == 0
    q0, t0 = _pose0(sc)
    loss, gq, gt = S.loss_grad(q0, t0)
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(loss[0], l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq, gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt, gt_o, rtol=2e-3, atol=2e-4)


@pytest.mark.parametrize("n,off", [(3, 0.0), (3, 0.02)])
def test_icc_kernel_source_vs_reference_link_executed(lib, fixtures3, sp, n, off):
    """The ICC kernel text (both iteration layouts) against the REFERENCE's IterativeCollisionCheckLink,
    executed: loss from its forward (K7 CUDA text underneath), gradients from its own backward methods
    (tests/golden/ref_cuda_links.npz, ref_cuda_link_gradients.npz; oracle/gen_golden_cuda.py)."""
    from conftest import golden
    g, gg = golden("ref_cuda_links.npz"), golden("ref_cuda_link_gradients.npz")
    sc = synthetic.make_icc_scene(n, seed=0, fixtures=fixtures3)
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=off, single_pass=sp)
    loss, gq, gt = S.loss_grad(g[f"icc_q_n{n}"], g[f"icc_t_n{n}"])
    np.testing.assert_allclose(loss[0], float(g[f"icc_loss_n{n}_off{off}"]), rtol=2e-5, atol=2e-6)
    wq, wt = gg[f"icc_gq_n{n}_off{off}"], gg[f"icc_gt_n{n}_off{off}"]
    np.testing.assert_allclose(gq, wq, rtol=2e-3, atol=3e-4 * float(np.abs(wq).max()))
    np.testing.assert_allclose(gt, wt, rtol=2e-3, atol=3e-4 * float(np.abs(wt).max()))


def test_icc_bin_overflow_list_gives_the_same_bits(lib, fixtures3, sp, monkeypatch):
    """Compact bins: a bin holds max(64, P_g / 8) records, the rest of a crowded bin goes to the grid's overflow
    list, which every tile of the grid scans with the bin-membership test.  Winners are exact (lowest-id ties)
    and the sums are fixed point, so WHERE a record is stored cannot change a bit: with the capacity forced
    down to 3 records per bin (MF_ICC_BIN_CAP, nearly everything overflows) loss, gradients and a 3-iteration
    refinement equal the default layout's bit for bit -- and the workspace is O(N * sum P)."""
    sc = synthetic.make_icc_scene(3, seed=2, fixtures=fixtures3)
    q0, t0 = _pose0(sc)

    def run():
        S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=sp)
        out = S.loss_grad(q0, t0)
        q, t = q0.copy(), t0.copy()
        m, v = np.zeros((3, 7), np.float32), np.zeros((3, 7), np.float32)
        losses = np.zeros((3, 1), np.float32)
        S.refine(q, t, m, v, 3, losses=losses)
        return out + (q, t, losses), S.ws.nbytes

    ref, nbytes = run()
    monkeypatch.setenv("MF_ICC_BIN_CAP", "3")
    got, _ = run()
    for a_, b_ in zip(ref, got):
        np.testing.assert_array_equal(a_, b_)
    assert np.abs(ref[1]).sum() > 0
    # round 2 reserved nbins * Ns * sum(P) records: 68 * 3 * P * 16 B
    P = sum(p.shape[0] for p in sc["points"])
    assert nbytes < 0.4 * 68 * 3 * P * 16 + 2 * 3 * 32 ** 3 * 8 + (1 << 20)


@pytest.mark.parametrize("n_obj", [40, 64])
def test_icc_scene_of_more_than_32_objects(lib, sp, n_obj):
    """kMaxSceneObjects is 64 since round 3 (the reference has no limit; round 2 rejected > 32): a 40-object scene
    (thinned point sets: the emulator runs one GPU thread at a time) gives the oracle's loss and gradients.
    40 objects: the 64-bit object masks of k_icc_accum and the second trip of k_icc_fused's moment reduction
    (one trip covers 37 objects) are both exercised; 64 objects (the limit): the scene pose table is 768 words,
    more than one word per lane of the 512-lane workgroups that stage it (objects 43.. were uninitialised LDS)."""
    sc = synthetic.make_icc_scene(n_obj, seed=5)
    sc = dict(sc)
    sc["points"] = [p[::20].copy() for p in sc["points"]]
    sc["sdf"] = [s[::20].copy() for s in sc["sdf"]]
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=sp)
    q0, t0 = _pose0(sc)
    loss, gq, gt = S.loss_grad(q0, t0)
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(loss[0], l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq, gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt, gt_o, rtol=2e-3, atol=2e-4)
    assert np.abs(gq).sum() > 0


def test_icc_scene_of_more_than_64_objects_on_the_single_pass_path(lib):
    """Round 6: the single-pass path takes scenes of up to 128 objects -- the collision moments' LDS rows (1664 bytes
    per other object and workgroup: 106 KB at 64) are re-used chunk by chunk of 64 objects, the voxels' collision
    terms wait in registers.  A 72-object scene (thinned point sets) gives the oracle's loss and gradients, including
    the gradients of the objects of the second chunk; the two-kernel path (64-bit object masks) refuses it."""
    import ctypes
    n_obj = 72
    sc = dict(synthetic.make_icc_scene(n_obj, seed=5))
    sc["points"] = [p[::20].copy() for p in sc["points"]]
    sc["sdf"] = [s[::20].copy() for s in sc["sdf"]]
    S = emul.EmulIccScenes(lib, [_dict(sc)], sdf_offset=0.02, single_pass=True)
    assert S.desc.max_scene_objects == n_obj and S.desc.grid_ne_binary == 1
    q0, t0 = _pose0(sc)
    loss, gq, gt = S.loss_grad(q0, t0)
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(loss[0], l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq, gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt, gt_o, rtol=2e-3, atol=2e-4)
    assert np.abs(gq_o[64:]).sum() > 0 and np.abs(gt_o[64:]).sum() > 0  # the second chunk's objects do collide
    general = type(S.desc)()
    ctypes.memmove(ctypes.byref(general), ctypes.byref(S.desc), ctypes.sizeof(general))
    general.grid_ne_binary = 0
    assert lib.mf_icc_workspace_bytes(ctypes.byref(general)) < 0

