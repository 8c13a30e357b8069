"""ICC / ICP refinement on the MI355X vs the oracle (C restatement cross-checked with
the NumPy one in tests/test_oracle_c.py).  Real fixture instances + synthetic scenes."""
import numpy as np
import pytest
import torch

from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def scene_args(sc, n=None):
    n = len(sc["points"]) if n is None else n
    return (sc["points"][:n], sc["sdf"][:n], sc["pitch"][:n], sc["origin"][:n],
            sc["grid_target"][:n], sc["grid_nontarget_empty"][:n])


def to_dev(args):
    pts, sdf, pitch, origin, gt, gne = args
    return ([dev(p) for p in pts], [dev(s) for s in sdf], dev(pitch), dev(origin), dev(gt), dev(gne))


def add_between(points, qa, ta, qb, tb):
    """ADD (metrics/average_distance.py:10-13) between two pose sets, float64."""
    Ta = O.transformation_matrix(qa.astype(np.float64), ta.astype(np.float64))
    Tb = O.transformation_matrix(qb.astype(np.float64), tb.astype(np.float64))
    out = []
    for i in range(len(points)):
        p = np.asarray(points[i], dtype=np.float64)
        pa = p @ Ta[i][:3, :3].T + Ta[i][:3, 3]
        pb = p @ Tb[i][:3, :3].T + Tb[i][:3, 3]
        out.append(np.linalg.norm(pa - pb, axis=1).mean())
    return np.array(out)


@pytest.fixture(scope="module")
def scene8(fixtures3):
    return mf.synthetic.make_icc_scene(8, seed=0, fixtures=fixtures3)


@pytest.mark.parametrize("n", [1, 3, 8])
def test_icc_link_loss_and_grad_vs_oracle(scene8, n):
    args = scene_args(scene8, n)
    link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02)
    link.to_gpu()
    loss = link(*to_dev(args))
    loss.backward()
    q0 = link.quaternion.detach().cpu().numpy()
    t0 = link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*args, q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(float(loss), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(link.quaternion.grad.cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)


def test_icc_step_by_step_api_matches_fused_refine(scene8):
    """loss.backward(); optimizer.update(); link.zerograds() (the reference's loop) and
    link.refine() (one hipGraph) walk the same trajectory."""
    n, iters = 3, 10
    args = to_dev(scene_args(scene8, n))
    a = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    opt = mf.optimizers.Adam(alpha=0.01).setup(a)
    a.translation.update_rule.hyperparam.alpha *= 0.1
    losses_a = []
    for _ in range(iters):
        loss = a(*args)
        loss.backward()
        opt.update()
        a.zerograds()
        losses_a.append(float(loss))
    b = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    losses_b, traj = b.refine(*args, n_iter=iters, return_history=True)
    np.testing.assert_allclose(losses_b.cpu().numpy(), losses_a, rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(b.quaternion.detach().cpu().numpy(), a.quaternion.detach().cpu().numpy(), atol=2e-4)
    np.testing.assert_allclose(b.translation.detach().cpu().numpy(), a.translation.detach().cpu().numpy(), atol=2e-4)


@pytest.mark.parametrize("n,iters", [(3, 30), (8, 100)])
def test_icc_refine_teacher_forced_vs_oracle(scene8, n, iters):
    """BASELINE config 3 (n=8, 100 iterations).  The ICC objective is non-smooth
    (arg-min / round / max) and Adam normalises the step, so ANY two float32
    implementations that differ in summation order walk apart after a handful of
    iterations -- the oracle's own NumPy and C restatements end ~7 mm apart after 30
    iterations (tests/test_oracle_c.py), and the reference's float atomics are not even
    run-to-run reproducible.  Parity is therefore pinned per iteration: starting from the
    oracle's state (pose + Adam moments) at iteration k, one fused GPU step must land on
    the oracle's iterate k+1 (pose within 1e-6, i.e. ADD << 1e-4 m; loss within 2e-5)."""
    args = scene_args(scene8, n)
    dargs = to_dev(args)
    link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    q0 = link.quaternion.detach().cpu().numpy().copy()
    t0 = link.translation.detach().cpu().numpy().copy()
    q_o, t_o, losses_o, traj_o, hist_o = OC.icc_refine(*args, q0, t0, n_iter=iters, sdf_offset=0.02,
                                                      return_adam=True)
    scenes = link._pack(*dargs)
    worst_pose, worst_add = 0.0, 0.0
    for k in range(iters - 1):
        q, t = dev(traj_o[k, :, :4]), dev(traj_o[k, :, 4:])
        m, v = dev(hist_o[k, 0]), dev(hist_o[k, 1])
        loss = torch.empty(1, 1).cuda()
        scenes.refine(q, t, m, v, 1, step0=k, alpha_q=0.01, alpha_t=0.001, losses=loss)
        np.testing.assert_allclose(float(loss), losses_o[k], rtol=2e-5, atol=2e-6, err_msg=f"iter {k}")
        got = torch.cat([q, t], 1).cpu().numpy()
        worst_pose = max(worst_pose, np.abs(got - traj_o[k + 1]).max())
        worst_add = max(worst_add, add_between(args[0], got[:, :4], got[:, 4:], traj_o[k + 1][:, :4],
                                               traj_o[k + 1][:, 4:]).max())
    # a step moves q by ~alpha=1e-2 and t by ~1e-3: 1e-5 is < 1 % of a step
    assert worst_pose < 1e-5, worst_pose
    assert worst_add < 1e-5, worst_add  # north_star: ADD within 1e-4 m


def test_icc_refine_free_running_quality(scene8):
    """Free-running 100 iterations: same first iterations as the oracle, a comparable
    final loss, and the synthetic objects (known ground truth) end closer to it."""
    args = scene_args(scene8)
    link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"], sdf_offset=0.02).to_gpu()
    q0 = link.quaternion.detach().cpu().numpy().copy()
    t0 = link.translation.detach().cpu().numpy().copy()
    losses, _ = link.refine(*to_dev(args), n_iter=100, return_history=True)
    losses = losses.cpu().numpy()
    q_o, t_o, losses_o, _ = OC.icc_refine(*args, q0, t0, n_iter=100, sdf_offset=0.02)
    np.testing.assert_allclose(losses[:4], losses_o[:4], rtol=1e-5, atol=1e-6)
    assert losses[-1] < losses[0] - 0.05
    assert abs(losses[-10:].mean() - losses_o[-10:].mean()) < 0.03
    qg, tg = link.quaternion.detach().cpu().numpy(), link.translation.detach().cpu().numpy()
    syn = [i for i, T in enumerate(scene8["transform_gt"]) if T is not None]
    Tgt = np.stack([scene8["transform_gt"][i] for i in syn]).astype(np.float64)
    pts = [args[0][i] for i in syn]

    def add_to_gt(q, t):
        T = O.transformation_matrix(q[syn].astype(np.float64), t[syn].astype(np.float64))
        return np.array([np.linalg.norm((pts[j] @ Tgt[j][:3, :3].T + Tgt[j][:3, 3]) - (pts[j] @ T[j][:3, :3].T + T[j][:3, 3]), axis=1).mean() for j in range(len(syn))])

    a0, ag, ao = add_to_gt(q0, t0), add_to_gt(qg, tg), add_to_gt(q_o, t_o)
    assert ag.mean() < a0.mean(), (a0, ag)
    assert ag.mean() < ao.mean() + 2e-3, (ag, ao)


def test_icc_refine_is_bitwise_reproducible(scene8):
    args = to_dev(scene_args(scene8, 8))
    outs = []
    for _ in range(2):
        link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"], sdf_offset=0.02).to_gpu()
        link.refine(*args, n_iter=20)
        outs.append(torch.cat([link.quaternion.data, link.translation.data], 1).cpu().numpy())
    np.testing.assert_array_equal(outs[0], outs[1])


@pytest.mark.parametrize("general", [False, True], ids=["single_pass", "two_kernel"])
def test_icc_compact_bins_overflow_list_gives_the_same_bits(scene8, monkeypatch, general):
    """Compact bins (a bin holds max(64, P_g / 8) records; the excess goes to the grid's overflow list that its
    tiles scan with the membership test): forcing the capacity down to 5 records per bin -- nearly every record
    overflows -- must not change one bit of a 20-iteration refinement, and the workspace of the 8-object
    scene is an order of magnitude below round 2's nbins x Ns x sum(P) x 16 B (242 MB)."""
    if general:
        monkeypatch.setenv("MF_ICC_GENERAL", "1")
    args = to_dev(scene_args(scene8, 8))
    outs, sizes = [], []
    for cap in (None, "5"):
        if cap:
            monkeypatch.setenv("MF_ICC_BIN_CAP", cap)
        link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"], sdf_offset=0.02).to_gpu()
        losses, _ = link.refine(*args, n_iter=20, return_history=True)
        outs.append((torch.cat([link.quaternion.data, link.translation.data], 1).cpu().numpy(), losses.cpu().numpy()))
        sizes.append(link._scenes.ws.numel() if hasattr(link, "_scenes") else None)
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    if sizes[0] is not None:
        assert sizes[0] < 60e6, sizes


def test_icc_multi_scene_batch_equals_single_scenes(fixtures3):
    scenes = [mf.synthetic.make_icc_scene(4, seed=s, fixtures=fixtures3 if s == 0 else None) for s in range(3)]
    dicts = [dict(points=s["points"], sdf=s["sdf"], pitch=s["pitch"], origin=s["origin"],
                  grid_target=s["grid_target"], grid_nontarget_empty=s["grid_nontarget_empty"]) for s in scenes]
    q0 = np.concatenate([np.stack([O.quaternion_from_matrix(T) for T in s["transform_init"]]) for s in scenes]).astype(np.float32)
    t0 = np.concatenate([s["transform_init"][:, :3, 3] for s in scenes]).astype(np.float32)
    batch = mf.contrib.IccScenes(dicts, sdf_offset=0.02)
    q, t = dev(q0), dev(t0)
    m, v = torch.zeros(12, 7).cuda(), torch.zeros(12, 7).cuda()
    losses = torch.empty(15, 3).cuda()
    batch.refine(q, t, m, v, 15, losses=losses)
    for s in range(3):
        single = mf.contrib.IccScenes([dicts[s]], sdf_offset=0.02)
        qs, ts = dev(q0[4 * s:4 * s + 4]), dev(t0[4 * s:4 * s + 4])
        ms, vs = torch.zeros(4, 7).cuda(), torch.zeros(4, 7).cuda()
        ls = torch.empty(15, 1).cuda()
        single.refine(qs, ts, ms, vs, 15, losses=ls)
        np.testing.assert_array_equal(q[4 * s:4 * s + 4].cpu().numpy(), qs.cpu().numpy())
        np.testing.assert_array_equal(t[4 * s:4 * s + 4].cpu().numpy(), ts.cpu().numpy())
        np.testing.assert_array_equal(losses[:, s].cpu().numpy(), ls[:, 0].cpu().numpy())


def test_icp_link_vs_oracle(fixtures3):
    f = fixtures3[2]
    src = f["pcd_cad"].astype(np.float32)
    tgt = (np.argwhere(f["grid_target"] >= 0.5).astype(np.float32) * np.float32(f["pitch"]) + f["origin"]).astype(np.float32)
    link = mf.contrib.IterativeClosestPointLink(f["transform_init"]).to_gpu()
    loss = link(dev(src), dev(tgt))
    loss.backward()
    q0 = link.quaternion.detach().cpu().numpy()
    t0 = link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o = OC.icp_loss_grad(src, tgt, q0, t0)
    np.testing.assert_allclose(float(loss), l_o, rtol=1e-5)
    np.testing.assert_allclose(link.quaternion.grad.cpu().numpy(), gq_o, rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gt_o, rtol=1e-3, atol=1e-5)
    # the reference driver: Adam(0.01), translation x0.1 (check_iterative_closest_point_link.py:40-66)
    opt = mf.optimizers.Adam(alpha=0.01).setup(link)
    link.translation.update_rule.hyperparam.alpha *= 0.1
    first = None
    for _ in range(20):
        link.zerograds()
        loss = link(dev(src), dev(tgt))
        loss.backward()
        opt.update()
        first = float(loss) if first is None else first
    assert float(loss) < first


def test_icc_single_object_scene_and_far_apart_objects():
    """N = 1 (no 'other' grid at all, iterative_collision_check_link.py:62-63) and a scene
    whose objects never overlap (every 'other' grid empty -> NaN guard path, :82)."""
    sc = mf.synthetic.make_icc_scene(2, seed=5)
    one = scene_args(sc, 1)
    link = mf.contrib.IterativeCollisionCheckLink(sc["transform_init"][:1], sdf_offset=0.02).to_gpu()
    loss = link(*to_dev(one))
    loss.backward()
    q0, t0 = link.quaternion.detach().cpu().numpy(), link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*one, q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(float(loss.detach()), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(link.quaternion.grad.cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
    # two objects 10 m apart
    T = sc["transform_init"].copy()
    T[1, :3, 3] += 10.0
    origin = sc["origin"].copy()
    origin[1] += 10.0
    args = (sc["points"], sc["sdf"], sc["pitch"], origin, sc["grid_target"], sc["grid_nontarget_empty"])
    link = mf.contrib.IterativeCollisionCheckLink(T, sdf_offset=0.02).to_gpu()
    loss = link(*to_dev(args))
    loss.backward()
    q0, t0 = link.quaternion.detach().cpu().numpy(), link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*args, q0, t0, sdf_offset=0.02)
    assert np.isfinite(l_o)
    np.testing.assert_allclose(float(loss.detach()), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)


def test_icc_ragged_multi_scene_batch_vs_oracle(fixtures3):
    """Scenes with 1, 3 and 5 objects in ONE batch: per-scene loss and gradients equal the
    oracle's for each scene evaluated alone (ragged scene tables, max_scene_objects = 5)."""
    sizes = [1, 3, 5]
    scenes = [mf.synthetic.make_icc_scene(n, seed=10 + n, fixtures=fixtures3 if n == 3 else None) for n in sizes]
    dicts = [dict(points=s["points"], sdf=s["sdf"], pitch=s["pitch"], origin=s["origin"],
                  grid_target=s["grid_target"], grid_nontarget_empty=s["grid_nontarget_empty"]) for s in scenes]
    q0 = np.concatenate([np.stack([O.quaternion_from_matrix(T) for T in s["transform_init"]]) for s in scenes]).astype(np.float32)
    t0 = np.concatenate([s["transform_init"][:, :3, 3] for s in scenes]).astype(np.float32)
    batch = mf.contrib.IccScenes(dicts, sdf_offset=0.02)
    loss, gq, gt = batch.loss_grad(dev(q0), dev(t0))
    lo = 0
    for k, (n, s) in enumerate(zip(sizes, scenes)):
        l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*scene_args(s), q0[lo:lo + n], t0[lo:lo + n], sdf_offset=0.02)
        np.testing.assert_allclose(float(loss[k]), l_o, rtol=2e-5, atol=1e-6)
        np.testing.assert_allclose(gq[lo:lo + n].cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
        np.testing.assert_allclose(gt[lo:lo + n].cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)
        lo += n


def test_occupancy_registration_vs_oracle_and_converges():
    """contrib/occupancy_registration.py:21-60: loss + gradient at the initial pose against the
    oracle's occupancy_grid_3d forward/backward chained by hand; then the Adam loop lowers it."""
    rs = np.random.RandomState(0)
    pitch, dim = 0.01, 16
    origin = (-0.075, -0.075, -0.075)
    model = rs.uniform(-0.03, 0.03, (300, 3)).astype(np.float32)
    T_gt = np.eye(4, dtype=np.float32)
    T_gt[:3, :3] = mf.synthetic.random_rotation(rs, 0.3)
    T_gt[:3, 3] = (0.01, -0.005, 0.008)
    occ = O.occupancy_grid_3d((model @ T_gt[:3, :3].T + T_gt[:3, 3]).astype(np.float32), pitch=pitch,
                              origin=origin, dims=(dim,) * 3, threshold=1.5)
    grid_target = np.stack([(occ > 0.3), (occ == 0)]).astype(np.float32)
    T0 = np.eye(4, dtype=np.float32)
    reg = mf.contrib.OccupancyRegistration(model, grid_target, pitch=pitch, origin=origin, threshold=1.5,
                                           transform_init=T0, alpha=0.01)
    link = reg._optimizer.target
    loss = link(points_source=reg._points_source, grid_target=reg._grid_target, pitch=pitch, origin=origin, threshold=1.5)
    loss.backward()
    # oracle: same composition in NumPy
    q = np.array([1, 0, 0, 0], np.float32)
    t = np.zeros(3, np.float32)
    pw = O.transform_points(model, O.transformation_matrix(q, t))
    g = O.occupancy_grid_3d(pw, pitch=pitch, origin=origin, dims=(dim,) * 3, threshold=1.5)
    occd, unocc = grid_target[0], grid_target[1]
    l_o = (unocc * g).sum() / g.sum() - (occd * g).sum() / occd.sum()
    np.testing.assert_allclose(float(loss.detach()), l_o, rtol=1e-5, atol=1e-6)
    gg = unocc / g.sum() - (unocc * g).sum() / g.sum() ** 2 - occd / occd.sum()
    gp = O.occupancy_grid_3d_backward(gg.astype(np.float32), pw, pitch=pitch, origin=origin, dims=(dim,) * 3, threshold=1.5)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gp.sum(axis=0), rtol=1e-3, atol=1e-5)
    link.cleargrads()
    first = float(loss.detach())
    T = reg.register(iteration=40)
    final = float(link(points_source=reg._points_source, grid_target=reg._grid_target, pitch=pitch,
                       origin=origin, threshold=1.5).detach())
    assert final < first - 0.05, (first, final)
    assert T.shape == (4, 4) and np.linalg.norm(T[:3, 3] - T_gt[:3, 3]) < np.linalg.norm(T_gt[:3, 3])


def test_reference_driver_loop_runs_under_the_chainer_facade(scene8):
    """The loop body of check_iterative_collision_check_link.py:29-79, in its own idiom
    (cuda.to_gpu, chainer.optimizers.Adam, Variable.array, cuda.to_cpu), over the façade."""
    import morefusion_amd as morefusion
    from morefusion_amd import chainer_compat as chainer
    from morefusion_amd.chainer_compat import cuda

    n = 3
    data = {k: v[:n] for k, v in scene8.items()}
    points = [cuda.to_gpu(p).float() for p in data["points"]]
    sdf = [cuda.to_gpu(s).float() for s in data["sdf"]]
    pitch = cuda.to_gpu(np.asarray(data["pitch"], np.float32))
    origin = cuda.to_gpu(np.asarray(data["origin"], np.float32))
    grid_target = cuda.to_gpu(np.asarray(data["grid_target"], np.float32))
    grid_nontarget_empty = cuda.to_gpu(np.asarray(data["grid_nontarget_empty"], np.float32))

    link = morefusion.contrib.IterativeCollisionCheckLink(data["transform_init"], sdf_offset=0.02)
    link.to_gpu()
    optimizer = chainer.optimizers.Adam(alpha=0.01)
    optimizer.setup(link)
    link.translation.update_rule.hyperparam.alpha *= 0.1
    losses = []
    for i in range(5):
        transform = morefusion.functions.transformation_matrix(link.quaternion, link.translation)
        transform = cuda.to_cpu(transform.array)
        assert transform.shape == (n, 4, 4) and isinstance(transform, np.ndarray)
        loss = link(points, sdf, pitch, origin, grid_target, grid_nontarget_empty)
        loss.backward()
        optimizer.update()
        link.zerograds()
        losses.append(float(cuda.to_cpu(loss.array)))
    ref = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    losses_ref, _ = ref.refine(*to_dev(scene_args(scene8, n)), n_iter=5, return_history=True)
    np.testing.assert_allclose(losses, losses_ref.cpu().numpy(), rtol=1e-4, atol=1e-6)


def test_icc_refine_teacher_forced_vs_committed_golden(fixtures3):
    """Same per-iteration pin as above, against the COMMITTED trajectory of the three recorded
    fixtures (tests/golden/oracle_icc_icp_trajectories.npz, oracle/gen_golden_icc.py) instead of
    an oracle run at test time."""
    from conftest import golden
    g = golden("oracle_icc_icp_trajectories.npz")
    sc = mf.synthetic.make_icc_scene(3, seed=0, fixtures=fixtures3)
    args = scene_args(sc)
    link = mf.contrib.IterativeCollisionCheckLink(sc["transform_init"], sdf_offset=0.02).to_gpu()
    scenes = link._pack(*to_dev(args))
    traj, adam, losses = g["icc_traj"], g["icc_adam"], g["icc_losses"]
    worst = 0.0
    for k in range(99):
        q, t = dev(traj[k, :, :4]), dev(traj[k, :, 4:])
        m, v = dev(adam[k, 0]), dev(adam[k, 1])
        loss = torch.empty(1, 1).cuda()
        scenes.refine(q, t, m, v, 1, step0=k, alpha_q=0.01, alpha_t=0.001, losses=loss)
        np.testing.assert_allclose(float(loss), losses[k], rtol=2e-5, atol=2e-6, err_msg=f"iter {k}")
        worst = max(worst, np.abs(torch.cat([q, t], 1).cpu().numpy() - traj[k + 1]).max())
    assert worst < 1e-5, worst
    # and the ICP driver's committed iterates: loss + gradient of the fused kernel
    f = fixtures3[2]
    target = dev((np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32))
    source = dev(f["pcd_cad"].astype(np.float32))
    for k in (0, 7, 29):
        icp = mf.contrib.IterativeClosestPointLink(np.eye(4, dtype=np.float32)).to_gpu()
        with torch.no_grad():
            icp.quaternion.copy_(dev(g["icp_traj"][k, :4]))
            icp.translation.copy_(dev(g["icp_traj"][k, 4:]))
        loss = icp(source, target)
        np.testing.assert_allclose(float(loss.detach()), g["icp_losses"][k], rtol=2e-5)


def test_icp_fused_loop_vs_committed_golden_and_autograd_loop(fixtures3):
    """icp_refine (mf_icp_refine: the driver loop of check_iterative_closest_point_link.py:40-70 on the
    device, every link of the ChainList in one batch) against (a) the same loop driven through
    autograd + optimizers.Adam per link for the first steps, and (b) the committed 30-iterate oracle
    trajectory of fixture 2, reached by a second call that continues the optimiser state."""
    from conftest import golden
    g = golden("oracle_icc_icp_trajectories.npz")
    sets = []
    for i, f in enumerate((fixtures3[2], fixtures3[0], fixtures3[1])):
        tgt = np.ascontiguousarray((np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32))
        src = np.ascontiguousarray(f["pcd_cad"].astype(np.float32)[:: (1, 3, 2)[i]])
        sets.append((f["transform_init"], dev(src), dev(tgt)))
    n0, n1 = 6, 24
    links = [mf.contrib.IterativeClosestPointLink(T).to_gpu() for T, _, _ in sets]
    srcs, tgts = [s for _, s, _ in sets], [t for _, _, t in sets]
    losses = mf.contrib.icp_refine(links, srcs, tgts, n_iter=n0, return_history=True).cpu().numpy()
    np.testing.assert_allclose(losses[:, 0], g["icp_losses"][:n0], rtol=1e-4)
    got = np.r_[links[0].quaternion.detach().cpu().numpy(), links[0].translation.detach().cpu().numpy()]
    np.testing.assert_allclose(got, g["icp_traj"][n0], atol=2e-5)
    for i, (T, src, tgt) in enumerate(sets):
        # (a1) the oracle's loop on this link
        qi = O.quaternion_from_matrix(T).astype(np.float32)
        ti = np.asarray(T)[:3, 3].astype(np.float32).copy()
        opt_o = O.ChainerAdam([qi, ti], [0.01, 0.001])
        for k in range(n0):
            l_o, gq, gt = OC.icp_loss_grad(src.cpu().numpy(), tgt.cpu().numpy(), qi, ti)
            np.testing.assert_allclose(losses[k, i], l_o, rtol=2e-4, err_msg=f"link {i} iter {k}")
            opt_o.update([gq, gt])
        got_i = np.r_[links[i].quaternion.detach().cpu().numpy(), links[i].translation.detach().cpu().numpy()]
        np.testing.assert_allclose(got_i, np.r_[qi, ti], atol=1e-4)
        # (a2) autograd + optimizers.Adam: the same losses; the poses only loosely (a component
        # whose gradient is rounding noise takes an Adam step of full size in either direction)
        ref = mf.contrib.IterativeClosestPointLink(T).to_gpu()
        opt = mf.optimizers.Adam(alpha=0.01).setup(ref)
        ref.translation.update_rule.hyperparam.alpha *= 0.1
        for k in range(n0):
            ref.zerograds()
            loss = ref(src, tgt)
            loss.backward()
            opt.update()
            np.testing.assert_allclose(losses[k, i], float(loss), rtol=2e-3, err_msg=f"link {i} iter {k}")
        np.testing.assert_allclose(links[i].quaternion.detach().cpu().numpy(), ref.quaternion.detach().cpu().numpy(), atol=5e-3)
        np.testing.assert_allclose(links[i].translation.detach().cpu().numpy(), ref.translation.detach().cpu().numpy(), atol=5e-4)
    # the second call continues the same optimiser (moments and step count live on the links)
    more = mf.contrib.icp_refine(links, srcs, tgts, n_iter=n1, return_history=True).cpu().numpy()
    assert links[0]._adam_t == n0 + n1
    np.testing.assert_allclose(more[:, 0], g["icp_losses"][n0:n0 + n1], rtol=1e-3)
    got = np.r_[links[0].quaternion.detach().cpu().numpy(), links[0].translation.detach().cpu().numpy()]
    np.testing.assert_allclose(got, g["icp_final"], atol=2e-4)  # 30 free-running float32 steps


def test_icc_fractional_no_entry_grid_two_kernel_path_vs_oracle(scene8):
    """No-entry grids with values strictly between 0 and 1: maximum(no-entry, other-occupancy) is a
    genuine comparison against the NORMALISED other grid (iterative_collision_check_link.py:83-85),
    the single-pass kernel's polynomial form does not apply; the wrapper detects it at pack time
    and the tile -> accum path reproduces the oracle.  The {0,1} grids of every other test take the
    single-pass path (asserted)."""
    n = 4
    args = list(scene_args(scene8, n))
    rs = np.random.RandomState(0)
    args[5] = (args[5] * rs.uniform(0.05, 1.0, args[5].shape)).astype(np.float32)
    link = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    loss = link(*to_dev(args))
    assert link._scenes.desc.grid_ne_binary == 0
    loss.backward()
    q0 = link.quaternion.detach().cpu().numpy()
    t0 = link.translation.detach().cpu().numpy()
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*args, q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(float(loss.detach()), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(link.quaternion.grad.cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(link.translation.grad.cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)
    losses, _ = link.refine(*to_dev(args), n_iter=5, return_history=True)
    _, _, losses_o, _ = OC.icc_refine(*args, q0, t0, n_iter=5, sdf_offset=0.02)
    np.testing.assert_allclose(losses.cpu().numpy()[:3], losses_o[:3], rtol=1e-4, atol=1e-6)
    binary = mf.contrib.IterativeCollisionCheckLink(scene8["transform_init"][:n], sdf_offset=0.02).to_gpu()
    binary(*to_dev(scene_args(scene8, n)))
    assert binary._scenes.desc.grid_ne_binary == 1
