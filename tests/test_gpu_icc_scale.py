"""ICC at BASELINE config 4's per-GPU size and beyond one GPU's usual scene size, on the MI355X.

(a) configs[3] "64 objects x 32^3 grids sharded across 8 GPUs, RCCL pose all-gather": one GPU's share at its
    largest -- 8 scenes x 8 objects in ONE ``IccScenes`` batch, 100 iterations, the refined poses through
    ``all_gather_poses_equal`` over the ``nccl`` backend (RCCL) at world size 1.  Per-scene results must be
    bit-equal to the single-scene runs, and teacher-forced steps must land on the C oracle's iterates.
(b) scenes of 40 and 64 objects (the reference, contrib/iterative_collision_check_link.py:31-99, has no object
    limit; this implementation's is 64): loss + gradients vs the C oracle on both iteration layouts (the
    single-pass kernel and the two-kernel path), and fused refinement steps vs the oracle's."""
import os

import numpy as np
import pytest
import torch

from oracle import oracle_c as OC
from oracle import oracle_np as O

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402


def dev(x):
    return torch.as_tensor(np.ascontiguousarray(x)).cuda()


def _dict(s):
    return dict(points=s["points"], sdf=s["sdf"], pitch=s["pitch"], origin=s["origin"],
                grid_target=s["grid_target"], grid_nontarget_empty=s["grid_nontarget_empty"])


def _args(s):
    return (s["points"], s["sdf"], s["pitch"], s["origin"], s["grid_target"], s["grid_nontarget_empty"])


def _pose0(scenes):
    q0 = np.concatenate([np.stack([O.quaternion_from_matrix(T) for T in s["transform_init"]]) for s in scenes])
    t0 = np.concatenate([s["transform_init"][:, :3, 3] for s in scenes])
    return q0.astype(np.float32), t0.astype(np.float32)


@pytest.fixture(scope="module")
def nccl_world_of_one():
    import torch.distributed as dist
    if dist.is_initialized():
        yield dist
        return
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    from morefusion_amd.parallel import free_port
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{free_port()}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    yield dist
    dist.destroy_process_group()


def test_config4_share_8_scenes_of_8_objects_through_rccl_gather(fixtures3, nccl_world_of_one):
    S, N, iters = 8, 8, 100
    scenes = [mf.synthetic.make_icc_scene(N, seed=20 + s, fixtures=fixtures3 if s == 0 else None) for s in range(S)]
    dicts = [_dict(s) for s in scenes]
    q0, t0 = _pose0(scenes)
    batch = mf.contrib.IccScenes(dicts, sdf_offset=0.02)
    assert batch.n_objects == S * N and batch.desc.grid_ne_binary == 1
    q, t = dev(q0), dev(t0)
    m, v = torch.zeros(S * N, 7).cuda(), torch.zeros(S * N, 7).cuda()
    losses = torch.empty(iters, S).cuda()
    batch.refine(q, t, m, v, iters, losses=losses)
    from morefusion_amd.parallel import all_gather_poses_equal
    local = torch.cat([q, t], 1)
    gathered = all_gather_poses_equal(local, always=True)      # RCCL all_gather_into_tensor, world of one
    assert gathered.data_ptr() != local.data_ptr() and gathered.shape == (S * N, 7)
    got = gathered.cpu().numpy()
    np.testing.assert_array_equal(got, local.cpu().numpy())
    assert np.isfinite(got).all() and np.isfinite(losses.cpu().numpy()).all()
    # (1) every scene of the batch walks exactly the trajectory it walks alone
    for s in (0, 3, 7):
        single = mf.contrib.IccScenes([dicts[s]], sdf_offset=0.02)
        qs, ts = dev(q0[N * s:N * s + N]), dev(t0[N * s:N * s + N])
        ms, vs = torch.zeros(N, 7).cuda(), torch.zeros(N, 7).cuda()
        ls = torch.empty(iters, 1).cuda()
        single.refine(qs, ts, ms, vs, iters, losses=ls)
        np.testing.assert_array_equal(got[N * s:N * s + N, :4], qs.cpu().numpy())
        np.testing.assert_array_equal(got[N * s:N * s + N, 4:], ts.cpu().numpy())
        np.testing.assert_array_equal(losses[:, s].cpu().numpy(), ls[:, 0].cpu().numpy())
    # (2) teacher-forced: from the oracle's state of ALL 64 objects at iteration k, one fused step of the whole
    #     batch lands on the oracle's iterate k+1 of every scene (tests/test_gpu_icc.py has the single-scene form)
    orc = [OC.icc_refine(*_args(s), q0[N * i:N * i + N], t0[N * i:N * i + N], n_iter=iters, sdf_offset=0.02,
                         return_adam=True) for i, s in enumerate(scenes)]
    traj = np.concatenate([o[3] for o in orc], axis=1)          # [iters, 64, 7]
    hist = np.concatenate([o[4] for o in orc], axis=2)          # [iters, 2, 64, 7]
    loss_o = np.stack([o[2] for o in orc], axis=1)              # [iters, 8]
    worst = 0.0
    for k in list(range(0, 12)) + list(range(12, iters - 1, 8)):
        qk, tk = dev(traj[k, :, :4]), dev(traj[k, :, 4:])
        mk, vk = dev(hist[k, 0]), dev(hist[k, 1])
        lk = torch.empty(1, S).cuda()
        batch.refine(qk, tk, mk, vk, 1, step0=k, losses=lk)
        np.testing.assert_allclose(lk.cpu().numpy()[0], loss_o[k], rtol=2e-5, atol=2e-6, err_msg=f"iter {k}")
        worst = max(worst, np.abs(torch.cat([qk, tk], 1).cpu().numpy() - traj[k + 1]).max())
    assert worst < 1e-5, worst


@pytest.mark.parametrize("n_obj,single_pass", [(40, True), (40, False), (64, True), (64, False), (96, True)],
                         ids=["40-single_pass", "40-two_kernel", "64-single_pass", "64-two_kernel", "96-single_pass"])
def test_scene_of_40_and_64_objects_vs_oracle(n_obj, single_pass):
    """(96 objects, round 6: the single-pass path re-uses the collision moments' LDS rows chunk by chunk of 64
    objects; the two-kernel path -- 64-bit object masks -- refuses more than 64, test_cabi.py.)"""
    sc = mf.synthetic.make_icc_scene(n_obj, seed=5)
    sc = dict(sc)
    sc["points"] = [p[::4].copy() for p in sc["points"]]       # (keeps the C oracle's O(N^2 P) pass short)
    sc["sdf"] = [s[::4].copy() for s in sc["sdf"]]
    q0, t0 = _pose0([sc])
    batch = mf.contrib.IccScenes([_dict(sc)], sdf_offset=0.02, single_pass=single_pass)
    assert batch.desc.grid_ne_binary == int(single_pass) and batch.desc.max_scene_objects == n_obj
    loss, gq, gt = batch.loss_grad(dev(q0), dev(t0))
    l_o, gq_o, gt_o, _ = OC.icc_loss_grad(*_args(sc), q0, t0, sdf_offset=0.02)
    np.testing.assert_allclose(float(loss[0]), l_o, rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(gq.cpu().numpy(), gq_o, rtol=2e-3, atol=2e-5)
    np.testing.assert_allclose(gt.cpu().numpy(), gt_o, rtol=2e-3, atol=2e-4)
    assert np.abs(gq_o[42:]).sum() > 0 if n_obj > 43 else True   # the objects past lane 511 / 12 contribute
    assert np.abs(gq_o[64:]).sum() > 0 if n_obj > 64 else True   # ... and those of the second chunk of LDS rows
    # teacher-forced fused steps (pose table of up to 768 words, 64-bit object masks, looped moment reduction)
    iters = 6
    _, _, losses_o, traj_o, hist_o = OC.icc_refine(*_args(sc), q0, t0, n_iter=iters, sdf_offset=0.02,
                                                   return_adam=True)
    for k in range(iters - 1):
        q, t = dev(traj_o[k, :, :4]), dev(traj_o[k, :, 4:])
        m, v = dev(hist_o[k, 0]), dev(hist_o[k, 1])
        lk = torch.empty(1, 1).cuda()
        batch.refine(q, t, m, v, 1, step0=k, losses=lk)
        np.testing.assert_allclose(float(lk), losses_o[k], rtol=2e-5, atol=2e-6, err_msg=f"iter {k}")
        np.testing.assert_allclose(torch.cat([q, t], 1).cpu().numpy(), traj_o[k + 1], rtol=0, atol=1e-5)
