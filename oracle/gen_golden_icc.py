#!/usr/bin/env python
"""Golden trajectories of the refinement drivers.

The reference's drivers (examples/ycb_video/pose_refinement/
check_iterative_collision_check_link.py:14-79, check_iterative_closest_point_link.py:14-70)
cannot execute here (chainer / cupy / trimesh viewer), so -- as SURVEY 8a prescribes -- the
trajectories of THIS repository's CPU restatement on the reference's three recorded fixtures
(tests/golden/fixture_pose_refinement_*.npz, synthetic SDF) are stored as a small golden file:
``tests/golden/oracle_icc_icp_trajectories.npz``.  It pins the restatement against silent
drift; it is NOT reference output (parity unpinned).

    python oracle/gen_golden_icc.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from morefusion_amd import synthetic  # noqa: E402  (scene assembly only: fixtures + synthetic sdf)
from oracle import oracle_c as OC  # noqa: E402
from oracle import oracle_np as O  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def main():
    fixtures = [dict(np.load(os.path.join(GOLD, f"fixture_pose_refinement_0000000{i}.npz"))) for i in range(3)]
    sc = synthetic.make_icc_scene(3, seed=0, fixtures=fixtures)  # exactly the three recorded instances
    args = (sc["points"], sc["sdf"], sc["pitch"], sc["origin"], sc["grid_target"], sc["grid_nontarget_empty"])
    q0 = np.stack([O.quaternion_from_matrix(T) for T in sc["transform_init"]]).astype(np.float32)
    t0 = sc["transform_init"][:, :3, 3].astype(np.float32)
    OC.set_threads(1)
    q, t, losses, traj, hist = OC.icc_refine(*args, q0, t0, n_iter=100, sdf_offset=0.02, return_adam=True)

    # A18: ICP of fixture 2 (target = occupied voxel centres, source = its CAD points), 30 steps
    f = fixtures[2]
    target = (np.argwhere(f["grid_target"] >= 0.5) * f["pitch"] + f["origin"]).astype(np.float32)
    source = f["pcd_cad"].astype(np.float32)
    qi = O.quaternion_from_matrix(f["transform_init"]).astype(np.float32)
    ti = f["transform_init"][:3, 3].astype(np.float32)
    opt = O.ChainerAdam([qi, ti], [0.01, 0.001])  # Adam(alpha=0.01), translation alpha x 0.1 (:40-43)
    icp_losses, icp_traj = [], []
    for _ in range(30):
        icp_traj.append(np.r_[qi, ti])
        loss, gq, gt = OC.icp_loss_grad(source, target, qi, ti)
        icp_losses.append(loss)
        opt.update([gq, gt])  # in place
    np.savez_compressed(
        os.path.join(GOLD, "oracle_icc_icp_trajectories.npz"),
        icc_losses=losses, icc_traj=traj, icc_adam=hist, icc_final=np.concatenate([q, t], axis=1),
        icp_losses=np.array(icp_losses, np.float32), icp_traj=np.array(icp_traj, np.float32),
        icp_final=np.r_[qi, ti].astype(np.float32))
    print("icc loss", losses[0], "->", losses[-1], " icp loss", icp_losses[0], "->", icp_losses[-1])


if __name__ == "__main__":
    main()
