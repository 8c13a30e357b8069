#!/usr/bin/env python
"""ICP driver -- counterpart of the reference's
examples/ycb_video/pose_refinement/check_iterative_closest_point_link.py:14-70:
one link per instance, target = occupied voxel centres of grid_target (:33-38),
summed loss, one chainer-style Adam over all links, translation alpha x0.1.
The fixture's pcd_cad stands in for the unavailable CAD `points.xyz`."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402


def main(iters=100, fused=True):
    """``fused``: run the whole loop on the device in one call (``contrib.icp_refine`` -> ``mf_icp_refine``,
    2 launches per iteration for all links; 100 iterations in ~2 ms); otherwise the reference's loop body
    through autograd + ``optimizers.Adam`` (~140 ms)."""
    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    instances = [dict(np.load(os.path.join(gold, f"fixture_pose_refinement_0000000{i}.npz"))) for i in range(3)]
    links, pcds_cad, pcds_depth = [], [], []
    for instance in instances:
        link = morefusion.contrib.IterativeClosestPointLink(instance["transform_init"].astype(np.float32))
        link.to_gpu()
        links.append(link)
        pcds_cad.append(torch.as_tensor(instance["pcd_cad"].astype(np.float32)).cuda())
        pcd_depth = np.argwhere(instance["grid_target"] >= 0.5)
        pcd_depth = pcd_depth.astype(np.float32) * instance["pitch"] + instance["origin"]
        pcds_depth.append(torch.as_tensor(pcd_depth.astype(np.float32)).cuda())
    if fused:
        losses = morefusion.contrib.icp_refine(links, pcds_cad, pcds_depth, n_iter=iters, alpha=0.01,
                                               translation_alpha_scale=0.1, return_history=True)
        total = losses.sum(dim=1).cpu().numpy()
        for i in list(range(0, iters, 20)) + [iters - 1]:
            print(f"iter {i:3d} loss {total[i]:.5f}")
        return
    chain = torch.nn.ModuleList(links)
    optimizer = morefusion.optimizers.Adam(alpha=0.01)
    optimizer.setup(chain)
    for link in links:
        link.translation.update_rule.hyperparam.alpha *= 0.1
    for i in range(iters):
        loss = 0
        for link, pcd_cad, pcd_depth in zip(links, pcds_cad, pcds_depth):
            loss = loss + link(pcd_cad, pcd_depth)
        loss.backward()
        optimizer.update()
        for link in links:
            link.zerograds()
        if i % 20 == 0 or i == iters - 1:
            print(f"iter {i:3d} loss {float(loss.detach()):.5f}")


if __name__ == "__main__":
    main(fused="--autograd" not in sys.argv)
