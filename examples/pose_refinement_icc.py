#!/usr/bin/env python
"""ICC driver -- this repository's counterpart of the reference's
examples/ycb_video/pose_refinement/check_iterative_collision_check_link.py:14-79
(same argument marshalling, hyper-parameters and iteration count; no viewer).

Inputs: the three real fixture instances the reference ships (tests/golden/) plus
synthetic primitives; their SDF values are synthetic (the YCB SDFs are a download).
  --mode step   : the reference's loop (loss.backward(); optimizer.update(); zerograds())
  --mode fused  : link.refine() -- the whole loop as one hipGraph on the device
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--mode", choices=["step", "fused"], default="fused")
    ap.add_argument("--objects", type=int, default=8)
    ap.add_argument("--iters", type=int, default=100)
    args = ap.parse_args()

    gold = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
    fixtures = [dict(np.load(os.path.join(gold, f"fixture_pose_refinement_0000000{i}.npz"))) for i in range(3)]
    data = morefusion.synthetic.make_icc_scene(args.objects, seed=0, fixtures=fixtures)

    to_gpu = lambda x: torch.as_tensor(np.ascontiguousarray(x)).cuda()  # noqa: E731
    points = [to_gpu(p) for p in data["points"]]
    sdf = [to_gpu(s) for s in data["sdf"]]
    pitch, origin = to_gpu(data["pitch"]), to_gpu(data["origin"])
    grid_target = to_gpu(data["grid_target"])
    grid_nontarget_empty = to_gpu(data["grid_nontarget_empty"])

    link = morefusion.contrib.IterativeCollisionCheckLink(data["transform_init"], sdf_offset=0.02)
    link.to_gpu()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.mode == "step":
        optimizer = morefusion.optimizers.Adam(alpha=0.01)
        optimizer.setup(link)
        link.translation.update_rule.hyperparam.alpha *= 0.1
        losses = []
        for i in range(args.iters):
            loss = link(points, sdf, pitch, origin, grid_target, grid_nontarget_empty)
            loss.backward()
            optimizer.update()
            link.zerograds()
            losses.append(float(loss.detach()))
    else:
        losses, _ = link.refine(points, sdf, pitch, origin, grid_target, grid_nontarget_empty,
                                n_iter=args.iters, return_history=True)
        losses = losses.cpu().tolist()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    transform = morefusion.functions.transformation_matrix(link.quaternion, link.translation)
    print(f"{args.mode}: {args.objects} objects x {args.iters} iterations in {dt * 1e3:.1f} ms "
          f"(first call includes graph capture); loss {losses[0]:.4f} -> {losses[-1]:.4f}")
    print("refined transform[0]:\n", transform[0].detach().cpu().numpy())


if __name__ == "__main__":
    main()
