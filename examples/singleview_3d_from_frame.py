#!/usr/bin/env python
"""RGB-D frame in -> refined poses out: the callback of the reference's ROS node
(ros/src/morefusion_ros/nodes/singleview_3d_pose_estimation.py:113-256) without ROS:
instance crops (HIP, no host loop) -> grid placement -> Model.predict -> arg-max confidence
-> 4x4 transforms.  Synthetic frame, random weights unless ``--model snapshot.npz``."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", help="chainer .npz checkpoint of the reference")
    args = ap.parse_args()

    frame = morefusion.synthetic.make_rgbd_frame(0)
    class_of_instance = dict(zip(frame["instance_ids"].tolist(), [2, 5, 9, 12, 15, 16, 19, 21]))
    to_gpu = lambda x: torch.as_tensor(x).cuda()  # noqa: E731
    crops = morefusion.geometry.instance_crops(
        to_gpu(frame["rgb"]), to_gpu(frame["depth"]), frame["K"], to_gpu(frame["label"]),
        frame["instance_ids"], image_size=256, min_valid=50)
    keep = crops["keep"].cpu().numpy()  # the node's `continue` for inactive / tiny instances
    instance_ids = frame["instance_ids"][keep]
    class_id = torch.tensor([class_of_instance[i] for i in instance_ids.tolist()], dtype=torch.int32).cuda()
    rgb, pcd = crops["rgb"][crops["keep"]], crops["pcd"][crops["keep"]]

    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True)
    if args.model:
        morefusion.serializers.load_npz(args.model, model)
    model = model.cuda().eval()
    grid_nontarget_empty = torch.zeros((len(instance_ids), 32, 32, 32), dtype=torch.bool, device="cuda")
    with torch.no_grad():  # pitch from the class table, origin = median - 15.5 pitch (model.py:195-207)
        quaternion, translation, confidence = model.predict(
            class_id=class_id, rgb=rgb, pcd=pcd, grid_nontarget_empty=grid_nontarget_empty)
    best = confidence.argmax(dim=1)
    ar = torch.arange(len(instance_ids), device=best.device)
    T = morefusion.functions.transformation_matrix(quaternion[ar, best], translation[ar, best])
    for ins, cls, t in zip(instance_ids, class_id.tolist(), T.cpu().numpy()):
        print(f"instance {ins} (class {cls}): translation {np.round(t[:3, 3], 4)}")


if __name__ == "__main__":
    main()
