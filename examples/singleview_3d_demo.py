#!/usr/bin/env python
"""singleview_3d inference -- counterpart of the reference's
examples/ycb_video/singleview_3d/demo.py:70-112: batch example dicts, call
model.predict(class_id, rgb, pcd, pitch, origin, grid_nontarget_empty), take the
arg-max-confidence pose per object.  Synthetic examples; ``--model snapshot.npz`` loads a
Chainer checkpoint of the reference (demo.py:51), otherwise random weights (no pretrained file
is reachable offline)."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import morefusion_amd as morefusion  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import Model  # noqa: E402


def main(batch_size=2, checkpoint=None):
    torch.manual_seed(0)
    model = Model(n_fg_class=21, with_occupancy=True)
    if checkpoint:
        morefusion.serializers.load_npz(checkpoint, model)
    model = model.cuda().eval()
    examples = morefusion.synthetic.make_singleview_batch(batch_size, seed=0)
    inputs = {k: torch.as_tensor(examples[k]).cuda()
              for k in ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty")}
    with torch.no_grad():
        quaternion_pred, translation_pred, confidence_pred = model.predict(**inputs)
    indices = confidence_pred.argmax(dim=1)
    ar = torch.arange(batch_size, device=indices.device)
    T = morefusion.functions.transformation_matrix(quaternion_pred[ar, indices], translation_pred[ar, indices])
    for i in range(batch_size):
        print(f"class {int(inputs['class_id'][i])}: conf {float(confidence_pred[i, indices[i]]):.3f}\n", T[i].cpu().numpy())


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", help="chainer .npz checkpoint (snapshot_model_best_auc.npz)")
    parser.add_argument("--batch-size", type=int, default=2)
    args = parser.parse_args()
    main(args.batch_size, args.model)
