"""instance_crops -- RGB-D frame + instance image -> the network's per-object inputs.

Device counterpart of the per-instance host loop at
ros/src/morefusion_ros/nodes/singleview_3d_pose_estimation.py:116-176 and
morefusion/datasets/rgbd_pose_estimation/base.py:112-137 (``pointcloud_from_depth`` ->
``masks_to_bboxes`` -> masked crops -> ``imgviz.centerize`` to 256 x 256): two HIP launches
for all instances (``mf_instance_stats``, ``mf_instance_crops``), no host synchronisation.
Instances the reference would skip (empty mask, fewer than ``min_valid`` valid points) are
reported through ``keep``; their crops are pure padding (rgb 0, points NaN).

``grid_origin`` is the grid placement that follows (base.py:150-153 median,
singleview_3d_pose_estimation.py:186-187 mean): ``center - (dim/2 - 0.5) * pitch``.
"""
import torch

from .. import _lib


def instance_crops(rgb, depth, K, instance_label, instance_ids, image_size=256, min_valid=50):
    """rgb [H,W,3] uint8, depth [H,W] float32 metres (NaN invalid), K 3x3 intrinsics (host),
    instance_label [H,W] int32, instance_ids [n] int32 -> dict(rgb [n,S,S,3] uint8,
    pcd [n,S,S,3] float32, keep [n] bool, bbox [n,4] int32 (y1,x1,y2,x2), n_valid [n])."""
    _lib.require_gpu(rgb, depth, instance_label)
    if rgb.dtype != torch.uint8 or rgb.ndim != 3 or rgb.shape[2] != 3:
        raise TypeError("rgb must be uint8 [H,W,3]")
    if depth.dtype != torch.float32 or depth.shape != rgb.shape[:2]:
        raise TypeError("depth must be float32 [H,W] in metres (NaN = invalid)")
    if instance_label.shape != depth.shape:
        raise TypeError("instance_label must be [H,W]")
    H, W = depth.shape
    dev = depth.device
    ids = torch.as_tensor(instance_ids, dtype=torch.int32).to(dev).contiguous()
    n, S = int(ids.numel()), int(image_size)
    if n > 256:
        raise ValueError("at most 256 instances per call")
    fx, fy, cx, cy = float(K[0][0]), float(K[1][1]), float(K[0][2]), float(K[1][2])
    rgb_c, depth_c = rgb.contiguous(), depth.contiguous()
    label_c = _lib.i32c(instance_label)
    stats = torch.empty((n, 6), dtype=torch.int32, device=dev)
    rgb_out = torch.empty((n, S, S, 3), dtype=torch.uint8, device=dev)
    pcd_out = torch.empty((n, S, S, 3), dtype=torch.float32, device=dev)
    keep = torch.empty((n,), dtype=torch.uint8, device=dev)
    if n:
        lib, st = _lib.lib(), _lib.stream_ptr()
        _lib.check(lib.mf_instance_stats(label_c.data_ptr(), depth_c.data_ptr(), H, W, ids.data_ptr(),
                                         n, stats.data_ptr(), st), "mf_instance_stats")
        _lib.check(lib.mf_instance_crops(rgb_c.data_ptr(), depth_c.data_ptr(), label_c.data_ptr(), H, W,
                                         fx, fy, cx, cy, ids.data_ptr(), stats.data_ptr(), n, S,
                                         int(min_valid), rgb_out.data_ptr(), pcd_out.data_ptr(),
                                         keep.data_ptr(), st), "mf_instance_crops")
    empty = stats[:, 4:5] == 0
    return dict(rgb=rgb_out, pcd=pcd_out, keep=keep.bool(),
                bbox=torch.where(empty, torch.zeros_like(stats[:, :4]), stats[:, :4]),
                n_valid=stats[:, 5])


def grid_origin(pcd, pitch, dim=32, center="median"):
    """pcd [n,S,S,3] (NaN = invalid), pitch [n] -> origin [n,3] of each object's dim^3 grid:
    nan-median (NumPy semantics: mean of the two middle values) or nan-mean of the points,
    minus (dim/2 - 0.5) * pitch.  Objects without any valid point give NaN."""
    n = pcd.shape[0]
    flat = pcd.reshape(n, -1, 3)
    valid = ~torch.isnan(flat).any(dim=2)
    cnt = valid.sum(dim=1)
    if center == "mean":
        c = torch.where(valid[..., None], flat, torch.zeros_like(flat)).sum(dim=1) / cnt[:, None]
    elif center == "median":
        srt = torch.sort(flat, dim=1).values  # NaN sorts last, per coordinate
        lo = ((cnt - 1).clamp(min=0) // 2)[:, None, None].expand(n, 1, 3)
        hi = (cnt // 2).clamp(max=flat.shape[1] - 1)[:, None, None].expand(n, 1, 3)
        c = (0.5 * (srt.gather(1, lo) + srt.gather(1, hi)))[:, 0]
        c = torch.where((cnt > 0)[:, None], c, torch.full_like(c, float("nan")))
    else:
        raise ValueError("center must be 'median' or 'mean'")
    pitch = torch.as_tensor(pitch, dtype=pcd.dtype, device=pcd.device).reshape(n, 1)
    return c - (dim / 2 - 0.5) * pitch
