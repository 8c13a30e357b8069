"""singleview_3d pose network: RGB crop + masked point cloud + no-entry grid -> per-point
(quaternion, translation, confidence).

Restates morefusion/contrib/singleview_3d/models/model.py:11-481 on torch.  The 3-D part
is the hot path: ``average_voxelization_3d`` (HIP) -> occupancy branch + conv3/conv4
(Conv3d -> MIOpen / MFMA) -> ``interpolate_voxel_grid`` (HIP) -> three per-point heads.
Differences that are deliberate:
  * the per-object host loop of ``predict`` (:195-229, one D2H sync per object) is one
    batched gather with a single sync for the point counts (NumPy RNG kept bit-for-bit:
    ``RandomState(1234).permutation(n)[:1000]`` in eval mode);
  * the last PSPNet level (full-resolution 3x3 conv, 1x1 head, log-softmax) is evaluated
    only at the 1000 sampled pixels per object (``PSPNetExtractor.forward_sampled``);
  * interpolated features are produced channels-first ([C, B*P]), the layout the heads
    consume, instead of [B*P, C] + transpose;
  * CAD models / voxel pitches come from an injectable ``models`` provider because the
    YCB downloads (datasets/ycb_video/models.py:33-42) are unreachable offline.
"""
import os
import uuid

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

from .... import _lib
from .... import functions as functions_module
from .... import geometry as geometry_module
from .... import metrics
from ....models import PSPNetExtractor, ResNet18, ResNet18Extractor
from ....synthetic import CLASS_IDS_SYMMETRIC, CLASS_PITCH
from .sparse_conv import SparseVoxelConv3d
from .volumetric_cl import ChannelsLastVolumetric


class PitchTableModels:
    """Stand-in for ``YCBVideoModels``: voxel pitch from the per-class table
    (ros/.../utils/data.h:12-32 == bbox_diagonal/32, models.py:113-115); CAD point clouds
    must be supplied (``pcds`` dict class_id -> [n,3]) for evaluate()/loss()."""

    def __init__(self, pcds=None):
        self._pcds = dict(pcds or {})

    def get_voxel_pitch(self, dimension, class_id):
        return CLASS_PITCH[int(class_id)] * 32.0 / dimension

    def get_pcd(self, class_id):
        if int(class_id) not in self._pcds:
            raise KeyError(f"no CAD point cloud registered for class {class_id}")
        return self._pcds[int(class_id)]


class Model(nn.Module):

    _lambda_confidence = 0.015
    _n_point = 1000
    _voxel_dim = 32

    def __init__(self, *, n_fg_class, pretrained_resnet18=False, with_occupancy=False, loss=None,
                 loss_scale=None, models=None):
        super().__init__()
        self._n_fg_class = n_fg_class
        self._with_occupancy = with_occupancy
        if loss is None:
            loss = "add/add_s"
        if loss in ("add+occupancy", "add/add_s+occupancy"):
            # model.py:437-470: that branch calls pseudo_occupancy_voxelization without its ``sdf``
            # argument (truncated_distance_function.py:181) and raises TypeError in the reference
            raise NotImplementedError(
                f"loss={loss!r}: the occupancy term of the reference cannot run (it omits the required "
                "sdf argument of pseudo_occupancy_voxelization); use 'add' or 'add/add_s'")
        if loss not in ("add", "add/add_s"):
            raise ValueError(f"unknown loss {loss!r}")
        self._loss = loss
        self._models = models or PitchTableModels()
        # evaluate the last PSPNet level only where the network samples it (model.py:222)
        self.sparse_pspnet_tail = True
        # inference: conv3 on fp32 MFMA over the occupied voxels only (csrc/sparseconv.hip)
        self.sparse_conv3 = True
        # inference: the whole volumetric part hand-written in channels-last layout (volumetric_cl.py:
        # occupancy convs, conv3 dense + sparse, conv4 as fp32-MFMA implicit GEMMs, samplers writing
        # into the heads' input matrix); False = round 2's channels-first path (stock conv4 / occ convs)
        self.channels_last_3d = True
        # under torch.autocast(bfloat16) -- training (BASELINE config 5) and --dtype bf16 inference -- the 3-D
        # convolutions and every 1x1 convolution run on the hand-written bf16 MFMA kernels, forward and backward
        # (csrc/gemm_bf16.hip through bf16_ops.py); False = stock MIOpen / hipBLASLt operators under autocast
        self.bf16_kernels = True

        # model.py:50-56: the ImageNet-pretrained chainercv2 ResNet-18 (frozen BatchNorm, no
        # gradient below res2) or the DenseFusion ResNet18.  The pretrained weights are a
        # download: the architecture is built with random init and filled by serializers.load_npz
        self.resnet_extractor = ResNet18Extractor() if pretrained_resnet18 else ResNet18()
        self.pspnet_extractor = PSPNetExtractor()
        self.conv1_rgb = nn.Conv1d(32, 64, 1)
        self.conv1_pcd = nn.Conv1d(3, 8, 1)
        self.conv2_rgb = nn.Conv1d(64, 128, 1)
        self.conv2_pcd = nn.Conv1d(8, 16, 1)
        c_vox = 144
        if with_occupancy:
            self.conv1_occ = nn.Conv3d(1, 8, 3, 1, padding=1)
            self.conv2_occ = nn.Conv3d(8, 16, 3, 1, padding=2, dilation=2)
            c_vox += 16
        self.conv3 = nn.Conv3d(c_vox, 256, 4, 2, padding=1)
        self.conv4 = nn.Conv3d(256, 512, 4, 2, padding=1)
        c_feat = 72 + 144 + 256 + 512
        for name, c_out in (("rot", 4), ("trans", 3), ("conf", 1)):
            setattr(self, f"conv1_{name}", nn.Conv1d(c_feat, 640, 1))
            setattr(self, f"conv2_{name}", nn.Conv1d(640, 256, 1))
            setattr(self, f"conv3_{name}", nn.Conv1d(256, 128, 1))
            setattr(self, f"conv4_{name}", nn.Conv1d(128, n_fg_class * c_out, 1))

    def __getstate__(self):
        """copy.deepcopy / pickle of a model that has predicted: the captured hipGraphs, packed-weight caches and
        workspaces are per-instance run-time state, rebuilt on demand."""
        state = dict(self.__dict__)
        for k in ("_graphed", "_volumetric_cl", "_sparse_conv3_op"):
            state.pop(k, None)
        return state

    # ---- 3-D feature extraction (model.py:93-164) --------------------------------------
    def _voxelize(self, values, points, return_counts=False):
        B, P, _ = values.shape
        assert P == self._n_point
        batch_indices = torch.arange(B, dtype=torch.int32, device=values.device).repeat_interleave(P)
        return functions_module.average_voxelization_3d(
            values.reshape(B * P, -1), points.reshape(B * P, 3).contiguous(), batch_indices,
            batch_size=B, origin=(0, 0, 0), pitch=1.0, dimensions=(self._voxel_dim,) * 3,
            return_counts=return_counts, check_nan=False)

    def _extract(self, values, points, grid_nontarget_empty):
        B, _, P = values.shape
        to_center = (self._voxel_dim / 2.0 - 0.5) - points
        batch_indices = torch.arange(B, dtype=torch.int32, device=values.device).repeat_interleave(P)
        indices = points.transpose(1, 2).reshape(B * P, 3).contiguous()
        batch_start = torch.arange(B + 1, dtype=torch.int32, device=values.device) * P

        h_rgb = F.relu(self.conv1_rgb(values))
        h_pcd = F.relu(self.conv1_pcd(to_center))
        feat1 = torch.cat((h_rgb, h_pcd), dim=1)
        h_rgb = F.relu(self.conv2_rgb(h_rgb))
        h_pcd = F.relu(self.conv2_pcd(h_pcd))
        feat2 = torch.cat((h_rgb, h_pcd), dim=1)
        h_occ = None
        if self._with_occupancy:
            g = grid_nontarget_empty.to(values.dtype)[:, None, :, :, :]
            h_occ = F.relu(self.conv1_occ(g))
            h_occ = F.relu(self.conv2_occ(h_occ))
        if self.sparse_conv3 and not torch.is_grad_enabled() and values.is_cuda:
            # inference: the voxelization's per-voxel chains feed conv3's GEMM rows directly --
            # the dense [B,144,32^3] tensor (151 MB at B = 8, <= 3 % non-zero) is never built
            if getattr(self, "_sparse_conv3_op", None) is None:
                self.__dict__["_sparse_conv3_op"] = SparseVoxelConv3d(self.conv3)
            h = self._sparse_conv3_op.from_points(
                feat2.transpose(1, 2).reshape(B * P, -1).float().contiguous(), indices, batch_indices,
                batch_size=B, h_dense=h_occ, dim=self._voxel_dim)
        else:
            voxelized = self._voxelize(values=feat2.transpose(1, 2).float().contiguous(),
                                       points=points.transpose(1, 2))
            if h_occ is not None:
                voxelized = torch.cat([voxelized.to(h_occ.dtype), h_occ], dim=1)
            h = F.relu(self.conv3(voxelized))
        assert h.shape == (B, 256, 16, 16, 16)
        feat3 = functions_module.interpolate_voxel_grid(h.float(), indices / 2.0, batch_indices,
                                                        channels_first=True, batch_start=batch_start)
        feat3 = feat3.reshape(256, B, P).transpose(0, 1)
        h = F.relu(self.conv4(h))
        assert h.shape == (B, 512, 8, 8, 8)
        feat4 = functions_module.interpolate_voxel_grid(h.float(), indices / 4.0, batch_indices,
                                                        channels_first=True, batch_start=batch_start)
        feat4 = feat4.reshape(512, B, P).transpose(0, 1)
        dt = feat1.dtype
        return torch.cat((feat1, feat2, feat3.to(dt), feat4.to(dt)), dim=1)

    @property
    def xp(self):
        """``model.xp`` (``model.xp.arange / argmax`` in demo.py / evaluate.py): arrays on the model's device."""
        from ....chainer_compat import link_xp
        return link_xp(self)

    # ---- point selection (model.py:191-230) ---------------------------------------------
    _eval_keep_cache = {}

    def _keep_indices(self, n_point):
        """The reference's subsample / pad of the n valid pixels (model.py:208-219).  In eval
        mode it seeds a fresh ``RandomState(1234)`` per object, i.e. it is a pure function of
        n: memoised, so the GPU does not idle behind a host-side MT19937 permutation."""
        if n_point == 0:
            raise ValueError("an example has no valid point")
        if not self.training:
            hit = Model._eval_keep_cache.get((n_point, self._n_point))
            if hit is not None:
                return hit
        random_state = np.random.mtrand._rand if self.training else np.random.RandomState(1234)
        if n_point >= self._n_point:
            keep = random_state.permutation(n_point)[: self._n_point]
        else:
            keep = np.r_[np.arange(n_point),
                         random_state.randint(0, n_point, self._n_point - n_point)]
        keep = keep.astype(np.int64)
        if not self.training and len(Model._eval_keep_cache) < 4096:
            Model._eval_keep_cache[(n_point, self._n_point)] = keep
        return keep

    def _select_points(self, pcd):
        """pcd [B,H,W,3] -> flat pixel indices [B,P]: the row-major list of the pixels without a
        NaN coordinate (``where(mask)``, model.py:195) from one launch of ``mf_valid_pixel_order``,
        then the reference's NumPy-RNG subsample / pad of it."""
        B, HW = pcd.shape[0], pcd.shape[1] * pcd.shape[2]
        _lib.require_gpu(pcd)
        pcd = _lib.f32c(pcd)
        order = torch.empty((B, HW), dtype=torch.int32, device=pcd.device)
        counts = torch.empty((B,), dtype=torch.int32, device=pcd.device)
        _lib.check(_lib.lib().mf_valid_pixel_order(pcd.data_ptr(), B, HW, order.data_ptr(), counts.data_ptr(),
                                                   _lib.stream_ptr()), "mf_valid_pixel_order")
        return self._subsample(order, counts.cpu().numpy())  # the one host sync (the RNG needs n_point)

    def _subsample(self, order, counts):
        """order [B,HW] (valid pixels first, row-major), counts [B] on the host -> [B,P] int64:
        ``iy[keep], ix[keep]`` of model.py:207-220 as flat indices."""
        keep = torch.from_numpy(np.stack([self._keep_indices(int(c)) for c in counts])).to(order.device)
        return torch.gather(order, 1, keep).long()

    def predict(self, *, class_id, rgb, pcd, pitch=None, origin=None, grid_nontarget_empty=None):
        B = rgb.shape[0]
        dev = rgb.device
        if pitch is None:
            pitch = torch.tensor([self._models.get_voxel_pitch(self._voxel_dim, int(c))
                                  for c in class_id.tolist()], dtype=torch.float32, device=dev)
        else:
            pitch = torch.as_tensor(pitch, dtype=torch.float32, device=dev)
        if origin is None:
            # model.py:202-207 (per-object median of the valid points), batched on device
            origin = geometry_module.grid_origin(pcd.float(), pitch, dim=self._voxel_dim)
        else:
            origin = torch.as_tensor(origin, dtype=torch.float32, device=dev)
        pix = self._select_points(pcd)  # [B,P]; the one host synchronisation
        return self._predict_device(torch.as_tensor(class_id, device=dev), rgb, pcd, pix, pitch,
                                    origin, grid_nontarget_empty)

    def predict_graphed(self, *, class_id, rgb, pcd, pitch=None, origin=None, grid_nontarget_empty=None,
                        pix=None, clone=True):
        """``predict`` with everything after the point selection replayed from ONE hipGraph per input shape
        (BASELINE config 2, batch = 1: ~300 short launches whose host-side launch cost exceeds their GPU time).
        The first call of a shape warms up (MIOpen solver search) and captures; later calls copy the inputs
        into the graph's static buffers (skipped for tensors that still live at the captured address) and
        replay.  ``pix`` [B,P]: a point selection computed ahead (``select_points_async``), which removes the
        call's only host synchronisation.  ``clone=False`` returns the graph's static output tensors (valid
        until the next call).  (Round 3's intermittent replay faults came from memset nodes in the captured graph;
        the kernels fill through launches since round 4 -- DESIGN.md 6 -- and replays are clean with MIOpen's find
        mode on or off.)"""
        if self.training or torch.is_grad_enabled():
            raise RuntimeError("predict_graphed is an inference path: call under torch.no_grad() in eval mode")
        dev = rgb.device
        if pitch is None:
            pitch = torch.tensor([self._models.get_voxel_pitch(self._voxel_dim, int(c))
                                  for c in class_id.tolist()], dtype=torch.float32, device=dev)
        pitch = torch.as_tensor(pitch, dtype=torch.float32, device=dev)
        if origin is None:
            origin = geometry_module.grid_origin(pcd.float(), pitch, dim=self._voxel_dim)
        origin = torch.as_tensor(origin, dtype=torch.float32, device=dev)
        if pix is None:
            pix = self._select_points(pcd)
        if getattr(self, "_graphed", None) is None:
            from .graphed import GraphedPredict
            self.__dict__["_graphed"] = GraphedPredict(self)
        outs = self._graphed(torch.as_tensor(class_id, device=dev), rgb, pcd, pix, pitch, origin,
                             grid_nontarget_empty)
        return tuple(o.clone() for o in outs) if clone else outs

    def select_points_async(self, pcd, stream=None):
        """EXPERIMENTAL (measured slower than the synchronous selection at batch 1: 2.1-2.3 vs 1.8 ms per frame, the
        cross-stream hand-over costs more than the 30 us it hides; kept for larger batches / pipelines that already
        own a side stream).  Point selection of a FUTURE frame on a side stream: launches ``mf_valid_pixel_order`` and the
        device-to-host copy of the counts into pinned memory, returns a handle whose ``result()`` finishes
        the host-side RNG subsample.  Issued while the network of the current frame runs, the host
        synchronisation of ``predict`` hides behind it (demo.py:80-100 processes frames in sequence)."""
        B, HW = pcd.shape[0], pcd.shape[1] * pcd.shape[2]
        _lib.require_gpu(pcd)
        stream = stream or torch.cuda.Stream(device=pcd.device)
        stream.wait_stream(torch.cuda.current_stream())
        model = self

        class _Pending:
            def __init__(self):
                with torch.cuda.stream(stream):
                    p = _lib.f32c(pcd)
                    self.order = torch.empty((B, HW), dtype=torch.int32, device=pcd.device)
                    counts = torch.empty((B,), dtype=torch.int32, device=pcd.device)
                    _lib.check(_lib.lib().mf_valid_pixel_order(p.data_ptr(), B, HW, self.order.data_ptr(),
                                                               counts.data_ptr(), stream.cuda_stream),
                               "mf_valid_pixel_order")
                    self.counts_host = torch.empty((B,), dtype=torch.int32, pin_memory=True)
                    self.counts_host.copy_(counts, non_blocking=True)
                    self.done = torch.cuda.Event()
                    self.done.record(stream)
                    self._keep = (p, counts)

            def result(self):
                self.done.synchronize()
                with torch.cuda.stream(stream):
                    pix = model._subsample(self.order, self.counts_host.numpy())
                cur = torch.cuda.current_stream()
                cur.wait_stream(stream)
                pix.record_stream(cur)  # allocated on the side stream, consumed on the caller's
                return pix

        return _Pending()

    def _predict_device(self, class_id, rgb, pcd, pix, pitch, origin, grid_nontarget_empty):
        """Everything after point selection: pure device work, no host synchronisation."""
        # inference on the channels-last kernels: the PSPNet tail hands over feature ROWS [n,32] (one launch)
        rows = (self.channels_last_3d and self.sparse_conv3 and self.sparse_pspnet_tail and rgb.is_cuda
                and not self.training  # (forward_sampled_rows has no dropout: train mode keeps the reference's)
                and not torch.is_grad_enabled() and not torch.is_autocast_enabled()
                and os.environ.get("MF_NO_TAIL_KERNEL") != "1")  # (A/B knob: the torch formulation of the tail)
        values, points = self._backbone_features(rgb, pcd, pix, rows=rows)
        return self._pose_from_features(class_id, values, points, pitch, origin, grid_nontarget_empty)

    def _backbone_features(self, rgb, pcd, pix, rows=False):
        """The stock 2-D part (ResNet18 + PSPNet on MIOpen) and the gathers at the sampled pixels:
        -> per-point image features [B,32,P] (``rows=True``: [B*P,32] from the fused tail kernel) and
        camera-frame points [B,3,P]."""
        B = rgb.shape[0]
        rgb = rgb.permute(0, 3, 1, 2)  # (uint8 or float: the extractor normalises the image as it arrives)
        pcd = pcd.float().permute(0, 3, 1, 2)
        if self.sparse_pspnet_tail and rows:
            values = self.pspnet_extractor.forward_sampled_rows(self.resnet_extractor(rgb), pix)
        elif self.sparse_pspnet_tail:
            # last PSPNet level evaluated only at the sampled pixels (identical features)
            values = self.pspnet_extractor.forward_sampled(self.resnet_extractor(rgb), pix)
        else:
            h_rgb = self.pspnet_extractor(self.resnet_extractor(rgb))
            values = torch.gather(h_rgb.reshape(B, h_rgb.shape[1], -1), 2,
                                  pix[:, None, :].expand(B, h_rgb.shape[1], -1))
        # NaN-masked pixels are never selected; nan_to_num keeps the gather capture-safe
        points = torch.gather(pcd.reshape(B, 3, -1), 2, pix[:, None, :].expand(B, 3, -1))
        return values, points

    def _pose_from_features(self, class_id, values, points, pitch, origin, grid_nontarget_empty):
        """The volumetric part (the hand-written path of the network): voxelize -> occupancy
        branch + conv3/conv4 -> trilinear sampling -> the three per-point heads."""
        if (self.bf16_kernels and values.is_cuda and torch.is_autocast_enabled()
                and torch.get_autocast_dtype("cuda") == torch.bfloat16):
            return self._pose_from_features_bf16(class_id, values, points, pitch, origin, grid_nontarget_empty)
        if self.channels_last_3d and self.sparse_conv3 and not torch.is_grad_enabled() and values.is_cuda:
            return self._pose_from_features_cl(class_id, values, points, pitch, origin, grid_nontarget_empty)
        B = values.shape[0]
        dev = values.device
        points = (points - origin[:, :, None]) / pitch[:, None, None]  # camera -> voxel frame
        h = self._extract(values, points, grid_nontarget_empty)

        outs = {}
        for name in ("rot", "trans", "conf"):
            x = F.relu(getattr(self, f"conv1_{name}")(h))
            x = F.relu(getattr(self, f"conv2_{name}")(x))
            x = F.relu(getattr(self, f"conv3_{name}")(x))
            outs[name] = getattr(self, f"conv4_{name}")(x).float()
        P = self._n_point
        cls_rot = outs["rot"].reshape(B, self._n_fg_class, 4, P)
        cls_trans = outs["trans"].reshape(B, self._n_fg_class, 3, P)
        cls_conf = torch.sigmoid(outs["conf"]).reshape(B, self._n_fg_class, P)

        points = points * pitch[:, None, None] + origin[:, :, None]  # voxel -> camera frame
        cls_trans = points[:, None, :, :] + cls_trans * pitch[:, None, None, None]

        fg_class_id = (class_id - 1).long()
        ar = torch.arange(B, device=dev)
        rot = cls_rot[ar, fg_class_id]
        # F.normalize of chainer (l2_normalization.py): x / (|x| + eps), eps = 1e-5 -- not torch's x / max(|x|, eps)
        rot = (rot / (rot.norm(dim=1, keepdim=True) + 1e-5)).transpose(1, 2)  # B4P -> BP4
        trans = cls_trans[ar, fg_class_id].transpose(1, 2)  # B3P -> BP3
        conf = cls_conf[ar, fg_class_id]
        return rot, trans, conf

    def _pose_from_features_bf16(self, class_id, values, points, pitch, origin, grid_nontarget_empty):
        """``_pose_from_features`` under bf16 autocast, differentiable: points-major bf16 rows and channels-last
        bf16 grids; conv3 / conv4, the occupancy convolutions and all sixteen 1x1 convolutions on the bf16 MFMA kernels,
        voxelization and trilinear sampling on their channels-last bf16 kernels -- forward, data and weight gradients
        (bf16_ops.py); the loss keeps its fp32 HIP op.
        values [B,32,P] image features, points [B,3,P] camera frame (model.py:93-164,232-275)."""
        from . import bf16_ops as K
        B, _, P = values.shape
        n, D, nf = B * P, self._voxel_dim, self._n_fg_class
        dev = values.device
        pts = ((points.float() - origin[:, :, None]) / pitch[:, None, None]).transpose(1, 2).reshape(n, 3).contiguous()
        to_center = (D / 2.0 - 0.5) - pts
        batch_indices = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(P)
        batch_start = torch.arange(B + 1, dtype=torch.int32, device=dev) * P          # rows of item b: sorted by item
        x_rgb = values.transpose(1, 2).reshape(n, values.shape[1])
        h1_rgb = K.linear(x_rgb, self.conv1_rgb)
        h1_pcd = K.linear(to_center, self.conv1_pcd)
        h2_rgb = K.linear(h1_rgb, self.conv2_rgb)
        h2_pcd = K.linear(h1_pcd, self.conv2_pcd)
        feat2 = torch.cat((h2_rgb, h2_pcd), dim=1)                                    # [n,144] bf16
        c3 = self.conv3.in_channels
        h_occ = None
        if self._with_occupancy:
            g8 = torch.zeros((B, D ** 3, 8), dtype=torch.bfloat16, device=dev)        # 1 channel + 7 zeros (16 B / voxel)
            g8[:, :, 0] = grid_nontarget_empty.reshape(B, D ** 3)
            h_occ = K.conv3d(K.conv3d(g8, self.conv1_occ, D), self.conv2_occ, D)      # [B,D^3,16] bf16
        if os.environ.get("MF_DENSE_CONV3", "0") not in ("", "0"):
            # round 4's form (A/B measurements): the dense [B,D^3,160] grid through the dense engines
            x3 = K.AverageVoxelizationCL.apply(feat2, pts, batch_indices, B, D, c3)   # cols 0:144
            if h_occ is not None:
                x3[:, :, 144:] = h_occ
            h3 = K.conv3d_k4s2(x3, self.conv3, D)                                     # [B,16^3,256] bf16
        else:
            # conv3 on the occupied voxels only (<= P of 32768 per object in 144 of the 160 channels): compact rows,
            # forward + data + weight gradients (csrc/sparseconv_bf16.hip); the occupancy channels stay dense
            h3 = K.SparseConv3.apply(feat2, h_occ, pts, batch_indices, self.conv3.weight, self.conv3.bias, B, D)
        Dh = D // 2
        feat3 = K.InterpolateVoxelGridCL.apply(h3, pts / 2.0, batch_indices, Dh, batch_start)  # [n,256] bf16
        h4 = K.conv3d_k4s2(h3, self.conv4, Dh)                                        # [B,8^3,512] bf16
        feat4 = K.InterpolateVoxelGridCL.apply(h4, pts / 4.0, batch_indices, D // 4, batch_start)  # [n,512]
        feat = torch.cat((h1_rgb, h1_pcd, h2_rgb, h2_pcd, feat3, feat4), dim=1)       # [n,984] bf16
        names = ("rot", "trans", "conf")
        w1 = torch.cat([getattr(self, f"conv1_{k}").weight for k in names])           # one GEMM for the 3 heads
        b1 = torch.cat([getattr(self, f"conv1_{k}").bias for k in names])
        h = K.Linear.apply(feat, w1, b1, True)                                        # [n,1920]
        outs = {}
        for i, k in enumerate(names):
            x = K.linear(h[:, 640 * i:640 * (i + 1)], getattr(self, f"conv2_{k}"))
            x = K.linear(x, getattr(self, f"conv3_{k}"))
            outs[k] = K.linear(x, getattr(self, f"conv4_{k}"), relu=False).float()
        # class selection + F.normalize + translation + sigmoid (model.py:262-273): one launch forward, one backward;
        # a class id outside 1 .. n_fg (0 = background) has no head and gives NaN, like the inference epilogue
        return K.PoseEpilogue.apply(outs["rot"], outs["trans"], outs["conf"], class_id, pts, pitch, origin, B, P, nf)

    def _pose_from_features_cl(self, class_id, values, points, pitch, origin, grid_nontarget_empty):
        """``_pose_from_features`` (camera-frame points) on the channels-last kernels (volumetric_cl.py)."""
        if getattr(self, "_volumetric_cl", None) is None:
            self.__dict__["_volumetric_cl"] = ChannelsLastVolumetric(self)
        return self._volumetric_cl.pose(class_id, values, points, pitch, origin, grid_nontarget_empty)

    # ---- training (model.py:277-481) ------------------------------------------------------
    def forward(self, *, class_id, rgb, pcd, quaternion_true, translation_true, pitch=None,
                origin=None, grid_target=None, grid_nontarget_empty=None, pix=None, cad=None, symmetric=None):
        if pix is not None:  # the host's share was done ahead (``forward_device``: capturable, also through DDP)
            return self.forward_device(class_id, rgb, pcd, pix, pitch, origin, grid_nontarget_empty, quaternion_true,
                                       translation_true, cad, symmetric)
        quaternion_pred, translation_pred, confidence_pred = self.predict(
            class_id=class_id, rgb=rgb, pcd=pcd, pitch=pitch, origin=origin,
            grid_nontarget_empty=grid_nontarget_empty)
        return self.loss(class_id=class_id, quaternion_true=quaternion_true,
                         translation_true=translation_true, quaternion_pred=quaternion_pred,
                         translation_pred=translation_pred, confidence_pred=confidence_pred)

    @torch.no_grad()
    def evaluate(self, *, class_id, quaternion_true, translation_true, quaternion_pred,
                 translation_pred, per_instance=False):
        """ADD / ADD-S of the given poses (model.py:325-375), returned as a dict of means.

        ``per_instance=True`` is the reference's evaluation-mode report: one
        ``{add,add_s,add_or_add_s}/{class_id:04d}/{uuid}`` entry per object, which
        ``training.PoseEstimationEvaluator`` regroups per class for the AUC."""
        T_true = functions_module.transformation_matrix(
            quaternion_true.float(), translation_true.float()).cpu().numpy()
        T_pred = functions_module.transformation_matrix(quaternion_pred, translation_pred).cpu().numpy()
        report = {}
        adds, add_ss, mixed = [], [], []
        for i, cid in enumerate(torch.as_tensor(class_id).tolist()):
            add, add_s = metrics.average_distance([self._models.get_pcd(cid)], [T_true[i]], [T_pred[i]])
            adds.append(add[0])
            add_ss.append(add_s[0])
            mixed.append(add_s[0] if cid in CLASS_IDS_SYMMETRIC else add[0])
            if per_instance:
                tag = f"{cid:04d}/{uuid.uuid1()}"
                report[f"add/{tag}"] = float(adds[-1])
                report[f"add_s/{tag}"] = float(add_ss[-1])
                report[f"add_or_add_s/{tag}"] = float(mixed[-1])
        if per_instance:
            return report
        return dict(add=float(np.mean(adds)), add_s=float(np.mean(add_ss)),
                    add_or_add_s=float(np.mean(mixed)))

    def loss_prepare(self, class_id, device):
        """The host side of ``loss`` (model.py:411-414): 500 random CAD points per object (host RNG, like the
        reference) and the ADD-S flags.  Returns (cad, symmetric): cad a [B,500,3] device tensor -- or a list of
        per-object arrays when the clouds differ in size -- and symmetric a [B] bool device tensor."""
        cids = [int(c) for c in torch.as_tensor(class_id).tolist()]
        cads = []
        for cid in cids:
            cad_pcd = self._models.get_pcd(cid)
            cads.append(np.asarray(cad_pcd[np.random.permutation(cad_pcd.shape[0])[:500]], dtype=np.float32))
        symmetric = torch.tensor([cid in CLASS_IDS_SYMMETRIC and self._loss != "add" for cid in cids],
                                 dtype=torch.bool, device=device)
        if len({c.shape[0] for c in cads}) == 1:
            return torch.as_tensor(np.stack(cads), device=device), symmetric
        return cads, symmetric

    def loss_device(self, cad, symmetric, quaternion_true, translation_true, quaternion_pred, translation_pred,
                    confidence_pred):
        """The device side of ``loss``: no host work, no synchronisation (capturable into a hipGraph when ``cad``
        is one tensor)."""
        B, P = quaternion_pred.shape[0], quaternion_pred.shape[1]
        dev = quaternion_pred.device
        from ....functions.geometry.transformation_matrix import transformation_matrix_batch
        T_pred = transformation_matrix_batch(  # (one fused launch forward / backward for the B * P predicted poses)
            quaternion_pred.reshape(B * P, 4), translation_pred.reshape(B * P, 3)).reshape(B, P, 4, 4)
        T_true = transformation_matrix_batch(quaternion_true.float(), translation_true.float())
        if torch.is_tensor(cad):
            add = functions_module.average_distance_batch(cad, T_true, T_pred, symmetric)  # [B,P]
        else:  # CAD clouds of different sizes (< 500 points): one call per object
            add = torch.stack([functions_module.average_distance(
                torch.as_tensor(cad[i], device=dev), T_true[i], T_pred[i], symmetric=bool(symmetric[i]))
                for i in range(B)])
        from ....functions.loss.confidence_loss import confidence_loss
        return confidence_loss(add, confidence_pred.float(), self._lambda_confidence)  # (one launch each way)

    def loss(self, *, class_id, quaternion_true, translation_true, quaternion_pred,
             translation_pred, confidence_pred):
        """DenseFusion pose loss (model.py:377-434): per object mean over the confident points
        of ``ADD(-S) * conf - lambda * log(conf)``, averaged over the batch.  All B objects go
        through ONE fused ADD / ADD-S kernel (functions.average_distance_batch) instead of the
        reference's per-object loop of transform / nn / gather launches."""
        cad, symmetric = self.loss_prepare(class_id, quaternion_pred.device)
        if not bool(symmetric.any()):
            symmetric = None if torch.is_tensor(cad) else symmetric
        return self.loss_device(cad, symmetric, quaternion_true, translation_true, quaternion_pred,
                                translation_pred, confidence_pred)

    def forward_device(self, class_id, rgb, pcd, pix, pitch, origin, grid_nontarget_empty, quaternion_true,
                       translation_true, cad, symmetric):
        """``forward`` with the host work done ahead (``_select_points`` -> pix, ``loss_prepare`` -> cad, symmetric):
        device work only, the form a training step is captured into a hipGraph in
        (examples/singleview_3d_train.py --graph)."""
        if pitch is None or origin is None or cad is None or symmetric is None:
            # (predict() derives pitch / origin from the class and the cloud's median on the HOST: model.py:195-205)
            raise ValueError("forward(pix=...) / forward_device is the device-only form: pitch, origin, cad and "
                             "symmetric must be given (Model._select_points, Model.loss_prepare; pitch / origin as "
                             "predict() computes them)")
        dev = rgb.device
        pitch = torch.as_tensor(pitch, dtype=torch.float32, device=dev)  # predict()'s casts
        origin = torch.as_tensor(origin, dtype=torch.float32, device=dev)
        q, t, c = self._predict_device(class_id, rgb, pcd, pix, pitch, origin, grid_nontarget_empty)
        return self.loss_device(cad, symmetric, quaternion_true, translation_true, q, t, c)
