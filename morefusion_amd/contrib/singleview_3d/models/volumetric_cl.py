"""Inference-time volumetric part of the pose network in CHANNELS-LAST / points-major layout.

Restates morefusion/contrib/singleview_3d/models/model.py:93-164,232-275 (``_extract`` + the three
per-point heads) for ``torch.no_grad()`` on the MI355X with every 3-D operator hand-written:

  point MLP          conv1/2_rgb, conv1/2_pcd through mf_linear_fwd, written into F's column blocks
  occupancy branch   mf_occupancy_convs_fwd            (csrc/conv3d.hip, VALU, scalar weights)
  conv3              mf_conv3d_k4s2_fwd on the 16 occupancy channels (fp32 MFMA implicit GEMM)
                     + mf_sparse_conv3d_k4s2_points_cl_fwd on the 144 voxelized channels
                       (point chains -> compact rows -> 8 parity-class MFMA GEMMs -> reduce + bias + ReLU)
  conv4              mf_conv3d_k4s2_fwd (256 -> 512, split-K, bias + ReLU in the finish pass)
  trilinear sampling mf_interpolate_voxel_grid_cl_fwd x2, written straight into the column blocks
                     [216:472] and [472:984] of the heads' input matrix F [n, 984]
  heads              mf_linear_fwd (csrc/linear.hip: grouped fp32-MFMA GEMM + bias + ReLU): layer 1 of the three
                     heads as ONE [n, 984] x [984, 1920] GEMM, layers 2-4 with the heads side by side, each
                     reading / writing its column block (under bf16 autocast: stock ``F.linear``)

Grids are [B, D^3, C] (a voxel's channels are contiguous: the implicit GEMM's K runs over
(tap, channel) without gathers, and a trilinear corner is one coalesced row read); nothing is
transposed between the stages and ``torch.cat`` of the four feature groups never happens.
The channels-first path of ``Model._extract`` (training; round 2's inference path) computes the same
values: ``tests/test_gpu_conv3d.py`` compares the two stage by stage.
"""
import torch
import torch.nn.functional as F

from .... import _lib
from .sparse_conv import SparseVoxelConv3d


# Row pitch of the heads' input matrix F: 984 feature columns padded to 992 floats = 31 x 128 bytes, so that every
# row -- and every 128-byte K chunk the GEMM kernels stage -- starts on a cache-line boundary (984 x 4 B = 30.75
# lines: every chunk straddled two).  Columns 984..991 are zero (and so are the packed weights' columns there).
F_COLS, F_LD = 984, 992


class ChannelsLastVolumetric:
    """Weight packs, workspaces and the launch sequence; one instance per Model (and device)."""

    def __init__(self, model):
        self.m = model
        self._packs = {}
        self._buf = {}
        self._sparse = SparseVoxelConv3d(model.conv3)
        self.mfma_linear = True  # 1x1 convolutions on csrc/linear.hip (False: stock F.linear)

    # ---- cached weight packs (re-packed when a parameter changes in place or is re-assigned) ----
    def _pack(self, name, tensors, build):
        key = tuple((t.data_ptr(), t._version) for t in tensors)
        hit = self._packs.get(name)
        if hit is None or hit[0] != key:
            hit = (key, build())
            self._packs[name] = hit
        return hit[1]

    def _conv_pack(self, name, conv, cin, c_off):
        def build():
            w = conv.weight.detach().float().contiguous()
            cout, w_cin = w.shape[0], w.shape[1]
            wt = torch.empty((cout, 64, cin), dtype=torch.float32, device=w.device)
            _lib.check(_lib.lib().mf_conv3d_k4s2_pack_weights(w.data_ptr(), cout, cin, w_cin, c_off, wt.data_ptr(),
                                                              _lib.stream_ptr()), "mf_conv3d_k4s2_pack_weights")
            bias = conv.bias.detach().float().contiguous() if conv.bias is not None else None
            return wt, bias
        return self._pack(name, [conv.weight] + ([conv.bias] if conv.bias is not None else []), build)

    def _occ_pack(self):
        m = self.m

        def build():
            out = []
            for conv in (m.conv1_occ, m.conv2_occ):
                w = conv.weight.detach().float()
                out.append(w.permute(2, 3, 4, 1, 0).contiguous().reshape(27, w.shape[1], w.shape[0]))
                out.append(conv.bias.detach().float().contiguous())
            return out
        return self._pack("occ", [m.conv1_occ.weight, m.conv1_occ.bias, m.conv2_occ.weight, m.conv2_occ.bias], build)

    def _linear_pack(self, name, conv):
        return self._pack(name, [conv.weight, conv.bias],
                          lambda: (conv.weight.detach().squeeze(-1).contiguous(), conv.bias.detach()))

    def _scratch(self, name, shape, device, dtype=torch.float32):
        """Reusable intermediate of one shape.  Keyed by (name, shape): a buffer is never replaced or freed once
        handed out, because a captured hipGraph (predict_graphed) of another batch size keeps its address."""
        key = (name, tuple(shape), str(device), dtype)
        t = self._buf.get(key)
        if t is None:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._buf[key] = t
        return t

    # ---- stages -----------------------------------------------------------------------------
    def conv_k4s2(self, name, conv, x_cl, B, D, cin, c_off=0, add=None, relu=True, bias=True, split=None):
        """x_cl [B, D^3, cin] -> [B, (D/2)^3, Cout] through the fp32-MFMA implicit GEMM."""
        L = _lib.lib()
        wt, b = self._conv_pack(name, conv, cin, c_off)
        cout = wt.shape[0]
        if split is None:
            split = L.mf_conv3d_k4s2_default_split(B, cin, cout, D)
        nbytes = L.mf_conv3d_k4s2_workspace_bytes(B, cout, D, split)
        ws = self._scratch(name + "_ws", (max(nbytes, 16),), x_cl.device, torch.uint8)
        out = torch.empty((B, (D // 2) ** 3, cout), dtype=torch.float32, device=x_cl.device)
        _lib.check(L.mf_conv3d_k4s2_fwd(x_cl.data_ptr(), wt.data_ptr(), _lib.ptr(b if bias else None), _lib.ptr(add),
                                        out.data_ptr(), ws.data_ptr(), B, cin, cout, D, int(split), int(relu),
                                        _lib.stream_ptr()), "mf_conv3d_k4s2_fwd")
        return out

    def occupancy(self, grid):
        """grid_nontarget_empty [B, D, D, D] -> relu(conv2_occ(relu(conv1_occ))) as [B, D^3, 16]."""
        B, D = grid.shape[0], grid.shape[1]
        g = _lib.f32c(grid)
        w1, b1, w2, b2 = self._occ_pack()
        h1 = self._scratch("occ_h1", (B, D ** 3, 8), g.device)
        h2 = torch.empty((B, D ** 3, 16), dtype=torch.float32, device=g.device)
        _lib.check(_lib.lib().mf_occupancy_convs_fwd(g.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(),
                                                     b2.data_ptr(), h1.data_ptr(), h2.data_ptr(), B, D,
                                                     _lib.stream_ptr()), "mf_occupancy_convs_fwd")
        return h2

    def sample(self, vox_cl, D, pts, batch_indices, out_block, ldo):
        """Trilinear samples of vox_cl [B, D^3, C] at pts [n,3] -> out_block (a column view of F)."""
        B, _, C = vox_cl.shape
        _lib.check(_lib.lib().mf_interpolate_voxel_grid_cl_fwd(
            vox_cl.data_ptr(), pts.data_ptr(), batch_indices.data_ptr(), pts.shape[0], B, C, D, D, D,
            out_block.data_ptr(), ldo, _lib.stream_ptr()), "mf_interpolate_voxel_grid_cl_fwd")

    def prep(self, values, points_cam, pitch, origin):
        """One launch (mf_point_prep): camera-frame points [B,3,P] -> voxel-frame rows ``pts`` [n,3]
        (model.py:236), ``tc4`` [n,4] = (to_center | 0) (:101), image features as rows [n,32], batch indices [n]."""
        B, _, P = points_cam.shape
        n, dev = B * P, values.device
        _lib.require_gpu(values, points_cam, pitch, origin)
        pts = torch.empty((n, 3), dtype=torch.float32, device=dev)
        tc4 = torch.empty((n, 4), dtype=torch.float32, device=dev)
        if values.ndim == 2:   # already rows [n, Cv] (PSPNetExtractor.forward_sampled_rows): nothing to transpose
            x_rows, Cv = _lib.f32c(values), 0
        else:
            Cv = values.shape[1]
            x_rows = torch.empty((n, Cv), dtype=torch.float32, device=dev)
        bi = torch.empty((n,), dtype=torch.int32, device=dev)
        _lib.check(_lib.lib().mf_point_prep(
            _lib.f32c(points_cam).data_ptr(), _lib.f32c(values).data_ptr(), _lib.f32c(origin).data_ptr(),
            _lib.f32c(pitch).data_ptr(), B, P, Cv, self.m._voxel_dim / 2.0 - 0.5, pts.data_ptr(), tc4.data_ptr(),
            x_rows.data_ptr(), bi.data_ptr(), _lib.stream_ptr()), "mf_point_prep")
        return pts, tc4, x_rows, bi

    def features(self, values, points_cam, pitch, origin, grid_nontarget_empty):
        """values [B,32,P] image features, points_cam [B,3,P] camera-frame coordinates, pitch [B], origin [B,3],
        no-entry grid [B,D,D,D] (or None) -> (F [B*P, 984] = feat1 | feat2 | feat3 | feat4 per point,
        voxel-frame points [n,3])."""
        m = self.m
        B, _, P = points_cam.shape
        n, D = B * P, m._voxel_dim
        dev = values.device
        pts, tc4, x_rgb, batch_indices = self.prep(values, points_cam, pitch, origin)
        to_center = tc4[:, :3]
        feat = torch.empty((n, F_LD), dtype=torch.float32, device=dev)
        feat[:, F_COLS:].zero_()

        if self.mfma_linear and not torch.is_autocast_enabled():
            self._linear("conv1_rgb", [m.conv1_rgb], x_rgb, 32, feat[:, 0:64], F_LD, relu=True)
            self._linear("conv1_pcd", [m.conv1_pcd], tc4, 4, feat[:, 64:72], F_LD, relu=True, k_pad=4)
            self._linear("conv2_rgb", [m.conv2_rgb], feat[:, 0:64], F_LD, feat[:, 72:200], F_LD, relu=True)
            self._linear("conv2_pcd", [m.conv2_pcd], feat[:, 64:72], F_LD, feat[:, 200:216], F_LD, relu=True)
        else:
            w, b = self._linear_pack("conv1_rgb", m.conv1_rgb)
            h_rgb = F.relu(F.linear(x_rgb, w, b))
            w, b = self._linear_pack("conv1_pcd", m.conv1_pcd)
            h_pcd = F.relu(F.linear(to_center.to(x_rgb.dtype), w, b))
            feat[:, 0:64] = h_rgb
            feat[:, 64:72] = h_pcd
            w, b = self._linear_pack("conv2_rgb", m.conv2_rgb)
            feat[:, 72:200] = F.relu(F.linear(h_rgb, w, b))
            w, b = self._linear_pack("conv2_pcd", m.conv2_pcd)
            feat[:, 200:216] = F.relu(F.linear(h_pcd, w, b))

        # conv3 = dense 16 occupancy channels (implicit GEMM) + sparse 144 voxelized channels
        dense = None
        if m._with_occupancy:
            h_occ = self.occupancy(grid_nontarget_empty)
            dense = self.conv_k4s2("conv3_occ", m.conv3, h_occ, B, D, cin=16, c_off=144, relu=False, bias=False)
        h3 = self._sparse.from_points_cl(feat[:, 72:216], F_LD, pts, batch_indices, B, dense, D)  # [B, 16^3, 256]
        pts2 = pts * 0.5  # == pts / 2.0 (a power of two: same bits); one launch serves both samplers' scales
        self.sample(h3, D // 2, pts2, batch_indices, feat[:, 216:472], F_LD)
        h4 = self.conv_k4s2("conv4", m.conv4, h3, B, D // 2, cin=256)                            # [B, 8^3, 512]
        self.sample(h4, D // 4, pts2 * 0.5, batch_indices, feat[:, 472:984], F_LD)
        return feat, pts

    # ---- 1x1 convolutions as grouped fp32-MFMA GEMMs -------------------------------------------
    def _gemm_pack(self, name, convs, k_pad=None):
        """Weights of the ``convs`` (one per group; equal shapes) as [G, Npad, K] (zero rows up to Npad % 128
        == 0, zero columns up to k_pad), biases [G, N]."""
        def build():
            ws, bs = [], []
            for c in convs:
                w = c.weight.detach().float().squeeze(-1)
                N, K = w.shape
                Kp = k_pad or K
                Np = -(-N // 128) * 128
                wp = torch.zeros((Np, Kp), dtype=torch.float32, device=w.device)
                wp[:N, :K] = w
                ws.append(wp)
                bs.append(c.bias.detach().float())
            return torch.stack(ws).contiguous(), torch.stack(bs).contiguous()
        return self._pack("gemm_" + name, [t for c in convs for t in (c.weight, c.bias)], build)

    def _linear(self, name, convs, a, lda, out, ldo, relu, k_pad=None, a_gs=0, o_gs=0):
        """out[:, g-th block] = act(a[:, g-th block] @ W_g^T + b_g) for every conv of ``convs`` in one launch.
        ``a`` / ``out``: views whose first element is the first group's block; group g starts a_gs / o_gs
        floats further; row pitches lda / ldo."""
        w, b = self._gemm_pack(name, convs, k_pad)
        G, Np, K = w.shape
        N = b.shape[1]
        _lib.check(_lib.lib().mf_linear_fwd(a.data_ptr(), a_gs, lda, w.data_ptr(), Np * K, K, b.data_ptr(), N,
                                            out.data_ptr(), o_gs, ldo, a.shape[0], N, Np, K, G, int(relu),
                                            _lib.stream_ptr()), "mf_linear_fwd")

    def heads(self, feat, B, P, raw=False):
        """F [B*P, 984] -> per-point class outputs rot [B,P,n_fg,4], trans [B,P,n_fg,3], conf [B,P,n_fg].
        ``raw=True``: the MFMA path's unsplit output rows (o [n, 3*np4], np4) for mf_pose_epilogue, or
        (None, 0) when the stock-GEMM path is active."""
        m = self.m
        nf = m._n_fg_class
        names = ("rot", "trans", "conf")
        if self.mfma_linear and not torch.is_autocast_enabled() and feat.dtype == torch.float32:
            n, dev = feat.shape[0], feat.device
            # layer 1 of the three heads = one GEMM against the stacked weights [1920, 984]
            def build1():
                w = torch.cat([getattr(m, f"conv1_{k}").weight.detach().float().squeeze(-1) for k in names])
                wp = torch.zeros((w.shape[0], F_LD), dtype=torch.float32, device=w.device)
                wp[:, :F_COLS] = w
                b = torch.cat([getattr(m, f"conv1_{k}").bias.detach().float() for k in names])
                return wp.contiguous()[None], b.contiguous()[None]
            w1, b1 = self._pack("gemm_heads1", [t for k in names for t in (getattr(m, f"conv1_{k}").weight,
                                                                            getattr(m, f"conv1_{k}").bias)], build1)
            h1 = self._scratch("heads_h1", (n, 1920), dev)
            h2 = self._scratch("heads_h2", (n, 768), dev)
            h3 = self._scratch("heads_h3", (n, 384), dev)
            np4 = -(-(nf * 4) // 128) * 128
            o = torch.empty((n, 3 * np4), dtype=torch.float32, device=dev)
            L = _lib.lib()
            assert feat.stride(0) == F_LD
            _lib.check(L.mf_linear_fwd(feat.data_ptr(), 0, F_LD, w1.data_ptr(), 0, F_LD, b1.data_ptr(), 0,
                                       h1.data_ptr(), 0, 1920, n, 1920, 1920, F_LD, 1, 1, _lib.stream_ptr()),
                       "mf_linear_fwd")
            self._linear("heads2", [getattr(m, f"conv2_{k}") for k in names], h1, 1920, h2, 768, True, a_gs=640, o_gs=256)
            self._linear("heads3", [getattr(m, f"conv3_{k}") for k in names], h2, 768, h3, 384, True, a_gs=256, o_gs=128)
            # layer 4: N = n_fg * {4, 3, 1} -> pad every head to the widest (equal shapes per launch)
            w4, b4 = self._pack("gemm_heads4", [t for k in names for t in (getattr(m, f"conv4_{k}").weight,
                                                                           getattr(m, f"conv4_{k}").bias)],
                                lambda: self._pad_heads4(names, nf, np4))
            _lib.check(L.mf_linear_fwd(h3.data_ptr(), 128, 384, w4.data_ptr(), np4 * 128, 128, b4.data_ptr(), np4,
                                       o.data_ptr(), np4, 3 * np4, n, np4, np4, 128, 3, 0, _lib.stream_ptr()),
                       "mf_linear_fwd")
            if raw:
                return o, np4
            rot = o[:, 0:nf * 4].reshape(B, P, nf, 4)
            trans = o[:, np4:np4 + nf * 3].reshape(B, P, nf, 3)
            conf = torch.sigmoid(o[:, 2 * np4:2 * np4 + nf]).reshape(B, P, nf)
            return rot, trans, conf
        if raw:
            return None, 0
        outs = {}
        for name in names:
            x = feat[:, :F_COLS]
            for i in (1, 2, 3):
                w, b = self._linear_pack(f"conv{i}_{name}", getattr(m, f"conv{i}_{name}"))
                x = F.relu(F.linear(x, w, b))
            w, b = self._linear_pack(f"conv4_{name}", getattr(m, f"conv4_{name}"))
            outs[name] = F.linear(x, w, b).float()
        return (outs["rot"].reshape(B, P, nf, 4), outs["trans"].reshape(B, P, nf, 3),
                torch.sigmoid(outs["conf"]).reshape(B, P, nf))

    def _pad_heads4(self, names, nf, np4):
        m = self.m
        w = torch.zeros((3, np4, 128), dtype=torch.float32, device=m.conv4_rot.weight.device)
        b = torch.zeros((3, np4), dtype=torch.float32, device=w.device)
        for g, k in enumerate(names):
            c = getattr(m, f"conv4_{k}")
            w[g, :c.out_channels] = c.weight.detach().float().squeeze(-1)
            b[g, :c.out_channels] = c.bias.detach().float()
        return w.contiguous(), b.contiguous()

    def pose(self, class_id, values, points_cam, pitch, origin, grid_nontarget_empty):
        """The whole volumetric part: -> (rot [B,P,4], trans [B,P,3], conf [B,P]) of each object's class."""
        B, _, P = points_cam.shape
        feat, pts = self.features(values, points_cam, pitch, origin, grid_nontarget_empty)
        o, np4 = self.heads(feat, B, P, raw=True)
        if o is not None:
            dev = values.device
            rot = torch.empty((B, P, 4), dtype=torch.float32, device=dev)
            trans = torch.empty((B, P, 3), dtype=torch.float32, device=dev)
            conf = torch.empty((B, P), dtype=torch.float32, device=dev)
            cid = class_id.to(device=dev, dtype=torch.int64).contiguous()
            _lib.check(_lib.lib().mf_pose_epilogue(
                o.data_ptr(), o.stride(0), np4, cid.data_ptr(), pts.data_ptr(), _lib.f32c(origin).data_ptr(),
                _lib.f32c(pitch).data_ptr(), B, P, self.m._n_fg_class, rot.data_ptr(), trans.data_ptr(), conf.data_ptr(),
                _lib.stream_ptr()), "mf_pose_epilogue")
            return rot, trans, conf
        cls_rot, cls_trans, cls_conf = self.heads(feat, B, P)          # [B,P,n_fg,c]
        fg = (class_id - 1).long()
        ar = torch.arange(B, device=values.device)
        rot = cls_rot[ar, :, fg]                                       # [B,P,4]
        rot = rot / (rot.norm(dim=2, keepdim=True) + 1e-5)             # chainer F.normalize: x / (|x| + eps)
        pts_b = pts.reshape(B, P, 3)
        points_back = pts_b * pitch[:, None, None] + origin[:, None, :]  # voxel -> camera frame (model.py:264)
        trans = points_back + cls_trans[ar, :, fg] * pitch[:, None, None]
        conf = cls_conf[ar, :, fg]
        return rot, trans, conf
