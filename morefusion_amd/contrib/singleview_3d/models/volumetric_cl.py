"""Inference-time volumetric part of the pose network in CHANNELS-LAST / points-major layout.

Restates morefusion/contrib/singleview_3d/models/model.py:93-164,232-275 (``_extract`` + the three
per-point heads) for ``torch.no_grad()`` on the MI355X with every 3-D operator hand-written:

  point MLP (conv1/2_rgb, conv1/2_pcd: stock GEMMs on [n, C] rows, n = B * P)
  occupancy branch   mf_occupancy_convs_fwd            (csrc/conv3d.hip, VALU, scalar weights)
  conv3              mf_conv3d_k4s2_fwd on the 16 occupancy channels (fp32 MFMA implicit GEMM)
                     + mf_sparse_conv3d_k4s2_points_cl_fwd on the 144 voxelized channels
                       (point chains -> compact rows -> 8 parity-class MFMA GEMMs -> reduce + bias + ReLU)
  conv4              mf_conv3d_k4s2_fwd (256 -> 512, split-K, bias + ReLU in the finish pass)
  trilinear sampling mf_interpolate_voxel_grid_cl_fwd x2, written straight into the column blocks
                     [216:472] and [472:984] of the heads' input matrix F [n, 984]
  heads              stock GEMMs on F (row-major [n, C]: ``F.linear``)

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


class ChannelsLastVolumetric:
    """Weight packs, workspaces and the launch sequence; one instance per Model (and device)."""

    def __init__(self, model):
        self.m = model
        self._packs = {}
        self._buf = {}
        self._sparse = SparseVoxelConv3d(model.conv3)

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
        t = self._buf.get(name)
        if t is None or t.shape != tuple(shape) or t.device != device or t.dtype != dtype:
            t = torch.empty(shape, dtype=dtype, device=device)
            self._buf[name] = t
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

    def features(self, values, points, grid_nontarget_empty):
        """values [B,32,P] image features, points [B,3,P] voxel-frame coordinates, no-entry grid
        [B,D,D,D] (or None) -> F [B*P, 984] = (feat1 | feat2 | feat3 | feat4) per point."""
        m = self.m
        B, _, P = values.shape
        n, D = B * P, m._voxel_dim
        dev = values.device
        _lib.require_gpu(values, points)
        pts = points.float().transpose(1, 2).reshape(n, 3).contiguous()
        x_rgb = values.transpose(1, 2).reshape(n, values.shape[1])
        to_center = (D / 2.0 - 0.5) - pts
        batch_indices = torch.arange(B, dtype=torch.int32, device=dev).repeat_interleave(P)
        feat = torch.empty((n, 984), dtype=torch.float32, device=dev)

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
        h3 = self._sparse.from_points_cl(feat[:, 72:216], 984, pts, batch_indices, B, dense, D)  # [B, 16^3, 256]
        self.sample(h3, D // 2, pts / 2.0, batch_indices, feat[:, 216:472], 984)
        h4 = self.conv_k4s2("conv4", m.conv4, h3, B, D // 2, cin=256)                            # [B, 8^3, 512]
        self.sample(h4, D // 4, pts / 4.0, batch_indices, feat[:, 472:984], 984)
        return feat

    def heads(self, feat, B, P):
        """F [B*P, 984] -> per-point class outputs rot [B,P,n_fg,4], trans [B,P,n_fg,3], conf [B,P,n_fg]."""
        m = self.m
        outs = {}
        for name in ("rot", "trans", "conf"):
            x = feat
            for i in (1, 2, 3):
                w, b = self._linear_pack(f"conv{i}_{name}", getattr(m, f"conv{i}_{name}"))
                x = F.relu(F.linear(x, w, b))
            w, b = self._linear_pack(f"conv4_{name}", getattr(m, f"conv4_{name}"))
            outs[name] = F.linear(x, w, b).float()
        nf = m._n_fg_class
        return (outs["rot"].reshape(B, P, nf, 4), outs["trans"].reshape(B, P, nf, 3),
                torch.sigmoid(outs["conf"]).reshape(B, P, nf))
