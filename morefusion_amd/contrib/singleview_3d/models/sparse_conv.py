"""Inference-time conv3 of the pose network on sparse voxelized features.

``SparseVoxelConv3d`` evaluates ``relu(conv3(cat[voxelized, h_occ]))``
(morefusion/contrib/singleview_3d/models/model.py:125-128; Convolution3D k=4 s=2 pad=1)
without convolving the <= 3 % occupied ``voxelized`` channels densely: those go through
``mf_sparse_conv3d_k4s2_fwd`` (csrc/sparseconv.hip: 8 parity-class GEMMs on fp32 MFMA +
an output-stationary reduce, 0.6 instead of 18.9 GFLOP per object); the dense occupancy
channels keep a stock dense convolution whose result the reduce kernel adds.
Forward only (training keeps the dense torch convolution and its autograd).
"""
import torch
import torch.nn.functional as F

from .... import _lib


class SparseVoxelConv3d:
    def __init__(self, conv):
        self.conv = conv  # torch.nn.Conv3d(Cs + Cd, Cout, 4, 2, padding=1)
        self._key = None
        self._ws = None

    def _prepare(self, Cs):
        w = self.conv.weight
        key = (w.data_ptr(), w._version, Cs)
        if self._key != key:
            Cout, Cin = w.shape[0], w.shape[1]
            self.Wp = torch.empty((8 * Cs * 8 * Cout,), dtype=torch.float32, device=w.device)
            wc = w.detach().float().contiguous()
            _lib.check(_lib.lib().mf_sparse_conv3d_pack_weights(
                wc.data_ptr(), Cout, Cs, Cin, 0, self.Wp.data_ptr(), _lib.stream_ptr()),
                "mf_sparse_conv3d_pack_weights")
            self.Wd = wc[:, Cs:].contiguous() if Cin > Cs else None
            self._key = key

    @torch.no_grad()
    def __call__(self, voxelized, counts, h_dense=None, max_rows=None, relu=True):
        """voxelized [B,Cs,D,D,D] (zeros where counts == 0), counts [B,D,D,D] int32,
        h_dense [B,Cd,D,D,D] or None -> relu(conv(cat)) [B,Cout,D/2,D/2,D/2]."""
        _lib.require_gpu(voxelized, counts)
        B, Cs, D = voxelized.shape[0], voxelized.shape[1], voxelized.shape[2]
        Cout = self.conv.out_channels
        self._prepare(Cs)
        if max_rows is None:
            max_rows = B * D ** 3
        dense = None
        if h_dense is not None:
            # (under autocast this convolution may run in bf16; the kernel adds a float32 tensor)
            dense = F.conv3d(h_dense.float(), self.Wd, None, stride=2, padding=1).float().contiguous()
        lib = _lib.lib()
        nbytes = lib.mf_sparse_conv3d_workspace_bytes(B, Cs, Cout, D, max_rows, 0)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws_retired = getattr(self, "_ws_retired", []) + [self._ws]  # a captured hipGraph may hold it
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=voxelized.device)
        x = voxelized.float().contiguous()
        counts = counts.to(torch.int32)
        assert x.dtype == torch.float32 and (dense is None or dense.dtype == torch.float32)
        out = torch.empty((B, Cout, D // 2, D // 2, D // 2), dtype=torch.float32, device=x.device)
        bias = self.conv.bias.detach().float().contiguous() if self.conv.bias is not None else None
        _lib.check(lib.mf_sparse_conv3d_k4s2_fwd(
            x.data_ptr(), counts.contiguous().data_ptr(), self.Wp.data_ptr(), _lib.ptr(dense),
            _lib.ptr(bias), out.data_ptr(), self._ws.data_ptr(), B, Cs, Cout, D, int(max_rows),
            int(relu), _lib.stream_ptr()), "mf_sparse_conv3d_k4s2_fwd")
        return out

    @torch.no_grad()
    def from_points(self, values, points, batch_indices, batch_size, h_dense=None, dim=32, origin=(0, 0, 0),
                    pitch=1.0, relu=True):
        """``relu(conv(cat[average_voxelization_3d(values, points), h_dense]))`` WITHOUT the dense
        voxelized tensor: values [n,Cs], points [n,3], batch_indices [n] int32 -> [B,Cout,D/2,...].
        The per-voxel chains of the voxelization write the GEMM's compact rows directly
        (``mf_sparse_conv3d_k4s2_points_fwd``); same output bits as ``__call__`` on the dense op."""
        _lib.require_gpu(values, points, batch_indices)
        n, Cs = values.shape
        B, D = int(batch_size), int(dim)
        Cout = self.conv.out_channels
        self._prepare(Cs)
        dense = None
        if h_dense is not None:
            dense = F.conv3d(h_dense.float(), self.Wd, None, stride=2, padding=1).float().contiguous()
        lib = _lib.lib()
        max_rows = max(int(n), 1)
        nbytes = lib.mf_sparse_conv3d_workspace_bytes(B, Cs, Cout, D, max_rows, n)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws_retired = getattr(self, "_ws_retired", []) + [self._ws]  # a captured hipGraph may hold it
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=values.device)
        vals, pts, bi = _lib.f32c(values), _lib.f32c(points), _lib.i32c(batch_indices)
        out = torch.empty((B, Cout, D // 2, D // 2, D // 2), dtype=torch.float32, device=values.device)
        bias = self.conv.bias.detach().float().contiguous() if self.conv.bias is not None else None
        o = _lib.as_float3(origin)
        _lib.check(lib.mf_sparse_conv3d_k4s2_points_fwd(
            vals.data_ptr(), pts.data_ptr(), bi.data_ptr(), n, o[0], o[1], o[2], float(pitch),
            self.Wp.data_ptr(), _lib.ptr(dense), _lib.ptr(bias), out.data_ptr(), self._ws.data_ptr(),
            B, Cs, Cout, D, max_rows, int(relu), _lib.stream_ptr()), "mf_sparse_conv3d_k4s2_points_fwd")
        return out

    @torch.no_grad()
    def from_points_cl(self, values, ldv, points, batch_indices, batch_size, dense_cl=None, dim=32, relu=True):
        """Channels-last form of ``from_points``: ``values`` is an [n, Cs] column block of a wider row-major
        matrix (row pitch ``ldv`` floats), ``dense_cl`` / the result are [B, (D/2)^3, Cout]."""
        _lib.require_gpu(values, points, batch_indices)
        n, Cs = values.shape
        B, D = int(batch_size), int(dim)
        Cout = self.conv.out_channels
        if values.dtype != torch.float32 or values.stride(1) != 1 or values.stride(0) != ldv:
            raise TypeError("values must be a float32 column block with row pitch ldv")
        self._prepare(Cs)
        lib = _lib.lib()
        max_rows = max(int(n), 1)
        nbytes = lib.mf_sparse_conv3d_workspace_bytes(B, Cs, Cout, D, max_rows, n)
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws_retired = getattr(self, "_ws_retired", []) + [self._ws]  # a captured hipGraph may hold it
            self._ws = torch.empty((nbytes,), dtype=torch.uint8, device=values.device)
        pts, bi = _lib.f32c(points), _lib.i32c(batch_indices)
        out = torch.empty((B, (D // 2) ** 3, Cout), dtype=torch.float32, device=values.device)
        bias = self.conv.bias.detach().float().contiguous() if self.conv.bias is not None else None
        _lib.check(lib.mf_sparse_conv3d_k4s2_points_cl_fwd(
            values.data_ptr(), int(ldv), pts.data_ptr(), bi.data_ptr(), n, 0.0, 0.0, 0.0, 1.0,
            self.Wp.data_ptr(), _lib.ptr(dense_cl), _lib.ptr(bias), out.data_ptr(), self._ws.data_ptr(),
            B, Cs, Cout, D, max_rows, int(relu), _lib.stream_ptr()), "mf_sparse_conv3d_k4s2_points_cl_fwd")
        return out
