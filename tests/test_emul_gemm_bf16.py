"""csrc/gemm_bf16.hip on the CPU (fiber emulator, lane-exact v_mfma_f32_32x32x16_bf16): the bf16 forward,
data-gradient and weight-gradient kernels of the 3-D convolutions (model.py:73-74,125-139) and the 1x1 convolution
chains (model.py:76-91,239-258) against torch's float32 operators evaluated on the SAME bf16-rounded operands
(products of two bf16 are exact in fp32, so the only difference is the summation order: tolerance 1e-4 relative
to the largest output, 1 bf16 ulp where the kernel rounds its output to bf16)."""
import ctypes
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from host_emul import emul

pytestmark = pytest.mark.skipif(not emul.available(), reason="g++ not available")


@pytest.fixture(scope="module")
def L():
    from morefusion_amd import _lib
    lib = emul.build(["gemm_bf16.hip"])
    for name, (argtypes, restype) in _lib._SIGNATURES.items():
        fn = getattr(lib, name, None)
        if fn is not None:
            fn.argtypes, fn.restype = argtypes, restype
    return lib


@pytest.fixture(params=["tile128", "tile256"], autouse=True)
def tile_choice(request, monkeypatch):
    """Every test runs on the 128 x 128 tile of the NT engine and again with MF_NT_BIG=2: the 256 x 256 form
    (k_gemm_nt_bf16_big: eight waves of 128 x 64) wherever its structure allows, whatever the tile count."""
    if request.param == "tile256":
        monkeypatch.setenv("MF_NT_BIG", "2")
    else:
        monkeypatch.setenv("MF_NT_BIG", "0")


def bf(x):
    """float32 tensor rounded to bf16 (kept as bf16)."""
    return x.to(torch.bfloat16)


def p(t):
    return t.data_ptr()


def close(got, want, tol=1e-4):
    scale = float(want.abs().max()) or 1.0
    err = float((got.float() - want.float()).abs().max())
    assert err <= tol * scale, (err, scale)


def close_bf16(got, want):
    """got: bf16 output of a kernel; want: fp32 reference -> within one bf16 rounding of the reference"""
    g, w = got.float(), want.float()
    assert float((g - w).abs().max()) <= float(w.abs().max()) * 2.0 ** -7


def test_cast_rows_and_relu_mask(L):
    torch.manual_seed(0)
    x = torch.randn(37, 21)
    out = torch.full((37, 24), 7.0, dtype=torch.bfloat16)
    assert L.mf_cast_rows_bf16(p(x), 21, p(out), 24, 37, 21, None) == 0
    assert torch.equal(out[:, :21], x.to(torch.bfloat16)) and float(out[:, 21:].abs().max()) == 0.0
    y, dy = bf(torch.randn(5, 64)), bf(torch.randn(5, 64))
    dz = torch.empty_like(dy)
    assert L.mf_relu_mask_bf16(p(y), p(dy), None, p(dz), y.numel(), None) == 0
    assert torch.equal(dz, torch.where(y > 0, dy, torch.zeros_like(dy)))
    g32 = torch.randn(5, 64)
    assert L.mf_relu_mask_bf16(p(y), None, p(g32), p(dz), y.numel(), None) == 0
    assert torch.equal(dz, torch.where(y > 0, g32.to(torch.bfloat16), torch.zeros_like(dy)))


@pytest.mark.parametrize("M,N,K,groups,relu", [(200, 136, 984, 1, 1), (70, 24, 8, 1, 0), (150, 128, 64, 3, 1),
                                               (300, 264, 200, 1, 0)])
def test_linear_bf16_forward_dgrad_wgrad(L, M, N, K, groups, relu):
    """out = act(A W^T + b) in column blocks of wider matrices (row pitches, group strides: how the heads' layers
    2-4 run side by side), its data gradient (the same kernel on W^T) and its weight gradient (TN engine)."""
    torch.manual_seed(1)
    lda, ldo = groups * K + 8, groups * N + 3
    A = bf(torch.randn(M, lda))
    W = bf(torch.randn(groups, N, K) / K ** 0.5)
    b = torch.randn(groups, N)
    for out_f32 in (1, 0):
        out = torch.full((M, ldo), -9.0, dtype=torch.float32 if out_f32 else torch.bfloat16)
        assert L.mf_linear_bf16(p(A), K, lda, p(W), N * K, K, p(b), N, p(out), N, ldo, M, N, K, groups, relu, out_f32,
                                0, None) == 0
        for g in range(groups):
            want = A[:, g * K:(g + 1) * K].float() @ W[g].float().t() + b[g]
            want = F.relu(want) if relu else want
            (close if out_f32 else close_bf16)(out[:, g * N:(g + 1) * N], want)
        assert float(out[:, groups * N:].float().min()) == -9.0   # nothing written past the last block
    import os
    assert L.mf_gemm_bf16_last_tile() == (256 if os.environ["MF_NT_BIG"] == "2" else 64)  # (which form of the engine ran)
    # accumulate into fp32
    acc = torch.ones(M, ldo)
    assert L.mf_linear_bf16(p(A), K, lda, p(W), N * K, K, None, 0, p(acc), N, ldo, M, N, K, groups, 0, 1, 1, None) == 0
    close(acc[:, :N], 1.0 + A[:, :K].float() @ W[0].float().t())
    # data gradient: dA = dY W  ==  linear(dY, W^T)
    dY = bf(torch.randn(M, N))
    Wt = W[0].t().contiguous()                               # [K, N]
    Np = (N + 7) // 8 * 8                                     # the transposed weight needs N % 8 == 0 columns
    Wt_p = torch.zeros(K, Np, dtype=torch.bfloat16)
    Wt_p[:, :N] = Wt
    dY_p = torch.zeros(M, Np, dtype=torch.bfloat16)
    dY_p[:, :N] = dY
    dA = torch.empty(M, K)
    assert L.mf_linear_bf16(p(dY_p), 0, Np, p(Wt_p), 0, Np, None, 0, p(dA), 0, K, M, K, Np, 1, 0, 1, 0, None) == 0
    close(dA, dY.float() @ W[0].float())
    # weight gradient, with and without the split over rows
    if N % 8 == 0:
        for split in (1, 3, 40):   # (40 slabs of a small result: the one-wave-per-weight finish)
            dW = torch.full((groups, N, K), 5.0)
            dYg = bf(torch.randn(M, groups * N))
            ws = torch.empty(max(split, 1) * groups * N * K)
            assert L.mf_linear_wgrad_bf16(p(dYg), N, groups * N, p(A), K, lda, p(dW), N * K, K, p(ws), M, N, K, groups,
                                          split, None) == 0
            for g in range(groups):
                close(dW[g], dYg[:, g * N:(g + 1) * N].float().t() @ A[:, g * K:(g + 1) * K].float())


@pytest.mark.parametrize("B,Cin,Cout,D,w_cin,c_off", [(1, 8, 136, 8, 8, 0), (2, 16, 128, 16, 24, 8), (1, 40, 64, 16, 40, 0),
                                                   (1, 24, 200, 8, 24, 0)])
def test_conv3d_k4s2_bf16_forward_dgrad_wgrad(L, B, Cin, Cout, D, w_cin, c_off):
    """Convolution3D(Cin, Cout, 4, 2, pad = 1) on channels-last bf16 grids: forward (+ bias + ReLU), data gradient
    (parity-class GEMMs, scattered back to voxel rows; bf16, fp32 and accumulating outputs) and weight gradient
    (framework layout, channel offset into a wider weight) vs torch.nn.functional.conv3d + autograd in float32."""
    torch.manual_seed(2)
    Do = D // 2
    x_cf = bf(torch.randn(B, Cin, D, D, D)).float().requires_grad_(True)
    Wfull = bf(torch.randn(Cout, w_cin, 4, 4, 4) / (Cin * 64) ** 0.5).float()
    W = Wfull[:, c_off:c_off + Cin].clone().requires_grad_(True)
    bias = torch.randn(Cout)
    y_ref = F.conv3d(x_cf, W, bias, stride=2, padding=1)
    dy_cf = bf(torch.randn_like(y_ref)).float()
    y_ref.backward(dy_cf)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()  # noqa: E731
    x_cl, dy_cl = cl(x_cf.detach()).to(torch.bfloat16), cl(dy_cf).to(torch.bfloat16)
    wt = torch.empty(Cout, 64, Cin, dtype=torch.bfloat16)
    wd = torch.empty(8, Cin, 8, Cout, dtype=torch.bfloat16)
    assert L.mf_conv3d_k4s2_pack_bf16(p(Wfull), Cout, Cin, w_cin, c_off, p(wt), p(wd), None) == 0
    assert torch.equal(wt, W.detach().reshape(Cout, Cin, 64).permute(0, 2, 1).to(torch.bfloat16))
    # forward
    for relu, out_f32 in ((0, 1), (1, 0)):
        out = torch.empty(B, Do ** 3, Cout, dtype=torch.float32 if out_f32 else torch.bfloat16)
        assert L.mf_conv3d_k4s2_bf16_fwd(p(x_cl), p(wt), p(bias), p(out), B, Cin, Cout, D, relu, out_f32, None) == 0
        want = cl(F.relu(y_ref.detach()) if relu else y_ref.detach()).reshape(B, Do ** 3, Cout)
        (close if out_f32 else close_bf16)(out, want)
    # forward with the reduction split over fp32 slabs of a workspace (what conv4 at 16 objects takes: 64 tiles for 256
    # CUs; forced here): the same values, bias / ReLU applied once by the finish pass, pitch / dtype of the output kept
    if Cout >= 192 and os.environ.get("MF_NT_BIG") == "2":
        os.environ["MF_NT_SPLITK"] = "3"
        try:
            nws = L.mf_conv3d_bf16_fwd_workspace_bytes(B, Cin, Cout, D, 4, 2, 1, 1)
            assert nws == 3 * B * Do ** 3 * Cout * 4
            ws = torch.empty(nws, dtype=torch.uint8)
            for relu, out_f32 in ((0, 1), (1, 0)):
                out = torch.full((B, Do ** 3, Cout + 8), 5.0, dtype=torch.float32 if out_f32 else torch.bfloat16)
                assert L.mf_conv3d_bf16_fwd_ws(p(x_cl), p(wt), p(bias), p(out), p(ws), nws, B, Cin, Cout, D, 4, 2, 1, 1,
                                               relu, out_f32, Cout + 8, None) == 0
                want = cl(F.relu(y_ref.detach()) if relu else y_ref.detach()).reshape(B, Do ** 3, Cout)
                (close if out_f32 else close_bf16)(out[:, :, :Cout], want)
                assert float((out[:, :, Cout:].float() - 5.0).abs().max()) == 0.0
        finally:
            del os.environ["MF_NT_SPLITK"]
        assert L.mf_conv3d_bf16_fwd_workspace_bytes(B, Cin, Cout, D, 4, 2, 1, 1) == 0  # (few tiles, short K: no split)
    # data gradient
    if Do ** 3 % 128 == 0:
        want = cl(x_cf.grad).reshape(B, D ** 3, Cin)
        dx = torch.empty(B, D ** 3, Cin)
        assert L.mf_conv3d_k4s2_bf16_dgrad(p(dy_cl), p(wd), p(dx), B, Cin, Cout, D, 1, 0, None) == 0
        close(dx, want)
        dxb = torch.empty(B, D ** 3, Cin, dtype=torch.bfloat16)
        assert L.mf_conv3d_k4s2_bf16_dgrad(p(dy_cl), p(wd), p(dxb), B, Cin, Cout, D, 0, 0, None) == 0
        close_bf16(dxb, want)
        dx.fill_(2.0)
        assert L.mf_conv3d_k4s2_bf16_dgrad(p(dy_cl), p(wd), p(dx), B, Cin, Cout, D, 1, 1, None) == 0
        close(dx, want + 2.0)
    # weight gradient into the [c_off, c_off + Cin) channels of the full weight's gradient
    for split in (1, 2):
        dW = torch.full((Cout, w_cin, 4, 4, 4), 3.0)
        ws = torch.empty(L.mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(Cin, Cout, split) // 4)
        assert L.mf_conv3d_k4s2_bf16_wgrad(p(dy_cl), p(x_cl), p(dW), p(ws), B, Cin, Cout, D, w_cin, c_off, split,
                                           None) == 0
        close(dW[:, c_off:c_off + Cin], W.grad)
        if w_cin > Cin:
            rest = torch.cat([dW[:, :c_off], dW[:, c_off + Cin:]], 1)
            assert float(rest.min()) == 3.0 and float(rest.max()) == 3.0


@pytest.mark.parametrize("name,Cin_real,Cout,ks,pad,dil", [("conv1_occ", 1, 8, 3, 1, 1), ("conv2_occ", 8, 16, 3, 2, 2)])
def test_occupancy_branch_convolutions_on_the_general_geometry(L, name, Cin_real, Cout, ks, pad, dil):
    """conv1_occ (1 -> 8, k3 p1; the single input channel travels as 8 with 7 zeros) and conv2_occ (8 -> 16, k3,
    dilation 2, p2) of model.py:69-72,120-124: forward, weight gradient (only the real input channels are written)
    and the data gradient as a forward convolution with the flipped / transposed operand."""
    torch.manual_seed(3)
    B, D, Cin = 1, 8, 8
    x_cf = torch.zeros(B, Cin, D, D, D)
    x_cf[:, :Cin_real] = bf(torch.randn(B, Cin_real, D, D, D)).float()
    x_cf.requires_grad_(True)
    W = bf(torch.randn(Cout, Cin_real, ks, ks, ks) / (Cin_real * ks ** 3) ** 0.5).float().requires_grad_(True)
    bias = torch.randn(Cout)
    y = F.conv3d(x_cf[:, :Cin_real], W, bias, stride=1, padding=pad, dilation=dil)
    dy_cf = bf(torch.randn_like(y)).float()
    y.backward(dy_cf)
    cl = lambda t: t.permute(0, 2, 3, 4, 1).contiguous()  # noqa: E731
    x_cl, dy_cl = cl(x_cf.detach()).to(torch.bfloat16), cl(dy_cf).to(torch.bfloat16)
    taps = ks ** 3
    wt = torch.empty(Cout, taps, Cin, dtype=torch.bfloat16)
    wf = torch.empty(Cin, taps, Cout, dtype=torch.bfloat16)
    assert L.mf_conv3d_bf16_pack(p(W.detach()), Cout, Cin, Cin_real, 0, ks, p(wt), None, p(wf), None) == 0
    assert float(wt[:, :, Cin_real:].abs().max() if Cin_real < Cin else 0.0) == 0.0
    # forward into a column block of a wider grid (pitch 24)
    out = torch.full((B, D ** 3, 24), -3.0, dtype=torch.bfloat16)
    assert L.mf_conv3d_bf16_fwd(p(x_cl), p(wt), p(bias), p(out[:, :, 4:]), B, Cin, Cout, D, ks, 1, pad, dil, 1, 0, 24,
                                None) == 0
    close_bf16(out[:, :, 4:4 + Cout], cl(F.relu(y.detach())).reshape(B, D ** 3, Cout))
    assert float(out[:, :, :4].float().max()) == -3.0 and float(out[:, :, 4 + Cout:].float().max()) == -3.0
    # weight gradient (33 slabs: the one-wave-per-weight finish of the small layers)
    for split in (1, 2, 33):
        dW = torch.full((Cout, Cin_real, ks, ks, ks), 9.0)
        ws = torch.empty(L.mf_conv3d_bf16_wgrad_workspace_bytes(Cin, Cout, ks, split) // 4)
        assert L.mf_conv3d_bf16_wgrad(p(dy_cl), p(x_cl), p(dW), p(ws), B, Cin, Cout, D, ks, 1, pad, dil, Cin_real, 0,
                                      split, None) == 0
        close(dW, W.grad)
    # data gradient = conv(dy, flipT) with pad' = dil (ks - 1) - pad
    if Cin_real == Cin:
        dx = torch.empty(B, D ** 3, Cin)
        assert L.mf_conv3d_bf16_fwd(p(dy_cl), p(wf), None, p(dx), B, Cout, Cin, D, ks, 1, dil * (ks - 1) - pad, dil, 0, 1,
                                    Cin, None) == 0
        close(dx, cl(x_cf.grad).reshape(B, D ** 3, Cin))
