"""The bf16 MFMA kernels of csrc/gemm_bf16.hip on the MI355X (BASELINE config 5: bf16 training; `--dtype bf16`):
conv3 / conv4 (model.py:73-74,125-139) and the 1x1 convolution chains (model.py:76-91,239-258), forward, data
gradient and weight gradient, against torch's float32 operators on the same bf16-rounded operands; and one
training step of the whole pose network on these kernels against the stock bf16-autocast step.
Tolerances: a bf16 output is within one bf16 rounding (2^-8 relative to the tensor's largest value) of the fp32
reference; gradients that pass through a bf16-rounded ReLU output within 2 %."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

import morefusion_amd as mf  # noqa: E402
from morefusion_amd.contrib.singleview_3d.models import bf16_ops as K  # noqa: E402


def rel(a, b):
    return float((a.detach().float() - b.detach().float()).abs().max() / (b.detach().float().abs().max() + 1e-30))


def grad_close(a, b):
    """Gradients behind a bf16-rounded ReLU output: an element of the output within one rounding of zero flips its
    mask, so single elements may be off by a few per cent of the largest value; the tensor as a whole agrees to
    better than 1 % in L2."""
    a, b = a.detach().float(), b.detach().float()
    l2 = float((a - b).norm() / (b.norm() + 1e-30))
    assert l2 < 1e-2 and rel(a, b) < 5e-2, (l2, rel(a, b))


@pytest.fixture(params=["by_size", "ping_pong"])
def engine_form(request, monkeypatch):
    """``ping_pong``: MF_NT_BIG=2 -- the 256 x 256 LDS-DMA ping-pong forms (k_gemm_nt_bf16_pp, k_gemm_tn_bf16_pp, round 6)
    wherever their structure allows, whatever the tile count (at the network's full batch they are chosen by size:
    test_ping_pong_engines_at_full_size_...); ``by_size``: the launcher's own choice at these small batches."""
    if request.param == "ping_pong":
        monkeypatch.setenv("MF_NT_BIG", "2")
    return request.param


@pytest.mark.parametrize("B,Cin,Cout,D,w_cin,c_off", [(2, 256, 512, 16, 256, 0), (1, 160, 256, 32, 160, 0),
                                                      (2, 16, 256, 32, 160, 144)])
def test_conv3d_k4s2_bf16_forward_backward_vs_fp32(B, Cin, Cout, D, w_cin, c_off, engine_form):
    torch.manual_seed(0)
    conv = torch.nn.Conv3d(w_cin, Cout, 4, 2, padding=1).cuda()
    x = torch.randn(B, D ** 3, Cin, device="cuda").to(torch.bfloat16).requires_grad_(True)
    out = K.conv3d_k4s2(x, conv, D, relu=True, c_off=c_off)
    g = torch.randn(out.shape, device="cuda").to(torch.bfloat16)
    out.backward(g)
    xr = x.detach().float().reshape(B, D, D, D, Cin).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
    wr = conv.weight.detach().to(torch.bfloat16).float()[:, c_off:c_off + Cin].contiguous().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    with torch.backends.cudnn.flags(enabled=True, benchmark=False):
        y = F.relu(F.conv3d(xr, wr, br, stride=2, padding=1))
        y.backward(g.float().reshape(B, D // 2, D // 2, D // 2, Cout).permute(0, 4, 1, 2, 3).contiguous())
    assert rel(out, y.permute(0, 2, 3, 4, 1).reshape(B, -1, Cout)) < 2 ** -7
    grad_close(x.grad, xr.grad.permute(0, 2, 3, 4, 1).reshape(B, -1, Cin))
    grad_close(conv.weight.grad[:, c_off:c_off + Cin], wr.grad)
    grad_close(conv.bias.grad, br.grad)
    if c_off:
        assert float(conv.weight.grad[:, :c_off].abs().max()) == 0.0


@pytest.mark.parametrize("n,Kin,N,relu", [(2000, 3, 8, True), (2000, 128, 63, False), (16000, 984, 1920, True),
                                           (1000, 640, 256, True)])
def test_linear_bf16_forward_backward_vs_fp32(n, Kin, N, relu, engine_form):
    torch.manual_seed(1)
    conv = torch.nn.Conv1d(Kin, N, 1).cuda()
    x = (torch.randn(n, Kin, device="cuda").to(torch.bfloat16) if Kin % 8 == 0 else torch.randn(n, Kin, device="cuda"))
    x = x.requires_grad_(True)
    out = K.linear(x, conv, relu=relu)
    g = torch.randn(n, N, device="cuda").to(torch.bfloat16)
    out.backward(g)
    xr = x.detach().to(torch.bfloat16).float().requires_grad_(True)
    wr = conv.weight.detach().reshape(N, Kin).to(torch.bfloat16).float().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y = xr @ wr.t() + br
    y = F.relu(y) if relu else y
    y.backward(g.float())
    assert rel(out, y) < 2 ** -7
    grad_close(x.grad, xr.grad)
    grad_close(conv.weight.grad.reshape(N, Kin), wr.grad)
    grad_close(conv.bias.grad, br.grad)


def test_ping_pong_engines_at_full_size_match_fp32_and_are_bitwise_reproducible():
    """The LDS-DMA ping-pong engines at the shapes that select them by size -- conv4 (256 -> 512 on 16^3) at the
    training batch of 16 objects: forward (split-K x 4 over fp32 slabs + finish pass), data gradient, weight gradient
    (TN form, split 2) -- against torch's float32 convolution on the same bf16-rounded operands (the objects of a
    slice), and a race screen: the kernels hand LDS stages between DMA requests and fragment reads of two wave groups
    on counted waits and raw barriers; an ordering hole would show as a rare wrong tile.  Every launch of a
    deterministic kernel must give the very same bits: 12 launches each, under load."""
    L, st = mf._lib.lib(), mf._lib.stream_ptr
    p = lambda t: t.data_ptr()  # noqa: E731
    torch.manual_seed(3)
    B, Cin, Cout, D = 16, 256, 512, 16
    Do = D // 2
    dev, bf = "cuda", torch.bfloat16
    x = torch.randn(B, D ** 3, Cin, device=dev).to(bf)
    dy = torch.randn(B, Do ** 3, Cout, device=dev).to(bf)
    W = torch.randn(Cout, Cin, 4, 4, 4, device=dev) / (64 * Cin) ** 0.5
    bias = torch.randn(Cout, device=dev)
    wt = torch.empty(Cout, 64, Cin, dtype=bf, device=dev)
    wd = torch.empty(8, Cin, 8, Cout, dtype=bf, device=dev)
    mf._lib.check(L.mf_conv3d_k4s2_pack_bf16(p(W), Cout, Cin, Cin, 0, p(wt), p(wd), st()), "pack")
    nws = L.mf_conv3d_bf16_fwd_workspace_bytes(B, Cin, Cout, D, 4, 2, 1, 1)
    assert nws == 4 * B * Do ** 3 * Cout * 4  # 64 tiles of 256 x 256 for 256 CUs: split-K x 4
    ws = torch.empty(nws, dtype=torch.uint8, device=dev)
    split = L.mf_conv3d_k4s2_bf16_wgrad_default_split(B, Cin, Cout, D)
    wsw = torch.empty(L.mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(Cin, Cout, split), dtype=torch.uint8, device=dev)

    def fwd():
        y = torch.empty(B, Do ** 3, Cout, dtype=bf, device=dev)
        mf._lib.check(L.mf_conv3d_bf16_fwd_ws(p(x), p(wt), p(bias), p(y), p(ws), nws, B, Cin, Cout, D, 4, 2, 1, 1, 1, 0, Cout, st()), "fwd")
        assert L.mf_gemm_bf16_last_tile() == 256
        return y

    def dgrad():
        dx = torch.empty(B, D ** 3, Cin, dtype=bf, device=dev)
        mf._lib.check(L.mf_conv3d_k4s2_bf16_dgrad(p(dy), p(wd), p(dx), B, Cin, Cout, D, 0, 0, st()), "dgrad")
        assert L.mf_gemm_bf16_last_tile() == 256
        return dx

    def wgrad():
        dW = torch.empty_like(W)
        mf._lib.check(L.mf_conv3d_k4s2_bf16_wgrad(p(dy), p(x), p(dW), p(wsw), B, Cin, Cout, D, Cin, 0, split, st()), "wgrad")
        return dW

    first = {}
    for name, fn in (("fwd", fwd), ("dgrad", dgrad), ("wgrad", wgrad)):
        first[name] = fn()
        for _ in range(11):
            assert torch.equal(fn(), first[name]), name
    # float32 reference: objects 0 and 15 for forward / data gradient, all of them for the weight gradient
    wr = W.to(bf).float().requires_grad_(True)
    for b in (0, B - 1):
        xr = x[b:b + 1].float().reshape(1, D, D, D, Cin).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
        with torch.backends.cudnn.flags(enabled=True, benchmark=False):
            pre = F.conv3d(xr, wr, bias, stride=2, padding=1)
            pre.backward(dy[b:b + 1].float().reshape(1, Do, Do, Do, Cout).permute(0, 4, 1, 2, 3).contiguous())
        assert rel(first["fwd"][b], F.relu(pre.detach()).permute(0, 2, 3, 4, 1).reshape(-1, Cout)) < 2 ** -7
        assert rel(first["dgrad"][b], xr.grad.permute(0, 2, 3, 4, 1).reshape(-1, Cin)) < 2 ** -7
    wr.grad = None
    xa = x.float().reshape(B, D, D, D, Cin).permute(0, 4, 1, 2, 3).contiguous()
    with torch.backends.cudnn.flags(enabled=True, benchmark=False):
        F.conv3d(xa, wr, None, stride=2, padding=1).backward(dy.float().reshape(B, Do, Do, Do, Cout).permute(0, 4, 1, 2, 3).contiguous())
    assert rel(first["wgrad"], wr.grad) < 2e-3


def test_training_step_on_bf16_kernels_matches_stock_autocast_step():
    """Model.forward + backward under bf16 autocast with ``bf16_kernels`` on (conv3 / conv4 / all 1x1 convolutions on
    csrc/gemm_bf16.hip) and off (MIOpen / hipBLASLt): same loss within 1 %, parameter gradients of every layer of the
    volumetric part aligned (cosine > 0.98: two bf16 evaluations of the same function), no stock 3-D convolution of
    conv3 / conv4 in the hand-written run."""
    from morefusion_amd.contrib.singleview_3d.models import Model, PitchTableModels
    torch.manual_seed(0)
    rs = np.random.RandomState(0)
    pcds = {c: rs.uniform(-0.05, 0.05, (800, 3)).astype(np.float32) for c in mf.synthetic.CLASS_PITCH}
    base = Model(n_fg_class=21, with_occupancy=True, models=PitchTableModels(pcds)).cuda().train()
    b = mf.synthetic.make_singleview_batch(2, seed=20)
    inputs = {k: torch.as_tensor(b[k]).cuda() for k in
              ("class_id", "rgb", "pcd", "pitch", "origin", "grid_nontarget_empty", "quaternion_true",
               "translation_true")}
    grads, losses = {}, {}
    for flag in (True, False):
        model = copy.deepcopy(base)
        model.bf16_kernels = flag
        model.pspnet_extractor.bf16_tail_kernels = flag
        np.random.seed(1)
        torch.manual_seed(1)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = model(**inputs)
        loss.backward()
        losses[flag] = float(loss.detach())
        grads[flag] = {n: p.grad.detach().float().flatten() for n, p in model.named_parameters() if p.grad is not None}
    assert np.isfinite(losses[True]) and abs(losses[True] - losses[False]) < 0.01 * abs(losses[False]), losses
    checked = 0
    for name, g_hw in grads[True].items():
        # the volumetric part (point MLP, occupancy convs, conv3/4, heads) and PSPNet's sampled tail
        if name.startswith(("conv", "pspnet_extractor.up3", "pspnet_extractor.conv1")):
            g_st = grads[False][name]
            cos = float(torch.dot(g_hw, g_st) / (g_hw.norm() * g_st.norm() + 1e-30))
            assert cos > 0.98, (name, cos)
            checked += 1
    assert checked >= 35


def test_occupancy_branch_on_the_bf16_kernels_vs_fp32():
    """conv1_occ (1 -> 8, k3 p1) -> ReLU -> conv2_occ (8 -> 16, k3, dilation 2, p2) -> ReLU (model.py:69-72,120-124)
    at the network's size (32^3, B = 2) through the general-geometry kernels: values and parameter gradients vs
    torch's float32 convolutions."""
    torch.manual_seed(2)
    B, D = 2, 32
    c1 = torch.nn.Conv3d(1, 8, 3, 1, padding=1).cuda()
    c2 = torch.nn.Conv3d(8, 16, 3, 1, padding=2, dilation=2).cuda()
    grid = (torch.rand(B, D, D, D, device="cuda") > 0.5).float()
    g8 = torch.zeros(B, D ** 3, 8, dtype=torch.bfloat16, device="cuda")
    g8[:, :, 0] = grid.reshape(B, -1)
    out = K.conv3d(K.conv3d(g8, c1, D), c2, D)
    g = torch.randn(out.shape, device="cuda").to(torch.bfloat16)
    out.backward(g)
    r1, r2 = copy.deepcopy(c1), copy.deepcopy(c2)
    for r in (r1, r2):
        r.weight.grad = r.bias.grad = None
        with torch.no_grad():
            r.weight.copy_(r.weight.to(torch.bfloat16).float())
    h1 = F.relu(r1(grid[:, None]))
    h1b = h1.to(torch.bfloat16).float()
    y = F.relu(r2(h1b.detach().requires_grad_(True)))
    assert rel(out, y.permute(0, 2, 3, 4, 1).reshape(B, -1, 16)) < 2 ** -7
    h1l = h1b.detach().requires_grad_(True)
    y2 = F.relu(r2(h1l))
    y2.backward(g.float().reshape(B, D, D, D, 16).permute(0, 4, 1, 2, 3))
    grad_close(c2.weight.grad, r2.weight.grad)
    grad_close(c2.bias.grad, r2.bias.grad)
    h1.backward((h1l.grad * (h1b > 0)).to(torch.bfloat16).float())
    grad_close(c1.weight.grad, r1.weight.grad)
    grad_close(c1.bias.grad, r1.bias.grad)


@pytest.mark.parametrize("B,X,C,P,sorted_rows", [(4, 16, 256, 1000, True), (4, 8, 512, 1000, True),
                                                  (3, 16, 64, 700, False)])
def test_channels_last_sampler_forward_backward_vs_oracle(B, X, C, P, sorted_rows):
    """InterpolateVoxelGridCL at the network's sizes (h3: 16^3 x 256, h4: 8^3 x 512, 1000 points per object;
    model.py:131,141) against the oracle's channels-first float32 restatement of interpolate_voxel_grid.py:61-215 on
    the same bf16-rounded inputs.  Points outside the grid and a NaN row included; with and without the items' row
    offsets (shuffled rows: the any-order scan of the backward)."""
    from oracle import oracle_np as O
    rs = np.random.RandomState(5)
    n = B * P
    vox = torch.from_numpy(rs.uniform(-1, 1, (B, X ** 3, C)).astype(np.float32)).to(torch.bfloat16)
    sp = rs.uniform(-0.6, X - 0.4, (n, 3)).astype(np.float32)
    sp[11] = np.nan
    bi = np.repeat(np.arange(B), P).astype(np.int32)
    if not sorted_rows:
        bi = bi[rs.permutation(n)]
    g = torch.from_numpy(rs.uniform(-1, 1, (n, C)).astype(np.float32)).to(torch.bfloat16)
    vg = vox.cuda().requires_grad_(True)
    bs = torch.arange(B + 1, dtype=torch.int32, device="cuda") * P if sorted_rows else None
    out = K.InterpolateVoxelGridCL.apply(vg, torch.from_numpy(sp).cuda(), torch.from_numpy(bi).cuda(), X, bs)
    out.backward(g.cuda())
    want = O.interpolate_voxel_grid(vox.float().transpose(1, 2).reshape(B, C, X, X, X).numpy(), sp, bi)
    assert float((out.float().cpu() - torch.from_numpy(want)).abs().max()) <= 2.0 ** -8 * float(np.abs(want).max()) + 1e-6
    gv = O.interpolate_voxel_grid_backward(g.float().numpy(), sp, bi, (B, C, X, X, X))
    gv_cl = torch.from_numpy(gv).reshape(B, C, X ** 3).transpose(1, 2)
    assert float((vg.grad.float().cpu() - gv_cl).abs().max()) <= 2.0 ** -7 * float(gv_cl.abs().max()) + 1e-5


@pytest.mark.parametrize("B,P", [(4, 1000), (1, 300)])
def test_sparse_conv3_forward_and_all_gradients_vs_float32_dense_conv3d(B, P):
    """SparseConv3 (round 5, csrc/sparseconv_bf16.hip) at the network's shapes -- 144 voxelized channels as compact
    rows of the occupied voxels + 16 dense occupancy channels, conv3 = Convolution3D(160, 256, 4, 2, pad=1) + ReLU
    (model.py:73,113-128) -- against torch's float32 ``conv3d`` over the dense grid built from the same bf16-rounded
    operands: values within one bf16 rounding, the ReLU masks agree on all but a sliver, and with the operator's mask
    the gradients to the point rows (through the voxel means), to the occupancy channels, to both channel ranges of
    the weight and to the bias agree to 1 % -- and the dense bf16 engines on the [B, 32^3, 160] grid (round 4's path)
    give the same result."""
    torch.manual_seed(0)
    rs = np.random.RandomState(1)
    D, Cs, Co, Cout = 32, 144, 16, 256
    n = B * P
    # a surface-like cloud: points on a sphere shell + noise, some outside the grid
    u = rs.normal(size=(n, 3)); u /= np.linalg.norm(u, axis=1, keepdims=True)
    pts = (15.5 + 11.0 * u + rs.normal(scale=0.4, size=(n, 3))).astype(np.float32)
    pts[:5] = rs.uniform(-3, D + 2, (5, 3))
    bi = np.repeat(np.arange(B, dtype=np.int32), P)
    dev = torch.device("cuda", 0)
    conv = torch.nn.Conv3d(Cs + Co, Cout, 4, 2, padding=1).to(dev)
    feat = torch.randn(n, Cs, device=dev).to(torch.bfloat16).requires_grad_(True)
    hocc = torch.randn(B, D ** 3, Co, device=dev).to(torch.bfloat16).requires_grad_(True)
    pts_d, bi_d = torch.from_numpy(pts).to(dev), torch.from_numpy(bi).to(dev)
    out = K.SparseConv3.apply(feat, hocc, pts_d, bi_d, conv.weight, conv.bias, B, D)
    g = torch.randn(out.shape, device=dev).to(torch.bfloat16)
    out.backward(g)
    got = dict(out=out.detach(), feat=feat.grad.clone(), occ=hocc.grad.clone(), w=conv.weight.grad.clone(),
               b=conv.bias.grad.clone())

    # float32 reference on the dense grid
    fr = feat.detach().float().requires_grad_(True)
    idx = np.round(pts).astype(np.int64)
    ok = ((idx >= 0) & (idx < D)).all(1)
    key = bi.astype(np.int64) * D ** 3 + (idx[:, 0] * D + idx[:, 1]) * D + idx[:, 2]
    keys, okt = torch.from_numpy(np.where(ok, key, 0)).to(dev), torch.from_numpy(ok).to(dev)
    cnt = torch.zeros(B * D ** 3, device=dev).index_add_(0, keys[okt], torch.ones(int(ok.sum()), device=dev))
    sums = torch.zeros(B * D ** 3, Cs, device=dev).index_add_(0, keys[okt], fr[okt])
    means = sums / cnt.clamp(min=1)[:, None]
    means_b = means + (means.detach().to(torch.bfloat16).float() - means.detach())
    hr = hocc.detach().float().requires_grad_(True)
    x = torch.cat((means_b.reshape(B, D, D, D, Cs), hr.reshape(B, D, D, D, Co)), dim=4)
    wr = conv.weight.detach().to(torch.bfloat16).float().requires_grad_(True)
    br = conv.bias.detach().clone().requires_grad_(True)
    y_pre = F.conv3d(x.permute(0, 4, 1, 2, 3), wr, br, stride=2, padding=1)
    y_cl = F.relu(y_pre).detach().permute(0, 2, 3, 4, 1).reshape(B, -1, Cout)
    assert rel(got["out"], y_cl) < 2 ** -7
    mask = got["out"].float() > 0
    assert float((mask != (y_cl > 0)).float().mean()) < 5e-3
    y_pre.backward((g.float() * mask).reshape(B, D // 2, D // 2, D // 2, Cout).permute(0, 4, 1, 2, 3))
    assert rel(got["feat"], fr.grad) < 1e-2 and float(got["feat"][~okt].abs().max()) == 0.0
    assert rel(got["occ"], hr.grad) < 1e-2
    assert rel(got["w"][:, :Cs], wr.grad[:, :Cs]) < 1e-2 and rel(got["w"][:, Cs:], wr.grad[:, Cs:]) < 1e-2
    assert rel(got["b"], br.grad) < 1e-2
    assert int((cnt > 1).sum()) > 10 and int(ok.sum()) < n

    # bitwise reproducible (round 6: the compact rows are handed out by a prefix scan over the voxel index, no atomics;
    # the order of the rows inside a class fixes the fp32 summation order of the weight gradient): two more runs
    # give the very same bits for the output and for EVERY gradient
    for _ in range(2):
        conv.zero_grad()
        f3 = feat.detach().clone().requires_grad_(True)
        h3 = hocc.detach().clone().requires_grad_(True)
        o3 = K.SparseConv3.apply(f3, h3, pts_d, bi_d, conv.weight, conv.bias, B, D)
        o3.backward(g)
        assert torch.equal(o3.detach(), got["out"]) and torch.equal(f3.grad, got["feat"]) and torch.equal(h3.grad, got["occ"])
        assert torch.equal(conv.weight.grad, got["w"]) and torch.equal(conv.bias.grad, got["b"])

    # round 4's dense path on the same operands: the same operator result up to bf16 rounding
    conv.zero_grad()
    f2 = feat.detach().clone().requires_grad_(True)
    h2 = hocc.detach().clone().requires_grad_(True)
    x3 = K.AverageVoxelizationCL.apply(f2, pts_d, bi_d, B, D, Cs + Co)
    x3 = torch.cat((x3[:, :, :Cs], h2), dim=2)
    out_d = K.conv3d_k4s2(x3, conv, D)
    out_d.backward(g)
    assert rel(out_d, got["out"]) < 2 ** -6

    def l2(a, b):
        return float((a.float() - b.float()).norm() / (b.float().norm() + 1e-30))

    # (two bf16 paths, each with its own ReLU mask next to zero: twice the flip noise of a comparison with float32)
    assert l2(f2.grad, got["feat"]) < 5e-2 and l2(conv.weight.grad, got["w"]) < 5e-2


def test_training_pose_epilogue_forward_and_gradients_vs_torch_composite():
    """K.PoseEpilogue (csrc/pointops.hip k_pose_epi3_fwd / _bwd) vs the torch composite of model.py:262-273 at the
    training batch's shape: values and the three heads' gradient rows; NaN for a class id without a head."""
    torch.manual_seed(5)
    B, P, nf = 16, 1000, 21
    n = B * P
    dev = "cuda"
    orot = torch.randn(n, 4 * nf, device=dev, requires_grad=True)
    otrn = torch.randn(n, 3 * nf, device=dev, requires_grad=True)
    ocnf = torch.randn(n, nf, device=dev, requires_grad=True)
    class_id = torch.randint(1, nf + 1, (B,), device=dev)
    pts = torch.rand(n, 3, device=dev) * 32
    pitch = torch.rand(B, device=dev) * 0.01 + 0.002
    origin = torch.randn(B, 3, device=dev) * 0.1

    def composite(cid):
        ar = torch.arange(B, device=dev)
        fg = (cid - 1).long()
        q = orot.reshape(B, P, nf, 4)[ar, :, fg]
        q = q / (q.norm(dim=2, keepdim=True) + 1e-5)
        pc = pts.reshape(B, P, 3) * pitch[:, None, None] + origin[:, None, :]
        t = pc + otrn.reshape(B, P, nf, 3)[ar, :, fg] * pitch[:, None, None]
        return q, t, torch.sigmoid(ocnf).reshape(B, P, nf)[ar, :, fg]

    q, t, c = K.PoseEpilogue.apply(orot, otrn, ocnf, class_id, pts, pitch, origin, B, P, nf)
    gq, gt, gc = torch.randn_like(q), torch.randn_like(t), torch.randn_like(c)
    torch.autograd.backward([q, t, c], [gq, gt, gc])
    got = [x.grad.clone() for x in (orot, otrn, ocnf)]
    for x in (orot, otrn, ocnf):
        x.grad = None
    qr, tr, cr = composite(class_id)
    torch.autograd.backward([qr, tr, cr], [gq, gt, gc])
    for a, b in ((q, qr), (t, tr), (c, cr)):
        torch.testing.assert_close(a, b, rtol=3e-6, atol=3e-7)
    for a, b in zip(got, (orot.grad, otrn.grad, ocnf.grad)):
        torch.testing.assert_close(a, b, rtol=3e-5, atol=3e-6)
    bad = class_id.clone()
    bad[0], bad[1] = 0, nf + 1
    q, t, c = K.PoseEpilogue.apply(orot, otrn, ocnf, bad, pts, pitch, origin, B, P, nf)
    assert torch.isnan(q[:2]).all() and torch.isnan(t[:2]).all() and torch.isnan(c[:2]).all()
    assert torch.isfinite(q[2:]).all()


def test_confidence_loss_forward_and_gradients_vs_torch_composite():
    """functions.loss.confidence_loss on the GPU (csrc/loss.hip) vs its torch composite (model.py:417-434)."""
    import importlib
    CL = importlib.import_module("morefusion_amd.functions.loss.confidence_loss")
    torch.manual_seed(3)
    for B, P in ((16, 1000), (3, 77)):
        add = torch.rand(B, P, device="cuda", requires_grad=True)
        conf = (torch.rand(B, P, device="cuda") - 0.2).requires_grad_(True)
        loss = CL.confidence_loss(add, conf, 0.015)
        loss.backward()
        ga, gc = add.grad.clone(), conf.grad.clone()
        a2, c2 = add.detach().cpu().requires_grad_(True), conf.detach().cpu().requires_grad_(True)
        ref = CL.confidence_loss(a2, c2, 0.015)   # CPU tensors: the composite
        ref.backward()
        np.testing.assert_allclose(float(loss.detach()), float(ref.detach()), rtol=3e-6)
        torch.testing.assert_close(ga.cpu(), a2.grad, rtol=2e-5, atol=1e-9)
        torch.testing.assert_close(gc.cpu(), c2.grad, rtol=2e-5, atol=1e-9)


def test_sampled_pspnet_tail_on_the_bf16_kernels_vs_torch_formulation():
    """PSPNetExtractor._tail_rows_bf16 at the training shape (128^2 map, 1000 pixels per object): the window rows
    bit-for-bit against the same float32 arithmetic in torch; features and gradients against the float32 torch
    formulation ``_tail`` (PReLU slope 1 -> no mask flips: max-norm 2 %; the trained slope: L2)."""
    from morefusion_amd.models import backbone2d, ops2d
    torch.manual_seed(11)
    B, H, W, P = 4, 128, 128, 1000
    net = backbone2d.PSPNetExtractor().cuda()
    u2 = torch.randn(B, 64, H, W, device="cuda").to(torch.bfloat16).float().contiguous(memory_format=torch.channels_last)
    pix = torch.randint(0, 4 * H * W, (B, P), device="cuda")
    pix[0, :4] = torch.tensor([0, 2 * W - 1, (2 * H - 1) * 2 * W, 4 * H * W - 1], device="cuda")
    rows = ops2d.tail_rows(u2.to(torch.bfloat16), pix)
    taps = net._tail_taps(pix, H, W)
    flat = u2.permute(0, 2, 3, 1).reshape(B, H * W, 64)

    def tap(iy, ix):
        return torch.gather(flat, 1, (iy * W + ix)[:, :, None].expand(B, P * 9, 64))
    ly, lx = taps["ly"][:, :, None], taps["lx"][:, :, None]
    up = (1 - ly) * ((1 - lx) * tap(taps["y0"], taps["x0"]) + lx * tap(taps["y0"], taps["x1"])) + \
        ly * ((1 - lx) * tap(taps["y1"], taps["x0"]) + lx * tap(taps["y1"], taps["x1"]))
    up = (up * taps["valid"][:, :, None]).reshape(B * P, 9, 64).permute(0, 2, 1).reshape(B * P, 576)
    assert torch.equal(rows, up.to(torch.bfloat16))

    def l2(a, b):
        return float((a.detach().float() - b.detach().float()).norm() / b.detach().float().norm())

    # slope 0.25 (the initial value): a pre-activation that the bf16 forward rounds across 0 takes the other slope -- a
    # quarter of a per cent of them do, each moving its 576 window gradients by 75 % -> a few per cent in L2
    for slope, err, tol in ((1.0, rel, 2e-2), (0.25, l2, 6e-2)):
        with torch.no_grad():
            net.up3.prelu.weight.fill_(slope)
        net.zero_grad()
        ua = u2.clone().requires_grad_(True)
        out = net._tail_rows_bf16(ua, pix)
        g = torch.randn_like(out)
        out.backward(g)
        got = {k: p.grad.clone() for k, p in net.named_parameters() if p.grad is not None}
        got_u = ua.grad.clone()
        net.zero_grad()
        ub = u2.clone().contiguous().requires_grad_(True)
        ref = net._tail(ub, taps)
        ref.backward(g.reshape(B, P, 32).transpose(1, 2))
        assert rel(out, ref.transpose(1, 2).reshape(B * P, 32)) < 2e-2
        assert l2(got_u, ub.grad) < (1e-2 if slope == 1.0 else tol) and err(got_u, ub.grad) < 6e-2
        assert set(got) == {"up3.conv.weight", "up3.conv.bias", "up3.prelu.weight", "conv1.weight", "conv1.bias"}
        for k, v in got.items():
            # (the slope's gradient is ONE number: a sum of 256 000 products of either sign that cancels to ~1e-4 of
            # its absolute mass, so the bf16 rounding of the addends shows at the per-cent level)
            assert err(v, dict(net.named_parameters())[k].grad) < (0.1 if k == "up3.prelu.weight" else tol), (slope, k)
