"""Time the bf16 MFMA kernels of csrc/gemm_bf16.hip at the network's training shapes (BASELINE config 5 per-GPU
share: 16 objects) and the stock library beside them (torch bf16: MIOpen conv3d forward / backward, hipBLASLt
matmuls).  HIP events, 20 launches after 3 warm-ups; TFLOP/s against the 2.5 PFLOP/s dense bf16 MFMA peak."""
import json
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morefusion_amd import _lib  # noqa: E402

PEAK = 2500.0


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def p(t):
    return t.data_ptr()


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    stock = "--no-stock" not in sys.argv
    quick = "--quick" in sys.argv  # conv3 / conv4 / the heads' first layer only
    L = _lib.lib()
    st = _lib.stream_ptr
    dev = "cuda"
    out = {"B": B}
    torch.manual_seed(0)
    for name, Cin, Cout, D in (("conv3", 160, 256, 32), ("conv4", 256, 512, 16)):
        Do = D // 2
        flop = 2.0 * B * Do ** 3 * Cout * 64 * Cin
        x = torch.randn(B, D ** 3, Cin, device=dev).to(torch.bfloat16)
        dy = torch.randn(B, Do ** 3, Cout, device=dev).to(torch.bfloat16)
        W = torch.randn(Cout, Cin, 4, 4, 4, device=dev) / (64 * Cin) ** 0.5
        bias = torch.randn(Cout, device=dev)
        wt = torch.empty(Cout, 64, Cin, dtype=torch.bfloat16, device=dev)
        wd = torch.empty(8, Cin, 8, Cout, dtype=torch.bfloat16, device=dev)
        L.mf_conv3d_k4s2_pack_bf16(p(W), Cout, Cin, Cin, 0, p(wt), p(wd), st())
        y = torch.empty(B, Do ** 3, Cout, dtype=torch.bfloat16, device=dev)
        dx = torch.empty(B, D ** 3, Cin, dtype=torch.bfloat16, device=dev)
        dW = torch.empty_like(W)
        split = L.mf_conv3d_k4s2_bf16_wgrad_default_split(B, Cin, Cout, D)
        ws = torch.empty(L.mf_conv3d_k4s2_bf16_wgrad_workspace_bytes(Cin, Cout, split), dtype=torch.uint8, device=dev)
        r = {"gflop": round(flop / 1e9, 1), "wgrad_split": split}
        nws = L.mf_conv3d_bf16_fwd_workspace_bytes(B, Cin, Cout, D, 4, 2, 1, 1)   # > 0: the layer splits its reduction
        fws = torch.empty(max(nws, 16), dtype=torch.uint8, device=dev)
        r["fwd_splitk_ws_mb"] = round(nws / 2 ** 20, 1)
        t = timeit(lambda: _lib.check(L.mf_conv3d_bf16_fwd_ws(p(x), p(wt), p(bias), p(y), p(fws), nws, B, Cin, Cout, D, 4, 2, 1, 1, 1, 0, Cout, st()), "fwd"))
        r["fwd_ms"], r["fwd_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        t = timeit(lambda: _lib.check(L.mf_conv3d_k4s2_bf16_dgrad(p(dy), p(wd), p(dx), B, Cin, Cout, D, 0, 0, st()), "dgrad"))
        r["dgrad_ms"], r["dgrad_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        t = timeit(lambda: _lib.check(L.mf_conv3d_k4s2_bf16_wgrad(p(dy), p(x), p(dW), p(ws), B, Cin, Cout, D, Cin, 0, split, st()), "wgrad"))
        r["wgrad_ms"], r["wgrad_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        t = timeit(lambda: _lib.check(L.mf_conv3d_k4s2_pack_bf16(p(W), Cout, Cin, Cin, 0, p(wt), p(wd), st()), "pack"))
        r["pack_ms"] = round(t, 4)
        # numerics vs torch on a slice (fp32 conv of the bf16-rounded operands)
        xc = x[:1].float().reshape(1, D, D, D, Cin).permute(0, 4, 1, 2, 3)
        ref = F.relu(F.conv3d(xc, W.to(torch.bfloat16).float(), bias, stride=2, padding=1)).permute(0, 2, 3, 4, 1).reshape(Do ** 3, Cout)
        r["fwd_max_rel_err"] = float((y[0].float() - ref).abs().max() / ref.abs().max())
        if stock:
            xs = x.reshape(B, D, D, D, Cin).permute(0, 4, 1, 2, 3).contiguous().requires_grad_(True)
            Ws = W.to(torch.bfloat16).requires_grad_(True)
            bs = bias.to(torch.bfloat16)
            r["stock_fwd_ms"] = round(timeit(lambda: F.conv3d(xs, Ws, bs, stride=2, padding=1), 5), 4)
            ys = F.conv3d(xs, Ws, bs, stride=2, padding=1)
            g = torch.randn_like(ys)
            r["stock_bwd_ms"] = round(timeit(lambda: torch.autograd.grad(ys, (xs, Ws), g, retain_graph=True), 5), 4)
        out[name] = r
        print(json.dumps({name: r}), flush=True)
    n = B * 1000
    for name, K, N in (("heads1", 992, 1920), ("heads2", 640, 256), ("mlp_conv2_rgb", 64, 128))[:1 if quick else 3]:
        flop = 2.0 * n * K * N
        A = torch.randn(n, K, device=dev).to(torch.bfloat16)
        W = (torch.randn(N, K, device=dev) / K ** 0.5).to(torch.bfloat16)
        Wt = W.t().contiguous()
        bias = torch.randn(N, device=dev)
        y = torch.empty(n, N, dtype=torch.bfloat16, device=dev)
        dA = torch.empty(n, K, dtype=torch.bfloat16, device=dev)
        dW = torch.empty(N, K, device=dev)
        split = 4
        ws = torch.empty(split * N * K, device=dev)
        r = {"gflop": round(flop / 1e9, 1)}
        t = timeit(lambda: _lib.check(L.mf_linear_bf16(p(A), 0, K, p(W), 0, K, p(bias), 0, p(y), 0, N, n, N, K, 1, 1, 0, 0, st()), "fwd"))
        r["fwd_ms"], r["fwd_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        t = timeit(lambda: _lib.check(L.mf_linear_bf16(p(y), 0, N, p(Wt), 0, N, None, 0, p(dA), 0, K, n, K, N, 1, 0, 0, 0, st()), "dgrad"))
        r["dgrad_ms"], r["dgrad_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        for s in (1, 4):
            t = timeit(lambda: _lib.check(L.mf_linear_wgrad_bf16(p(y), 0, N, p(A), 0, K, p(dW), 0, K, p(ws), n, N, K, 1, s, st()), "wgrad"))
            r[f"wgrad_s{s}_ms"], r[f"wgrad_s{s}_tflops"] = round(t, 4), round(flop / t / 1e9, 1)
        ref = F.relu(A[:256].float() @ W.float().t() + bias)
        r["fwd_max_rel_err"] = float((y[:256].float() - ref).abs().max() / ref.abs().max())
        if stock:
            r["stock_fwd_ms"] = round(timeit(lambda: F.relu(F.linear(A, W, bias.to(torch.bfloat16)))), 4)
            r["stock_dgrad_ms"] = round(timeit(lambda: y @ W), 4)
            r["stock_wgrad_ms"] = round(timeit(lambda: y.t() @ A), 4)
        out[name] = r
        print(json.dumps({name: r}), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
