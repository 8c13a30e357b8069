"""The occupancy branch's 3 x 3 x 3 convolutions at the training batch (16 objects, 32^3): the narrow kernel
(k_conv_k3_narrow_bf16) against the implicit-GEMM engine (mf_conv3d_bf16_fwd), HIP events."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from morefusion_amd import _lib  # noqa: E402
from tools.time_gemm_bf16 import timeit  # noqa: E402


def main():
    L, st, p = _lib.lib(), _lib.stream_ptr, (lambda t: t.data_ptr())
    B, D, dev, bf = 16, 32, "cuda", torch.bfloat16
    torch.manual_seed(0)
    for name, Cin, Cout, dil, transpose in (("conv1_occ fwd", 8, 8, 1, 0), ("conv2_occ fwd", 8, 16, 2, 0), ("conv2_occ dgrad", 8, 16, 2, 1)):
        W = torch.randn(Cout, Cin, 3, 3, 3, device=dev) / (27 * Cin) ** 0.5
        bias = torch.randn(Cout, device=dev)
        ci, co = (Cout, Cin) if transpose else (Cin, Cout)   # channels read / written by this launch
        x = torch.randn(B, D ** 3, ci, device=dev).to(bf)
        wp = torch.empty(int(L.mf_conv3d_k3_narrow_bf16_pack_elems(ci)), dtype=bf, device=dev)
        _lib.check(L.mf_conv3d_k3_narrow_bf16_pack(p(W), Cout, Cin, Cin, 0, transpose, p(wp), st()), "pack")
        y = torch.empty(B, D ** 3, co, dtype=bf, device=dev)
        t_n = timeit(lambda: _lib.check(L.mf_conv3d_k3_narrow_bf16(p(x), p(wp), None if transpose else p(bias), p(y), B, ci, co, D, dil, 0 if transpose else 1, st()), "narrow"))
        # the engine: forward operand / flipT operand
        wt = torch.empty(Cout, 27, Cin, dtype=bf, device=dev)
        wf = torch.empty(Cin, 27, Cout, dtype=bf, device=dev)
        _lib.check(L.mf_conv3d_bf16_pack(p(W), Cout, Cin, Cin, 0, 3, p(wt), None, p(wf), st()), "pack2")
        y2 = torch.empty_like(y)
        if transpose:
            t_e = timeit(lambda: _lib.check(L.mf_conv3d_bf16_fwd(p(x), p(wf), None, p(y2), B, Cout, Cin, D, 3, 1, dil, dil, 0, 0, Cin, st()), "engine"))
        else:
            t_e = timeit(lambda: _lib.check(L.mf_conv3d_bf16_fwd(p(x), p(wt), p(bias), p(y2), B, Cin, Cout, D, 3, 1, dil, dil, 1, 0, Cout, st()), "engine"))
        err = float((y.float() - y2.float()).abs().max() / y2.float().abs().max())
        print(f"{name:16s} narrow {t_n * 1e3:7.1f} us   engine {t_e * 1e3:7.1f} us   max rel diff {err:.2e}  equal bits: {bool(torch.equal(y, y2))}")


if __name__ == "__main__":
    main()
