"""Does aten::miopen_convolution_relu / _add_relu (MIOpen fusion) exist here, is it exact and is it faster than
conv + bias (+ residual) + ReLU as separate launches?  Shapes: the ResNet18 body at B = 8 (256^2 crops)."""
import json
import time

import torch
import torch.nn.functional as F

torch.backends.cudnn.benchmark = True
dev = "cuda"
out = {}
for name, (B, Cin, Cout, H, dil, stride) in {
        "res2": (8, 64, 64, 64, 1, 1), "res3": (8, 128, 128, 32, 1, 1), "res4": (8, 256, 256, 32, 2, 1),
        "res5": (8, 512, 512, 32, 4, 1), "res2_b1": (1, 64, 64, 64, 1, 1), "res5_b1": (1, 512, 512, 32, 4, 1)}.items():
    for cl in (False, True):
        x = torch.randn(B, Cin, H, H, device=dev)
        w = torch.randn(Cout, Cin, 3, 3, device=dev) * 0.05
        b = torch.randn(Cout, device=dev)
        z = torch.randn(B, Cout, H // stride, H // stride, device=dev)
        if cl:
            x, w, z = (t.contiguous(memory_format=torch.channels_last) for t in (x, w, z))
        pad = [dil, dil]

        def plain():
            return F.relu(F.conv2d(x, w, b, stride, dil, dil))

        def fused():
            return torch.ops.aten.miopen_convolution_relu(x, w, b, [stride, stride], pad, [dil, dil], 1)

        def plain_add():
            return F.relu(F.conv2d(x, w, b, stride, dil, dil) + z)

        def fused_add():
            return torch.ops.aten.miopen_convolution_add_relu(x, w, z, 1.0, b, [stride, stride], pad, [dil, dil], 1)

        rec = {}
        for tag, f, ref in (("relu", fused, plain), ("add_relu", fused_add, plain_add)):
            try:
                a, r = f(), ref()
                rec[tag + "_max_abs_diff"] = float((a - r).abs().max())
                for nm, fn in ((tag + "_fused_us", f), (tag + "_plain_us", ref)):
                    for _ in range(5):
                        fn()
                    torch.cuda.synchronize()
                    t0 = time.perf_counter()
                    for _ in range(50):
                        fn()
                    torch.cuda.synchronize()
                    rec[nm] = round((time.perf_counter() - t0) / 50 * 1e6, 1)
            except Exception as e:  # noqa: BLE001
                rec[tag + "_error"] = repr(e)[:200]
        out[f"{name}{'_cl' if cl else ''}"] = rec
        print(name, cl, json.dumps(rec), flush=True)
print(json.dumps(out))
