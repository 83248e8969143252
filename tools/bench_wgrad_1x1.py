"""Pointwise (1x1) weight gradients of the C1 step: cin blocks per workgroup (XMC_WGRAD_CB = 1 / 2 / 3 -> 1 / 2 / 4 blocks).
usage (GPU box): PYTHONPATH=. python tools/bench_wgrad_1x1.py"""
import os
import torch
from xmcgan_image_generation_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
SHAPES = [(56, 16, 1024, 4224), (56, 4, 1536, 1536), (56, 8, 1536, 768), (56, 16, 768, 384), (56, 32, 384, 192), (56, 64, 192, 96),
          (112, 64, 96, 96), (112, 32, 96, 192), (112, 16, 192, 384), (112, 8, 384, 768), (112, 4, 768, 1536), (56, 16, 1024, 768)]
base = ops.wgrad_variant
for (n, h, cin, cout) in SHAPES:
    x = torch.randn((n, h, h, cin), generator=g).cuda().bfloat16()
    dy = torch.randn((n, h, h, cout), generator=g).cuda().bfloat16()
    dw = torch.zeros((cout, 1, cin), device="cuda")
    db = torch.zeros((cout,), device="cuda")
    best = {}
    for r in range(4):
        for cb in (1, 2, 3):
            ops.wgrad_variant = base | (cb << 9)
            for _ in range(2):
                ops.conv_wgrad(x, dy, dw, db, ks=1, sync=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.conv_wgrad(x, dy, dw, db, ks=1, sync=True)
            e1.record()
            torch.cuda.synchronize()
            best[cb] = min(best.get(cb, 1e9), e0.elapsed_time(e1) / 10)
    fl = 2.0 * n * h * h * cin * cout
    print(f"{n}x{h}^2 {cin}>{cout}: " + "  ".join(f"cb{[0, 1, 2, 4][cb]} {best[cb] * 1e3:6.1f} us {fl / best[cb] / 1e9:5.0f} TF/s" for cb in (1, 2, 3)))
