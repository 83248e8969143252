#!/usr/bin/env python
"""wgrad microbenchmark for the flag combinations the step actually uses (x_relu / dy_ups / x_ups)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from xmcgan_image_generation_amd.ops import HipOps

def timeit(fn, iters=10):
    fn(); torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters

ops = HipOps(dtype=torch.bfloat16)
g = torch.Generator().manual_seed(0)
for (n, h, cin, cout) in [(112, 128, 96, 96), (112, 64, 96, 192), (112, 64, 192, 192), (56, 128, 96, 96)]:
    x = torch.randn((n, h, h, cin), generator=g).to(torch.bfloat16).cuda()
    dw = torch.zeros((cout, 9, cin), device="cuda"); db = torch.zeros((cout,), device="cuda")
    gf = 2.0 * n * h * h * 9 * cin * cout / 1e9
    for flags in [dict(), dict(x_relu=True), dict(dy_ups=True, alpha=0.25), dict(x_relu=True, dy_ups=True, alpha=0.25)]:
        hd = h // 2 if flags.get("dy_ups") else h
        dy = torch.randn((n, hd, hd, cout), generator=g).to(torch.bfloat16).cuda()
        t = timeit(lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, **flags))
        t2 = timeit(lambda: ops.conv_wgrad(x, dy, dw, None, ks=3, **flags))
        print(f"n={n} h={h} {cin}>{cout} {str(flags):52s} {t:7.3f} ms {gf/t:7.1f} TF/s | no-bias {t2:7.3f} ms {gf/t2:7.1f} TF/s")
