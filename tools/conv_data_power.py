"""Does operand DATA change the convolution kernels' speed?  (The MFMA rate probe drops from 2.49 to 1.70 PFLOP/s bf16
when its operands go from constants to random bits: the chip is power-limited under matrix load.)  Times the same
3x3 launch on zeros / small-integer / N(0,1) activations, interleaved.
usage (GPU box): PYTHONPATH=. python tools/conv_data_power.py"""
import math
import torch
from xmcgan_image_generation_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
for (n, h, cin, cout) in ((56, 32, 384, 384), (56, 64, 192, 192), (56, 16, 768, 768)):
    w = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    pw, _ = ops.prep_conv_weight(w.cuda())
    wz, _ = ops.prep_conv_weight(torch.zeros_like(w).cuda())
    xs = {
        "zeros x, zeros w": (torch.zeros((n, h, h, cin), device="cuda", dtype=torch.bfloat16), wz),
        "zeros x, randn w": (torch.zeros((n, h, h, cin), device="cuda", dtype=torch.bfloat16), pw),
        "ones x, randn w": (torch.ones((n, h, h, cin), device="cuda", dtype=torch.bfloat16), pw),
        "relu(randn) x, randn w": (torch.randn((n, h, h, cin), generator=g).cuda().relu().bfloat16(), pw),
        "randn x, randn w": (torch.randn((n, h, h, cin), generator=g).cuda().bfloat16(), pw),
    }
    dy = torch.randn((n, h, h, cout), generator=g).cuda().bfloat16()
    dyz = torch.zeros_like(dy)
    dw = torch.zeros((cout, 9, cin), device="cuda")
    fl = 2 * n * h * h * cin * cout * 9
    best = {k: 1e9 for k in xs}
    bw = {"wgrad randn": 1e9, "wgrad zeros": 1e9}
    for r in range(6):
        for k, (x, ww) in xs.items():
            for _ in range(3):
                ops.conv(x, ww, None, ks=3)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv(x, ww, None, ks=3)
            e1.record()
            torch.cuda.synchronize()
            best[k] = min(best[k], e0.elapsed_time(e1) / 20)
        for k, (x, d) in (("wgrad randn", (xs["randn x, randn w"][0], dy)), ("wgrad zeros", (xs["zeros x, zeros w"][0], dyz))):
            for _ in range(3):
                ops.conv_wgrad(x, d, dw, ks=3, sync=True)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                ops.conv_wgrad(x, d, dw, ks=3, sync=True)
            e1.record()
            torch.cuda.synchronize()
            bw[k] = min(bw[k], e0.elapsed_time(e1) / 20)
    print(f"--- {n}x{h}x{h} {cin}>{cout}")
    for k, v in list(best.items()) + list(bw.items()):
        print(f"{k:26s} {v:7.3f} ms {fl / v / 1e9:6.0f} TF/s")
