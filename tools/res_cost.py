"""What does the residual read in a convolution epilogue cost?  (cf. tools/mask_cost.py)
usage (GPU box): PYTHONPATH=. python tools/res_cost.py"""
import math
import torch
from xmcgan_image_generation_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 20


def rnd(*shape):
    return torch.randn(shape, generator=g).cuda().bfloat16()


# (name, n, h, cin, cout, ks, res_ups)
for name, n, h, cin, cout, ks, res_ups in (("G c1 128^2 96>96 res_ups", 56, 128, 96, 96, 3, True), ("G c1 64^2 192>192 res_ups", 56, 64, 192, 192, 3, True),
                                          ("G c1 32^2 384>384 res_ups", 56, 32, 384, 384, 3, True), ("D c0.dgrad 64^2 192>96 res_ups", 112, 64, 192, 96, 3, True),
                                          ("D c0.dgrad 32^2 384>192 res_ups", 112, 32, 384, 192, 3, True), ("ResNet c3 64^2 64>256 res", 112, 64, 64, 256, 1, False),
                                          ("ResNet c3 32^2 128>512 res", 112, 32, 128, 512, 1, False)):
    w = torch.randn((cout, ks * ks, cin), generator=g) / math.sqrt(ks * ks * cin)
    wf, _ = ops.prep_conv_weight(w.cuda(), None, True)
    x = rnd(n, h, h, cin)
    res = rnd(n, h // 2, h // 2, cout) if res_ups else rnd(n, h, h, cout)
    a = timed(lambda: ops.conv(x, wf, None, ks=ks, res=res, res_ups=res_ups))
    b = timed(lambda: ops.conv(x, wf, None, ks=ks))
    print(f"{name:34s} with res {a * 1e3:6.0f} us   without {b * 1e3:6.0f} us")
