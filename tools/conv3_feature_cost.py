"""Cost of the optional features of the 3x3 weight-streaming launches at the C1 shapes (cf. tools/pw_feature_cost.py).
usage (GPU box): PYTHONPATH=. python tools/conv3_feature_cost.py"""
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
    return e0.elapsed_time(e1) / 20 * 1e3


def rnd(*shape):
    return torch.randn(shape, generator=g).cuda().bfloat16()


for (n, h, cin, cout) in ((56, 128, 96, 96), (112, 64, 96, 192), (56, 64, 192, 192), (112, 32, 192, 384), (56, 32, 384, 384), (56, 16, 768, 768)):
    w = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    wf, _ = ops.prep_conv_weight(w.cuda(), None, True)
    x, res, bias = rnd(n, h, h, cin), rnd(n, h, h, cout), torch.randn(cout, generator=g).cuda()
    base = timed(lambda: ops.conv(x, wf, None, ks=3))
    row = [f"{n}x{h}^2 {cin}>{cout}: plain {base:5.0f}"]
    for name, kw in (("bias", dict()), ("relu_in", dict(relu_in=True)), ("relu_out", dict(relu_out=True)), ("bits", dict(emit_bits=True)),
                     ("res", dict(res=res)), ("D c0", dict(relu_in=True, relu_out=True, emit_bits=True))):
        b = bias if name in ("bias", "D c0") else None
        row.append(f"{name} {timed(lambda: ops.conv(x, wf, b, ks=3, **kw)):5.0f}")
    print("  ".join(row))
