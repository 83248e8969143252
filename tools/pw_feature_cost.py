"""Cost of the optional epilogue features of the pointwise kernel on ResNet-50-like launches (cf. tools/mask_cost.py).
usage (GPU box): PYTHONPATH=. python tools/pw_feature_cost.py"""
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


for (n, h, cin, cout, hv) in ((112, 64, 64, 256, 56), (112, 64, 256, 64, 56), (112, 32, 512, 128, 28), (112, 32, 128, 512, 28), (112, 16, 1024, 256, 14)):
    w = torch.randn((cout, 1, cin), generator=g) / math.sqrt(cin)
    wf, _ = ops.prep_conv_weight(w.cuda(), None, True)
    x, res, bias = rnd(n, h, h, cin), rnd(n, h, h, cout), torch.randn(cout, generator=g).cuda()
    base = timed(lambda: ops.conv(x, wf, None, ks=1))
    row = [f"{n}x{h}^2 {cin}>{cout}: plain {base:5.0f}"]
    for name, kw in (("bias", dict()), ("relu_out", dict(relu_out=True)), ("valid", dict(valid=hv)), ("bits", dict(emit_bits=True)),
                     ("res", dict(res=res)), ("all", dict(res=res, relu_out=True, valid=hv, emit_bits=True))):
        b = bias if name in ("bias", "all") else None
        row.append(f"{name} {timed(lambda: ops.conv(x, wf, b, ks=1, **kw)):5.0f}")
    print("  ".join(row))
