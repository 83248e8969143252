"""The discriminator's data-gradient launches with their ReLU masks as bits (the in-step configuration), C1 shapes at 2B = 112:
"out" = c1.dgrad(dout, ups, alpha = 1/4, mask = h1 bits) on the phases-as-waves kernel, "3x3" = c0.dgrad(dh1, mask = x bits,
res = dxp, res_ups) on conv_stream_kernel; each with and without the mask.  usage: PYTHONPATH=. python tools/bench_mask_bits.py"""
import math
import torch
from xmcgan_image_generation_amd.ops import HipOps

ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
N = 112


def bits_of(t):
    m = (t > 0).view(*t.shape[:-1], t.shape[-1] // 16, 16).to(torch.int32)
    w = (m << torch.arange(16, device=t.device, dtype=torch.int32)).sum(-1)
    return w.to(torch.int16)


def timed(fn):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3


print(f"{'layer':28s} | out-form dgrad: no mask   bits | 3x3 dgrad (mask + res_ups): no mask   bits   (us)")
for name, lo, c in (("D 128>64  96", 64, 96), ("D 64>32  192", 32, 192), ("D 32>16  384", 16, 384), ("D 16>8   768", 8, 768), ("D 8>4   1536", 4, 1536)):
    hi = 2 * lo
    w = torch.randn((c, 9, c), generator=g) / math.sqrt(9 * c)
    wf, wd = ops.prep_conv_weight(w.cuda(), None, True, phase="pool")
    wf3, wd3 = ops.prep_conv_weight(w.cuda(), None, True)
    dout = torch.randn((N, lo, lo, c), generator=g).cuda().bfloat16()
    h1 = torch.randn((N, hi, hi, c), generator=g).cuda().bfloat16()
    h1.bits = bits_of(h1)
    plain = h1.clone()
    r = []
    r.append(timed(lambda: ops.conv(dout, wd, None, ks=3, ups=True, alpha=0.25)))
    r.append(timed(lambda: ops.conv(dout, wd, None, ks=3, ups=True, alpha=0.25, mask=h1)))
    dh1 = torch.randn((N, hi, hi, c), generator=g).cuda().bfloat16()
    dxp = torch.randn((N, lo, lo, c), generator=g).cuda().bfloat16()
    r.append(timed(lambda: ops.conv(dh1, wd3, None, ks=3, res=dxp, res_ups=True, res_scale=0.25)))
    r.append(timed(lambda: ops.conv(dh1, wd3, None, ks=3, mask=h1, res=dxp, res_ups=True, res_scale=0.25)))
    a = ops.conv(dout, wd, None, ks=3, ups=True, alpha=0.25, mask=h1)
    b = ops.conv(dout, wd, None, ks=3, ups=True, alpha=0.25, mask=plain)
    assert torch.equal(a, b), "bits mask != bf16 mask"
    print(f"{name:28s} | {r[0]:24.1f} {r[1]:6.1f} | {r[2]:36.1f} {r[3]:6.1f}")
