"""conv3x3 next to a 2x resampling: the 3x3 kernel (ups gather / pooled epilogue) vs four 2x2 convolutions on the low-resolution
grid (conv_phase_kernel), C1 shapes (B = 56; D at 2B = 112), forward and data gradient, interleaved rounds.
TF/s are ALGORITHMIC (the 3x3 formulation's 2 M K N); the phase launches execute 4/9 of them.
usage (GPU box): PYTHONPATH=. python tools/bench_phase.py [--iters 5]"""
import argparse
import math
import torch
from xmcgan_image_generation_amd.ops import HipOps

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--batch", type=int, default=56)
ap.add_argument("--only-phase", action="store_true", help="time only the phase-decomposed forward / data-gradient launches (ablation builds)")
args = ap.parse_args()
ops = HipOps(torch.bfloat16)
g = torch.Generator().manual_seed(0)
B = args.batch
# (name, kind, n, low-res side, cin, cout): "ups": x at low res; "pool": x at 2 * low res
LAYERS = [("G 4>8   1536>1536", "ups", B, 4, 1536, 1536), ("G 8>16  1536>768", "ups", B, 8, 1536, 768),
          ("G 16>32 768>384", "ups", B, 16, 768, 384), ("G 32>64 384>192", "ups", B, 32, 384, 192),
          ("G 64>128 192>96", "ups", B, 64, 192, 96),
          ("D 128>64 96>96", "pool", 2 * B, 64, 96, 96), ("D 64>32 192>192", "pool", 2 * B, 32, 192, 192),
          ("D 32>16 384>384", "pool", 2 * B, 16, 384, 384), ("D 16>8  768>768", "pool", 2 * B, 8, 768, 768),
          ("D 8>4   1536>1536", "pool", 2 * B, 4, 1536, 1536)]


def timed(fn):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10


print(f"{'layer':22s} {'GF':>6s} | fwd 3x3 ms TF/s | fwd phase ms TF/s    x | dgrad 3x3 ms TF/s | dgrad phase ms TF/s    x | wgrad 3x3 ms TF/s | wgrad phase ms TF/s    x")
tot = [0.0] * 6
for name, kind, n, lo, cin, cout in LAYERS:
    hi = 2 * lo
    w = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    ops.phase_conv = True
    wfp, wdp = ops.prep_conv_weight(w.cuda(), None, True, phase=kind)
    wf3, wd3 = ops.prep_conv_weight(w.cuda(), None, True)
    wsel = lambda: (wfp, wdp) if ops.phase_conv else (wf3, wd3)
    bias = torch.zeros(cout, device="cuda")
    if kind == "ups":
        x = torch.randn((n, lo, lo, cin), generator=g).cuda().bfloat16()
        dy = torch.randn((n, hi, hi, cout), generator=g).cuda().bfloat16()
        fwd = lambda: ops.conv(x, wsel()[0], bias, ks=3, ups=True)
        bwd = lambda: ops.conv(dy, wsel()[1], None, ks=3, pool_out=True, alpha=4.0) if ops.can_pool_out(dy, wsel()[1]) else ops.pool2(ops.conv(dy, wsel()[1], None, ks=3), 1.0)
    else:
        x = torch.randn((n, hi, hi, cin), generator=g).cuda().bfloat16()
        dy = torch.randn((n, lo, lo, cout), generator=g).cuda().bfloat16()
        res = torch.randn((n, lo, lo, cout), generator=g).cuda().bfloat16()
        fwd = lambda: ops.conv(x, wsel()[0], bias, ks=3, pool_out=True, relu_in=True, res=res) if ops.can_pool_out(x, wsel()[0]) else ops.pool2(ops.conv(x, wsel()[0], bias, ks=3, relu_in=True), 0.25, res=res)
        bwd = lambda: ops.conv(dy, wsel()[1], None, ks=3, ups=True, alpha=0.25, mask=x)
    fl = 2.0 * n * hi * hi * cin * cout * 9
    dw = torch.zeros((cout, 9, cin), device="cuda")
    db = torch.zeros((cout,), device="cuda")
    if kind == "ups":
        wg = lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, x_ups=True, sync=True)
    else:
        wg = lambda: ops.conv_wgrad(x, dy, dw, db, ks=3, x_relu=True, dy_ups=True, alpha=0.25, sync=True)
    best = [1e9] * 6
    for r in range(args.iters):
        for k, (ph, fn) in enumerate(((False, fwd), (True, fwd), (False, bwd), (True, bwd), (False, wg), (True, wg))):
            if args.only_phase and k not in (1, 3):
                best[k] = 1.0
                continue
            ops.phase_conv = ph
            best[k] = min(best[k], timed(fn))
    for k in range(6):
        tot[k] += best[k]
    print(f"{name:22s} {fl / 1e9:6.1f} | {best[0]:7.3f} {fl / best[0] / 1e9:6.0f} | {best[1]:9.3f} {fl / best[1] / 1e9:6.0f} {best[0] / best[1]:5.2f} |"
          f" {best[2]:9.3f} {fl / best[2] / 1e9:6.0f} | {best[3]:11.3f} {fl / best[3] / 1e9:6.0f} {best[2] / best[3]:5.2f} |"
          f" {best[4]:9.3f} {fl / best[4] / 1e9:6.0f} | {best[5]:11.3f} {fl / best[5] / 1e9:6.0f} {best[4] / best[5]:5.2f}")
print(f"{'TOTAL':22s}        | {tot[0]:7.3f}        | {tot[1]:9.3f}        {tot[0] / tot[1]:5.2f} | {tot[2]:9.3f}        | {tot[3]:11.3f}        {tot[2] / tot[3]:5.2f} |"
      f" {tot[4]:9.3f}        | {tot[5]:11.3f}        {tot[4] / tot[5]:5.2f}")
