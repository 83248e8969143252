#!/usr/bin/env python
"""Per-layer microbenchmark of the implicit-GEMM convolution kernels on the C1 layer shapes.
usage (on the GPU box): python tools/bench_conv.py [--dtype bf16|f32] [--iters 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd.ops import HipOps  # noqa: E402

# (tag, n, h_in, cin, cout, ks, ups)
LAYERS = [
    ("G 4>8   1536>1536 up", 56, 4, 1536, 1536, 3, True),
    ("G 8     1536>1536", 56, 8, 1536, 1536, 3, False),
    ("G 8>16  1536>768 up", 56, 8, 1536, 768, 3, True),
    ("G 16    768>768", 56, 16, 768, 768, 3, False),
    ("G 16>32 768>384 up", 56, 16, 768, 384, 3, True),
    ("G 32    384>384", 56, 32, 384, 384, 3, False),
    ("G 32>64 384>192 up", 56, 32, 384, 192, 3, True),
    ("G 64    192>192", 56, 64, 192, 192, 3, False),
    ("G 64>128 192>96 up", 56, 64, 192, 96, 3, True),
    ("G 128   96>96", 56, 128, 96, 96, 3, False),
    ("G 128   96>3", 56, 128, 96, 3, 3, False),
    ("G 16    1024>768 1x1", 56, 16, 1024, 768, 1, False),
    ("D 128   3>96", 112, 128, 3, 96, 3, False),
    ("D 128   96>96", 112, 128, 96, 96, 3, False),
    ("D 64    96>192", 112, 64, 96, 192, 3, False),
    ("D 64    192>192", 112, 64, 192, 192, 3, False),
    ("D 32    192>384", 112, 32, 192, 384, 3, False),
    ("D 32    384>384", 112, 32, 384, 384, 3, False),
    ("D 16    384>768", 112, 16, 384, 768, 3, False),
    ("D 16    768>768", 112, 16, 768, 768, 3, False),
    ("D 8     768>1536", 112, 8, 768, 1536, 3, False),
    ("D 8     1536>1536", 112, 8, 1536, 1536, 3, False),
    ("D 4     1536>1536", 112, 4, 1536, 1536, 3, False),
]


def timeit(fn, iters):
    fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--only", default=None)
    ap.add_argument("--packed", action="store_true", help="weight-streaming kernel on fragment-packed weights")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ops = HipOps(dtype=dt, stream_conv=args.packed)
    g = torch.Generator().manual_seed(0)
    print(f"{'layer':26s} {'GF':>7s} | {'fwd ms':>8s} {'TF/s':>7s} | {'dgrad ms':>8s} {'TF/s':>7s} | {'wgrad ms':>8s} {'TF/s':>7s}")
    tot = [0.0, 0.0, 0.0, 0.0]
    for tag, n, h, cin, cout, ks, ups in LAYERS:
        if args.only and args.only not in tag:
            continue
        ho = 2 * h if ups else h
        x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
        w = (torch.randn((cout, ks * ks, cin), generator=g) / (ks * ks * cin) ** 0.5).cuda()
        wf, wd = ops.prep_conv_weight(w)
        dy = torch.randn((n, ho, ho, cout), generator=g).to(dt).cuda()
        dw = torch.zeros_like(w)
        gf = 2.0 * n * ho * ho * ks * ks * cin * cout / 1e9
        t_f = timeit(lambda: ops.conv(x, wf, None, ks=ks, ups=ups), args.iters)
        t_d = timeit(lambda: ops.conv(dy, wd, None, ks=ks), args.iters)
        t_w = timeit(lambda: ops.conv_wgrad(x, dy, dw, ks=ks, x_ups=ups), args.iters)
        print(f"{tag:26s} {gf:7.1f} | {t_f:8.3f} {gf / t_f:7.1f} | {t_d:8.3f} {gf / t_d:7.1f} | {t_w:8.3f} {gf / t_w:7.1f}")
        tot[0] += gf
        tot[1] += t_f
        tot[2] += t_d
        tot[3] += t_w
    print(f"{'TOTAL':26s} {tot[0]:7.1f} | {tot[1]:8.3f} {tot[0] / tot[1]:7.1f} | {tot[2]:8.3f} {tot[0] / tot[2]:7.1f} | {tot[3]:8.3f} {tot[0] / tot[3]:7.1f}")


if __name__ == "__main__":
    main()
