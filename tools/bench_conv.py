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
    ap.add_argument("--tile-ab", action="store_true", help="3x3 fwd + dgrad with the 128-cout tile forced vs the launcher's choice (96-cout tiles for Cout = 96 / 192)")
    ap.add_argument("--all96", action="store_true", help="with --tile-ab: 96-cout tiles for EVERY Cout % 96 == 0 layer in the second column")
    ap.add_argument("--fp8", action="store_true", help="MX-fp8 3x3 convolution (fwd + dgrad) next to the bf16 kernel")
    ap.add_argument("--wgrad-tunes", default=None,
                    help="comma list of LDS-DMA wgrad tuning values (xmc_wgrad_desc.variant >> 4): wgrad only, one column each")
    ap.add_argument("--wgrad-raw", action="store_true", help="--wgrad-tunes values are RAW xmc_wgrad_desc.variant values (e.g. 1,2049: "
                    "the launcher's choice vs bit 11 = no 96-cout tiles); x_relu off")
    args = ap.parse_args()
    dt = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    ops = HipOps(dtype=dt, stream_conv=args.packed)
    g = torch.Generator().manual_seed(0)
    if args.tile_ab:
        print(f"{'layer':26s} {'GF':>7s} | fwd 128-tile ms TF/s | fwd 96-tile ms TF/s || dgrad 128-tile ms TF/s | dgrad 96-tile ms TF/s")
        tot = [0.0] * 5
        for tag, n, h, cin, cout, ks, ups in LAYERS:
            if (args.only and args.only not in tag) or ks != 3 or cin < 32 or cout < 32:
                continue
            ho = 2 * h if ups else h
            x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
            w = (torch.randn((cout, 9, cin), generator=g) / (9 * cin) ** 0.5).cuda()
            wf, wd = ops.prep_conv_weight(w)
            dy = torch.randn((n, ho, ho, cout), generator=g).to(dt).cuda()
            gf = 2.0 * n * ho * ho * 9 * cin * cout / 1e9
            acc = [0.0] * 4
            for r in range(4):
                for i, (inp, wt, u, force) in enumerate(((x, wf, ups, True), (x, wf, ups, False), (dy, wd, False, True), (dy, wd, False, False))):
                    ops.force_tile128 = force
                    ops.force_tile96 = (not force) and args.all96
                    t = timeit(lambda: ops.conv(inp, wt, None, ks=3, ups=u), args.iters)
                    if r > 0:
                        acc[i] += t / 3
            print(f"{tag:26s} {gf:7.1f} | {acc[0]:10.3f} {gf / acc[0]:6.0f} | {acc[1]:10.3f} {gf / acc[1]:6.0f} || {acc[2]:10.3f} {gf / acc[2]:6.0f} | {acc[3]:10.3f} {gf / acc[3]:6.0f}")
            for i, v in enumerate([gf] + acc):
                tot[i] += v
        print(f"{'TOTAL':26s} {tot[0]:7.1f} | {tot[1]:10.3f} {tot[0] / tot[1]:6.0f} | {tot[2]:10.3f} {tot[0] / tot[2]:6.0f} || {tot[3]:10.3f} {tot[0] / tot[3]:6.0f} | {tot[4]:10.3f} {tot[0] / tot[4]:6.0f}")
        return
    if args.fp8:
        import ctypes as C
        from xmcgan_image_generation_amd._lib import ConvDesc, XMC_BF16
        print(f"{'layer':26s} {'GF':>7s} | {'bf16 fwd':>9s} {'TF/s':>6s} | {'quant ms':>8s} | {'mx8 conv':>8s} {'TF/s':>6s} | {'mx8 total':>9s} {'x bf16':>6s} || "
              f"{'bf16 dgr':>9s} | {'quant':>6s} | {'mx8 conv':>8s} {'TF/s':>6s} | {'x bf16':>6s}")
        tot = [0.0] * 8
        for tag, n, h, cin, cout, ks, ups in LAYERS:
            if (args.only and args.only not in tag) or ks != 3 or cin < 32 or cout < 32 or (2 * h if ups else h) < 8:
                continue
            ho = 2 * h if ups else h
            x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
            w = (torch.randn((cout, 9, cin), generator=g) / (9 * cin) ** 0.5).cuda()
            wf, wd = ops.prep_conv_weight(w)
            dy = torch.randn((n, ho, ho, cout), generator=g).to(dt).cuda()
            gf = 2.0 * n * ho * ho * 9 * cin * cout / 1e9
            res = []
            for (inp, wt, u) in ((x, wf, ups), (dy, wd, False)):
                ops.fp8 = False
                fns = [lambda: ops.conv(inp, wt, None, ks=3, ups=u)]
                w8, wsc = ops.pack_mx8(wt)
                x8 = ops.quantize_mx8(inp)
                nn, hh, _, cc = inp.shape
                y = torch.empty((nn, 2 * hh if u else hh, 2 * hh if u else hh, wt.cout), dtype=dt, device="cuda")
                d = ConvDesc(nn, hh, hh, cc, wt.cout, 3, int(u), 0, 0, 0, XMC_BF16, 1.0, 1.0, 1, 0, 0, 0, 0, 0)
                wsb = ops.lib.xmc_conv2d_mx8_workspace_bytes(C.byref(d))
                ws = torch.empty((max(wsb, 4) // 4,), dtype=torch.float32, device="cuda")
                p_ = lambda t: C.c_void_p(t.data_ptr())
                fns.append(lambda: ops.quantize_mx8(inp))
                fns.append(lambda: ops.lib.xmc_conv2d_mx8(C.byref(d), p_(x8), p_(w8), p_(wsc), None, None, None, p_(y),
                                                          None, 0, p_(ws) if wsb else None, ops._stream()))
                acc = [0.0] * 3
                for r in range(4):                     # interleaved rounds (DVFS: see the wgrad sweep)
                    for i, f in enumerate(fns):
                        t = timeit(f, args.iters)
                        if r > 0:
                            acc[i] += t / 3
                res.append(acc)
            (b16, q, c8), (b16d, qd, c8d) = res
            print(f"{tag:26s} {gf:7.1f} | {b16:9.3f} {gf / b16:6.0f} | {q:8.3f} | {c8:8.3f} {gf / c8:6.0f} | {q + c8:9.3f} {b16 / (q + c8):6.2f} || "
                  f"{b16d:9.3f} | {qd:6.3f} | {c8d:8.3f} {gf / c8d:6.0f} | {b16d / (qd + c8d):6.2f}")
            for i, v in enumerate((gf, b16, q, c8, b16d, qd, c8d)):
                tot[i] += v
        gf, b16, q, c8, b16d, qd, c8d = tot[:7]
        print(f"{'TOTAL':26s} {gf:7.1f} | {b16:9.3f} {gf / b16:6.0f} | {q:8.3f} | {c8:8.3f} {gf / c8:6.0f} | {q + c8:9.3f} {b16 / (q + c8):6.2f} || "
              f"{b16d:9.3f} | {qd:6.3f} | {c8d:8.3f} {gf / c8d:6.0f} | {b16d / (qd + c8d):6.2f}")
        return
    if args.wgrad_tunes:
        tunes = [int(t) for t in args.wgrad_tunes.split(",")]
        print(f"{'layer':26s} {'GF':>7s} | " + " | ".join(f"tune {t:2d} ms   TF/s" for t in tunes))
        tot = [0.0] * (len(tunes) + 1)
        for tag, n, h, cin, cout, ks, ups in LAYERS:
            if (args.only and args.only not in tag) or cin < 32 or cout < 32:
                continue
            ho = 2 * h if ups else h
            x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
            dy = torch.randn((n, ho, ho, cout), generator=g).to(dt).cuda()
            dw = torch.zeros((cout, ks * ks, cin), device="cuda")
            db = torch.zeros((cout,), device="cuda")
            gf = 2.0 * n * ho * ho * ks * ks * cin * cout / 1e9
            row = []
            # interleaved rounds: the chip clocks down under sustained load (DVFS), so a variant timed first after the
            # idle gap of the tensor set-up would look faster than the same variant timed later
            acc = [0.0] * len(tunes)
            rounds = 4
            for r in range(rounds + 1):
                for i, t in enumerate(tunes):
                    ops.wgrad_variant = t if args.wgrad_raw else 1 | (t << 4)
                    tw = timeit(lambda: ops.conv_wgrad(x, dy, dw, db, ks=ks, x_ups=ups, x_relu=not args.wgrad_raw), args.iters)
                    if r > 0:                     # round 0 = warm-up
                        acc[i] += tw / rounds
            for i, tw in enumerate(acc):
                row.append(f"{tw:10.3f} {gf / tw:6.1f}")
                tot[i + 1] += tw
            tot[0] += gf
            print(f"{tag:26s} {gf:7.1f} | " + " | ".join(row))
        print(f"{'TOTAL':26s} {tot[0]:7.1f} | " + " | ".join(f"{t:10.3f} {tot[0] / t:6.1f}" for t in tot[1:]))
        return
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
