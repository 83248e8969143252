"""Pointwise (1x1) convolution on the FEW-PIXEL, many-channel shapes of the frozen ResNet-50's stages 3-4 (14^2 / 7^2
maps: M = 22k / 5.5k pixels forward, half of that backward): time per launch for the kernel variants (w_packed bits
12-15: 1 <32,3>, 2 <64,3>, 3 <32,4>; + 4 forces 256-pixel tiles, + 8 forces 128-pixel tiles) with and without split-K.
    PYTHONPATH=. python tools/pw_small_m.py [--variants 5,9]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd.ops import HipOps  # noqa: E402

SHAPES = [  # (images, side, cin, cout, res)   side 16 / 8 stand in for the compact 14 / 7 maps (86 images ~ 112 * 49/64)
    (86, 32, 512, 128, False), (86, 32, 128, 512, True),
    (86, 16, 1024, 256, False), (86, 16, 256, 1024, True), (43, 16, 1024, 256, False), (43, 16, 256, 1024, True),
    (86, 16, 512, 1024, False), (86, 16, 1024, 512, False),
    (86, 8, 2048, 512, False), (86, 8, 512, 2048, True), (43, 8, 2048, 512, False), (43, 8, 512, 2048, True),
    (86, 8, 1024, 2048, False),
]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="5,9")
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    variants = [int(v) for v in a.variants.split(",")]
    ops = HipOps(dtype=torch.bfloat16)
    print(f"{'shape':34s} " + " ".join(f"{'v' + str(v) + (' nosplit' if ns else ''):>12s}" for v in variants for ns in (0, 1)) + "   (us; TF/s of the best)")
    for n, h, cin, cout, has_res in SHAPES:
        w = torch.randn((cout, 1, cin), device="cuda") / cin ** 0.5
        wf, _ = ops.prep_conv_weight(w, None, False)
        x = torch.randn((n, h, h, cin), device="cuda").bfloat16()
        res = torch.randn((n, h, h, cout), device="cuda").bfloat16() if has_res else None
        b = torch.randn((cout,), device="cuda")
        cells = []
        for v in variants:
            for ns in (0, 1):
                ops.pw_variant, ops.no_split_k = v, bool(ns)
                f = lambda: ops.conv(x, wf, b, ks=1, res=res, relu_out=True)
                for _ in range(3):
                    f()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(a.iters):
                    f()
                e.record()
                torch.cuda.synchronize()
                cells.append(s.elapsed_time(e) / a.iters * 1e3)
        m = n * h * h
        best = min(cells)
        byts = 2 * m * (cin + cout * (1 + has_res)) + 2 * cin * cout
        print(f"{f'M={m} {cin}->{cout}' + (' +res' if has_res else ''):34s} " + " ".join(f"{c:12.1f}" for c in cells) +
              f"   {2.0 * m * cin * cout / best / 1e6:6.0f} TF/s, {byts / best / 1e6:5.2f} TB/s")


if __name__ == "__main__":
    main()
