"""Is a weight-gradient launch bit-stable beside a busy neighbour stream?  (the fp8 mode's race.  The MX-fp8 mode keeps
the discriminator's h1 as a PRE-activation, so its weight gradients run with x_relu = 1 -- the ReLU applied to the staged patch in
LDS -- a path the bf16 mode never takes since round 3 stored the ReLU-ed tensor.)
    PYTHONPATH=. python tools/wgrad_concurrency.py [--reps 20]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    side = torch.cuda.Stream()
    nx = torch.randn((32, 64, 64, 192), generator=g).to(dt).cuda()
    nw = (torch.randn((192, 9, 192), generator=g) / 42).cuda()
    nwf, _ = ops.prep_conv_weight(nw)

    def noise(k=8):
        with torch.cuda.stream(side):
            for _ in range(k):
                ops.conv(nx, nwf, None, ks=3)
    cases = [  # tag, n, h, cin, cout, ks, x_ups, x_relu, dy_ups
        ("3x3 64^2 192>192", 32, 64, 192, 192, 3, False, False, False),
        ("3x3 64^2 192>192 x_relu", 32, 64, 192, 192, 3, False, True, False),
        ("3x3 128^2 96>96 x_relu", 32, 128, 96, 96, 3, False, True, False),
        ("3x3 16^2 768>768 x_relu", 32, 16, 768, 768, 3, False, True, False),
        ("3x3 8^2 1536>1536 x_relu", 32, 8, 1536, 1536, 3, False, True, False),
        ("pooled 64>32 192>192 x_relu dy_ups", 32, 64, 192, 192, 3, False, True, True),
        ("pooled 64>32 192>192 dy_ups", 32, 64, 192, 192, 3, False, False, True),
        ("pooled 16>8 768>768 x_relu dy_ups", 32, 16, 768, 768, 3, False, True, True),
        ("ups 32>64 384>192 x_ups", 32, 32, 384, 192, 3, True, False, False),
        ("1x1 64^2 96>192 x_relu", 32, 64, 96, 192, 1, False, True, False),
    ]
    bad = 0
    for tag, n, h, cin, cout, ks, x_ups, x_relu, dy_ups in cases:
        x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
        ho = 2 * h if x_ups else (h // 2 if dy_ups else h)
        dy = torch.randn((n, ho, ho, cout), generator=g).to(dt).cuda()

        def run():
            dw = torch.zeros((cout, ks * ks, cin), device="cuda")
            db = torch.zeros((cout,), device="cuda")
            ops.conv_wgrad(x, dy, dw, db, ks=ks, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, sync=True)
            return dw, db
        torch.cuda.synchronize()
        rw, rb = run()
        torch.cuda.synchronize()
        diff = 0
        worst = 0.0
        for _ in range(a.reps):
            noise()
            dw, db = run()
            torch.cuda.synchronize()
            if not (torch.equal(dw, rw) and torch.equal(db, rb)):
                diff += 1
                worst = max(worst, float((dw - rw).abs().max() / rw.abs().max()))
        print(f"{tag:40s} differing runs of {a.reps}: {diff}" + (f"   (max |diff| / max |dw| = {worst:.2e})" if diff else ""), flush=True)
        bad += bool(diff)
    print("RACE FOUND" if bad else "every weight-gradient launch is bit-stable beside a busy neighbour")


if __name__ == "__main__":
    main()
