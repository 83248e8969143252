#!/usr/bin/env python
"""Conditional-BatchNorm apply / backward passes alone, per layer geometry of the C1 generator (batch 56, bf16): time and
tensor bytes per second.  XMC_CBN_RUN=0 selects the pixel-per-thread / thread-per-cell kernels (A/B in two processes).
usage (GPU box): PYTHONPATH=. python tools/bench_cbn.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timed(fn, n=30):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    b = 56
    geos = [(4, 1536, 1), (8, 1536, 1), (8, 1536, 1), (16, 768, 1), (16, 768, 16), (32, 384, 16), (32, 384, 16), (64, 192, 16),
            (64, 192, 16), (128, 96, 16), (128, 96, 1)]
    print(f"XMC_CBN_RUN={os.environ.get('XMC_CBN_RUN', '1')}")
    print(f"{'h':>4s} {'c':>5s} {'hc':>3s} {'MB':>7s} | {'fwd us':>8s} {'TB/s':>5s} | {'bwd us':>8s} {'TB/s':>5s}")
    for h, c, hc in geos:
        x = torch.randn((b, h, h, c), device="cuda").to(torch.bfloat16)
        dy = torch.randn((b, h, h, c), device="cuda").to(torch.bfloat16)
        gb = (torch.randn((b, hc, hc, 2 * c), device="cuda") * 0.3).contiguous()
        mean = torch.zeros(c, device="cuda")
        rstd = torch.ones(c, device="cuda")
        mb = x.numel() * 2 / 1e6
        tf = timed(lambda: ops.cbn_act_fwd(x, mean, rstd, gb, hc))
        tb = timed(lambda: ops.cbn_act_bwd(dy, x, mean, rstd, gb, hc))
        print(f"{h:4d} {c:5d} {hc:3d} {mb:7.1f} | {tf:8.1f} {2 * mb / tf:5.2f} | {tb:8.1f} {5 * mb / tb:5.2f}")


if __name__ == "__main__":
    main()
