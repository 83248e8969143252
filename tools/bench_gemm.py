#!/usr/bin/env python
"""Microbenchmark of the bf16-MFMA strided GEMM on the word_loss shapes (GPU box).
usage: python tools/bench_gemm.py [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd.ops import HipOps  # noqa: E402

SHAPES = [  # (name, a shape, b shape, ta, tb)
    ("S    (14336x768)x(952x768)^T", (14336, 768), (952, 768), False, True),
    ("drn  (14336x952)x(952x768)", (14336, 952), (952, 768), False, False),
    ("G    56x(256x768)x(256x768)^T", (56, 256, 768), (56, 256, 768), False, True),
    ("h    56x(256x256)x(256x952)", (56, 256, 256), (56, 256, 952), False, False),
    ("dg   56x(256x952)x(256x952)^T", (56, 256, 952), (56, 256, 952), False, True),
]


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(0)
    for name, sa, sb, ta, tb in SHAPES:
        a = torch.randn(sa, generator=g).cuda()
        b = torch.randn(sb, generator=g).cuda()
        out = ops.gemm(a, b, ta=ta, tb=tb, fast=True)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(iters):
            ops.gemm(a, b, ta=ta, tb=tb, fast=True, out=out)
        e.record()
        torch.cuda.synchronize()
        ms = s.elapsed_time(e) / iters
        m, k = (sa[-1], sa[-2]) if ta else (sa[-2], sa[-1])
        n = sb[-2] if tb else sb[-1]
        batch = sa[0] if len(sa) == 3 else 1
        fl = 2.0 * m * n * k * batch
        print(f"{name:34s} {ms * 1e3:8.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s")


if __name__ == "__main__":
    main()
