"""One pointwise (1x1) convolution shape, repeated: timing with HIP events; run under rocprofv3 --pmc for counters.
    python tools/bench_pw.py --n 112 --h 64 --cin 64 --cout 256 --res [--mask] [--relu-out] [--iters 20] [--pw-variant V]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd.ops import HipOps  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    for k, d in (("n", 112), ("h", 64), ("cin", 64), ("cout", 256), ("iters", 20), ("pw_variant", 0), ("ks", 1)):
        ap.add_argument("--" + k.replace("_", "-"), type=int, default=d)
    ap.add_argument("--res", action="store_true")
    ap.add_argument("--mask", action="store_true")
    ap.add_argument("--relu-out", action="store_true")
    ap.add_argument("--no-bias", action="store_true")
    ap.add_argument("--no-split", action="store_true", help="no split-K workspace (tuning)")
    ap.add_argument("--plain", action="store_true", help="un-packed weights (the LDS-staged patch kernel)")
    a = ap.parse_args()
    ops = HipOps(dtype=torch.bfloat16, stream_conv=not a.plain)
    ops.no_split_k = a.no_split
    if a.pw_variant:
        ops.pw_variant = a.pw_variant
    w = torch.randn((a.cout, a.ks * a.ks, a.cin), device="cuda") / (a.cin * a.ks * a.ks) ** 0.5
    wf, _ = ops.prep_conv_weight(w, None, False)
    x = torch.randn((a.n, a.h, a.h, a.cin), device="cuda").bfloat16()
    res = torch.randn((a.n, a.h, a.h, a.cout), device="cuda").bfloat16() if a.res else None
    mask = torch.randn((a.n, a.h, a.h, a.cout), device="cuda").bfloat16() if a.mask else None
    b = None if a.no_bias else torch.randn((a.cout,), device="cuda")
    f = lambda: ops.conv(x, wf, b, ks=a.ks, res=res, mask=mask, relu_out=a.relu_out)
    for _ in range(3):
        y = f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(a.iters):
        y = f()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / a.iters
    m = a.n * a.h * a.h
    byts = 2 * m * (a.cin + a.cout * (1 + bool(a.res) + bool(a.mask)))
    print(f"{a.n}x{a.h}x{a.h} {a.cin}->{a.cout} ks={a.ks} res={a.res} mask={a.mask}: {ms * 1e3:.1f} us, "
          f"{2.0 * m * a.cin * a.cout * a.ks * a.ks / ms / 1e9:.1f} TF/s, {byts / ms / 1e9:.2f} TB/s of tensor bytes ({byts / 1e6:.0f} MB)")


if __name__ == "__main__":
    main()
