"""Per-launch timing of the frozen ResNet-50 feature path (utils/pretrained_model_utils.ResNet50Features): forward on
2B images + data gradient on B, every operator-table call bracketed by HIP events (fourth pass reported: allocator and event pool warm).
    python tools/bench_resnet.py [--batch 56] [--dtype bfloat16]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd.ops import HipOps  # noqa: E402
from xmcgan_image_generation_amd.utils import pretrained_model_utils as P  # noqa: E402
from xmcgan_image_generation_amd.utils import resnet_v1 as RV  # noqa: E402

NAMES = ["conv", "stem_conv", "stem_dgrad", "resize_to_canvas", "resize_to_canvas_bwd", "stem_im2col", "stem_col2im", "maxpool3x3s2", "maxpool3x3s2_bwd",
         "zero_margin_", "subsample2", "subsample2_bwd", "add_relu", "relu_bwd", "reduce_mid", "gemm", "bcast_relu_bwd"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=56)
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--detail", action="store_true", help="one line per convolution launch")
    ap.add_argument("--pw-variant", type=int, default=0, help="pointwise-kernel tuning hook: 0 auto, 1 <32,3>, 2 <64,3>, 3 <32,4>")
    a = ap.parse_args()
    dt = getattr(torch, a.dtype)
    ops = HipOps(dtype=dt)
    if a.pw_variant:
        ops.pw_variant = a.pw_variant
    p, s = RV.init_resnet50(1, head_scale=0.05)
    net = P.ResNet50Features(ops, p, s)
    b = a.batch
    x = (torch.rand((2 * b, 128, 128, 3)) * 2 - 1).to(dt).cuda()
    dl = torch.randn((b, 1000)).cuda()
    recs = []
    orig = {n: getattr(ops, n) for n in NAMES}

    def wrap(name):
        f = orig[name]

        def g(*args, **kw):
            fl = getattr(ops, "acct_flops", None) if name == "conv" else None
            s0, e0 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s0.record()
            y = f(*args, **kw)
            e0.record()
            byts = sum(t.numel() * t.element_size() for t in list(args) + list(kw.values()) + [y] if torch.is_tensor(t))
            tag = name
            if name == "conv":
                xx, w = args[0], args[1]
                packed = hasattr(w, "taps")
                tag = f"conv{kw.get('ks')}x{kw.get('ks')} {tuple(xx.shape)}->{w.cout if packed else w.shape[0]}" + \
                      (" relu_in" if kw.get("relu_in") else "") + (" mask" if kw.get("mask") is not None else "") + \
                      (" res" if kw.get("res") is not None else "")
            recs.append((name, tag, fl, byts, s0, e0))
            return y
        return g

    for rep in range(4):
        if rep == 2:
            for n in NAMES:
                setattr(ops, n, wrap(n))
        recs.clear()                                      # pass 2 creates the events (first use is slow), pass 3 counts
        torch.cuda.synchronize()
        t0, t1, t2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        t0.record()
        logits, tape = net.forward(x, reuse_buffers=os.environ.get("XMC_BENCH_REUSE", "1") != "0")   # the training step's mode
        t1.record()
        net.backward(tape, dl, b, 2 * b)
        t2.record()
        torch.cuda.synchronize()
        print(f"pass {rep}: forward({2 * b}) {t0.elapsed_time(t1):.3f} ms, backward({b}) {t1.elapsed_time(t2):.3f} ms")
    agg = collections.OrderedDict()
    for name, tag, fl, byts, s0, e0 in recs:
        key = tag if (a.detail or name != "conv") else ("conv3x3" if "conv3x3" in tag else "conv1x1")
        d = agg.setdefault(key, [0, 0.0, 0.0, 0.0])
        d[0] += 1
        d[1] += s0.elapsed_time(e0)
        d[2] += fl or 0.0
        d[3] += byts
    tot = sum(d[1] for d in agg.values())
    print(f"{'op':64s} calls      ms    TF/s    GB/s (tensor bytes)")
    for k, d in agg.items():
        print(f"{k:64s} {d[0]:5d} {d[1]:7.3f} {d[2] / d[1] / 1e9 if d[1] else 0:7.1f} {d[3] / d[1] / 1e6 if d[1] else 0:7.0f}")
    print(f"{'TOTAL (event-bracketed, includes host gaps)':64s}       {tot:7.3f}")


if __name__ == "__main__":
    main()
