#!/usr/bin/env python
"""Per-call timing of one op family inside a real C1 step (HIP events around each call).
usage (GPU box): python tools/prof_ops.py gemm|conv|conv_wgrad|... [--batch 56]"""
import argparse
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd import synthetic, train_utils, xmc_gan  # noqa: E402
from xmcgan_image_generation_amd.configs import coco_xmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("op")
    ap.add_argument("--batch", type=int, default=56)
    args = ap.parse_args()
    cfg = coco_xmc.get_c1_config()
    cfg.pretrained_image_contrastive = False
    cfg.batch_size = args.batch
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    batch = {k: torch.as_tensor(v).cuda() for k, v in
             synthetic.make_batch(cfg, per_device_batch=args.batch).items()}
    ops = gen(train=True).ops
    for _ in range(2):
        state, _ = train_utils.train_step(0, state, batch, xmc_gan, gen, disc, cfg, {})
    recs = []
    orig = getattr(ops, args.op)

    def wrapped(*a, **kw):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        r = orig(*a, **kw)
        e.record()
        shapes = tuple(tuple(t.shape) if hasattr(t, "shape") else (("packed", t.cout, t.taps, t.cin) if hasattr(t, "cout") else t)
                       for t in a if t is not None)
        recs.append((shapes, tuple(sorted((k, v) for k, v in kw.items() if isinstance(v, (bool, int, float)))), s, e))
        return r
    setattr(ops, args.op, wrapped)
    state, _ = train_utils.train_step(0, state, batch, xmc_gan, gen, disc, cfg, {})
    torch.cuda.synchronize()
    agg = defaultdict(lambda: [0, 0.0])
    for sh, kw, s, e in recs:
        a = agg[(sh, kw)]
        a[0] += 1
        a[1] += s.elapsed_time(e)
    tot = sum(v[1] for v in agg.values())
    for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print(f"{v[1]:8.3f} ms  x{v[0]:3d}  {k[0]} {dict(k[1])}")
    print(f"total {tot:.3f} ms in {len(recs)} calls")


if __name__ == "__main__":
    main()
