#!/usr/bin/env python
"""Is the one-sided fp8 - bf16 difference of g_loss a property of the fp8 mode or of ONE network?  tests/test_gpu_mx8.py varies
the data over a fixed initialisation; this varies the initialisation (C1 network, per-device batch 8, first step).
usage (GPU box): PYTHONPATH=. python tools/fp8_bias_over_inits.py [--inits 8]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--inits", type=int, default=8)
    a = ap.parse_args()
    from tests.test_gpu_mx8 import _run_steps
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd.configs import coco_xmc
    keys = ("d_loss", "g_loss", "c_loss_d", "c_loss_g")

    def cfg_of(fp8):
        cfg = coco_xmc.get_c1_config()
        cfg.pretrained_image_contrastive = False
        cfg.batch_size = 8
        cfg.conv_fp8 = fp8
        return cfg
    cfg = cfg_of(False)
    signed = {k: [] for k in keys}
    for s in range(a.inits):
        init = (*syn.init_generator(cfg, seed=100 + s, bias_scale=0.05), *syn.init_discriminator(cfg, seed=200 + s, bias_scale=0.05))
        tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=8, rank=s).items()}
        m16, _ = _run_steps(cfg_of(False), init, [tb])
        m8, _ = _run_steps(cfg_of(True), init, [tb])
        scale = max(abs(m16[0][k]) for k in keys)
        for k in keys:
            signed[k].append((m8[0][k] - m16[0][k]) / scale)
        print(f"init {s}: bf16 " + " ".join(f"{k} {m16[0][k]:.3f}" for k in keys) + " | fp8 - bf16 (relative to the largest loss) "
              + " ".join(f"{k} {signed[k][-1]:+.4f}" for k in keys), flush=True)
    for k in keys:
        v = np.array(signed[k])
        print(f"{k}: mean {v.mean():+.4f}  std {v.std():.4f}  positive {int((v > 0).sum())} of {len(v)}  worst {np.abs(v).max():.4f}")


if __name__ == "__main__":
    main()
