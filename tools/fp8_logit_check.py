#!/usr/bin/env python
"""D's logits on the same images in the bf16 mode and in the MX-fp8 mode, for one initialisation (a large d_loss difference of
tools/fp8_bias_over_inits.py looked at per sample).   usage: PYTHONPATH=. python tools/fp8_logit_check.py --init 2"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--init", type=int, default=2)
    a = ap.parse_args()
    from xmcgan_image_generation_amd import synthetic as syn, train_utils
    from xmcgan_image_generation_amd.configs import coco_xmc
    s = a.init
    out = {}
    for fp8 in (False, True):
        for env in ((), (("XMC_FP8_RELU_STORED", "0"),), (("XMC_MASK_BITS", "0"),)) if fp8 else ((),):
            cfg = coco_xmc.get_c1_config()
            cfg.pretrained_image_contrastive = False
            cfg.batch_size = 8
            cfg.conv_fp8 = fp8
            init = (*syn.init_generator(cfg, seed=100 + s, bias_scale=0.05), *syn.init_discriminator(cfg, seed=200 + s, bias_scale=0.05))
            tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=8, rank=s).items()}
            gen, disc, state = train_utils.create_train_state(cfg, 0)
            state = train_utils.load_flax_params(state, *init)
            g, d = gen(train=True), disc(train=True)
            cond = {k: tb[k][:8] for k in ("sentence_embedding", "embedding", "max_len")}
            img, _, _ = g.forward(state.g_optimizer.target, state.generator_state["batch_stats"], cond, tb["z"][:8], train=True, need_tape=False)
            real = tb["image"][:8].to(img.dtype)
            logit, lv, _, _ = d.forward(state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"],
                                        torch.cat([real, img], 0), cond, need_tape=False, fake_losses=True)
            key = ("fp8" if fp8 else "bf16") + "".join(f" {k}={v}" for k, v in env)
            out[key] = (logit.float().view(-1).cpu().numpy(), img.float().cpu().numpy())
            print(key, "logits real", np.round(out[key][0][:8], 2).tolist(), "fake", np.round(out[key][0][8:], 2).tolist(), flush=True)
            del state, gen, disc, g, d
            torch.cuda.empty_cache()
    b = out["bf16"]
    for k, v in out.items():
        if k != "bf16":
            print(f"{k}: max |logit difference| real {np.abs(v[0][:8] - b[0][:8]).max():.3f} fake {np.abs(v[0][8:] - b[0][8:]).max():.3f}; "
                  f"generated images: max |difference| {np.abs(v[1] - b[1]).max():.4f}")


if __name__ == "__main__":
    main()
