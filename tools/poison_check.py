"""Uninitialised-read check of the training step: every torch.empty / empty_like allocation on the GPU is filled with 0xFF
bytes (NaN in float32 / bf16 / e4m3 / e8m0) before use; a kernel that reads memory nobody wrote then turns the losses into
NaN (or changes them), instead of silently depending on what the caching allocator handed out.
    PYTHONPATH=. python tools/poison_check.py [--fp8] [--dtype bfloat16|float32] [--pretrained]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--fp8", action="store_true")
    ap.add_argument("--dtype", default="bfloat16")
    ap.add_argument("--pretrained", action="store_true")
    ap.add_argument("--batch", type=int, default=8)
    ap.add_argument("--config", default="c1", choices=["c1", "c3"], help="c3 = the 256 px network (BASELINE configs #4 / #5)")
    ap.add_argument("--serial", action="store_true", help="no stream overlap (separates a race from an uninitialised read)")
    a = ap.parse_args()
    if a.serial:
        os.environ.update(XMC_OVERLAP_BWD="0", XMC_PREFETCH_G="0", XMC_OVERLAP_PREP="0")
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_c1_config() if a.config == "c1" else coco_xmc.get_c3_config()
    cfg.batch_size = a.batch
    cfg.dtype = a.dtype
    cfg.conv_fp8 = a.fp8
    cfg.pretrained_image_contrastive = a.pretrained
    ad = {}
    if a.pretrained:
        from xmcgan_image_generation_amd.utils import pretrained_model_utils, resnet_v1
        rp, rs = resnet_v1.init_resnet50(seed=7, head_scale=0.05)
        st = {"params": rp, "batch_stats": rs}
        ad = {"image_model": pretrained_model_utils.ImageModel(st), "image_model_state": st}

    def run(poison):
        o_empty, o_like = torch.empty, torch.empty_like
        if poison:
            def fill(t):
                if t.is_cuda and t.numel():
                    t.view(-1).view(torch.uint8).fill_(0xFF) if t.is_contiguous() else t.fill_(float("nan") if t.is_floating_point() else -1)
                return t
            torch.empty = lambda *x, **k: fill(o_empty(*x, **k))
            torch.empty_like = lambda *x, **k: fill(o_like(*x, **k))
        try:
            torch.manual_seed(0)
            gen, disc, state = train_utils.create_train_state(cfg, 0)
            tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=a.batch).items()}
            outs = []
            for i in range(2):
                state, m = train_utils.train_step(i, state, tb, xmc_gan, gen, disc, cfg, ad)
                outs.append({k: float(v) for k, v in m.items()})
            torch.cuda.synchronize()
            return outs
        finally:
            torch.empty, torch.empty_like = o_empty, o_like

    clean = run(False)
    again = run(False)
    print("two clean runs identical:", clean == again)
    if clean != again:
        for i, (c, p) in enumerate(zip(clean, again)):
            for k in c:
                if c[k] != p[k]:
                    print(f"  step {i} {k:22s} run 1 {c[k]:.9g}  run 2 {p[k]:.9g}")
    pois = run(True)
    ok = clean == again
    for i, (c, p) in enumerate(zip(clean, pois)):
        for k in c:
            same = c[k] == p[k]
            ok = ok and same
            print(f"step {i} {k:22s} clean {c[k]:.9g}  poisoned {p[k]:.9g}  {'==' if same else 'DIFFERENT'}")
    print("poison check", "OK: no loss depends on uninitialised memory" if ok else "FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
