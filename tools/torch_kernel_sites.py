"""Which Python lines of the training step still launch torch (at::native / copy) kernels, and what they cost.

One eager C1 step under a TorchDispatchMode: every aten op that touches device tensors (views excluded) is charged to
the innermost frame inside this package, with the bytes of its tensor arguments + result.
    PYTHONPATH=. python tools/torch_kernel_sites.py [--pretrained on|off] [--batch 56]"""
import argparse
import collections
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from xmcgan_image_generation_amd import synthetic as syn  # noqa: E402
from xmcgan_image_generation_amd import train_utils, xmc_gan  # noqa: E402
from xmcgan_image_generation_amd.configs import coco_xmc  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pretrained", default="on")
    ap.add_argument("--batch", type=int, default=56)
    a = ap.parse_args()
    cfg = coco_xmc.get_c1_config()
    cfg.batch_size = a.batch
    cfg.pretrained_image_contrastive = a.pretrained == "on"
    ad = {}
    if cfg.pretrained_image_contrastive:
        from xmcgan_image_generation_amd.utils import pretrained_model_utils, resnet_v1
        rp, rs = resnet_v1.init_resnet50(seed=7, head_scale=0.05)
        st = {"params": rp, "batch_stats": rs}
        ad = {"image_model": pretrained_model_utils.ImageModel(st), "image_model_state": st}
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=a.batch).items()}
    for _ in range(3):
        state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, ad)
    torch.cuda.synchronize()
    import traceback
    from torch.utils._python_dispatch import TorchDispatchMode
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__))) + "/"
    sites = collections.defaultdict(lambda: [0, 0])

    class Log(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            out = func(*args, **(kwargs or {}))
            name = str(func)
            base = name.replace("aten.", "").split(".")[0]      # the operator's own name: "cat", "repeat", "t", "is_same_size" ...
            # (round 5: this was a substring test on the full name, and "t.default" also matched cat.default / repeat.default)
            if base in ("view", "as_strided", "slice", "select", "reshape", "detach", "alias", "unsqueeze", "squeeze", "permute",
                        "transpose", "expand", "empty", "empty_like", "empty_strided", "t", "_unsafe_view", "split", "unbind",
                        "narrow", "size", "stride", "numel", "split_with_sizes", "unsafe_split", "sym_size", "sym_stride",
                        "sym_numel", "_reshape_alias", "view_as", "expand_as", "lift_fresh") or base.startswith("is_"):
                return out
            byts = sum(t.numel() * t.element_size() for t in (list(args) + [out]) if torch.is_tensor(t) and t.is_cuda)
            if byts == 0:
                return out
            st = [f for f in traceback.extract_stack() if f.filename.startswith(root) and "tools/" not in f.filename]
            fr = st[-1] if st else None
            key = (f"{fr.filename.replace(root, '')}:{fr.lineno} {fr.line[:70]}" if fr else "<outside>", name.replace("aten.", ""))
            sites[key][0] += 1
            sites[key][1] += byts
            return out

    with Log():
        state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, ad)
    torch.cuda.synchronize()
    print(f"torch ops touching device tensors in one eager step: {sum(v[0] for v in sites.values())}")
    print(f"{'site':130s} {'op':24s} {'n':>4s} {'MB':>9s}")
    for (frame, name), (cnt, byts) in sorted(sites.items(), key=lambda kv: -kv[1][1])[:70]:
        print(f"{frame[:130]:130s} {name[:24]:24s} {cnt:4d} {byts / 1e6:9.2f}")


if __name__ == "__main__":
    main()
