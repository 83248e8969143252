"""ResNet-50 v1 parameter tree (reference ``xmcgan/utils/resnet_v1.py:60-186``): shapes, Flax names and the
initialisers ``model.init`` uses (host side, NumPy).  The frozen feature network itself -- forward on 2B images and
the data gradient onto the generated half -- is ``utils/pretrained_model_utils.ResNet50Features``.

Flax names: ``init_conv`` (7, 7, 3, 64), ``init_bn``, ``stage{1..4}/block{k}/{conv1, bn1, conv2, bn2, conv3, bn3,
proj_conv, proj_bn}`` (``proj_*`` in the first block of every stage), ``head`` (2048 -> num_classes, zero-initialised
kernel: resnet_v1.py:170-171); BatchNorm leaves ``scale`` / ``bias`` in ``params`` and ``mean`` / ``var`` in
``batch_stats``.
"""
from __future__ import annotations

import numpy as np

STAGE_SIZES = [3, 4, 6, 3]


def resnet50_shapes(num_classes: int = 1000):
    """-> (params shape tree, batch_stats shape tree)"""
    p, s = {}, {}

    def bn(c):
        return {"scale": (c,), "bias": (c,)}, {"mean": (c,), "var": (c,)}
    p["init_conv"] = {"kernel": (7, 7, 3, 64)}
    p["init_bn"], s["init_bn"] = bn(64)
    cin = 64
    for i, n in enumerate(STAGE_SIZES):
        f = 64 * 2 ** i
        sp, ss = {}, {}
        for k in range(n):
            bp, bs = {}, {}
            bp["conv1"] = {"kernel": (1, 1, cin, f)}
            bp["bn1"], bs["bn1"] = bn(f)
            bp["conv2"] = {"kernel": (3, 3, f, f)}
            bp["bn2"], bs["bn2"] = bn(f)
            bp["conv3"] = {"kernel": (1, 1, f, 4 * f)}
            bp["bn3"], bs["bn3"] = bn(4 * f)
            if k == 0:                                   # residual.shape != x.shape (resnet_v1.py:78-82)
                bp["proj_conv"] = {"kernel": (1, 1, cin, 4 * f)}
                bp["proj_bn"], bs["proj_bn"] = bn(4 * f)
            sp[f"block{k + 1}"], ss[f"block{k + 1}"] = bp, bs
            cin = 4 * f
        p[f"stage{i + 1}"], s[f"stage{i + 1}"] = sp, ss
    p["head"] = {"kernel": (cin, num_classes), "bias": (num_classes,)}
    return p, s


def forward_flops_per_image(num_classes: int = 1000) -> float:
    """algorithmic FLOPs (2 x MACs at the true 112 / 56 / 28 / 14 / 7 feature-map sizes) of one 224 x 224 forward pass"""
    fl = 2.0 * 112 * 112 * 147 * 64
    cin, h = 64, 56
    for i, n in enumerate(STAGE_SIZES):
        f = 64 * 2 ** i
        for k in range(n):
            hin = h
            if i > 0 and k == 0:
                h //= 2
            fl += 2.0 * hin * hin * cin * f + 2.0 * h * h * 9 * f * f + 2.0 * h * h * f * 4 * f
            if k == 0:
                fl += 2.0 * h * h * cin * 4 * f
            cin = 4 * f
    return fl + 2.0 * cin * num_classes


def count_params(tree) -> int:
    """number of scalars in a tree of arrays (or of shape tuples, as ``resnet50_shapes`` returns)"""
    return sum(count_params(v) if isinstance(v, dict) else int(np.prod(v if isinstance(v, tuple) else np.shape(v)))
               for v in tree.values())


def init_resnet50(seed: int = 42, num_classes: int = 1000, head_scale: float = 0.0, randomize_bn: bool = False):
    """Random-init trees as ``model.init`` produces them (pretrained_model_utils.py:33-56): conv kernels
    lecun_normal (flax nn.Conv default), BatchNorm scale 1 / bias 0 / mean 0 / var 1, head kernel zeros.
    ``head_scale`` / ``randomize_bn`` (tests only) make the head and the BatchNorm statistics non-trivial."""
    rng = np.random.default_rng(seed)
    ps, ss = resnet50_shapes(num_classes)

    def fill(shape_tree, path=""):
        out = {}
        for k, v in shape_tree.items():
            p = f"{path}/{k}" if path else k
            if isinstance(v, dict):
                out[k] = fill(v, p)
            elif k == "kernel" and len(v) == 4:
                fan_in = v[0] * v[1] * v[2]
                out[k] = (rng.standard_normal(v) * np.sqrt(1.0 / fan_in)).astype(np.float32)
            elif k == "kernel":
                out[k] = (rng.standard_normal(v) * head_scale).astype(np.float32)
            elif k == "scale":
                out[k] = (1.0 + 0.2 * rng.standard_normal(v)).astype(np.float32) if randomize_bn else np.ones(v, np.float32)
            elif k == "var":
                out[k] = (0.5 + rng.random(v)).astype(np.float32) if randomize_bn else np.ones(v, np.float32)
            elif k in ("bias", "mean"):
                out[k] = (0.1 * rng.standard_normal(v)).astype(np.float32) if randomize_bn else np.zeros(v, np.float32)
            else:
                raise KeyError(p)
        return out
    return fill(ps), fill(ss)
