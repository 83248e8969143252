"""Synthetic COCO-shaped batches and random-init parameter trees (host side, NumPy).

There is no network for datasets or checkpoints, so benchmarks and parity tests run on data
of the reference's shapes drawn from fixed NumPy seeds:

* batch contract -- reference ``xmcgan/libml/coco_dataset.py:127-167`` (keys ``image``,
  ``image_aug``, ``embedding``, ``max_len``, ``sentence_embedding``, ``z``);
* parameter tree -- Flax auto-naming of ``xmcgan/nets/xmc_net.py`` / ``nets/common.py`` /
  ``libml/layers.py``, conv kernels HWIO, dense kernels (in, out);
* initialisers -- ``glorot_normal()`` (truncated, fan_avg) for every kernel
  (``xmc_net.py:70,75,181,186``), zero biases, ``u0 ~ N(0, 0.01^2)`` (``layers.py:86-91``),
  BatchNorm running mean 0 / var 1.
"""
from __future__ import annotations

import numpy as np

MAX_WORDS = 17      # reference xmcgan/libml/dataset_constants.py:20
EMB_DIM = 768       # BERT-base hidden size (preprocess_data.py)

G_CHANNELS = {128: [16, 8, 4, 2, 1], 256: [16, 8, 8, 4, 2, 1]}
D_CHANNELS = {128: ([2, 4, 8, 16, 16], [True, True, True, True, False]),
              256: ([2, 4, 8, 8, 16, 16], [True, True, True, True, True, False])}


# ------------------------------------------------------------------------------ shape trees
def _conv(k, cin, cout):
    return {"kernel": (k, k, cin, cout), "bias": (cout,)}


def _dense(cin, cout):
    return {"kernel": (cin, cout), "bias": (cout,)}


def generator_shapes(cfg):
    """(params shapes, batch_stats shapes) of the Generator (xmc_net.py:160-248)."""
    gf, zd = cfg["gf_dim"], cfg["z_dim"]
    chans = G_CHANNELS[cfg["image_size"]]
    cond_g = 2 * zd                      # [Dense_0(sentence) (z_dim), z]
    cond_s = EMB_DIM + cond_g            # [region_context, global_cond]
    p, s = {}, {}
    p["Dense_0"] = _dense(EMB_DIM, zd)
    p["Dense_1"] = _dense(zd, gf * 16 * 4 * 4)
    cin = gf * 16
    for i in range(2):
        cout = gf * chans[i]
        nm = f"GenBlock_{i}"
        p[nm] = {
            "ConditionalBatchNorm_0": {"Dense_0": _dense(cond_g, cin), "Dense_1": _dense(cond_g, cin)},
            "Conv_0": _conv(3, cin, cout),
            "ConditionalBatchNorm_1": {"Dense_0": _dense(cond_g, cout), "Dense_1": _dense(cond_g, cout)},
            "Conv_1": _conv(3, cout, cout),
            "Conv_2": _conv(1, cin, cout),
        }
        s[nm] = {"ConditionalBatchNorm_0": {"BatchNorm_0": {"mean": (cin,), "var": (cin,)}},
                 "ConditionalBatchNorm_1": {"BatchNorm_0": {"mean": (cout,), "var": (cout,)}}}
        cin = cout
    p["Conv_0"] = _conv(1, cin, EMB_DIM)
    for i in range(2, len(chans)):
        cout = gf * chans[i]
        nm = f"GenSpatialBlock_{i - 2}"
        p[nm] = {
            "LocalConditionalBatchNorm_0": {"Conv_0": _conv(1, cond_s, cin), "Conv_1": _conv(1, cond_s, cin)},
            "Conv_0": _conv(3, cin, cout),
            "LocalConditionalBatchNorm_1": {"Conv_0": _conv(1, cond_s, cout), "Conv_1": _conv(1, cond_s, cout)},
            "Conv_1": _conv(3, cout, cout),
            "Conv_2": _conv(1, cin, cout),
        }
        s[nm] = {"LocalConditionalBatchNorm_0": {"BatchNorm_0": {"mean": (cin,), "var": (cin,)}},
                 "LocalConditionalBatchNorm_1": {"BatchNorm_0": {"mean": (cout,), "var": (cout,)}}}
        cin = cout
    p["LocalConditionalBatchNorm_0"] = {"Conv_0": _conv(1, cond_s, cin), "Conv_1": _conv(1, cond_s, cin)}
    s["LocalConditionalBatchNorm_0"] = {"BatchNorm_0": {"mean": (cin,), "var": (cin,)}}
    p["Conv_1"] = _conv(3, cin, 3)
    return p, s


def discriminator_shapes(cfg):
    """(params shapes, spectral_norm_stats shapes) of the Discriminator (xmc_net.py:45-142)."""
    df = cfg["df_dim"]
    chans, downs = D_CHANNELS[cfg["image_size"]]
    p, s = {}, {}

    def sn(shape):
        return {"u0": (1, shape["kernel"][-1])}

    blk = {"SpectralConv_0": _conv(3, 3, df), "SpectralConv_1": _conv(3, df, df),
           "SpectralConv_2": _conv(1, 3, df)}
    p["DiscOptimizedBlock_0"] = blk
    s["DiscOptimizedBlock_0"] = {k: sn(v) for k, v in blk.items()}
    cin = df
    res = cfg["image_size"] // 2
    cond_c = None
    for i, (c, d) in enumerate(zip(chans, downs)):
        cout = df * c
        blk = {"SpectralConv_0": _conv(3, cin, cout), "SpectralConv_1": _conv(3, cout, cout)}
        if d or cin != cout:
            blk["SpectralConv_2"] = _conv(1, cin, cout)
        p[f"DiscBlock_{i}"] = blk
        s[f"DiscBlock_{i}"] = {k: sn(v) for k, v in blk.items()}
        cin = cout
        if d:
            res //= 2
        if res == cfg["cond_size"]:
            cond_c = cout
    p["SpectralDense_0"] = _dense(cin, 1)
    p["SpectralDense_1"] = _dense(EMB_DIM, cin)
    heads = ["SpectralDense_0", "SpectralDense_1"]
    if cfg.get("word_contrastive", True):            # the x_cond 1x1 conv only exists with the word head (xmc_net.py:112-114)
        p["SpectralConv_0"] = _conv(1, cond_c, EMB_DIM)
        heads.append("SpectralConv_0")
    for k in heads:
        s[k] = sn(p[k])
    return p, s


def tree_map(fn, tree, *rest):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v, *[r[k] for r in rest]) for k, v in tree.items()}
    return fn(tree, *rest)


def tree_leaves(tree, prefix=""):
    """Deterministic (insertion-order) flattening: list of (path, leaf)."""
    out = []
    for k, v in tree.items():
        path = f"{prefix}/{k}" if prefix else k
        if isinstance(v, dict):
            out.extend(tree_leaves(v, path))
        else:
            out.append((path, v))
    return out


def count_params(shape_tree):
    return int(sum(int(np.prod(s)) for _, s in tree_leaves(shape_tree)))


# ---------------------------------------------------------------------------------- initialisers
def _glorot_truncated(rng, shape):
    """jax.nn.initializers.glorot_normal(): variance_scaling(1.0, 'fan_avg', 'truncated_normal')."""
    if len(shape) == 2:
        fan_in, fan_out = shape
    else:
        rf = int(np.prod(shape[:-2]))
        fan_in, fan_out = shape[-2] * rf, shape[-1] * rf
    std = np.sqrt(2.0 / (fan_in + fan_out)) / 0.87962566103423978
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2.0
    while bad.any():                       # resample the tails: truncation at +-2 sigma
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2.0
    return (x * std).astype(np.float32)


def _init_params(shape_tree, rng, bias_scale=0.0):
    def leaf(path, shape):
        if path.endswith("kernel"):
            return _glorot_truncated(rng, shape)
        if bias_scale:
            return (rng.standard_normal(shape) * bias_scale).astype(np.float32)
        return np.zeros(shape, np.float32)
    flat = {p: leaf(p, s) for p, s in tree_leaves(shape_tree)}
    return _unflatten(shape_tree, flat)


def _unflatten(shape_tree, flat, prefix=""):
    out = {}
    for k, v in shape_tree.items():
        path = f"{prefix}/{k}" if prefix else k
        out[k] = _unflatten(v, flat, path) if isinstance(v, dict) else flat[path]
    return out


def init_generator(cfg, seed=42, bias_scale=0.0):
    """-> (params, batch_stats) as nested dicts of float32 ndarrays (Flax layout)."""
    rng = np.random.default_rng(seed)
    ps, ss = generator_shapes(cfg)
    params = _init_params(ps, rng, bias_scale)
    stats = tree_map(lambda s: None, ss)
    flat = {}
    for path, shape in tree_leaves(ss):
        flat[path] = (np.ones if path.endswith("var") else np.zeros)(shape, np.float32)
    return params, _unflatten(ss, flat)


def init_discriminator(cfg, seed=43, bias_scale=0.0):
    """-> (params, spectral_norm_stats)."""
    rng = np.random.default_rng(seed)
    ps, ss = discriminator_shapes(cfg)
    params = _init_params(ps, rng, bias_scale)
    flat = {path: (rng.standard_normal(shape) * 0.01).astype(np.float32)
            for path, shape in tree_leaves(ss)}
    return params, _unflatten(ss, flat)


# ------------------------------------------------------------------------------------- batches
def make_batch(cfg, per_device_batch=None, rank=0, seed=1234, dtype=np.float32):
    """One per-device batch for ``train_step``: leading dim = B * d_step_per_g_step.

    image U[0,1); embedding N(0,1) with all 17 rows non-zero (cosine_similarity in the
    reference has no epsilon, attention_lib.py:23-27); max_len integer in [4, 17] stored as
    float (coco_dataset.py:141,158); sentence_embedding = sum over all rows / max_len
    (coco_dataset.py:142); z N(0,1) (coco_dataset.py:165-166).
    """
    b = per_device_batch if per_device_batch is not None else cfg["batch_size"]
    n = b * cfg["d_step_per_g_step"]
    hw = cfg["image_size"]
    rng = np.random.default_rng(seed + rank)
    image = rng.random((n, hw, hw, 3), dtype=np.float32)
    emb = rng.standard_normal((n, MAX_WORDS, EMB_DIM)).astype(np.float32)
    max_len = rng.integers(4, MAX_WORDS + 1, size=(n, 1)).astype(np.float32)
    sent = (emb.sum(axis=1) / max_len).astype(np.float32)
    z = rng.standard_normal((n, cfg["z_dim"])).astype(np.float32)
    out = dict(image=image, image_aug=image.copy(), embedding=emb, max_len=max_len,
               sentence_embedding=sent, z=z)
    return {k: v.astype(dtype) for k, v in out.items()}
