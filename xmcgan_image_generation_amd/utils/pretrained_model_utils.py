"""Frozen ResNet-50 feature network of the pretrained image-contrastive term.

Mirrors the reference's ``xmcgan/utils/pretrained_model_utils.py``: ``get_pretrained_model`` (:59-99, checkpoint =
``np.save`` of ``{"params", "batch_stats"}`` Flax trees) and ``get_pretrained_embs`` (:102-127: bilinear resize to
224 x 224, ``ResNet50(train=False)`` -> ``(pool (N, 7, 7, 2048), outputs (N, 1000))``), plus what ``jax.vjp`` gives the
reference for free: the data gradient of the logits onto the (generated) input images.

MI355X-first structure.  ResNet's feature maps are 112^2, 56^2, 28^2, 14^2, 7^2 -- not powers of two, which the tiled
convolution kernels assume -- so every feature map lives on a power-of-two CANVAS (128^2 ... 8^2; valid region
top-left).  A stride-1 SAME convolution on a canvas with a ZERO margin equals the SAME convolution of the valid
region, hence all 53 convolutions run on the library's existing kernels (``xmc_conv2d_nhwc``; the bf16 3x3 layers on
the weight-streaming kernel) at 1.31x the pixels; stride-2 convolutions are the stride-1 result sub-sampled
(3x3: odd positions, 1x1: even positions, as flax's SAME padding places them); eval-mode BatchNorm is folded into
the convolution weights and biases once (the network is frozen); the 7x7 stride-2 stem is an im2col + 1x1
convolution.  The margins are kept zero by the convolution epilogue itself (``valid``: the 1x1 that FEEDS a 3x3, and the
block outputs), which also applies the post-activation residual ReLU (``relu_out``) and, in the backward pass, its mask
on the summed gradient (``mask_after_res``): no separate pointwise pass inside a bottleneck block.
The network follows ``ops.dtype`` (bf16 in the training configs; the reference evaluates ResNet in float32 --
the float32 parity mode does too).
"""
from __future__ import annotations

import os

import numpy as np
import torch

from . import resnet_v1

_DEFAULT_RESNET_PATH = "data/resnet_pretrained.npy"          # pretrained_model_utils.py:28
RESNET_IMG_SIZE = 224
VALID_MODELS = ["resnet50"]
_EPS = 1e-5
_SKIP3 = os.environ.get("XMC_RESNET_SKIP3", "1") != "0"            # A/B switch: 3x3 launches skip the tiles that lie in the canvas margin
_STEM_FUSED = os.environ.get("XMC_RESNET_STEM_FUSED", "1") != "0"  # A/B switch: the stem as one implicit-GEMM launch (no im2col columns)
_DUAL = os.environ.get("XMC_RESNET_DUAL", "1") != "0"          # A/B switch: projection shortcut folded into the block's last 1x1 launch


def get_pretrained_model(model_name: str = "resnet50", checkpoint_path=_DEFAULT_RESNET_PATH, seed: int = 42):
    """-> (params, batch_stats) NumPy trees in Flax layout.  ``checkpoint_path``: the reference's ``.npy`` pickle of
    ``{"params", "batch_stats"}`` (pretrained_model_utils.py:93-98).  A path that does not exist raises, as the reference's
    ``np.load`` does: silently training with ``model.init``'s zero head would make the term the constant 2 ln B with zero
    gradient while still paying for the ResNet-50 passes.  ``checkpoint_path=None`` is the explicit opt-in
    to random initialisation (tests, benchmarks without the download)."""
    if model_name not in VALID_MODELS:
        raise ValueError(f"Model {model_name} not supported.")
    if checkpoint_path is not None and os.path.exists(checkpoint_path):
        data = np.load(checkpoint_path, allow_pickle=True).item()
        return data["params"], data["batch_stats"]
    if checkpoint_path is not None:
        raise FileNotFoundError(f"{checkpoint_path}: ResNet-50 checkpoint of pretrained_image_contrastive not found "
                                f"(pass checkpoint_path=None / config.pretrained_model_path=None for random initialisation, or "
                                f"set config.pretrained_image_contrastive=False)")
    return resnet_v1.init_resnet50(seed)


def _fold(kernel, bn_p, bn_s):
    """HWIO kernel + eval-mode BatchNorm -> (master (cout, taps, cin) float32, bias (cout,))"""
    k = np.asarray(kernel, np.float64)
    a = np.asarray(bn_p["scale"], np.float64) / np.sqrt(np.asarray(bn_s["var"], np.float64) + _EPS)
    w = np.transpose(k, (3, 0, 1, 2)).reshape(k.shape[3], k.shape[0] * k.shape[1], k.shape[2]) * a[:, None, None]
    b = np.asarray(bn_p["bias"], np.float64) - np.asarray(bn_s["mean"], np.float64) * a
    return w.astype(np.float32), b.astype(np.float32)


class _Conv:
    def __init__(self, ops, w, b, ks, k_true=None, stride=1, fwd_only=False):
        self.ops, self.ks = ops, ks
        self.mac_per_pixel = (k_true if k_true is not None else w.shape[1] * w.shape[2]) * w.shape[0]
        dev = ops.device
        self.b = torch.as_tensor(b).to(dev)
        # a 3x3 stride-2 layer also gets the phase copies of the strided form (bf16 weight-streaming path only): the layer
        # then runs at its TRUE output resolution instead of "stride 1, then sub-sample" (4x the pixels)
        self.wf, self.wd = ops.prep_conv_weight(torch.as_tensor(w).to(dev).contiguous(), None, True,
                                                **({"phase": "s2"} if stride == 2 and ks == 3 else {}))
        if fwd_only:
            self.wd = None

    # ``true_hw``: side of the convolution's TRUE output map (the canvas is larger, and a stride-2 layer is computed at
    # stride 1).  Accounting only: bench.py's per-launch counter takes ``ops.acct_flops`` (the algorithmic FLOPs of the
    # launch that follows) instead of the launch geometry, and clears it
    def fwd(self, x, true_hw=None, **kw):
        self.ops.acct_flops = None if true_hw is None else 2.0 * x.shape[0] * true_hw * true_hw * self.mac_per_pixel
        return self.ops.conv(x, self.wf, self.b, ks=self.ks, **kw)

    def dgrad(self, dy, true_hw=None, **kw):
        self.ops.acct_flops = None if true_hw is None else 2.0 * dy.shape[0] * true_hw * true_hw * self.mac_per_pixel
        return self.ops.conv(dy, self.wd, None, ks=self.ks, **kw)


class ResNet50Features:
    """ResNet-50 v1 in inference mode (resnet_v1.py:129-172) with the data gradient onto a batch slice."""

    def __init__(self, ops, params, batch_stats):
        self.ops = ops
        self._bufs = {}
        p, s = params, batch_stats
        # stem: (7, 7, 3, 64) -> 1x1 convolution over the im2col columns k = tap * 3 + ch, padded 147 -> 160
        w, b = _fold(p["init_conv"]["kernel"], p["init_bn"], s["init_bn"])               # (64, 49, 3)
        w160 = np.zeros((64, 1, 160), np.float32)
        w160[:, 0, :147] = w.reshape(64, 147)
        self.stem = _Conv(ops, w160, b, 1, k_true=147)
        # the same layer as one implicit-GEMM launch (bf16 training step; ops.stem_conv): weights in its fragment order
        self.stem_frag = ops.pack_stem_weight(w) if (_STEM_FUSED and hasattr(ops, "pack_stem_weight")) else None
        self.stem_dfrag = ops.pack_stem_dgrad_weight(w) if (_STEM_FUSED and hasattr(ops, "pack_stem_dgrad_weight")) else None
        self.blocks = []
        for i, n in enumerate(resnet_v1.STAGE_SIZES):
            for k in range(n):
                bp, bs = p[f"stage{i + 1}"][f"block{k + 1}"], s[f"stage{i + 1}"][f"block{k + 1}"]
                blk = dict(stride=2 if (i > 0 and k == 0) else 1,
                           c1=_Conv(ops, *_fold(bp["conv1"]["kernel"], bp["bn1"], bs["bn1"]), 1),
                           c2=_Conv(ops, *_fold(bp["conv2"]["kernel"], bp["bn2"], bs["bn2"]), 3, stride=2 if (i > 0 and k == 0) else 1),
                           c3=_Conv(ops, *_fold(bp["conv3"]["kernel"], bp["bn3"], bs["bn3"]), 1),
                           proj=_Conv(ops, *_fold(bp["proj_conv"]["kernel"], bp["proj_bn"], bs["proj_bn"]), 1)
                           if "proj_conv" in bp else None)
                if blk["proj"] is not None and _DUAL:
                    # relu(bn3(conv3(h)) + proj_bn(proj_conv(x))) as ONE pointwise launch over the concatenated channels [h | x(s y, s x)]
                    # (ops.conv(x2=...), forward only: the two data gradients keep their own weights)
                    (w3, b3), (wp, bpj) = _fold(bp["conv3"]["kernel"], bp["bn3"], bs["bn3"]), _fold(bp["proj_conv"]["kernel"], bp["proj_bn"], bs["proj_bn"])
                    blk["c3p"] = _Conv(ops, np.concatenate([w3, wp], axis=2), b3 + bpj, 1, fwd_only=True)
                    # ... and the block's data gradient mask(conv1^T(dh1) + scatter2(proj^T(g))) likewise: [dh1 | g'] [W1^T | Wp^T]^T
                    w1, _ = _fold(bp["conv1"]["kernel"], bp["bn1"], bs["bn1"])                         # (cm, 1, cin), (c4, 1, cin)
                    blk["c1pd"] = _Conv(ops, np.concatenate([np.transpose(w1, (2, 1, 0)), np.transpose(wp, (2, 1, 0))], axis=2),
                                        np.zeros((w1.shape[2],), np.float32), 1, fwd_only=True)
                    blk["c1pd"].b = None
                self.blocks.append(blk)
        dev = ops.device
        self.head_w = torch.as_tensor(np.asarray(p["head"]["kernel"], np.float32)).to(dev).contiguous()   # (2048, classes)
        self.head_b = torch.as_tensor(np.asarray(p["head"]["bias"], np.float32)).to(dev).contiguous()

    # ------------------------------------------------------------------------------------ forward
    def _buf(self, key, shape):
        """zero-initialised buffer that persists from call to call (compact mode: the pointwise launches write only the
        valid corner of a canvas; its margin stays zero because nothing ever writes it)"""
        k = (key,) + tuple(shape)
        b = self._bufs.get(k)
        if b is None:
            b = self._bufs[k] = torch.zeros(tuple(shape), dtype=self.ops.dtype, device=self.ops.device)
        return b

    def forward(self, images, need_tape=True, reuse_buffers=False):
        """images (N, H, H, 3) in the activation dtype -> (logits (N, classes) float32, tape).  ``reuse_buffers`` (the
        training step): the 1x1 layers run in the pointwise kernel's COMPACT mode -- they walk the valid corner of each
        canvas only (56^2 of 64^2: 23 % fewer pixels) and write into buffers owned by this object, which the NEXT call
        overwrites: the tape is valid until then."""
        ops = self.ops
        # (bf16 weight-streaming path only: the float32 parity mode has no compact kernel and keeps fresh tensors per call)
        cp = bool(reuse_buffers and (ops.resnet_step_mode() if hasattr(ops, "resnet_step_mode") else False))
        ck = (lambda key, shape: dict(compact=True, out=self._buf(key, shape))) if cp else (lambda key, shape: {})
        n, hs = images.shape[0], images.shape[1]
        x0 = ops.resize_to_canvas(images, RESNET_IMG_SIZE, 256)          # (the identity when the images are 224 already)
        if cp and self.stem_frag is not None:
            ops.acct_flops = None
            s0 = ops.stem_conv(x0, self.stem_frag, self.stem.b, RESNET_IMG_SIZE, 112, self._buf("s0", (n, 128, 128, 64)))
        else:
            col = ops.stem_im2col(x0, RESNET_IMG_SIZE, 128)                             # (N, 128, 128, 160)
            s0 = self.stem.fwd(col, 112, **(dict(valid=112, **ck("s0", (n, 128, 128, 64))) if cp else {}))   # init_conv + init_bn, valid 112
        x, pool_idx = ops.maxpool3x3s2(s0, 112)                                         # valid 56 on a 64 canvas (no ReLU: :155-156)
        hv, tapes = 56, []
        for bi, blk in enumerate(self.blocks):
            st = blk["stride"]
            ho = hv // st
            hc, hco = x.shape[1], x.shape[1] // st                                      # canvas sides at the block's input / output
            h1 = blk["c1"].fwd(x, hv, relu_out=True, valid=hv, emit_bits=True,
                               **ck(("h1", bi), (n, hc, hc, blk["c1"].b.numel())))                         # relu(bn1(conv1)), zero margin
            s2 = st == 2 and hasattr(ops, "can_stride2") and ops.can_stride2(blk["c2"].wf, h1.shape[1], h1.shape[2])
            if s2:
                h2 = blk["c2"].fwd(h1, ho, relu_out=True, stride2=True, emit_bits=True)                 # relu(bn2(conv2)), natively at stride 2
            else:
                # relu(bn2(conv2)) at stride 1 (compact: its consumer reads the valid corner only -- margin tiles are skipped)
                h2 = blk["c2"].fwd(h1, ho, relu_out=True, emit_bits=st != 2, **(dict(compact=True, valid=hv) if cp and st == 1 and _SKIP3 else {}))
                if st == 2:
                    h2 = ops.subsample2(h2, 1)                                          # 3x3 stride 2 SAME: centres at 2o + 1
            xs = x
            if cp and "c3p" in blk:
                out = blk["c3p"].fwd(h2, ho, x2=x, x2_stride=st, relu_out=True, valid=ho, emit_bits=True,
                                     **ck(("out", bi), (n, hco, hco, blk["c3"].b.numel())))
                tapes.append((x, h1, h2, out, hv))
                x, hv = out, ho
                continue
            if blk["proj"] is not None:
                if st == 2:
                    xs = ops.subsample2(x, 0)                                           # 1x1 stride 2 SAME: reads 2o
                r = blk["proj"].fwd(xs, ho, **(dict(valid=ho, **ck(("r", bi), (n, hco, hco, blk["proj"].b.numel()))) if cp else {}))
            else:
                r = x
            out = blk["c3"].fwd(h2, ho, res=r, relu_out=True, valid=ho, emit_bits=True,
                                **ck(("out", bi), (n, hco, hco, blk["c3"].b.numel())))                     # relu(residual + bn3(conv3)) :86
            tapes.append((x, h1, h2, out, hv))
            x, hv = out, ho
        c = x.shape[-1]
        pooled = ops.reduce_mid(x.view(n, -1, c), scale=1.0 / (hv * hv))                # jnp.mean(pool, (1, 2)) :167
        logits = self.head_b.unsqueeze(0).repeat(n, 1)
        ops.gemm(pooled, self.head_w, beta=1.0, out=logits)                             # head :168-171
        tape = dict(tapes=tapes, pool_idx=pool_idx, x5=x, hs=hs, n=n, compact=cp) if need_tape else None
        return logits, tape

    # ----------------------------------------------------------------------------------- backward
    def backward(self, tape, dlogits, lo, hi):
        """d(loss) / d images[lo:hi] given d(loss) / d logits[lo:hi] (float32 (hi - lo, classes)); dgrad only."""
        ops = self.ops
        n = hi - lo
        cp = tape.get("compact", False)
        ck = (lambda key, shape: dict(compact=True, out=self._buf(key, shape))) if cp else (lambda key, shape: {})
        x5 = tape["x5"][lo:hi]
        c = x5.shape[-1]
        dpool = ops.gemm(dlogits, self.head_w, tb=True, alpha=1.0 / 49.0)               # (n, 2048): mean over 7 x 7
        g = ops.bcast_relu_bwd(dpool, x5.reshape(n, -1, c)).view(x5.shape)              # through the last ReLU (margin: x5 == 0)
        for bi, (blk, (x, h1, h2, out, hv)) in reversed(list(enumerate(zip(self.blocks, tape["tapes"])))):
            bs = ops.bslice if hasattr(ops, "bslice") else (lambda t, a, b: t[a:b])      # keeps the ReLU-mask bits
            x, h1, h2 = bs(x, lo, hi), bs(h1, lo, hi), bs(h2, lo, hi)
            st = blk["stride"]
            ho = hv // st
            dh2 = blk["c3"].dgrad(g, ho, mask=h2, valid=ho, **ck(("dh2", bi), (n,) + tuple(h2.shape[1:])))   # through conv3 and the ReLU after bn2;
                                                                                        # the 3x3 dgrad must see a zero margin
            if st == 2 and hasattr(ops, "can_stride2") and ops.can_stride2(blk["c2"].wd, dh2.shape[1], dh2.shape[2]):
                dh1 = blk["c2"].dgrad(dh2, ho, mask=h1, stride2=True)                   # adjoint of the strided layer, 4 output phases
            else:
                if st == 2:
                    dh2 = ops.subsample2_bwd(dh2, 1)
                dh1 = blk["c2"].dgrad(dh2, ho, mask=h1, **(dict(compact=True, valid=hv) if cp and st == 1 and _SKIP3 else {}))   # through conv2 and the ReLU after bn1
            first = blk is self.blocks[0]
            if cp and "c1pd" in blk:
                # + the projection shortcut's gradient, scattered to the even pixels of a stride-2 block, in the SAME launch; then
                # through the previous block's output ReLU (the first block's input is the max-pool output: no ReLU there)
                cm, c4, cb = dh1.shape[-1], g.shape[-1], x.shape[-1]
                ops.acct_flops = 2.0 * n * cb * (hv * hv * cm + ho * ho * c4)
                g = ops.conv(dh1, blk["c1pd"].wf, None, ks=1, x2=g, x2_stride=1 if st == 1 else -2, mask=None if first else x, valid=hv,
                             **ck(("g", bi), (n,) + tuple(x.shape[1:])))
                continue
            if blk["proj"] is not None:
                dsc = blk["proj"].dgrad(g, ho, **(dict(valid=ho, **ck(("dsc", bi), (n, g.shape[1], g.shape[2], x.shape[-1]))) if cp else {}))
                if st == 2:
                    dsc = ops.subsample2_bwd(dsc, 0)
            else:
                dsc = g
            # + the shortcut gradient; then through the previous block's output ReLU (x is its post-ReLU output; the
            # first block's input is the max-pool output: no ReLU there)
            g = blk["c1"].dgrad(dh1, hv, res=dsc, mask=None if first else x, mask_after_res=not first,
                                **(dict(valid=hv, **ck(("g", bi), (n,) + tuple(x.shape[1:]))) if cp else {}))
        ds0 = ops.maxpool3x3s2_bwd(g, tape["pool_idx"][lo:hi], 112)
        if cp and self.stem_dfrag is not None:
            dx0 = ops.stem_dgrad(ds0, self.stem_dfrag, 112, 256)                         # valid 224 corner of a (n, 256, 256, 3) canvas
        else:
            dcol = self.stem.dgrad(ds0, 112, **(dict(valid=112, **ck("dcol", (n, 128, 128, 160))) if cp else {}))   # (n, 128, 128, 160)
            dx0 = ops.stem_col2im(dcol, 256, RESNET_IMG_SIZE)
        return ops.resize_to_canvas_bwd(dx0, tape["hs"], RESNET_IMG_SIZE)


class ImageModel:
    """What ``create_additional_data`` hands to ``train_g_d`` as ``image_model``: the checkpoint trees plus one
    ``ResNet50Features`` per operator table (built on first use: the folded weights live on that table's device)."""

    def __init__(self, state):
        self.state = state
        self._bound = {}

    def bind(self, ops) -> ResNet50Features:
        key = id(ops)
        if key not in self._bound:
            self._bound[key] = ResNet50Features(ops, self.state["params"], self.state["batch_stats"])
        return self._bound[key]


def get_pretrained_embs(state, model, images, ops=None):
    """pretrained_model_utils.py:102-127 on the HIP path: -> (pool (N, 7, 7, 2048), outputs (N, classes)).  ``state`` is
    unused (the folded weights live in ``model``: a ``ResNet50Features``, or an ``ImageModel`` together with ``ops``);
    kept for the reference's argument order."""
    if images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("images should be of shape (H, W, 3).")
    if isinstance(model, ImageModel):
        model = model.bind(ops)
    logits, tape = model.forward(images.to(model.ops.dtype).contiguous(), need_tape=True)
    pool = tape["x5"][:, :7, :7, :]
    return pool, logits
