"""XMC-GAN losses and update rule: ``train_d`` / ``train_g_d`` (reference ``xmcgan/xmc_gan.py``).

Same call surface as the reference (``train_d(rng, state, batch, generator, discriminator,
config)``, ``train_g_d(..., config, additional_data)``, ``create_additional_data``,
``calculate_contrastive_loss``); ``jax.vjp`` + the two pullbacks (xmc_gan.py:162-167) are
replaced by an explicit schedule over the HIP kernels:

    train_d   : G fwd (no tape) -> D fwd -> D backward (d-stream, 2B samples, dgrad + wgrad)
    train_g_d : G fwd (tape)    -> D fwd -> D backward (d-stream) -> D backward (g-stream: fake
                half only, dgrad only) -> G backward

``lax.pmean`` (xmc_gan.py:170-171,251) becomes an RCCL all-reduce of the flat gradient arena
(``dp.GradSync``), issued on a side HIP stream so that the discriminator's gradient exchange
overlaps the g-stream / generator backward; the 1/world factor is folded into the Adam kernel.
"""
from __future__ import annotations

import os

import torch

from .libml import attention_lib as attn_lib
from .libml import losses
from .nets import xmc_net
from .utils import pretrained_model_utils

_OVERLAP_PREP = os.environ.get("XMC_OVERLAP_PREP", "1") != "0"       # A/B switch for benchmarks
# train_g_d's two pullbacks only share the forward tape: run the g-stream (D dgrad on the fake half + G backward) on a
# side HIP stream beside the d-stream (D dgrad + wgrad on 2B samples) -- A/B switch
_OVERLAP_BWD = os.environ.get("XMC_OVERLAP_BWD", "1") != "0"
# weight-gradient launches of the d-stream on their own HIP stream (ops.wgrad_async): in train_d, where nothing else
# runs beside the backward pass, and in the data-parallel schedule (where the two pullbacks stay in program order so
# that D's gradient exchange can start early).  Measured on MI355X (C1, ms/step, hipGraph replay / eager):
# serial 41.0 / 41.1, overlap only 39.2 / 39.3, async everywhere 40.2 / 39.1, both 39.9 / 39.3.
_ASYNC_WGRAD_D = os.environ.get("XMC_WGRAD_ASYNC_D", "0") != "0"
# data-parallel replicas: run train_g_d's two pullbacks beside each other as on one GPU (the discriminator's exchange is
# then issued after the d-stream, the generator's buckets from the g-stream) instead of in program order -- A/B switch.
# Measured under torchrun at world size 1 with the RCCL all-reduces inside the captured graph (C1, ms/step):
# plain 42.4, program order 44.7, overlapped 43.3 (profiles/r03_bench_c1_torchrun_world1_graph*.json).
_DP_OVERLAP = os.environ.get("XMC_DP_OVERLAP", "1") != "0"
# generator forward of train_g_d issued during train_d's backward (train_step passes the next batch down) -- A/B switch
_PREFETCH_G = os.environ.get("XMC_PREFETCH_G", "1") != "0"
_PREFETCH_EARLY = os.environ.get("XMC_PREFETCH_EARLY", "1") != "0"    # ... from the end of D's trunk, beside its heads too (A/B)
# ... or beside D's optimiser update instead (A/B): the update streams 2.5 GB at the HBM rate with the matrix cores idle, the
# generator forward is matrix-bound -- two full-chip convolution streams side by side (the default) share the same units
_PREFETCH_AT_ADAM = os.environ.get("XMC_PREFETCH_AT_ADAM", "0") != "0"
# ... or from the very start of train_d, beside its own generator forward and the whole discriminator pass (A/B)
_PREFETCH_AT_START = os.environ.get("XMC_PREFETCH_AT_START", "0") != "0"
_PREP_SIDE = 1 if _PREFETCH_AT_START else 0          # (the prefetch owns side stream 0 from the start: D's preparation on stream 1)

# config.conv_fp8 on the overlapped schedule (default since the end of round 4).  Earlier in round 4 the MX-fp8 step was NOT
# run-to-run reproducible when its kernels shared the CUs with another stream's (two runs differed in the 4th digit of the losses
# with any one overlap on; serial runs were bit-identical), and the mode ran on one stream.  Found: not a stream race but one
# instruction form in the MX kernel's residual add (a crossed v_pk_add_f32; csrc/common.h) that lost the residual of
# ~0.01 % of a launch's outputs beside a weight-gradient launch.  XMC_FP8_OVERLAP=0 restores the single-stream schedule (A/B).
_FP8_OVERLAP = os.environ.get("XMC_FP8_OVERLAP", "1") != "0"


def _ovl(ops, flag):
    """is this stream overlap on for this operator table?"""
    return flag and (_FP8_OVERLAP or not getattr(ops, "fp8", False))


METRIC_KEYS = ("d_loss", "g_loss", "c_loss_d", "c_loss_g", "c_loss_g_pretrained")


def create_additional_data(config):
    """xmc_gan.py:43-55: with ``pretrained_image_contrastive`` the frozen ResNet-50 and its variables (SURVEY.md 8(f)
    N1).  ``image_model`` binds itself to the operator table on first use (the reference's Flax module needs no
    device state; this one owns folded, fragment-ordered weights in HBM).  ``config.pretrained_model_path`` names the
    ``.npy`` checkpoint (default: the reference's ``data/resnet_pretrained.npy``)."""
    additional_data = {}
    if config.get("pretrained_image_contrastive", False):
        path = config.get("pretrained_model_path", pretrained_model_utils._DEFAULT_RESNET_PATH)
        params, batch_stats = pretrained_model_utils.get_pretrained_model(checkpoint_path=path)
        state = {"params": params, "batch_stats": batch_stats}
        additional_data.update({"image_model": pretrained_model_utils.ImageModel(state), "image_model_state": state})
    return additional_data


def calculate_contrastive_loss(result_dict):
    """xmc_gan.py:58-71."""
    c_loss_d = result_dict["real_word_loss"] + result_dict["real_sentence_loss"]
    c_loss_g = (result_dict["fake_word_loss"] + result_dict["fake_sentence_loss"]
                + result_dict["image_contrastive_loss"])
    return c_loss_d, c_loss_g


def prefetch_pretrained_real(model, batch_g, ops):
    """A/B (round 6, XMC_RESNET_REAL_EARLY): the frozen ResNet-50's pass over the REAL images of the coming train_g_d depends on
    nothing the step computes -- issued at the start of train_step on its own HIP stream, it fills the CUs that train_d's launch-bound
    phases leave idle; ``_pretrained_forward`` joins that stream and runs the generated half only."""
    feats = model.bind(ops)
    src = batch_g["image"]
    with ops.side(2):
        real = ops.cast(xmc_net._to_dev(ops, src), ops.dtype).contiguous()
        out_r, _ = feats.forward(real, need_tape=False, reuse_buffers=True)
    model.early = (src, getattr(src, "_version", None), out_r)


def _pretrained_forward(model, real_images, fake_images, ops, real_src=None):
    """ResNet-50 forward of calculate_contrastive_loss_on_pretrained (xmc_gan.py:85-88) on [real; fake] -> (outputs, tape).
    ResNet-50 runs in inference mode, so the reference's two calls (real, fake) equal one call on the concatenated batch."""
    feats = model.bind(ops)
    early, model.early = getattr(model, "early", None), None
    if early is not None:
        ops.join_side((early[2],), which=2)          # (always: the stream forked inside this step / this capture)
        if early[0] is real_src and early[1] == getattr(real_src, "_version", None):
            out_f, rtape = feats.forward(fake_images.to(ops.dtype).contiguous(), need_tape=True, reuse_buffers=True)
            rtape["fake_at"] = 0
            return feats, torch.cat([early[2], out_f], dim=0), rtape
    if _RESNET_SPLIT:
        # A/B (round 6): the two halves as two passes of B images -- a block's tensors (2 x 90 MB + 2 x 22 MB at 56^2) then fit the
        # 256 MB Infinity Cache together; the real half keeps no tape, the generated half's pass overwrites its buffers
        if real_images.dim() != 4 or real_images.shape[3] != 3:
            raise ValueError("images should be of shape (H, W, 3).")
        out_r, _ = feats.forward(real_images.to(ops.dtype).contiguous(), need_tape=False, reuse_buffers=True)
        out_f, rtape = feats.forward(fake_images.to(ops.dtype).contiguous(), need_tape=True, reuse_buffers=True)
        rtape["fake_at"] = 0
        return feats, torch.cat([out_r, out_f], dim=0), rtape
    images = torch.cat([real_images, fake_images], dim=0)
    if images.dim() != 4 or images.shape[3] != 3:
        raise ValueError("images should be of shape (H, W, 3).")
    # reuse_buffers: the tape lives in buffers the next step's forward overwrites (this step's pullback has run by then)
    outputs, rtape = feats.forward(images.to(ops.dtype).contiguous(), need_tape=True, reuse_buffers=True)   # get_pretrained_embs
    return feats, outputs, rtape


def _pretrained_loss(ops, feats, outputs, rtape, b, loss_acc=None):
    """attention.contrastive_loss(real_outputs, fake_outputs) (xmc_gan.py:89) -> (loss (1,), pullback)"""
    acc = loss_acc if loss_acc is not None else ops.zeros((1,))
    tape = attn_lib.contrastive_loss_fwd(ops, outputs[:b], outputs[b:], acc)

    def pullback():
        _, dfake = attn_lib.contrastive_loss_bwd(ops, tape, want_a=False)
        lo = rtape.get("fake_at", b)                 # where the generated images sit in the pass that owns the tape
        return feats.backward(rtape, dfake, lo, lo + b)
    return acc, pullback


def calculate_contrastive_loss_on_pretrained(model, state, real_images, fake_images, ops=None, loss_acc=None):
    """xmc_gan.py:74-90 -> (loss (1,) float32, pullback).  ``pullback()`` returns d loss / d fake_images (the real
    images carry no gradient)."""
    ops = ops if ops is not None else model.ops
    feats, outputs, rtape = _pretrained_forward(model, real_images, fake_images, ops)
    return _pretrained_loss(ops, feats, outputs, rtape, real_images.shape[0], loss_acc)


def _leaf_tensors(obj):
    """every tensor inside a nest of tuples / lists / dicts"""
    return xmc_net._tensors_of(obj)


def _leaves(tree, prefix=""):
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from _leaves(v, f"{prefix}/{k}")
        else:
            yield f"{prefix}/{k}", v


def _nets(generator, discriminator):
    """``generator`` / ``discriminator`` are the partials returned by create_train_state
    (callables taking ``train=``), as in the reference."""
    return generator(train=True), discriminator(train=True)


def _flush(state):
    """Apply a deferred discriminator update (see train_d(defer_update=True))."""
    if getattr(state, "pending", None) is not None:
        state.pending()
        state = state.replace(pending=None)
    return state


_ID_KEYS = ("sentence_embedding", "embedding", "max_len", "z")


def _batch_identity(batch):
    """what a prefetched generator forward was computed from: the conditioning OBJECTS themselves (held, so their ids cannot
    be recycled) and the tensors' in-place version counters.  Storage addresses would not do: the caching allocator hands
    the address of a freed batch to the next one of the same shape."""
    return tuple((k, batch[k], getattr(batch[k], "_version", None)) for k in _ID_KEYS if k in batch)


def _same_batch(ident, batch):
    keys = tuple(k for k in _ID_KEYS if k in batch)
    return (tuple(k for k, _, _ in ident) == keys
            and all(obj is batch[k] and ver == getattr(batch[k], "_version", None) for k, obj, ver in ident))


def _generator_forward(rng, config, state, batch, g, need_tape):
    cond = {k: batch[k] for k in ("sentence_embedding", "embedding", "max_len")}
    if "z" in batch:                                                        # xmc_gan.py:132-136,225-229
        z = batch["z"]
    else:
        # ``rng`` is this half step's own stream (train_step derives one per half step, as train_utils.py:121 splits
        # the key); drawn on the device in the activation dtype.  Host-seeded: not replayable by a captured graph
        # (GraphedTrainStep rejects z-less batches).
        ops = g.ops
        b0 = torch.as_tensor(batch["sentence_embedding"]).shape[0]
        gen = torch.Generator(device=ops.device).manual_seed(int(rng))
        z = torch.randn((b0, config.z_dim), generator=gen, device=ops.device, dtype=torch.float32).to(ops.dtype)
    return g.forward(state.g_optimizer.target, state.generator_state["batch_stats"], cond, z, train=True,
                     need_tape=need_tape)


def _forward(rng, config, state, batch, g, d, need_g_tape, image_model=None, after_trunk=None, want_metrics=True):
    ops = g.ops
    cond = {k: batch[k] for k in ("sentence_embedding", "embedding", "max_len")}
    deferred = getattr(state, "pending", None) is not None
    prep_on_main = _PREFETCH_AT_ADAM and getattr(state, "prefetched_g", None) is not None and not deferred
    if prep_on_main:
        # the prefetched generator forward is still running on the side stream (it started beside D's optimiser update): D's
        # preparation reads what that update wrote -- on THIS stream, beside the rest of the forward pass
        new_sn = d.prepare(state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"])
    elif _ovl(ops, _OVERLAP_PREP) and not deferred:
        with ops.side(_PREP_SIDE):    # D's spectral-norm prep does not depend on the images: overlap it with G forward
            new_sn = d.prepare(state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"])
    else:
        new_sn = None
    pre = getattr(state, "prefetched_g", None)
    if pre is not None:
        # train_step already ran a generator forward on the side stream, beside train_d's backward: join that stream
        # before anything else touches G's state, and use the result only for the batch it was computed from
        ops.join_side()
        state = state.replace(prefetched_g=None)
        if not (need_g_tape and _same_batch(pre[0], batch)):
            pre = None
    respre_in = pre[2] if (pre is not None and len(pre) > 2) else None       # the ResNet-50 forward prefetched with it (or None)
    if pre is not None:
        img, new_g_stats, g_tape = pre[1]
    else:
        img, new_g_stats, g_tape = _generator_forward(rng, config, state, batch, g, need_g_tape)
    if deferred:
        # the previous train_d's D-gradient all-reduce ran under the generator forward above; the D parameters are
        # first needed now: finish that update, then this half step may reuse the gradient arena
        state = _flush(state)
        state.d_optimizer.arena.zero_grads()
    real = ops.cast(xmc_net._to_dev(ops, batch["image"]), ops.dtype)
    all_images = torch.cat([real, img], dim=0)                               # xmc_gan.py:140,233
    if _ovl(ops, _OVERLAP_PREP) and not deferred and not prep_on_main:
        ops.join_side(d.prepared_tensors() + [t for _, t in _leaves(new_sn)], _PREP_SIDE)
    pre = respre_in if (image_model is not None and respre_in is not None) else None
    if image_model is not None and pre is None:
        # the frozen ResNet-50's forward needs only the images: on the side stream (where its pullback will run), beside
        # the discriminator's forward below -- HBM-bound pointwise layers under MFMA-bound 3x3 convolutions
        if _ovl(ops, _OVERLAP_BWD) and hasattr(ops, "side"):
            with ops.side():
                pre = _pretrained_forward(image_model, real, img, ops, batch["image"])
        else:
            pre = _pretrained_forward(image_model, real, img, ops, batch["image"])
    logit, loss_vec, new_sn, d_tape = d.forward(state.d_optimizer.target,
                                                state.discriminator_state["spectral_norm_stats"], all_images,
                                                cond, need_tape=True, fake_losses=need_g_tape, prepared=new_sn,
                                                **({"after_trunk": after_trunk} if after_trunk is not None else {}))
    b = img.shape[0]
    hinge = ops.zeros((2,))
    dld, dlg = losses.hinge_loss(ops, logit, b, hinge[0:1], hinge[1:2])      # xmc_gan.py:144-145
    if not want_metrics:                     # train_d discards them (xmc_gan.py:238-256 returns the state only)
        out = None
    elif hasattr(ops, "loss_assemble"):      # one launch instead of ten scalar adds in front of the backward pass
        m4 = ops.loss_assemble(loss_vec, hinge)
        out = dict(d_loss=m4[0], g_loss=m4[1], c_loss_d=m4[2], c_loss_g=m4[3])
    else:
        rd = {k: loss_vec[i] for i, k in enumerate(xmc_net.LOSS_SLOTS)}
        c_loss_d, c_loss_g = calculate_contrastive_loss(rd)
        out = dict(d_loss=hinge[0] + c_loss_d, g_loss=hinge[1] + c_loss_g, c_loss_d=c_loss_d, c_loss_g=c_loss_g)
    return state, out, dld, dlg, g_tape, d_tape, new_g_stats, new_sn, pre


def _apply_adam(ops, opt, config, lr, grad_scale, ema=None, fix_args=None, net=None):
    """flax.optim.Adam.apply_gradient (+ EMA).  ``fix_args`` (Discriminator.sn_fix_args()): the gradient through sigma of the
    spectrally-normalised weights rides in the optimiser kernel -- its scalar <G, W> is formed HERE, on the gradient the
    update consumes (after the replicas' exchange: the term is linear in G, and u, v, sigma are identical on every replica).
    ``net`` (the Generator / Discriminator that owns the arena; round 5, ops.fuse_prep): the update of its batched-preparation
    weights also EMITS their prepared copies for the next forward pass (xmc_adam_wprep_tiles)."""
    a = opt.arena
    a.note_steps(1)
    decay = config.polyak_decay if ema is not None else 0.0
    if getattr(a, "first_write", False):
        a.audit_writes()                     # (first update only) every leaf of the arena was written by this half step
    prep = None
    if a.step_state is not None and getattr(ops, "fuse_opt", False):
        if net is not None and getattr(ops, "fuse_prep", False) and hasattr(net, "adam_prep"):
            prep = net.adam_prep(a)
        fix = kvec = None
        if fix_args is not None:
            mp, bank, scal, u, v = fix_args
            kvec = ops.sn_bank_dot(bank, a.params, a.grads, scal)
            fix = (prep["skip_map"] if prep is not None else mp, bank, kvec, scal, u, v)
        zero = not getattr(a, "first_write", False)
        a.grads_clean = ops.adam_ema_dev_sn(a.params, a.grads, a.m, a.v, ema, a.step_state, lr=lr, beta1=config.beta1,
                                            beta2=config.beta2, grad_scale=grad_scale, ema_decay=decay, fix=fix, zero_grads=zero,
                                            **({"skip_map": prep["skip_map"]} if prep is not None and fix is None else {}))
        if prep is not None:                 # ... and the tensors that kernel skipped: update + prepared copies in one pass
            ops.adam_wprep(prep["wp"], prep["out"], a.params, a.grads, a.m, a.v, ema, a.step_state, lr=lr, beta1=config.beta1,
                           beta2=config.beta2, grad_scale=grad_scale, ema_decay=decay, zero_grads=zero,
                           fix=(kvec, fix_args[2], fix_args[3], fix_args[4]) if fix_args is not None else None)
    elif a.step_state is not None:          # device-side step counter (hipGraph-replayable)
        ops.adam_ema_dev(a.params, a.grads, a.m, a.v, ema, a.step_state, lr=lr, beta1=config.beta1,
                         beta2=config.beta2, grad_scale=grad_scale, ema_decay=decay)
    else:
        ops.adam_ema(a.params, a.grads, a.m, a.v, ema, lr=lr, beta1=config.beta1, beta2=config.beta2,
                     step=a.opt_step, grad_scale=grad_scale, ema_decay=decay)
    a.version += 1
    if prep is not None:
        net.note_adam_prepared(a, fix_args[3] if fix_args is not None else None)


def _fix_args(d):
    return d.sn_fix_args() if hasattr(d, "sn_fix_args") else None


_RESNET_BWD_MAIN = int(os.environ.get("XMC_RESNET_BWD_MAIN", "1"))         # A/B switch (train_g_d: where the ResNet-50 pullback runs: 0 side stream, 1 main, 2 a third stream)
_RESNET_FWD_PREFETCH = os.environ.get("XMC_RESNET_FWD_PREFETCH", "0") != "0"   # A/B switch (train_d: the ResNet forward behind the prefetched G forward)
_RESNET_REAL_EARLY = os.environ.get("XMC_RESNET_REAL_EARLY", "0") != "0"   # A/B switch (train_utils.train_step -> prefetch_pretrained_real)
_RESNET_SPLIT = os.environ.get("XMC_RESNET_SPLIT", "0") != "0"        # A/B switch (_pretrained_forward)
_BUCKET_D = os.environ.get("XMC_DP_BUCKET_D", "1") != "0"             # A/B switch
_EARLY_ADAM_D = os.environ.get("XMC_EARLY_ADAM_D", "1") != "0"        # train_g_d: D's update beside G's backward pass (A/B)


def _d_bucketer(grad_sync, d_arena, fix_args):
    """Replicas: the discriminator's gradient exchange (xmc_gan.py:170,251) in slices issued from inside the backward pass --
    the two deepest blocks (86 % of the 352 MB) are final after the first few layers -- instead of in one piece after it.
    Possible only when the gradient through sigma is applied AFTER the exchange (in the optimiser kernel: ``fix_args``).
    -> (kwargs for Discriminator.backward_d, "were slices sent?")"""
    sent = []
    if grad_sync is None or fix_args is None or not _BUCKET_D or getattr(grad_sync, "exclusive", False):
        return {}, lambda: False

    def on_ready(lo, hi):
        if hi > lo:
            grad_sync.all_reduce(d_arena.grads[lo:hi], "d", append=bool(sent))
            sent.append((lo, hi))
    return {"on_ready": on_ready}, lambda: bool(sent)


def train_d(rng, state, batch, generator, discriminator, config, grad_sync=None, defer_update=False,
            next_g_batch=None, next_g_rng=None, next_image_model=None):
    """Discriminator-only half step (xmc_gan.py:194-256).  ``rng`` is unused: ``z`` comes with the
    batch (coco_dataset.py:165-166), exactly as in the reference.

    ``defer_update`` (replicas only): return with the gradient all-reduce still in flight and the Adam step
    recorded in ``state.pending``; ``train_step`` uses it so that the exchange overlaps the generator forward of
    the following ``train_g_d`` (which does not read D's parameters).

    ``next_g_batch`` (single GPU, build-side): the batch of the FOLLOWING train_g_d.  Its generator forward depends
    on nothing this half step changes (G's parameters and running statistics are untouched: xmc_gan.py:231), so it
    is issued on the side HIP stream beside the discriminator backward below and handed over in
    ``state.prefetched_g`` together with the identity of the batch it was computed from (``train_g_d`` recomputes the
    forward when it is called with a different batch); ``next_g_rng`` is that half step's rng."""
    g, d = _nets(generator, discriminator)
    next_g_rng = rng if next_g_rng is None else next_g_rng
    ops = g.ops
    if hasattr(ops, "begin_pool"):
        ops.begin_pool()
    d_arena = state.d_optimizer.arena
    deferred_in = getattr(state, "pending", None) is not None
    if not deferred_in:
        d_arena.zero_grads()
    prefetched = None
    do_prefetch = next_g_batch is not None and grad_sync is None and _ovl(ops, _PREFETCH_G) and hasattr(ops, "side")
    state_in = state

    def prefetch():
        nonlocal prefetched
        with ops.side():
            fwd = _generator_forward(next_g_rng, config, state_in, next_g_batch, g, True)
            respre = None
            if next_image_model is not None and _RESNET_FWD_PREFETCH:
                # A/B (round 6): the frozen ResNet-50's forward of the coming train_g_d needs only that half step's real images and
                # the images just generated: behind the prefetched generator forward on the side stream, beside this half step's
                # backward pass and optimiser update -- train_g_d's discriminator forward then has the chip to itself
                real = ops.cast(xmc_net._to_dev(ops, next_g_batch["image"]), ops.dtype)
                respre = _pretrained_forward(next_image_model, real, fwd[0], ops, next_g_batch["image"])
            prefetched = (_batch_identity(next_g_batch), fwd, respre)
    # round 5 (_PREFETCH_EARLY): the prefetched forward starts when the discriminator's TRUNK is done, not after its heads
    early = do_prefetch and _PREFETCH_EARLY and not deferred_in and not _PREFETCH_AT_ADAM
    if do_prefetch and _PREFETCH_AT_START and not deferred_in:
        prefetch()
        early = False
    state, out, dld, _, _, d_tape, _new_g_stats, new_sn, _ = _forward(rng, config, state, batch, g, d, need_g_tape=False,
                                                                      after_trunk=prefetch if early else None, want_metrics=False)
    keep_async = getattr(ops, "wgrad_async", False)
    if (_ASYNC_WGRAD_D or grad_sync is not None) and hasattr(ops, "wgrad_async"):
        ops.wgrad_async = True
    if do_prefetch and prefetched is None and not _PREFETCH_AT_ADAM:
        prefetch()
    fix_args = _fix_args(d)                  # u, v, sigma of THIS half step's forward (a deferred update runs after the next prepare)
    d_ready, d_sent = _d_bucketer(grad_sync, d_arena, fix_args)
    d.backward_d(d_tape, dld, **d_ready)
    if hasattr(ops, "wgrad_async"):
        ops.wgrad_async = keep_async
    scale = 1.0
    if grad_sync is not None:
        if getattr(grad_sync, "exclusive", False) and hasattr(ops, "join_wgrad"):
            ops.join_wgrad()                 # every producer of the arena has finished before the exchange starts
        scale = 1.0 / grad_sync.world if d_sent() else grad_sync.all_reduce(d_arena.grads, "d")   # lax.pmean, xmc_gan.py:251
        if defer_update and not getattr(grad_sync, "exclusive", False):
            opt = state.d_optimizer

            def finish():
                grad_sync.wait("d")
                _apply_adam(ops, opt, config, config.d_lr, scale, fix_args=fix_args, net=d)
            return state.replace(discriminator_state={"spectral_norm_stats": new_sn}, pending=finish)
        grad_sync.wait("d")
    if do_prefetch and prefetched is None:
        prefetch()                           # (_PREFETCH_AT_ADAM: the side stream forks here, behind the backward pass)
    _apply_adam(ops, state.d_optimizer, config, config.d_lr, scale, fix_args=fix_args, net=d)
    # G's new batch_stats are discarded (xmc_gan.py:231); D's new u0 are kept (:253-255)
    return state.replace(discriminator_state={"spectral_norm_stats": new_sn}, prefetched_g=prefetched)


def train_g_d(rng, state, batch, generator, discriminator, config, additional_data, grad_sync=None):
    """Generator + discriminator half step (xmc_gan.py:93-191)."""
    g, d = _nets(generator, discriminator)
    ops = g.ops
    if hasattr(ops, "begin_pool"):
        ops.begin_pool()
    d_arena, g_arena = state.d_optimizer.arena, state.g_optimizer.arena
    if getattr(state, "pending", None) is None:
        d_arena.zero_grads()                 # (with a deferred D update the arena is still being exchanged: _forward)
    g_arena.zero_grads()
    image_model = additional_data["image_model"] if config.get("pretrained_image_contrastive", False) else None
    state, out, dld, dlg, g_tape, d_tape, new_g_stats, new_sn, pre = _forward(rng, config, state, batch, g, d,
                                                                              need_g_tape=True, image_model=image_model)
    b = g_tape["b"]
    d_scale = g_scale = 1.0
    c_pre = None

    def image_pullback(dlg_fake):
        """pullback (0, 1) down to the generated images: the discriminator's g-stream plus, with
        ``pretrained_image_contrastive``, the frozen ResNet-50 term (xmc_gan.py:148-154)"""
        nonlocal c_pre
        dimg = d.backward_g(d_tape, dlg_fake)
        if pre is not None:
            c_pre, pull = _pretrained_loss(ops, *pre, b)
            ops.add_into(dimg, pull())
        return dimg
    excl = grad_sync is not None and getattr(grad_sync, "exclusive", False)
    if _ovl(ops, _OVERLAP_BWD) and (grad_sync is None or _DP_OVERLAP or excl) and hasattr(ops, "side"):
        dlg_f = dlg[b:].contiguous()
        on_ready = None
        if grad_sync is not None and not excl:    # G's exchange (xmc_gan.py:171) in three buckets, issued from the g-stream
            g_scale = 1.0 / grad_sync.world
            on_ready = lambda lo, hi: grad_sync.all_reduce(g_arena.grads[lo:hi], "g", append=True)
        async_wg, ops.wgrad_async = ops.wgrad_async, False     # the g-stream's weight gradients stay on its own stream
        d_part_done = None
        if pre is not None and _RESNET_BWD_MAIN and hasattr(ops, "record_event"):
            # A/B (round 6): the frozen ResNet-50's pullback on the MAIN stream, in front of D's backward pass and beside D's g-stream
            # on the side stream -- so that the two long MFMA-bound passes (D's backward: 4.8 TFLOP, G's: 4.6) then run side by side
            # for their whole length instead of [g-stream + ResNet + G] against [D] alone
            ops.join_side(_leaf_tensors(pre[2]) + [pre[1]])                  # the ResNet forward ran on the side stream (_forward)
            with ops.side():
                dimg = d.backward_g(d_tape, dlg_f)                           # pullback (0, 1), D part
                if grad_sync is None and _EARLY_ADAM_D:
                    d_part_done = ops.record_event()
            if _RESNET_BWD_MAIN == 2:                                        # A/B: on a THIRD stream, beside D's backward pass as well
                with ops.side(2):
                    c_pre, pull = _pretrained_loss(ops, *pre, b)
                    dres = pull()
                    res_done = ops.record_event()
            else:
                c_pre, pull = _pretrained_loss(ops, *pre, b)
                dres = pull()                                                #                  ResNet part, main stream
                res_done = ops.record_event()
            side = ops._sides[0]
            dres.record_stream(side)
            with torch.cuda.stream(side):                                    # (no wait for the main stream: only for the event)
                ops.wait_event(res_done)
                ops.add_into(dimg, dres)
                g.backward(g_tape, dimg, on_ready)                           #                  G part
        else:
            with ops.side():                                   # (stream graph main -> {side, wgrad}: no cross edges)
                dimg = image_pullback(dlg_f)                                 # pullback (0, 1), D (+ ResNet) part
                if grad_sync is None and _EARLY_ADAM_D and hasattr(ops, "record_event"):
                    d_part_done = ops.record_event()                         # the g-stream is done with D's parameters here
                g.backward(g_tape, dimg, on_ready)                           #                  G part
        ops.wgrad_async = async_wg
        d_ready, d_sent = _d_bucketer(grad_sync, d_arena, _fix_args(d))
        d.backward_d(d_tape, dld, **d_ready)                                 # pullback (1, 0), beside it
        if grad_sync is not None and not excl:
            d_scale = 1.0 / grad_sync.world if d_sent() else grad_sync.all_reduce(d_arena.grads, "d")   # lax.pmean, xmc_gan.py:170
        d_updated = False
        if d_part_done is not None:
            # D's optimiser step (HBM-bound: 0.6 ms of Adam + the <G, W> pass) needs only D's finished gradients and nobody
            # reading D's parameters any more: it runs HERE, beside the generator's backward pass on the side stream (MFMA-bound),
            # instead of after the join
            ops.wait_event(d_part_done)
            _apply_adam(ops, state.d_optimizer, config, config.d_lr, d_scale, fix_args=_fix_args(d), net=d)
            d_updated = True
        ops.join_side()
        if excl:
            # exclusive schedule: both pullbacks (the single-GPU overlap of the two backward passes is kept) have finished --
            # now the two exchanges, each arena in one piece, with nothing else on the GPU; both are waited for before D's update
            if hasattr(ops, "join_wgrad"):
                ops.join_wgrad()
            d_scale = grad_sync.all_reduce(d_arena.grads, "d")               # lax.pmean, xmc_gan.py:170
            g_scale = grad_sync.all_reduce(g_arena.grads, "g")               #            xmc_gan.py:171
            grad_sync.wait("d")
            grad_sync.wait("g")
        return _finish_g_d(ops, state, config, out, c_pre, new_g_stats, new_sn, d_scale, g_scale, grad_sync, _fix_args(d),
                           d_updated=d_updated, nets=(g, d))
    keep_async = getattr(ops, "wgrad_async", False)
    if grad_sync is not None and hasattr(ops, "wgrad_async"):
        ops.wgrad_async = True               # data-parallel schedule: weight gradients beside the dgrad chain
    d_ready, d_sent = _d_bucketer(grad_sync, d_arena, _fix_args(d))
    d.backward_d(d_tape, dld, **d_ready)                                     # pullback (1, 0)
    if grad_sync is not None and not excl:
        d_scale = 1.0 / grad_sync.world if d_sent() else grad_sync.all_reduce(d_arena.grads, "d")   # overlaps the g-stream below
    if pre is not None and getattr(ops, "_side", None) is not None:
        ops.join_side()                                                      # the ResNet-50 forward _forward put on the side stream
    dimg = image_pullback(dlg[b:].contiguous())                              # pullback (0, 1), D (+ ResNet) part
    on_ready = None
    if grad_sync is not None and not excl:
        # G's gradient exchange (xmc_gan.py:171) in three buckets, each issued the moment the backward pass has
        # finished its slice of the arena: only the last bucket (GenBlock_0 + the input denses) is exposed
        g_scale = 1.0 / grad_sync.world
        on_ready = lambda lo, hi: grad_sync.all_reduce(g_arena.grads[lo:hi], "g", append=True)
    g.backward(g_tape, dimg, on_ready)                                       #                  G part
    if hasattr(ops, "wgrad_async"):
        ops.wgrad_async = keep_async
    if excl:                                 # exclusive schedule: both exchanges after both pullbacks, waited for before the updates
        if hasattr(ops, "join_wgrad"):
            ops.join_wgrad()
        d_scale = grad_sync.all_reduce(d_arena.grads, "d")
        g_scale = grad_sync.all_reduce(g_arena.grads, "g")
        grad_sync.wait("d")
        grad_sync.wait("g")
    return _finish_g_d(ops, state, config, out, c_pre, new_g_stats, new_sn, d_scale, g_scale, grad_sync, _fix_args(d), nets=(g, d))


def _finish_g_d(ops, state, config, out, c_pre, new_g_stats, new_sn, d_scale, g_scale, grad_sync, fix_args=None, d_updated=False,
                nets=(None, None)):
    """Optimiser updates, EMA, new state and metrics of train_g_d (xmc_gan.py:170-190).  ``d_updated``: the caller already
    applied D's update (beside the generator's backward pass)."""
    if grad_sync is not None:
        grad_sync.wait("d")
    if not d_updated:
        _apply_adam(ops, state.d_optimizer, config, config.d_lr, d_scale, fix_args=fix_args, net=nets[1])
    if grad_sync is not None:
        grad_sync.wait("g")
    ema = state.ema_buffer if config.get("ema", True) else None
    _apply_adam(ops, state.g_optimizer, config, config.g_lr, g_scale, ema, net=nets[0])  # + EMA, xmc_gan.py:174-177
    new_state = state.replace(step=state.step + 1,
                              generator_state={"batch_stats": new_g_stats},
                              discriminator_state={"spectral_norm_stats": new_sn})
    metrics = dict(out)
    if c_pre is not None:                                                    # xmc_gan.py:149-156
        metrics["c_loss_g_pretrained"] = c_pre[0]
        metrics["g_loss"] = metrics["g_loss"] + c_pre[0]
    else:
        metrics["c_loss_g_pretrained"] = torch.zeros((), device=out["d_loss"].device)
    if grad_sync is not None:                # TrainMetrics.gather_from_model_output (xmc_gan.py:185-190): mean over replicas
        metrics = grad_sync.mean_metrics(metrics)
    return new_state, metrics
