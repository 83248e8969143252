"""XMC-GAN generator and discriminator on the HIP operator table.

Keeps the module API of the reference's ``xmcgan/nets/xmc_net.py`` -- ``Generator`` (:145-248)
and ``Discriminator`` (:28-142) with ``.init(rng, inputs)`` / ``.apply(variables, inputs,
mutable=...)``, the same parameter-tree names, state collections (``batch_stats``,
``spectral_norm_stats``) and the 15-key statistics dict -- and adds the explicit
``forward`` / ``backward`` pair that ``xmc_gan.train_d`` / ``train_g_d`` drive instead of
``jax.vjp``.
"""
from __future__ import annotations

import torch

from .. import synthetic as syn
from ..libml import attention_lib as attn_lib
from ..libml.layers import (ConvSite, DenseSite, FlatTree, ParamArena, ParamTree, flat_running_stats,
                            prefill_running_stats, tree_get, tree_set)
from . import common

_OPS_FACTORY = None
_XC_REAL_HALF = __import__("os").environ.get("XMC_XC_REAL_HALF", "1") != "0"       # A/B switch (Discriminator.backward_d)
# A/B switch (Discriminator.forward).  Measured (profiles/r05_ab_heads_2stream.txt): G/D-only 25.30 / 25.20 -> 25.15 / 25.19 ms, the
# default workload 30.01 / 29.96 -> 30.09 / 30.08 (the frozen ResNet-50's forward already shares the chip there): level -> OFF
_SCOND_DIRECT = __import__("os").environ.get("XMC_SCOND_DIRECT", "1") != "0"      # A/B switch (Generator.forward / backward)
_HEADS_2STREAM = __import__("os").environ.get("XMC_HEADS_2STREAM", "0") != "0"


def _tensors_of(obj, out=None):
    """every tensor inside a nest of tuples / lists / dicts (a loss tape): for ops.join_side's record_stream"""
    out = [] if out is None else out
    if torch.is_tensor(obj):
        out.append(obj)
    elif isinstance(obj, dict):
        for v in obj.values():
            _tensors_of(v, out)
    elif isinstance(obj, (tuple, list)):
        for v in obj:
            _tensors_of(v, out)
    return out


def set_ops_factory(fn):
    """Install the operator-table factory ``fn(dtype) -> ops`` (tests inject a CPU mock here;
    the product default builds ``HipOps`` and fails loudly without the HIP library / a GPU)."""
    global _OPS_FACTORY
    _OPS_FACTORY = fn


def make_ops(dtype):
    if _OPS_FACTORY is not None:
        return _OPS_FACTORY(dtype)
    from ..ops import HipOps
    return HipOps(dtype=dtype)


def _to_dev(ops, t, dtype=torch.float32):
    t = torch.as_tensor(t)
    return t.to(device=ops.device, dtype=dtype).contiguous()


def _tree_to_dev(ops, tree):
    return syn.tree_map(lambda a: _to_dev(ops, a), tree)


class SnTree(FlatTree):
    """spectral_norm_stats tree whose ``u0`` leaves are views of ONE flat buffer (``.flat``), in the order
    of the discriminator's spectral bank -- the next power iteration consumes it without a gather."""


def check_config(config):
    """Reject the hyper-parameters of the reference's config (xmcgan/configs/coco_xmc.py) that this build does not
    implement instead of silently ignoring them.  ``word_contrastive`` / ``sentence_contrastive`` /
    ``image_contrastive`` (xmc_net.py:105-125) are honoured by the discriminator."""
    if config.get("g_spectral_norm", False):
        raise ValueError("g_spectral_norm=True (xmc_net.py:170-191) is not supported: the generator uses plain "
                         "nn.Conv / nn.Dense as in every reference config")
    if not config.get("d_spectral_norm", True):
        raise ValueError("d_spectral_norm=False (xmc_net.py:74-80) is not supported: the discriminator is built on "
                         "SpectralConv / SpectralDense as in every reference config")
    if config.get("batch_norm_group_size", -1) > 0:
        raise ValueError("batch_norm_group_size > 0 (cross-replica BatchNorm groups, xmc_net.py:197-200 / "
                         "utils/device_utils.py:18-26) is not supported: BatchNorm statistics are per replica")
    if config.get("image_size") not in (128, 256):
        raise ValueError("image_size must be 128 or 256 (channel_dims are only defined for those, xmc_net.py:81-86,202-205)")
    if config.get("architecture", "xmc_net") != "xmc_net":
        raise ValueError(f"Architecture {config.get('architecture')} is not supported.")


class _Net:
    def __init__(self, config, train, dtype=torch.float32, activation_fn=None, ops=None):
        check_config(config)
        self.config, self.train, self.dtype = config, train, dtype
        self.ops = ops if ops is not None else make_ops(dtype)
        self._arena = None

    def _bind(self, params):
        """Resolve the arena behind a parameter tree (copying a foreign tree into a fresh one)."""
        arena = getattr(params, "arena", None)
        if arena is None:
            buf = getattr(params, "buffer", None)
            if self._arena is None:
                self._arena = ParamArena(self.ops, self.shapes()[0], with_opt=False)
            if buf is not None and buf.numel() == self._arena.size:
                self._arena.params.copy_(buf)
                self._arena.version += 1
            else:
                self._arena.load_flax(params)
            arena = self._arena
        if getattr(self, "_built_for", None) is not arena:
            self._build(arena)
            self._built_for = arena
        return arena


# =================================================================================== generator
class Generator(_Net):
    """Generator network (reference xmc_net.py:145-248)."""

    def shapes(self):
        return syn.generator_shapes(self.config)

    def init(self, rng, inputs):
        """-> {"params": ParamTree, "batch_stats": tree}; ``rng`` is an integer seed."""
        params, stats = syn.init_generator(self.config, seed=int(rng))
        arena = ParamArena(self.ops, self.shapes()[0])
        arena.load_flax(params)
        return {"params": arena.tree(), "batch_stats": _tree_to_dev(self.ops, stats)}

    def _build(self, arena):
        ops, cfg = self.ops, self.config
        chans = syn.G_CHANNELS[cfg["image_size"]]
        self.d0 = DenseSite(ops, arena, "Dense_0")
        self.d1 = DenseSite(ops, arena, "Dense_1")
        self.gblocks = [common.GenBlock(ops, arena, f"GenBlock_{i}", local=False) for i in range(2)]
        self.xcond = ConvSite(ops, arena, "Conv_0")
        self.sblocks = [common.GenBlock(ops, arena, f"GenSpatialBlock_{i - 2}", local=True)
                        for i in range(2, len(chans))]
        self.fnorm = common.CondNorm(ops, arena, "LocalConditionalBatchNorm_0", local=True)
        self.rgb = ConvSite(ops, arena, "Conv_1")
        self.local_gb = common.FusedLocalGB(ops, arena, [n for blk in self.sblocks for n in (blk.n0, blk.n1)] + [self.fnorm])
        self.global_gb = common.FusedGlobalGB(ops, arena, [n for blk in self.gblocks for n in (blk.n0, blk.n1)])
        # one batched pass prepares every packable convolution weight (ops.wprep_*: fragment-ordered copies + the 16-tap phase
        # copies of the upsampling layers) whenever the parameters changed -- ~20 prep launches per step before
        self.wp, self._wp_ver = None, -1
        self._wp_out = self._skip_map = None          # fuse_prep: persistent copy buffers; the flat optimiser kernel's skip map
        if getattr(ops, "fold_sigma", False):
            self.wp_sites = [s for blk in self.gblocks + self.sblocks for s in (blk.c0, blk.c1, blk.c2)] + [self.xcond]
            self.wp_sites = [s for s in self.wp_sites if s.cout % 32 == 0 and s.cin % 32 == 0 and ops._packable(s.taps, s.cin)
                             and ops._packable(s.taps, s.cout)]
            if self.wp_sites:
                self.wp = ops.wprep_create([dict(w_off=arena.offset(s.path + "/kernel"), cout=s.cout, cin=s.cin, taps=s.taps,
                                                 phase=s.phase, spectral=False) for s in self.wp_sites])

    def bn_sites(self):
        return [n.bn for blk in self.gblocks + self.sblocks for n in (blk.n0, blk.n1)] + [self.fnorm.bn]

    # ---- round 5 (ops.fuse_prep): the optimiser kernel emits the prepared weights
    def _wp_persistent(self):
        """persistent (bufs, part) of the batched preparation -- written by ``wprep_run`` or by the optimiser kernel"""
        if not getattr(self.ops, "fuse_prep", False) or self.wp is None:
            return None
        if self._wp_out is None:
            self._wp_out = self.ops.wprep_alloc(self.wp)
        return self._wp_out

    def adam_prep(self, arena):
        """-> dict(wp, out, skip_map) when this network's prepared weights can be emitted by the optimiser update of ``arena``
        (xmc_gan._apply_adam), else None"""
        if (not getattr(self.ops, "fuse_prep", False) or self.wp is None or getattr(self, "_built_for", None) is not arena
                or self._wp_ver != arena.version):
            return None
        if self._skip_map is None:
            self._skip_map = self.ops.wprep_skip_map(self.wp, arena.size)
        return dict(wp=self.wp, out=self._wp_persistent(), skip_map=self._skip_map)

    def note_adam_prepared(self, arena, u_new=None):
        """the optimiser wrote the copies of the updated parameters (``arena.version`` already advanced)"""
        self._wp_ver = arena.version
        for st in self.wp_sites:
            st._ver = arena.version

    def refresh_prepared(self, params):
        """re-prepare from the masters NOW (eager): parameters were changed behind a captured graph's back"""
        arena = self._bind(params)
        if self.wp is not None and getattr(self.ops, "fuse_prep", False):
            self.ops.wprep_run(self.wp, arena.params, out=self._wp_persistent())
            self.note_adam_prepared(arena)

    def flat_batch_stats(self, params, batch_stats):
        """``batch_stats`` as a FlatTree (leaves = views of one buffer in BatchNorm-site order); idempotent."""
        if getattr(batch_stats, "flat", None) is not None:
            return batch_stats
        self._bind(params)
        return flat_running_stats(self.bn_sites(), _tree_to_dev(self.ops, batch_stats))

    # ------------------------------------------------------------------------------- forward
    def forward(self, params, batch_stats, cond_dict, z, *, train, need_tape):
        """-> (image (B, H, W, 3) in [0,1], new_batch_stats, tape or None)."""
        ops, cfg = self.ops, self.config
        arena = self._bind(params)
        sent = _to_dev(ops, cond_dict["sentence_embedding"])
        words = _to_dev(ops, cond_dict["embedding"])
        max_len = _to_dev(ops, cond_dict["max_len"])
        z = _to_dev(ops, z)
        b = z.shape[0]
        new_stats = FlatTree()
        if train:
            prefill_running_stats(self.bn_sites(), batch_stats, new_stats)
        if self.wp is not None and self._wp_ver != arena.version:
            # (fuse_prep: the optimiser kernel wrote the copies of the CURRENT parameters into the persistent buffers and
            # advanced _wp_ver -- adam_prep / note_adam_prepared -- so a training step never gets here after its first forward)
            bufs, _ = ops.wprep_run(self.wp, arena.params, out=self._wp_persistent())
            for k, st in enumerate(self.wp_sites):
                st.set_prepared(*ops.wprep_weights(self.wp, k, bufs), None, None, None, folded=True)
            self._wp_ver = arena.version
        for blk in self.gblocks + self.sblocks:
            blk.prepare()
        self.xcond.prepare()
        self.fnorm.prepare()
        self.rgb.prepare()
        self.local_gb.prepare()

        gs = self.d0.fwd(sent)                                              # xmc_net.py:213
        gcond = torch.cat([gs, z], dim=1)                                   # :214
        x = ops.cast(self.d1.fwd(z).view(b, 4, 4, -1), ops.dtype)           # :215-216
        ggb = self.global_gb.fwd(gcond) if self.global_gb.ok else None      # gamma | beta of the four global cBN sites, one product
        tapes = []
        for blk in self.gblocks:                                            # :217-219
            x, t = blk.fwd(x, gcond, batch_stats, new_stats, train)
            tapes.append(t)
        x16 = x
        xc = self.xcond.fwd(x16)                                            # :220
        ss = xc.shape[1]
        words_n = attn_lib.normalize_words(ops, words)
        region = xc.view(b, ss * ss, -1)
        e = region.shape[-1]
        if hasattr(ops, "attn_g_sliced") and ops.attn_g_sliced(region, words_n.shape[1]) and _SCOND_DIRECT:
            # round 5: the attention kernel writes its context straight into the spatial condition's first E channels (row pitch
            # E + 2 z_dim) and one strided copy broadcasts the global condition into the rest -- no concatenation pass
            scond = ops.empty((b, ss, ss, e + gcond.shape[1]))
            s3 = scond.view(b, ss * ss, -1)
            _, atape = attn_lib.attention_for_g_fwd(ops, region, words_n, max_len, float(cfg["gamma_for_g"]), ctx_out=s3[..., :e])
            s3[..., e:].copy_(gcond.view(b, 1, -1).expand(-1, ss * ss, -1))        # (float32 -> activation dtype in the copy)
        else:
            ctx, atape = attn_lib.attention_for_g_fwd(ops, region, words_n, max_len, float(cfg["gamma_for_g"]))     # :225-229
            scond = torch.cat([ctx.view(b, ss, ss, -1),
                               ops.cast(gcond, ops.dtype).view(b, 1, 1, -1).expand(-1, ss, ss, -1)],
                              dim=-1).contiguous()                          # :231-235
        gball = self.local_gb.fwd(scond)                                    # gamma / beta of all local cBN sites
        for blk in self.sblocks:                                            # :236-241
            x, t = blk.fwd(x, scond, batch_stats, new_stats, train)
            tapes.append(t)
        a, ftape = self.fnorm.fwd(x, scond, batch_stats, new_stats, train)  # :242-244
        pre = self.rgb.fwd(a)                                               # :245
        img = ops.tanh_out_fwd(pre)                                         # :246-247
        tape = None
        if need_tape:
            tape = dict(sent=sent, z=z, gcond=gcond, x16=x16, tapes=tapes, atape=atape, scond=scond, a=a,
                        ftape=ftape, img=img, b=b, ss=ss, attn=atape[2], gball=gball, ggb=ggb)
        self.last_attn = atape[2]
        return img, new_stats, tape

    # ------------------------------------------------------------------------------ backward
    def backward(self, tape, dimg, on_ready=None):
        """Accumulates d g_loss / d params into the arena's gradient buffer.

        ``on_ready(lo, hi)`` (optional) is called as soon as the gradient slice [lo, hi) of the flat arena is final,
        latest layers first: the data-parallel exchange of that bucket (xmc_gan.py:171 ``lax.pmean(g_grad)``) then
        runs under the rest of the backward pass.  Buckets follow the arena order (= parameter-tree order):
        [spatial blocks .. RGB conv] after the last local-cBN projection gradient, [GenBlock_1, x_cond conv] after
        GenBlock_1, [Dense_0, Dense_1, GenBlock_0] at the end."""
        ops = self.ops
        arena = self.d0.arena
        cut_sp, cut_b1 = arena.prefix_offset("GenSpatialBlock_0"), arena.prefix_offset("GenBlock_1")
        b, ss = tape["b"], tape["ss"]
        dpre = ops.tanh_out_bwd(dimg, tape["img"])
        self.rgb.wgrad_rgb_out(tape["a"], dpre)
        da = self.rgb.dgrad_rgb_out(dpre)
        self.local_gb.begin_bwd(tape["gball"])
        dx, _ = self.fnorm.bwd(tape["ftape"], da, None)
        nsb = len(self.sblocks)
        for k in range(nsb - 1, -1, -1):
            dx, _ = self.sblocks[k].bwd(tape["tapes"][2 + k], dx, None)
        dscond = self.local_gb.bwd(tape["scond"])                           # d(spatial condition) of all local sites
        if on_ready is not None:
            ops.join_wgrad()
            on_ready(cut_sp, arena.size)
        e = tape["atape"][0].shape[-1]
        dctx = dscond.view(b, ss * ss, -1)[..., :e]                         # a column slice: the MFMA kernel takes the row pitch
        if not (hasattr(ops, "attn_g_sliced") and ops.attn_g_sliced(tape["atape"][0], tape["atape"][1].shape[1]) and _SCOND_DIRECT):
            dctx = dctx.contiguous()
        dgc_sp = ops.reduce_mid(dscond.view(b, ss * ss, -1)[..., e:].contiguous())      # (B, 2*z_dim)
        dxc = attn_lib.attention_for_g_bwd(ops, tape["atape"], dctx).view(b, ss, ss, e)
        self.xcond.wgrad(tape["x16"], dxc)
        dx = self.xcond.dgrad(dxc, res=dx)
        dgcond = dgc_sp
        if self.global_gb.ok:
            self.global_gb.begin_bwd(tape["ggb"])
        for k in (1, 0):
            dx, dgcond = self.gblocks[k].bwd(tape["tapes"][k], dx, dgcond)
            if k == 1 and on_ready is not None:
                ops.join_wgrad()
                on_ready(cut_b1, cut_sp)
        if self.global_gb.ok:
            dgcond = self.global_gb.bwd(tape["gcond"], dgcond)              # (in place: + the four global sites' share)
        self.d1.bwd(tape["z"], ops.cast(dx, torch.float32).view(b, -1), need_dx=False)
        zd = tape["z"].shape[1]
        self.d0.bwd(tape["sent"], dgcond[:, :zd].contiguous(), need_dx=False)
        ops.join_wgrad()
        if on_ready is not None:
            on_ready(0, cut_b1)

    # ----------------------------------------------------------------------------- flax-style
    def apply(self, variables, inputs, mutable=False):
        cond_dict, z = inputs
        img, new_stats, _ = self.forward(variables["params"], variables.get("batch_stats"), cond_dict, z,
                                         train=self.train, need_tape=False)
        if mutable:
            return img, {"batch_stats": new_stats}
        return img


# =============================================================================== discriminator
class Discriminator(_Net):
    """Discriminator network (reference xmc_net.py:28-142)."""

    def shapes(self):
        return syn.discriminator_shapes(self.config)

    def init(self, rng, inputs):
        params, sn = syn.init_discriminator(self.config, seed=int(rng))
        arena = ParamArena(self.ops, self.shapes()[0])
        arena.load_flax(params)
        return {"params": arena.tree(), "spectral_norm_stats": _tree_to_dev(self.ops, sn)}

    def _build(self, arena):
        ops, cfg = self.ops, self.config
        df = cfg["df_dim"]
        chans, downs = syn.D_CHANNELS[cfg["image_size"]]
        self.b0 = common.DiscOptimizedBlock(ops, arena, "DiscOptimizedBlock_0")
        self.blocks = []
        cin, res = df, cfg["image_size"] // 2
        self.cond_idx = None
        for i, (c, d) in enumerate(zip(chans, downs)):
            self.blocks.append(common.DiscBlock(ops, arena, f"DiscBlock_{i}", cin, df * c, d))
            cin = df * c
            if d:
                res //= 2
            if res == cfg["cond_size"]:
                self.cond_idx = i
        self.sd0 = DenseSite(ops, arena, "SpectralDense_0", spectral=True)
        self.sd1 = DenseSite(ops, arena, "SpectralDense_1", spectral=True)
        # the three contrastive heads are optional (xmc_net.py:105-125); the x_cond 1x1 conv only exists with the word head
        self.use_word = bool(cfg.get("word_contrastive", True))
        self.use_sent = bool(cfg.get("sentence_contrastive", True))
        self.use_img = bool(cfg.get("image_contrastive", True))
        self.xc = ConvSite(ops, arena, "SpectralConv_0", spectral=True) if self.use_word else None
        self.conv_sites = list(self.b0.sites)
        for blk in self.blocks:
            self.conv_sites += blk.sites
        if self.use_word:
            self.conv_sites.append(self.xc)
        # one descriptor table over every spectrally-normalised weight: the power iteration, the
        # W / sigma weight copies and the gradient through sigma each run as a few batched launches
        self.sn_sites = self.conv_sites + [self.sd0, self.sd1]
        entries = []
        for s in self.sn_sites:
            is_conv = isinstance(s, ConvSite)
            rows, cols = (s.cout, s.taps * s.cin) if is_conv else tuple(s.w.shape)
            entries.append(dict(w_off=arena.offset(s.path + "/kernel"), rows=rows, cols=cols,
                                u_axis=0 if is_conv else 1, taps=s.taps if is_conv else 1, is_conv=is_conv,
                                phase=getattr(s, "phase", None)))
        self.bank = ops.sn_bank_create(entries)
        # fold_sigma (ops): the packable convolution weights are prepared by ONE batched pass that also produces the first
        # product of their power iteration (ops.wprep_*); the rest (dense kernels, the two 3-channel convolutions) keep the
        # per-kind kernels through a sub-bank that shares the full bank's u / v slices
        self.wp = self.irr = None
        if getattr(ops, "fold_sigma", False):
            reg, irr = [], []
            for i, (s, e) in enumerate(zip(self.sn_sites, self.bank["entries"])):
                if e["is_conv"] and s.cout % 32 == 0 and s.cin % 32 == 0 and ops._packable(s.taps, s.cin) and ops._packable(s.taps, s.cout):
                    reg.append(dict(w_off=e["w_off"], cout=s.cout, cin=s.cin, taps=s.taps, phase=getattr(s, "phase", None),
                                    spectral=True, u_off=e["u_off"], v_off=e["v_off"], site=i))
                else:
                    irr.append(dict(e, site=i))
            if reg:
                self.wp = ops.wprep_create(reg)
                self.irr = ops.sn_bank_create(irr, keep_uv=True) if irr else None
                self._ones_scal = torch.ones((2 * max(len(irr), 1),), dtype=torch.float32, device=ops.device)
                self.sn_map = ops.sn_bank_map(self.bank, arena.size) if getattr(ops, "fuse_opt", False) else None
        self._wp_out = self._skip_map = self._wp_fresh = None     # fuse_prep (round 5): see Generator
        # the bucketed gradient exchange (backward_d's on_ready slices) relies on the arena following the parameter tree:
        # [DiscOptimizedBlock_0, DiscBlock_0 .. n, SpectralDense_0, SpectralDense_1, (SpectralConv_0)] -- a tree in another order
        # would all-reduce slices whose gradients are not final (round-4 advisor finding): checked here, once per build
        order = ["DiscOptimizedBlock_0"] + [f"DiscBlock_{i}" for i in range(len(self.blocks))] + ["SpectralDense_0", "SpectralDense_1"]
        if self.use_word:
            order.append("SpectralConv_0")
        offs = [arena.prefix_offset(k) for k in order]
        self.bucket_order_ok = all(a < b for a, b in zip(offs, offs[1:])) and offs[0] == 0

    def _pack_u0(self, sn_stats):
        """u0 of every spectral site gathered into the bank's flat layout (slices start 16-byte aligned)."""
        flat = torch.zeros((self.bank["nu"],), dtype=torch.float32, device=self.ops.device)
        for s, e in zip(self.sn_sites, self.bank["entries"]):
            flat[e["u_off"]:e["u_off"] + e["nu"]].copy_(_to_dev(self.ops, tree_get(sn_stats, s.path)["u0"]).reshape(-1))
        return flat

    def flat_sn_stats(self, params, sn_stats):
        """``spectral_norm_stats`` as an SnTree (``u0`` leaves = views of one buffer in spectral-bank order); idempotent."""
        if getattr(sn_stats, "flat", None) is not None:
            return sn_stats
        self._bind(params)
        flat = self._pack_u0(sn_stats)
        out = SnTree()
        out.flat = flat
        for s, e in zip(self.sn_sites, self.bank["entries"]):
            tree_set(out, s.path, {"u0": flat[e["u_off"]:e["u_off"] + e["nu"]].view(1, -1)})
        return out

    def prepare(self, params, sn_stats, need_dgrad=True):
        """Spectral-norm power iteration + W/sigma weight copies of every D layer (layers.py:209-221), batched
        over all 21 weights (ops.sn_bank_*).  Depends only on the D parameters and u0, not on the images:
        xmc_gan issues it on a side stream under the generator forward.  -> new spectral_norm_stats."""
        ops = self.ops
        arena = self._bind(params)
        u0 = getattr(sn_stats, "flat", None)
        if u0 is None:                                   # a foreign tree (init / checkpoint): gather once
            u0 = self._pack_u0(sn_stats)
        if self.wp is not None and need_dgrad:
            return self._prepare_folded(arena, u0)
        u_new, v, scal = ops.sn_bank_power_iter(self.bank, arena.params, u0)
        wf, wd = ops.sn_bank_prep(self.bank, arena.params, scal, need_dgrad)
        new_sn = SnTree()
        new_sn.flat = u_new
        for i, (s, e) in enumerate(zip(self.sn_sites, self.bank["entries"])):
            u = u_new[e["u_off"]:e["u_off"] + e["nu"]].view(1, -1)
            vv = v[e["v_off"]:e["v_off"] + e["nv"]]
            sc = scal[2 * i:2 * i + 2]
            tree_set(new_sn, s.path, {"u0": u})
            if e["is_conv"]:
                f, d = ops.sn_bank_weights(self.bank, i, wf, wd)
                s.set_prepared(f, d, u, vv, sc)
            else:
                s.u, s.v, s.scal = u, vv, sc
        self._sn_ctx = (arena, u_new, v, scal, wf, wd)
        return new_sn

    def _prepare_folded(self, arena, u0):
        """``prepare`` with 1 / sigma folded into the launches' alpha: the weight copies are a cast of W, written by the pass
        that also reads W for the first product of the power iteration (2 reads of the arena instead of 3-4)."""
        ops = self.ops
        out = self._wp_persistent()
        if out is not None and self._wp_fresh == (arena.version, u0.data_ptr()):
            # ops.fuse_prep: the optimiser update that produced these parameters also wrote their copies and W^T u (u = that half
            # step's new u = this u0) into the persistent buffers -- the masters are not read again
            bufs, part = out
        else:
            bufs, part = ops.wprep_run(self.wp, arena.params, u0, out=out)
        self._wp_fresh = None
        u_new, v, scal = ops.sn_bank_power_iter_fused(self.bank, self.irr, self.wp, arena.params, u0, part)
        iw = ops.sn_bank_prep(self.irr, arena.params, self._ones_scal, True) if self.irr is not None else (None, None)
        new_sn = SnTree()
        new_sn.flat = u_new
        reg_of = {e["site"]: k for k, e in enumerate(self.wp["entries"])}
        irr_of = {e["site"]: k for k, e in enumerate(self.irr["entries"])} if self.irr is not None else {}
        for i, (s, e) in enumerate(zip(self.sn_sites, self.bank["entries"])):
            u = u_new[e["u_off"]:e["u_off"] + e["nu"]].view(1, -1)
            vv = v[e["v_off"]:e["v_off"] + e["nv"]]
            sc = scal[2 * i:2 * i + 2]
            tree_set(new_sn, s.path, {"u0": u})
            if i in reg_of:
                f, d = ops.wprep_weights(self.wp, reg_of[i], bufs)
                s.set_prepared(f, d, u, vv, sc, folded=True)
            elif e["is_conv"]:
                f, d = ops.sn_bank_weights(self.irr, irr_of[i], *iw)
                s.set_prepared(f, d, u, vv, sc, folded=True)
            else:
                s.u, s.v, s.scal = u, vv, sc
        self._sn_ctx = (arena, u_new, v, scal, *bufs, part, *(t for t in iw if t is not None))
        return new_sn

    def prepared_tensors(self):
        return list(self._sn_ctx[1:])

    # ---- round 5 (ops.fuse_prep): the optimiser kernel emits the prepared weights (see Generator)
    def _wp_persistent(self):
        if not getattr(self.ops, "fuse_prep", False) or self.wp is None:
            return None
        if self._wp_out is None:
            self._wp_out = self.ops.wprep_alloc(self.wp)
        return self._wp_out

    def adam_prep(self, arena):
        if (self._wp_persistent() is None or getattr(self, "_built_for", None) is not arena or self.sn_fix_args() is None
                or getattr(self, "_sn_ctx", None) is None or self._sn_ctx[0] is not arena):
            return None
        if self._skip_map is None:
            self._skip_map = self.ops.wprep_skip_map(self.wp, arena.size, base=self.sn_map)
        return dict(wp=self.wp, out=self._wp_out, skip_map=self._skip_map)

    def note_adam_prepared(self, arena, u_new=None):
        self._wp_fresh = (arena.version, u_new.data_ptr()) if u_new is not None else None

    def refresh_prepared(self, params, sn_stats):
        """a captured graph's first ``prepare`` trusts the persistent buffers: rebuild them from the masters NOW (eager) after
        the parameters or u0 were changed behind its back"""
        arena = self._bind(params)
        u0 = getattr(sn_stats, "flat", None)
        if self._wp_persistent() is not None and u0 is not None:
            self.ops.wprep_run(self.wp, arena.params, u0, out=self._wp_out)

    def finish_grads(self):
        """Gradient through sigma for every spectral weight (one batched pass, layers.py:217-219)."""
        arena, u_new, v, scal = self._sn_ctx[:4]
        if self.sn_fix_args() is not None:
            return                       # applied by the optimiser kernel while it reads the gradient (xmc_gan._apply_adam)
        self.ops.sn_bank_grad_fix(self.bank, arena.params, arena.grads, u_new, v, scal)

    def sn_fix_args(self):
        """(map, bank, scal, u, v) for ops.adam_ema_dev_sn when the gradient through sigma rides in the optimiser kernel"""
        if getattr(self, "sn_map", None) is None or not getattr(self.ops, "fuse_opt", False) or self.wp is None:
            return None
        _, u_new, v, scal = self._sn_ctx[:4]
        return self.sn_map, self.bank, scal, u_new, v

    def forward(self, params, sn_stats, images, cond_dict, *, need_tape, need_dgrad=True, fake_losses=True,
                prepared=None, want_stats=False, after_trunk=None):
        """images (2B, H, W, 3): real first, generated second (xmc_gan.py:140).  ``fake_losses=False``
        skips the generator-side contrastive terms (fake word / fake sentence / image): train_d only
        consumes c_loss_d (xmc_gan.py:240-241), XLA dead-code-eliminates the rest.

        -> (logit (2B,) float32, loss tensor (see LOSS_SLOTS), new_sn_stats, tape)
        """
        ops, cfg = self.ops, self.config
        arena = self._bind(params)
        sent = _to_dev(ops, cond_dict["sentence_embedding"])
        words = _to_dev(ops, cond_dict["embedding"])
        max_len = _to_dev(ops, cond_dict["max_len"])
        x = ops.cast(_to_dev(ops, images, images.dtype if torch.is_tensor(images) else torch.float32), ops.dtype)
        n2, b = x.shape[0], sent.shape[0]
        new_sn = prepared if prepared is not None else self.prepare(params, sn_stats, need_dgrad)

        x, t0 = self.b0.fwd(x)                                              # xmc_net.py:89
        btapes = []
        x_cond = None
        for i, blk in enumerate(self.blocks):                               # :90-95
            x, t = blk.fwd(x)
            btapes.append(t)
            if i == self.cond_idx:
                x_cond = x
        x5 = x
        c5 = x5.shape[-1]
        if after_trunk is not None:
            # the caller's hook between the trunk (chip-filling convolutions) and the heads (a chain of ~60 small launches):
            # xmc_gan.train_d starts the NEXT half step's generator forward on the side stream here, so that its convolutions
            # cover the heads' latency as well as the backward pass
            after_trunk()
        x_pool = ops.reduce_mid(x5.view(n2, -1, c5), relu=True)             # :97-98 (SUM)
        sent_cond = self.sd1.fwd(sent)                                      # :100
        logit = ops.proj_head_fwd(x_pool, self.sd0.w.view(-1), self.sd0.inv_sigma, self.sd0.b, sent_cond)
        losses = ops.zeros((len(LOSS_SLOTS),))
        hstats = ops.zeros((len(LOSS_SLOTS), 2)) if want_stats else None    # (accuracy, entropy) per head
        real_feat, fake_feat = x_pool[:b], x_pool[b:]                       # :106-107
        ls = lambda k: losses[LOSS_SLOTS.index(k):LOSS_SLOTS.index(k) + 1]
        st = lambda k: hstats[LOSS_SLOTS.index(k)] if want_stats else None
        t_fs = t_rs = t_fw = t_rw = t_ic = None
        # round 5: the heads are five independent chains of small launches (~60 per forward, each at its launch floor).  With the
        # generator-side terms on (train_g_d) the FAKE chains (fake sentence / fake word / image-contrastive) run on the side
        # stream beside the REAL ones -- two latency-bound chains in flight instead of one (_HEADS_2STREAM=0: one stream, A/B)
        two = bool(fake_losses and _HEADS_2STREAM and hasattr(ops, "side") and hasattr(ops, "join_side"))
        xc_shape = xc3 = words_n = wprep = None
        if self.use_word:                                                   # :112-121
            xc = self.xc.fwd(x_cond)                                        # :114
            xc_shape = xc.shape
            r = cfg["cond_size"] ** 2
            xc3 = xc.view(n2, r, -1)
            words_n = attn_lib.normalize_words(ops, words)
            wprep = attn_lib.prepare_words(ops, words_n, xc3.dtype, r)

        def fake_heads():
            nonlocal t_fs, t_fw, t_ic
            if self.use_sent:                                               # :105-111
                t_fs = attn_lib.contrastive_loss_fwd(ops, fake_feat, sent_cond, ls("fake_sentence_loss"),
                                                     stats=st("fake_sentence_loss"))
            if self.use_word:
                t_fw = attn_lib.word_loss_fwd(ops, xc3[b:], words_n, max_len, ls("fake_word_loss"),
                                              stats=st("fake_word_loss"), wprep=wprep)
            if self.use_img:                                                # :122-125
                t_ic = attn_lib.contrastive_loss_fwd(ops, fake_feat, real_feat, ls("image_contrastive_loss"),
                                                     stats=st("image_contrastive_loss"))
        if two:
            with ops.side(1):            # its own stream: the step's side stream may be busy (frozen ResNet-50 forward)
                fake_heads()
        elif fake_losses:
            fake_heads()
        if self.use_sent:
            t_rs = attn_lib.contrastive_loss_fwd(ops, real_feat, sent_cond, ls("real_sentence_loss"),
                                                 stats=st("real_sentence_loss"))
        if self.use_word:
            t_rw = attn_lib.word_loss_fwd(ops, xc3[:b], words_n, max_len, ls("real_word_loss"), stats=st("real_word_loss"),
                                          wprep=wprep)
        if two:
            ops.join_side(_tensors_of((t_fs, t_fw, t_ic)), 1)
        tape = None
        if need_tape:
            tape = dict(t0=t0, btapes=btapes, x5=x5, x_pool=x_pool, sent=sent, sent_cond=sent_cond, x_cond=x_cond,
                        xc_shape=xc_shape, t_fs=t_fs, t_rs=t_rs, t_fw=t_fw, t_rw=t_rw, t_ic=t_ic, b=b, n2=n2)
        self.last_aux = dict(x_pool=x_pool)
        self.last_stats = hstats
        for key, t, field in (("real_sentence_logits", t_rs, "logits"), ("real_word_sim_t", t_rw, "sim_t"),
                              ("fake_sentence_logits", t_fs, "logits"), ("image_contrastive_logits", t_ic, "logits"),
                              ("fake_word_sim_t", t_fw, "sim_t")):
            if t is not None:
                self.last_aux[key] = t[field]
        return logit, losses, new_sn, tape

    # ------------------------------------------------------------------------------ backward
    def backward_d(self, tape, dlogit, on_ready=None):
        """Pullback of d_loss = hinge_d + real_word_loss + real_sentence_loss (xmc_gan.py:58-71,153)
        onto the discriminator parameters; ``dlogit`` (2B,) is d hinge_d / d logit.

        ``on_ready(lo, hi)`` (optional; only when the gradient through sigma rides in the optimiser kernel, ``sn_fix_args``):
        called as soon as the slice [lo, hi) of the gradient arena is final, so that the replicas' exchange of it
        (xmc_gan.py:170,251 ``lax.pmean(d_grad)``) runs under the rest of the backward pass.  The arena follows the parameter
        tree: [DiscOptimizedBlock_0, DiscBlock_0 .. 4, SpectralDense_0 / 1, SpectralConv_0]; the two deepest blocks hold 86 %
        of the bytes and finish first: buckets [DiscBlock_4 .. SpectralDense_1] after DiscBlock_4, [DiscBlock_3] after
        DiscBlock_3, the rest at the end."""
        ops = self.ops
        if on_ready is not None and (self.sn_fix_args() is None or not getattr(self, "bucket_order_ok", False) or len(self.blocks) < 3):
            on_ready = None                  # the batched sigma pass rewrites the whole arena at the end: nothing is final early
                                             # (or the arena does not follow the tree order the slices assume: one exchange at the end)
        b, n2 = tape["b"], tape["n2"]
        x_pool, sent_cond = tape["x_pool"], tape["sent_cond"]
        # projection head + SpectralDense_0
        dpool, dsent_cond = ops.proj_head_bwd(dlogit, x_pool, self.sd0.w.view(-1), self.sd0.inv_sigma, sent_cond,
                                              True)
        fw = getattr(self.sd0.arena, "first_write", False)      # one producing launch per leaf: written, not accumulated
        if fw:
            self.sd0.arena.note_write("SpectralDense_0/kernel")
            self.sd0.arena.note_write("SpectralDense_0/bias")
        ops.gemm(x_pool, dlogit.view(n2, 1), ta=True, beta=0.0 if fw else 1.0, out=self.sd0.arena.grad("SpectralDense_0/kernel"))
        ops.reduce_mid(dlogit.view(1, n2, 1), accumulate=not fw,
                       out=self.sd0.arena.grad("SpectralDense_0/bias").view(1, 1))
        # real sentence contrastive: grads to real_feat and sent_cond
        if tape["t_rs"] is not None:
            attn_lib.contrastive_loss_bwd(ops, tape["t_rs"], add_a=dpool[:b], add_b=dsent_cond)
        self.sd1.bwd(tape["sent"], dsent_cond, need_dx=False)
        # real word loss -> real half of x_cond's 1x1 conv output (the fake half gets no gradient from d_loss)
        dxc = None
        if tape["t_rw"] is not None and not _XC_REAL_HALF:
            shp = tape["xc_shape"]
            dxc = ops.zeros_act((n2, *shp[1:]))
            attn_lib.word_loss_bwd(ops, tape["t_rw"], out=dxc[:b].view(b, -1, shp[-1]))
            self._backward_trunk(tape, dpool, dxc, 0, n2, wgrad=True, need_dimg=False, on_ready=on_ready)
        elif tape["t_rw"] is not None:
            # d_loss reaches x_cond's 1x1 convolution through the REAL half only (the generated half's word loss belongs to
            # g_loss): its weight gradient and data gradient run on those B samples -- no zero-filled (2B, ...) cotangent, half
            # the work of both launches (round 5; _backward_trunk: ``xc_rows``)
            shp = tape["xc_shape"]
            dxc = ops.empty((b, *shp[1:]))
            attn_lib.word_loss_bwd(ops, tape["t_rw"], out=dxc.view(b, -1, shp[-1]))
            self._backward_trunk(tape, dpool, dxc, 0, n2, wgrad=True, need_dimg=False, on_ready=on_ready, xc_rows=(0, b))
        else:
            self._backward_trunk(tape, dpool, None, 0, n2, wgrad=True, need_dimg=False, on_ready=on_ready)
        ops.join_wgrad()
        self.finish_grads()
        if on_ready is not None:
            arena = self.sd0.arena
            nb = len(self.blocks)
            on_ready(0, arena.prefix_offset(f"DiscBlock_{nb - 2}"))
            if self.use_word:
                on_ready(arena.prefix_offset("SpectralConv_0"), arena.size)

    def backward_g(self, tape, dlogit_fake):
        """Pullback of g_loss = hinge_g + fake_word + fake_sentence + image_contrastive
        (xmc_gan.py:58-71,154) onto the GENERATED images only: the discriminator has no batch
        coupling in its trunk, so only the fake half (B samples) is back-propagated and no weight
        gradient is formed.  -> d g_loss / d fake images (B, H, W, 3)."""
        ops = self.ops
        b, n2 = tape["b"], tape["n2"]
        x_pool, sent_cond = tape["x_pool"], tape["sent_cond"]
        dpf, _ = ops.proj_head_bwd(dlogit_fake, x_pool[b:], self.sd0.w.view(-1), self.sd0.inv_sigma, sent_cond, False)
        for t in (tape["t_fs"], tape["t_ic"]):
            if t is not None:
                attn_lib.contrastive_loss_bwd(ops, t, want_b=False, add_a=dpf)
        dxc = None
        if tape["t_fw"] is not None:
            dxc = attn_lib.word_loss_bwd(ops, tape["t_fw"]).reshape(b, *tape["xc_shape"][1:])
        return self._backward_trunk(tape, dpf, dxc, b, n2, wgrad=False, need_dimg=True)

    def _backward_trunk(self, tape, dpool, dxc, lo, hi, wgrad, need_dimg, on_ready=None, xc_rows=None):
        """``xc_rows`` = (r0, r1): ``dxc`` covers only the samples [r0, r1) of the slice [lo, hi) (the others receive no gradient
        through the x_cond branch)"""
        ops = self.ops
        arena, nb = self.sd0.arena, len(self.blocks)
        x5 = tape["x5"][lo:hi]
        n, c5 = x5.shape[0], x5.shape[-1]
        dx = ops.bcast_relu_bwd(dpool, x5.reshape(n, -1, c5)).view(x5.shape)
        for i in range(len(self.blocks) - 1, -1, -1):
            if i == self.cond_idx and dxc is not None:              # fan-in of the x_cond branch
                r0, r1 = xc_rows if xc_rows is not None else (0, hi - lo)
                if wgrad:
                    self.xc.wgrad(tape["x_cond"][lo + r0:lo + r1], dxc)
                if (r0, r1) == (0, hi - lo):
                    dx = self.xc.dgrad(dxc, res=dx)
                else:                                               # in place on those samples' rows of dx (out aliases res:
                    self.xc.dgrad(dxc, res=dx[r0:r1], out=dx[r0:r1])   # every element is read, then written, by one thread)
                    for twin in ("mx8", "bits"):                    # fp8 packets / mask bits emitted WITH dx describe the old values:
                        if hasattr(dx, twin):                       # a later packet reader would drop this branch's gradient
                            delattr(dx, twin)
            dx = self.blocks[i].bwd(tape["btapes"][i], dx, lo, hi, wgrad)
            if on_ready is not None and i >= nb - 2 and nb >= 3:
                ops.join_wgrad()
                if i == nb - 1:              # the last block and the two dense heads behind it in the arena
                    end = arena.prefix_offset("SpectralConv_0") if self.use_word else arena.size
                    on_ready(arena.prefix_offset(f"DiscBlock_{i}"), end)
                else:
                    on_ready(arena.prefix_offset(f"DiscBlock_{i}"), arena.prefix_offset(f"DiscBlock_{i + 1}"))
        return self.b0.bwd(tape["t0"], dx, lo, hi, wgrad, need_dimg)

    # ----------------------------------------------------------------------------- flax-style
    def apply(self, variables, inputs, mutable=False):
        images, cond_dict = inputs
        logit, losses, new_sn, _ = self.forward(variables["params"], variables["spectral_norm_stats"], images,
                                                cond_dict, need_tape=False, need_dgrad=False, want_stats=True)
        stats = {}
        for i, k in enumerate(LOSS_SLOTS):                      # the 15-key statistic_dict (xmc_net.py:126-141)
            head = k[:-len("_loss")]
            stats[k] = losses[i]
            stats[head + "_acc"] = self.last_stats[i, 0]
            stats[head + "_entropy"] = self.last_stats[i, 1]
        out = (logit.view(-1, 1), stats)
        if mutable:
            return out, {"spectral_norm_stats": new_sn if self.train else variables["spectral_norm_stats"]}
        return out


LOSS_SLOTS = ["fake_word_loss", "real_word_loss", "fake_sentence_loss", "real_sentence_loss",
              "image_contrastive_loss"]
# the reference's statistic_dict (xmc_net.py:126-141): loss, accuracy and entropy of every contrastive head
# (the accuracy / entropy pair is logging-only and only computed by ``apply``).
STAT_KEYS = [f"{a}_{b}" for a in ("fake_word", "real_word", "fake_sentence", "real_sentence", "image_contrastive")
             for b in ("loss", "acc", "entropy")]
