"""Parameter storage and layer "sites" of the XMC-GAN step (host side).

Mirrors the modules of the reference's ``xmcgan/libml/layers.py`` -- ``SpectralConv``
(:125-241), ``SpectralDense`` (:49-113), ``ConditionalBatchNorm`` (:244-258),
``LocalConditionalBatchNorm`` (:261-273) -- and flax ``nn.Conv`` / ``nn.Dense`` /
``nn.BatchNorm`` as configured in ``xmcgan/nets/xmc_net.py:58-80,170-201``, but as explicit
forward / backward calls on an operator table (``ops``), not as traced modules:

* every network's trainable parameters live in ONE flat float32 arena (plus same-shaped
  arenas for gradients and Adam moments), so the optimiser is one kernel launch and the
  data-parallel gradient exchange is a handful of large RCCL all-reduces on slices of one
  buffer (sized for 288 GB of HBM, not for per-tensor collectives);
* conv kernels are stored [cout][kh*kw][cin] (K contiguous per output channel -- the layout
  the MFMA implicit-GEMM wants); the Flax layout (kh, kw, cin, cout) is exposed as a
  permuted *view* so the parameter tree keeps the reference's names and shapes.
"""
from __future__ import annotations

import re

import numpy as np
import os

import torch

from .. import synthetic as syn


class ParamTree(dict):
    """Nested dict of (views of) parameters in Flax layout; the root carries ``.arena``."""
    arena = None


def _kind(path, shape):
    if path.endswith("kernel"):
        return "conv" if len(shape) == 4 else "dense"
    return "vec"


_PAIR = re.compile(r"(.*(?:Local)?ConditionalBatchNorm_\d+)/(Conv|Dense)_([01])/(kernel|bias)$")


class ParamArena:
    """Flat float32 arena for one network: params, grads, Adam moments share the layout.

    gamma / beta of every (Local)ConditionalBatchNorm (``Conv_0``/``Conv_1`` or ``Dense_0``/``Dense_1``,
    reference layers.py:252-254,269-270) read the same input, so they are stored as ONE physical tensor
    ``<module>/GB`` -- conv master (2C, 1, cin), dense kernel TRANSPOSED (2C, in), bias (2C) -- and run as one
    convolution / GEMM with 2C outputs; the Flax-named members are slice views of it.
    """

    ALIGN = 64   # elements; keeps every tensor 256-byte aligned

    def __init__(self, ops, shape_tree, with_opt=True):
        self.ops = ops
        self.shape_tree = shape_tree
        self.specs = {}          # path -> (offset, internal shape, kind, flax shape, member-of)
        self.merged = {}         # merged path -> (offset, internal shape)
        off = 0

        def alloc(n):
            nonlocal off
            o = off
            off += (n + self.ALIGN - 1) // self.ALIGN * self.ALIGN
            return o

        # The gamma / beta projections of the LOCAL conditional BatchNorm sites all read the same spatial condition and run as
        # ONE convolution (nets/common.py FusedLocalGB): their merged kernels, then their merged biases, are allocated behind
        # every other tensor, in tree order -- contiguous, so the fused weight (and its gradient) is a VIEW of the arena instead of
        # a concatenation rebuilt every step (and a gradient scattered back slice by slice).
        # Round 5: the GLOBAL sites (Dense_0 / Dense_1 on the (B, 2 z_dim) condition) likewise run as ONE product
        # (nets/common.py FusedGlobalGB).  Their merged kernels are stored TRANSPOSED -- (2C, in), the layout of the conv masters --
        # so that the sites stack along the rows; kernels, then biases, are allocated in FRONT of every other tensor: the fused
        # gradient is the last thing the generator's backward pass produces, i.e. it belongs to the last slice of the
        # data-parallel exchange, [0, first tensor of GenBlock_1) (``prefix_offset`` skips these relocated tensors).
        tail = {"kernel": [], "bias": []}
        head = {"kernel": [], "bias": []}
        self.relocated = set()
        leaves = list(syn.tree_leaves(shape_tree))
        for path, shape in leaves:                   # the head group first: offsets are handed out in order
            m = _PAIR.match(path)
            if m is None or m.group(2) != "Dense":
                continue
            mpath = f"{m.group(1)}/GB/{m.group(4)}"
            if mpath not in self.merged:
                mshape = (2 * shape[1], shape[0]) if m.group(4) == "kernel" else (2 * shape[0],)
                self.merged[mpath] = (None, mshape)
                head[m.group(4)].append(mpath)
        for which in ("kernel", "bias"):
            for mpath in head[which]:
                mshape = self.merged[mpath][1]
                self.merged[mpath] = (alloc(int(np.prod(mshape))), mshape)
                self.relocated.add(mpath)
        for path, shape in leaves:
            kind = _kind(path, shape)
            ishape = (shape[3], shape[0] * shape[1], shape[2]) if kind == "conv" else tuple(shape)
            m = _PAIR.match(path)
            if m is None:
                self.specs[path] = (alloc(int(np.prod(shape))), ishape, kind, tuple(shape), None)
                continue
            mpath, idx = f"{m.group(1)}/GB/{m.group(4)}", int(m.group(3))
            if mpath not in self.merged:             # (a local site: allocated below)
                mshape = (2 * ishape[0], ishape[1], ishape[2]) if kind == "conv" else (2 * ishape[0],)
                self.merged[mpath] = (None, mshape)
                tail[m.group(4)].append(mpath)
            self.specs[path] = (None, ishape, "dense_t" if kind == "dense" else kind, tuple(shape), (mpath, idx))
        for which in ("kernel", "bias"):
            for mpath in tail[which]:
                mshape = self.merged[mpath][1]
                self.merged[mpath] = (alloc(int(np.prod(mshape))), mshape)
                self.relocated.add(mpath)
        self.size = off
        self.n_params = sum(int(np.prod(s[3])) for s in self.specs.values())
        self.params = ops.zeros((self.size,))
        self.grads = ops.zeros((self.size,)) if with_opt else None
        self.m = ops.zeros((self.size,)) if with_opt else None
        self.v = ops.zeros((self.size,)) if with_opt else None
        # Adam step counter: the kernels read it from DEVICE memory (``step_state``: [t as int32 bits, 1/(1-b1^t),
        # 1/(1-b2^t), pad], advanced by xmc_adam_ema_dev) so that a captured hipGraph replays consecutive steps;
        # ``opt_step`` is the host mirror (checkpoints, flax ``state.step``).
        self._opt_step = 0
        self.step_state = (torch.zeros((4,), dtype=torch.float32, device=self.params.device)
                           if with_opt and hasattr(ops, "adam_ema_dev") else None)
        self.version = 0         # bumped whenever params change (invalidates prepared weights)
        # True while the gradient arena is known to be all zeros: the fused optimiser kernel (ops.adam_ema_dev_sn) zeroes
        # the gradient it consumed, so the next half step needs no fill; ``zero_grads`` clears the flag because its caller
        # is about to accumulate
        self.grads_clean = with_opt
        # round 5 (ops.first_write): the producers of a half step WRITE their gradients (one producing launch per tensor) instead
        # of adding into a zeroed arena -- nothing ever fills or zeroes ``grads``; what a step leaves there is overwritten by
        # the next one.  ``_audit``: the leaves written so far, checked against the arena's leaves by the FIRST optimiser update
        # (a leaf nobody wrote would feed the optimiser the previous step's gradient), then dropped.
        # XMC_AUDIT_WRITES=1 (debugging aid): keep auditing EVERY update, and count the writes -- a leaf written twice
        # between two updates (gradient accumulation, a backward pass run in two slices) is an error in this mode as well:
        # accumulation needs XMC_FIRST_WRITE=0.
        self.first_write = bool(with_opt and getattr(ops, "first_write", False))
        self._audit = {} if self.first_write else None
        self._audit_always = __import__("os").environ.get("XMC_AUDIT_WRITES", "0") != "0"

    @property
    def opt_step(self):
        return self._opt_step

    @opt_step.setter
    def opt_step(self, t):
        """Set the step counter on the host AND on the device (checkpoint restore; never inside a graph capture)."""
        self._opt_step = int(t)
        if self.step_state is not None:
            self.step_state.view(torch.int32)[0] = int(t)

    def note_steps(self, k=1):
        """The device counter was advanced ``k`` times by the Adam kernels: advance the host mirror."""
        self._opt_step += k

    # ------------------------------------------------------------------------------ views
    def view(self, path, buf=None):
        buf = self.params if buf is None else buf
        if path in self.merged:
            off, mshape = self.merged[path]
            return buf[off:off + int(np.prod(mshape))].view(mshape)
        off, ishape, kind, _, member = self.specs[path]
        if member is None:
            return buf[off:off + int(np.prod(ishape))].view(ishape)
        mv = self.view(member[0], buf)
        c = ishape[1] if kind == "dense_t" else ishape[0]
        lo, hi = member[1] * c, (member[1] + 1) * c
        return mv[lo:hi].t() if kind == "dense_t" else mv[lo:hi]     # (a transposed view of the (2C, in) merged kernel's rows)

    def grad(self, path):
        return self.view(path, self.grads)

    def offset(self, path):
        """Element offset of a (non-merged-member) tensor inside the flat arena."""
        if path in self.merged:
            return self.merged[path][0]
        off = self.specs[path][0]
        assert off is not None, f"{path} is a member of a merged tensor"
        return off

    def prefix_offset(self, prefix):
        """Element offset of the first tensor stored under the tree node ``prefix`` (arena order = tree order)."""
        offs = [o for p, (o, *_rest) in self.specs.items() if o is not None and (p == prefix or p.startswith(prefix + "/"))]
        offs += [o for p, (o, _shape) in self.merged.items()
                 if (p == prefix or p.startswith(prefix + "/")) and p not in self.relocated]
        assert offs, prefix
        return min(offs)

    def flax_view(self, path, buf=None):
        v = self.view(path, buf)
        _, _, kind, fshape, _ = self.specs[path]
        if kind == "conv":
            return v.view(fshape[3], fshape[0], fshape[1], fshape[2]).permute(1, 2, 3, 0)
        return v

    def tree(self, buf=None):
        """Flax-layout ParamTree of views over ``buf`` (default: the parameters)."""
        flat = {p: self.flax_view(p, buf) for p in self.specs}
        root = ParamTree(_unflatten(self.shape_tree, flat))
        root.arena = self if buf is None else None
        root.buffer = self.params if buf is None else buf
        return root

    def load_flax(self, tree, buf=None):
        """Copy a Flax-layout tree (numpy or torch leaves) into the arena."""
        for path, leaf in syn.tree_leaves(tree):
            t = torch.as_tensor(np.asarray(leaf) if not torch.is_tensor(leaf) else leaf).to(
                device=self.params.device, dtype=torch.float32)
            self.flax_view(path, buf).copy_(t)
        self.version += 1

    def zero_grads(self):
        if self.first_write:
            return                                   # every leaf is written, not accumulated: nothing to clear
        if not self.grads_clean:
            self.grads.zero_()
        self.grads_clean = False

    def note_write(self, path):
        """a producer wrote the gradient of leaf ``path`` (first-write audit; free once the first update has checked it)"""
        if self._audit is not None:
            self._audit[path] = self._audit.get(path, 0) + 1

    def audit_writes(self):
        """first optimiser update of a first-write arena: every leaf must have been written by this half step"""
        if self._audit is None:
            return
        leaves = {p for p, sp in self.specs.items() if sp[4] is None} | set(self.merged)
        missing = sorted(leaves - set(self._audit))
        twice = sorted(p for p, k in self._audit.items() if k > 1)
        self._audit = {} if self._audit_always else None
        if twice and self._audit_always:
            raise RuntimeError(f"first-write gradient arena: {twice[:6]}{' ...' if len(twice) > 6 else ''} written more than once before "
                               f"this update -- the earlier gradient was overwritten, not accumulated (XMC_FIRST_WRITE=0 accumulates)")
        if missing:
            raise RuntimeError(f"first-write gradient arena: no producer wrote {missing[:6]}{' ...' if len(missing) > 6 else ''} "
                               f"in this half step -- the optimiser would consume a stale gradient (XMC_FIRST_WRITE=0 restores "
                               f"the zero-and-accumulate path)")


def _unflatten(shape_tree, flat, prefix=""):
    out = {}
    for k, v in shape_tree.items():
        path = f"{prefix}/{k}" if prefix else k
        out[k] = _unflatten(v, flat, path) if isinstance(v, dict) else flat[path]
    return out


def tree_get(tree, path):
    for k in path.split("/"):
        tree = tree[k]
    return tree


def _tree_find(tree, path):
    for k in path.split("/"):
        if not isinstance(tree, dict) or k not in tree:
            return None
        tree = tree[k]
    return tree


class FlatTree(dict):
    """A state tree (batch_stats / spectral_norm_stats) whose leaves are views of ONE flat buffer ``.flat`` in a fixed
    site order: the next forward pass copies / consumes it with one launch, and a captured hipGraph writes the new
    statistics back into the persistent buffer with one copy."""
    flat = None


def flat_running_stats(sites, batch_stats, flat=None):
    """``batch_stats`` re-laid as a FlatTree over the running statistics of ``sites`` (BatchNormSite), order
    [mean, var] per site.  ``flat``: use this buffer (already holding the values) instead of gathering."""
    if flat is None:
        leaves = []
        for s in sites:
            st = tree_get(batch_stats, s.path)
            leaves += [st["mean"].reshape(-1), st["var"].reshape(-1)]
        flat = torch.cat(leaves)
    out, off = FlatTree(), 0
    out.flat = flat
    for s in sites:
        n = tree_get(batch_stats, s.path)["mean"].numel()
        tree_set(out, s.path, {"mean": flat[off:off + n], "var": flat[off + n:off + 2 * n]})
        off += 2 * n
    return out


def prefill_running_stats(sites, batch_stats, new_batch_stats):
    """Fresh copies of the running statistics of all ``sites`` (BatchNormSite) as views of ONE buffer filled by one
    copy (a FlatTree input) or one batched gather (a foreign tree), installed in ``new_batch_stats`` for
    ``BatchNormSite.stats`` to update in place -- instead of two ``clone`` launches per site per forward pass."""
    src = getattr(batch_stats, "flat", None)
    fresh = flat_running_stats(sites, batch_stats, src.clone() if src is not None else None)
    new_batch_stats.update(fresh)
    new_batch_stats.flat = fresh.flat


def tree_set(tree, path, value):
    keys = path.split("/")
    for k in keys[:-1]:
        tree = tree.setdefault(k, {})
    tree[keys[-1]] = value


# ------------------------------------------------------------------------------------- sites
class ConvSite:
    """One convolution call site: flax ``nn.Conv`` (generator) or ``SpectralConv`` (discriminator).

    ``prepare`` runs the power iteration (spectral sites; reference layers.py:209-221) and writes
    the activation-dtype forward / dgrad weight copies; ``fwd`` / ``dgrad`` / ``wgrad`` launch the
    implicit-GEMM kernels; ``finish`` applies the gradient through sigma.
    """

    def __init__(self, ops, arena, path, spectral=False):
        self.ops, self.arena, self.path, self.spectral = ops, arena, path, spectral
        self.w = arena.view(path + "/kernel")                  # (cout, taps, cin)
        self.b = arena.view(path + "/bias")
        self.cout, self.taps, self.cin = self.w.shape
        self.ks = int(round(self.taps ** 0.5))
        self._ver = -1
        self.wf = self.wd = self.u = self.v = self.scal = None
        self.phase = None        # "ups" / "pool": set by the block that puts this site next to a 2x resampling (ops.attach_phase_weights)
        # fold_sigma (ops): the prepared weights are a cast of W and 1 / (sigma + eps) rides in the launch's alpha
        self.alpha_dev = None

    def prepare(self, sn_state=None, new_sn_state=None, need_dgrad=True):
        """Must be called once per forward pass before ``fwd`` (weights may have changed)."""
        ops = self.ops
        inv = None
        self.alpha_dev = None
        if self.spectral:
            u0 = tree_get(sn_state, self.path)["u0"]
            self.u, self.v, self.scal = ops.spectral_power_iter(self.w.view(self.cout, -1), u0, 0)
            tree_set(new_sn_state, self.path, {"u0": self.u})
            inv = self.scal[1:2]
        elif self._ver == self.arena.version and (self.wd is not None or not need_dgrad):
            return
        self.wf, self.wd = ops.prep_conv_weight(self.w, inv, need_dgrad, phase=self.phase)
        self._ver = self.arena.version

    def set_prepared(self, wf, wd, u, v, scal, folded=False):
        """Install the outputs of the batched spectral pass (ops.sn_bank_*) for this site.  ``folded``: the weights are a
        pure cast of W (ops.wprep_*, phase copies already attached) and 1 / (sigma + eps) goes into every launch's alpha."""
        self.wf, self.wd, self.u, self.v, self.scal = wf, wd, u, v, scal
        self._ver = self.arena.version
        self.alpha_dev = scal[1:2] if (folded and scal is not None) else None
        if isinstance(wf, self.ops_packed()) and wf.data is None:       # phase-only site: how to make the plain copies if ever needed
            inv = None if folded or scal is None else scal[1:2]
            wf.lazy = (self.w, inv, 0)
            if wd is not None:
                wd.lazy = (self.w, inv, 1)
        if not folded:
            self.ops.attach_phase_weights(self.w, scal[1:2] if scal is not None else None, wf, wd, self.phase)

    def ops_packed(self):
        from ..ops import PackedWeight
        return PackedWeight

    def _akw(self, kw):
        if self.alpha_dev is not None:
            kw["alpha_dev"] = self.alpha_dev
        return kw

    def fwd(self, x, **kw):
        return self.ops.conv(x, self.wf, self.b, ks=self.ks, **self._akw(kw))

    def fwd_pool(self, x, res=None, **kw):
        """avg_pool2x2(conv(x)) + res: fused into the convolution's epilogue where the kernel supports it (the
        full-resolution tensor is then never written), otherwise convolution followed by the pooling kernel."""
        if self.ops.can_pool_out(x, self.wf, kw.get("ups", False)):
            return self.ops.conv(x, self.wf, self.b, ks=self.ks, pool_out=True, res=res, **self._akw(kw))
        kw.pop("emit_mx8", None)                     # the hints are about the POOLED tensor: not this launch's output
        kw.pop("emit_bits", None)
        return self.ops.pool2(self.fwd(x, **kw), 0.25, res=res)

    def dgrad(self, dy, **kw):
        return self.ops.conv(dy, self.wd, None, ks=self.ks, **self._akw(kw))

    def dgrad_sumpool(self, dy):
        """sum_pool2x2(dgrad(dy)): the adjoint of ``conv(nearest_upsample2(.))`` -- pooled in the epilogue when the
        kernel can (sum = 4 x average), else dgrad followed by the pooling kernel."""
        if self.ops.can_pool_out(dy, self.wd):
            return self.ops.conv(dy, self.wd, None, ks=self.ks, pool_out=True, alpha=4.0, **self._akw({}))
        return self.ops.pool2(self.dgrad(dy), 1.0)

    def _fw(self):
        """first-write arena: this site's gradient launches overwrite (and are noted for the arena's audit)"""
        fw = getattr(self.arena, "first_write", False)
        if fw:
            self.arena.note_write(self.path + "/kernel")
            self.arena.note_write(self.path + "/bias")
        return fw

    def wgrad(self, x, dy, **kw):
        """The kernel gradient and the (fused) bias gradient alpha * sum_p dy'(p): accumulated, or written (first-write arena)."""
        if self._fw():
            kw["overwrite"] = True
        self.ops.conv_wgrad(x, dy, self.arena.grad(self.path + "/kernel"), self.arena.grad(self.path + "/bias"),
                            ks=self.ks, **kw)

    def finish(self):
        if self.spectral:
            g = self.arena.grad(self.path + "/kernel").view(self.cout, -1)
            self.ops.spectral_grad_fix(g, self.w.view(self.cout, -1), self.u, self.v, self.scal, 0)

    # ---- RGB-like (<= 3 channel) operands: run as a 1x1 convolution on a tap-expanded 32-channel tensor
    #      (ops.expand_taps), i.e. on the MFMA kernels instead of the scalar-gather fallbacks.
    def fwd_rgb_in(self, x, emit_bits=False, relu_out=False):
        """conv(x) for cin <= 3.  -> (y, xcol) with xcol the expanded input (kept for wgrad_rgb_in).  ``emit_bits``: y is the
        ReLU mask of a later data gradient (ops.conv): the 32-channel weight then goes through the fragment-packed
        pointwise kernel, whose epilogue writes the mask bits."""
        k = self.taps * self.cin
        if getattr(self, "_w32_src", None) is not self.wf:
            w32 = torch.zeros((self.cout, 1, 32), dtype=self.wf.dtype, device=self.wf.device)
            w32[:, 0, :k] = self.wf.reshape(self.cout, k)
            self._w32, self._w32_src, self._w32p = w32, self.wf, None
        xcol = self.ops.expand_taps(x, self.ks, 1)
        if emit_bits and getattr(self.ops, "mask_bits", False) and self.ops._packable(1, 32) and self.cout % 16 == 0:
            if self._w32p is None:
                self._w32p = self.ops.pack_conv_weight(self._w32)
            return self.ops.conv(xcol, self._w32p, self.b, ks=1, emit_bits=True, relu_out=relu_out, **self._akw({})), xcol
        return self.ops.conv(xcol, self._w32, self.b, ks=1, relu_out=relu_out, **self._akw({})), xcol

    def dgrad_rgb_out(self, dy):
        """dgrad for cout <= 3: the (cin <- cout) convolution has a 3-channel INPUT (dy) -- same expansion."""
        k = self.taps * self.cout
        if getattr(self, "_wd32_src", None) is not self.wd:
            w32 = torch.zeros((self.cin, 1, 32), dtype=self.wd.dtype, device=self.wd.device)
            w32[:, 0, :k] = self.wd.reshape(self.cin, k)
            self._wd32, self._wd32_src = w32, self.wd
        return self.ops.conv(self.ops.expand_taps(dy, self.ks, 1), self._wd32, None, ks=1, **self._akw({}))

    def wgrad_rgb_in(self, xcol, dy, **kw):
        k = self.taps * self.cin
        if self._fw():
            dw32 = torch.empty((self.cout, 1, 32), dtype=torch.float32, device=dy.device)
            self.ops.conv_wgrad(xcol, dy, dw32, self.arena.grad(self.path + "/bias"), ks=1, sync=True, overwrite=True, **kw)
            self.arena.grad(self.path + "/kernel").view(self.cout, k).copy_(dw32[:, 0, :k])
            return
        dw32 = torch.zeros((self.cout, 1, 32), dtype=torch.float32, device=dy.device)
        self.ops.conv_wgrad(xcol, dy, dw32, self.arena.grad(self.path + "/bias"), ks=1, sync=True, **kw)
        self.arena.grad(self.path + "/kernel").view(self.cout, k).add_(dw32[:, 0, :k])

    def wgrad_rgb_out(self, x, dy):
        """Weight / bias gradient for cout <= 3: dW[o][tap][c] = sum_q x[q][c] dy[q - d(tap)][o]."""
        k = self.taps * self.cout
        dyx = self.ops.expand_taps(dy, self.ks, -1)
        if self._fw():
            dw32 = torch.empty((32, 1, self.cin), dtype=torch.float32, device=dy.device)
            self.ops.conv_wgrad(x, dyx, dw32, None, ks=1, sync=True, overwrite=True)
            self.arena.grad(self.path + "/kernel").copy_(dw32[:k, 0, :].view(self.taps, self.cout, self.cin).permute(1, 0, 2))
            self.ops.reduce_mid(dy.reshape(1, -1, self.cout), accumulate=False,
                                out=self.arena.grad(self.path + "/bias").view(1, self.cout))
            return
        dw32 = torch.zeros((32, 1, self.cin), dtype=torch.float32, device=dy.device)
        self.ops.conv_wgrad(x, dyx, dw32, None, ks=1, sync=True)
        self.arena.grad(self.path + "/kernel").add_(
            dw32[:k, 0, :].view(self.taps, self.cout, self.cin).permute(1, 0, 2))
        self.ops.reduce_mid(dy.reshape(1, -1, self.cout), accumulate=True,
                            out=self.arena.grad(self.path + "/bias").view(1, self.cout))


_DENSE_FAST = os.environ.get("XMC_DENSE_FAST", "1") != "0"            # A/B switch: dense layers on the bf16 MFMA in the bf16 mode


class DenseSite:
    """flax ``nn.Dense`` / ``SpectralDense`` on the float32 strided GEMM (kernel (in, out))."""

    def __init__(self, ops, arena, path, spectral=False):
        self.ops, self.arena, self.path, self.spectral = ops, arena, path, spectral
        # a merged gamma | beta kernel (``<module>/GB``) is stored transposed, (out, in): ``w`` is its (in, out) view and every
        # product below takes strides -- only the weight gradient has to be written in the stored orientation
        self.wt = (path + "/kernel") in arena.merged
        self.w = arena.view(path + "/kernel").t() if self.wt else arena.view(path + "/kernel")       # (in, out)
        self.b = arena.view(path + "/bias")
        self.u = self.v = self.scal = None
        self.cout, self.cin = self.w.shape[1], self.w.shape[0]

    def prepare(self, sn_state=None, new_sn_state=None):
        if self.spectral:
            u0 = tree_get(sn_state, self.path)["u0"]
            self.u, self.v, self.scal = self.ops.spectral_power_iter(self.w, u0, 1)
            tree_set(new_sn_state, self.path, {"u0": self.u})

    @property
    def inv_sigma(self):
        return self.scal[1:2] if self.spectral else None

    def fwd(self, x):
        # fast: in the bf16 training mode the operands are rounded to bf16 and multiplied with float32 accumulation --
        # what flax's nn.Dense(dtype=bfloat16) computes (layers.py:49-113 cast kernel and input to the module dtype); the
        # float32 parity mode ignores the flag.
        ops = self.ops
        out = self.b.unsqueeze(0).repeat(x.shape[0], 1)        # bias broadcast, then C += x W
        return ops.gemm(x, self.w, alpha_dev=self.inv_sigma, beta=1.0, out=out, fast=_DENSE_FAST)

    def bwd(self, x, dy, need_dx=True):
        """Accumulates dW (wrt the normalised kernel for spectral sites -- ``finish`` fixes it) and
        db; returns dx."""
        ops = self.ops
        fw = getattr(self.arena, "first_write", False)          # one producing launch per leaf: written, not accumulated
        if fw:
            self.arena.note_write(self.path + "/kernel")
            self.arena.note_write(self.path + "/bias")
        if self.wt:                                             # dW^T (out, in) = dy^T x
            ops.gemm(dy, x, ta=True, beta=0.0 if fw else 1.0, out=self.arena.grad(self.path + "/kernel"), fast=_DENSE_FAST)
        else:
            ops.gemm(x, dy, ta=True, beta=0.0 if fw else 1.0, out=self.arena.grad(self.path + "/kernel"), fast=_DENSE_FAST)
        ops.reduce_mid(dy.reshape(1, dy.shape[0], -1), accumulate=not fw,
                       out=self.arena.grad(self.path + "/bias").view(1, -1))
        if need_dx:
            return ops.gemm(dy, self.w, tb=True, alpha_dev=self.inv_sigma, fast=_DENSE_FAST)
        return None

    def finish(self):
        if self.spectral:
            self.ops.spectral_grad_fix(self.arena.grad(self.path + "/kernel"), self.w, self.u, self.v,
                                       self.scal, 1)


class BatchNormSite:
    """flax ``nn.BatchNorm(use_scale=False, use_bias=False, momentum=.9, eps=1e-5)``
    (xmc_net.py:192-201).  Statistics are float32; running stats live in the batch_stats tree."""

    def __init__(self, ops, path):
        self.ops, self.path = ops, path          # path of the BatchNorm_0 collection entry

    def stats(self, x, batch_stats, new_batch_stats, train):
        ops = self.ops
        st = tree_get(batch_stats, self.path)
        if not train:
            tree_set(new_batch_stats, self.path, st)
            return ops.bn_from_running(st["mean"], st["var"])
        pre = _tree_find(new_batch_stats, self.path)
        if pre is not None:                      # the network copied ALL running statistics with one batched copy
            rm, rv = pre["mean"], pre["var"]
        else:
            rm, rv = st["mean"].clone(), st["var"].clone()
        mean, rstd = ops.bn_batch_stats(x, rm, rv, True)
        tree_set(new_batch_stats, self.path, {"mean": rm, "var": rv})
        return mean, rstd
