"""Generator / discriminator residual blocks with explicit forward and backward.

Mirrors ``xmcgan/nets/common.py`` of the reference: ``GenBlock`` (:136-160),
``GenSpatialBlock`` (:163-186), ``DiscBlock`` (:58-79), ``DiscOptimizedBlock`` (:117-133),
``upsample`` (:48-51), ``dsample`` (:23-55).  Differences are algebraic, not numerical in exact
arithmetic:

* nearest-2x upsampling is never materialised -- it is fused into the convolution's gather
  (``ups``) and, in backward, into the dgrad epilogue / a 2x2 sum-pool;
* ``conv1x1(upsample(x)) == upsample(conv1x1(x))`` is used for the conditioning maps of
  LocalConditionalBatchNorm (gamma/beta are computed once at the 16x16 conditioning
  resolution and indexed with a shift), and ``pool(conv1x1(x)) == conv1x1(pool(x))`` for the
  discriminator shortcuts;
* ReLU is fused into the consumer convolution's gather (``relu_in``) and its backward into
  the producer dgrad's epilogue (``mask``).
"""
from __future__ import annotations

import os

import torch

from ..libml.layers import BatchNormSite, ConvSite, DenseSite


# =================================================================================== generator
class CondNorm:
    """ConditionalBatchNorm (layers.py:244-258) or LocalConditionalBatchNorm (:261-273) + ReLU.

    gamma and beta come from two Dense layers on the (B, 2*z_dim) global condition, or from two 1x1
    convolutions on the (B, 16, 16, 1024) spatial condition evaluated ONCE at the conditioning
    resolution (the reference evaluates them on the upsampled map).  Both read the same input, so they
    run as ONE dense / convolution with 2C outputs on the merged ``<module>/GB`` parameters
    (``ParamArena``); the normalisation kernels read gamma / beta as the two halves of that output.
    All LOCAL sites of the generator read the same spatial condition: ``FusedLocalGB`` evaluates their
    projections as one convolution and hands every site a column slice of its output.
    """

    def __init__(self, ops, arena, path, local):
        self.ops, self.local, self.path = ops, local, path
        self.gb = ConvSite(ops, arena, path + "/GB") if local else DenseSite(ops, arena, path + "/GB")
        self.bn = BatchNormSite(ops, path + "/BatchNorm_0")
        self.fused = None                            # FusedLocalGB / FusedGlobalGB (set by the generator)

    def prepare(self):
        if self.local and self.fused is None:
            self.gb.prepare()

    def fwd(self, x, cond, batch_stats, new_stats, train):
        ops = self.ops
        if self.local:                              # cond (B, hc, hc, 1024) activation dtype
            hc = cond.shape[1]
            gb = self.fused.gb_of(self) if self.fused is not None else self.gb.fwd(cond, out_f32=True)   # (B*hc*hc, 2C) float32
        else:                                       # cond (B, 2*z_dim) float32
            hc = 1
            gb = self.fused.gb_of(self) if self.fused is not None else self.gb.fwd(cond)                  # (B, 2C)
        mean, rstd = self.bn.stats(x, batch_stats, new_stats, train)
        y = ops.cbn_act_fwd(x, mean, rstd, gb, hc, relu=True)
        return y, (x, mean, rstd, gb, hc, cond)

    def bwd(self, tape, dy, dcond):
        """Returns (dx, dcond) -- dcond accumulates d(condition) across all norm sites (fused local sites leave
        it untouched: their share is produced by ``FusedLocalGB.bwd`` once every site has written its slice)."""
        ops = self.ops
        x, mean, rstd, gb, hc, cond = tape
        if self.fused is not None:
            dx, _ = ops.cbn_act_bwd(dy, x, mean, rstd, gb, hc, relu=True, dgb_out=self.fused.dgb_of(self))
            return dx, dcond
        dx, dgb = ops.cbn_act_bwd(dy, x, mean, rstd, gb, hc, relu=True)
        if self.local:
            d = ops.cast(dgb.view(x.shape[0], hc, hc, -1), ops.dtype)
            self.gb.wgrad(cond, d)
            dcond = self.gb.dgrad(d, res=dcond)
        else:
            d1 = self.gb.bwd(cond, dgb.view(x.shape[0], -1))
            dcond = d1 if dcond is None else ops.add(dcond, d1)
        return dx, dcond


class FusedLocalGB:
    """The gamma/beta 1x1 projections of ALL LocalConditionalBatchNorm sites as one convolution.

    Every local site projects the same (B, 16, 16, 1024) spatial condition (xmc_net.py:236-244): one
    (B*256 x 1024) x (1024 x sum 2C_i) product instead of 7 small ones in the forward pass, and one weight-
    gradient + one data-gradient launch instead of 14 in the backward pass.  The per-site master parameters stay
    where the reference names them (``.../LocalConditionalBatchNorm_k/GB``); their concatenation is rebuilt (one
    batched copy) whenever the generator's parameters change, and the fused weight gradient is added back
    slice by slice."""

    def __init__(self, ops, arena, sites):
        self.ops, self.arena, self.sites = ops, arena, sites
        self.off, o = {}, 0
        for s in sites:
            self.off[id(s)] = (o, s.gb.cout)
            o += s.gb.cout
            s.fused = self
        self.total = o
        self.cin = sites[0].gb.cin
        self._ver = -1
        self.wf = self.wd = self.bias = None
        self.gball = self.dgball = None
        # ParamArena allocates the local sites' merged kernels (then biases) back to back: when no alignment gap separates them
        # (every 2C a multiple of the arena's 64-element granule -- true from gf_dim = 32 up) the fused weight, its bias and
        # their gradients are plain views of the arena
        ko = [arena.offset(s.gb.path + "/kernel") for s in sites]
        bo = [arena.offset(s.gb.path + "/bias") for s in sites]
        self.contig = (all(ko[i] + sites[i].gb.cout * self.cin == ko[i + 1] for i in range(len(sites) - 1))
                       and all(bo[i] + sites[i].gb.cout == bo[i + 1] for i in range(len(sites) - 1))
                       and os.environ.get("XMC_GB_CONTIG", "1") != "0")             # (A/B switch)
        self._ko, self._bo = ko[0], bo[0]

    def _views(self, buf):
        return (buf[self._ko:self._ko + self.total * self.cin].view(self.total, 1, self.cin), buf[self._bo:self._bo + self.total])

    def prepare(self):
        if self._ver == self.arena.version and self.wf is not None:
            return
        if self.contig:
            w, self.bias = self._views(self.arena.params)
        else:
            w = torch.cat([s.gb.w for s in self.sites], dim=0)              # (sum 2C, 1, cin) float32 masters
            self.bias = torch.cat([s.gb.b for s in self.sites])
        self.wf, self.wd = self.ops.prep_conv_weight(w, None, True)
        self._ver = self.arena.version

    def fwd(self, cond):
        b, hc = cond.shape[0], cond.shape[1]
        f32 = not getattr(self.ops, "gb_bf16", False)                       # bf16 mode: the maps stay in the activation dtype
        self.gball = self.ops.conv(cond, self.wf, self.bias, ks=1, out_f32=f32).view(b * hc * hc, self.total)
        return self.gball

    def gb_of(self, site):
        o, n = self.off[id(site)]
        return self.gball[:, o:o + n]

    def begin_bwd(self, gball):
        self.gball = gball
        self.dgball = torch.empty_like(gball)        # every site's cbn backward fills its own columns

    def dgb_of(self, site):
        o, n = self.off[id(site)]
        return self.dgball[:, o:o + n]

    def bwd(self, cond):
        """-> d(cond); accumulates the GB weight / bias gradients of every site."""
        ops = self.ops
        b, hc = cond.shape[0], cond.shape[1]
        d = ops.cast(self.dgball.view(b, hc, hc, self.total), ops.dtype)
        fw = getattr(self.arena, "first_write", False)
        if fw:
            for s in self.sites:
                self.arena.note_write(s.gb.path + "/kernel")
                self.arena.note_write(s.gb.path + "/bias")
        if self.contig:                              # the fused gradient IS a slice of the gradient arena
            gw, gbias = self._views(self.arena.grads)
            ops.conv_wgrad(cond, d, gw, gbias, ks=1, sync=True, **({"overwrite": True} if fw else {}))
            return ops.conv(d, self.wd, None, ks=1)
        if fw:                                       # the fused gradient is WRITTEN, then copied (not added) to its seven masters
            dw = torch.empty((self.total, 1, self.cin), dtype=torch.float32, device=d.device)
            db = torch.empty((self.total,), dtype=torch.float32, device=d.device)
        else:
            dw = torch.zeros((self.total, 1, self.cin), dtype=torch.float32, device=d.device)
            db = ops.zeros((self.total,))
        ops.conv_wgrad(cond, d, dw, db, ks=1, sync=True, **({"overwrite": True} if fw else {}))   # consumed right below on this stream
        for s in self.sites:
            o, n = self.off[id(s)]
            gk, gbias = s.gb.arena.grad(s.gb.path + "/kernel"), s.gb.arena.grad(s.gb.path + "/bias")
            if fw:
                gk.copy_(dw[o:o + n])
                gbias.copy_(db[o:o + n])
            else:
                gk.add_(dw[o:o + n])
                gbias.add_(db[o:o + n])
        return ops.conv(d, self.wd, None, ks=1)


class FusedGlobalGB:
    """The gamma | beta Dense projections of ALL global ConditionalBatchNorm sites as one product (round 5).

    The four sites of GenBlock_0 / GenBlock_1 project the same (B, 2 z_dim) condition (xmc_net.py:217-219): one
    (B x 256) x (256 x sum 2C_i) product + one bias broadcast instead of four of each per forward pass, and one weight-gradient
    product, one bias reduction and one data-gradient product (accumulating into the spatial sites' share of d(condition)) instead
    of 4 x (two products + a reduction + an add) in the backward pass -- ~25 launch-floor launches per step.  ``ParamArena`` stores
    the sites' merged kernels transposed, (2C_i, in), back to back: the fused weight, its bias and their gradients are views."""

    def __init__(self, ops, arena, sites):
        self.ops, self.arena, self.sites = ops, arena, sites
        self.off, o = {}, 0
        for s in sites:
            self.off[id(s)] = (o, s.gb.cout)
            o += s.gb.cout
        self.total, self.cin = o, sites[0].gb.cin
        ko = [arena.offset(s.gb.path + "/kernel") for s in sites]
        bo = [arena.offset(s.gb.path + "/bias") for s in sites]
        self.ok = (all(ko[i] + sites[i].gb.cout * self.cin == ko[i + 1] for i in range(len(sites) - 1))
                   and all(bo[i] + sites[i].gb.cout == bo[i + 1] for i in range(len(sites) - 1))
                   and os.environ.get("XMC_GLOBAL_GB_FUSED", "1") != "0")           # (A/B switch; alignment gaps at tiny widths)
        self._ko, self._bo = ko[0], bo[0]
        self.gball = self.dgball = None
        if self.ok:
            for s in sites:
                s.fused = self

    def _views(self, buf):
        return buf[self._ko:self._ko + self.total * self.cin].view(self.total, self.cin), buf[self._bo:self._bo + self.total]

    def fwd(self, cond):
        """cond (B, 2 z_dim) float32 -> (B, sum 2C) float32; every site reads its columns"""
        w, bias = self._views(self.arena.params)
        out = bias.unsqueeze(0).repeat(cond.shape[0], 1)
        self.gball = self.ops.gemm(cond, w, tb=True, beta=1.0, out=out, fast=_dense_fast())
        return self.gball

    def gb_of(self, site):
        o, n = self.off[id(site)]
        return self.gball[:, o:o + n]

    def begin_bwd(self, gball):
        self.gball = gball
        self.dgball = torch.empty_like(gball)        # every site's cbn backward fills its own columns

    def dgb_of(self, site):
        o, n = self.off[id(site)]
        return self.dgball[:, o:o + n]

    def bwd(self, cond, dcond):
        """-> d(cond) (added to ``dcond`` in place when given); writes the GB weight / bias gradients of every site."""
        ops, d = self.ops, self.dgball
        fw = getattr(self.arena, "first_write", False)
        if fw:
            for s in self.sites:
                self.arena.note_write(s.gb.path + "/kernel")
                self.arena.note_write(s.gb.path + "/bias")
        gw, gbias = self._views(self.arena.grads)
        w, _ = self._views(self.arena.params)
        ops.gemm(d, cond, ta=True, beta=0.0 if fw else 1.0, out=gw, fast=_dense_fast())            # dW^T (sum 2C, in) = d^T cond
        ops.reduce_mid(d.reshape(1, d.shape[0], -1), accumulate=not fw, out=gbias.view(1, -1))
        if dcond is None:
            return ops.gemm(d, w, fast=_dense_fast())
        return ops.gemm(d, w, beta=1.0, out=dcond, fast=_dense_fast())


def _dense_fast():
    from ..libml import layers
    return layers._DENSE_FAST


class GenBlock:
    """GenBlock / GenSpatialBlock: norm-relu-up-conv3 - norm-relu-conv3 (+ up-conv1 shortcut)."""

    def __init__(self, ops, arena, path, local):
        self.ops, self.local = ops, local
        nm = "LocalConditionalBatchNorm" if local else "ConditionalBatchNorm"
        self.n0 = CondNorm(ops, arena, f"{path}/{nm}_0", local)
        self.n1 = CondNorm(ops, arena, f"{path}/{nm}_1", local)
        self.c0 = ConvSite(ops, arena, f"{path}/Conv_0")
        self.c0.phase = "ups"                        # conv3x3(upsample(.)): four 2x2 convolutions at the input resolution
        self.c1 = ConvSite(ops, arena, f"{path}/Conv_1")
        self.c2 = ConvSite(ops, arena, f"{path}/Conv_2")

    def prepare(self):
        for s in (self.n0, self.n1, self.c0, self.c1, self.c2):
            s.prepare()

    def fwd(self, x, cond, batch_stats, new_stats, train):
        a0, t0 = self.n0.fwd(x, cond, batch_stats, new_stats, train)
        # shortcut conv1x1(upsample(x)) == upsample(conv1x1(x)) bit for bit: evaluated at the INPUT
        # resolution (1/4 of the MACs and bytes) and nearest-upsampled inside c1's residual epilogue.  Issued right behind the
        # normalisation that read x (round 5: the non-temporal-store experiment showed how much the step lives on L2 / Infinity-Cache
        # hits between neighbouring launches), not three launches later
        sc = self.c2.fwd(x)
        h1 = self.c0.fwd(a0, ups=True)                            # conv3x3(upsample(a0))
        a1, t1 = self.n1.fwd(h1, cond, batch_stats, new_stats, train)
        out = self.c1.fwd(a1, res=sc, res_ups=True)
        return out, (x, a0, a1, t0, t1)

    def bwd(self, tape, dout, dcond):
        ops = self.ops
        x, a0, a1, t0, t1 = tape
        self.c1.wgrad(a1, dout)
        da1 = self.c1.dgrad(dout)
        dout_p = ops.pool2(dout, 1.0)                             # 1x1 conv commutes with the adjoint (third reader of dout in a row)
        dh1, dcond = self.n1.bwd(t1, da1, dcond)
        self.c0.wgrad(a0, dh1, x_ups=True)
        da0 = self.c0.dgrad_sumpool(dh1)                          # adjoint of nearest upsample, fused
        dx, dcond = self.n0.bwd(t0, da0, dcond)
        self.c2.wgrad(x, dout_p)
        dx = self.c2.dgrad(dout_p, res=dx)
        return dx, dcond


# =============================================================================== discriminator
def _bslice(ops, t, lo, hi):
    """batch slice that keeps the ReLU-mask bits a convolution epilogue attached to ``t`` (ops.conv emit_bits)"""
    return ops.bslice(t, lo, hi) if hasattr(ops, "bslice") else t[lo:hi]


_RELU_STORED = os.environ.get("XMC_RELU_STORED", "1") != "0"          # A/B switch
# round 5: the DiscBlock INPUTS too (relu(x) written by the block's pooling pass).  Same-box A/B in the step: 25.78 / 26.00 ms off,
# 25.91 / 26.04 on (G/D-only, profiles/r05_ab_relu_x.txt) -- the in-LDS ReLU of c0's weight gradient is covered by the CU's other
# workgroup at these channel counts, and the extra activation-sized write costs what it saves.  OFF; kept as a tested path.
_RELU_X = os.environ.get("XMC_RELU_X", "0") != "0"


def _relu_stored(ops, c0=None):
    """the discriminator blocks keep h1 = relu(conv0(.)) instead of conv0(.) (identical mathematics: see
    DiscOptimizedBlock.fwd).  Round 5: in the MX-fp8 mode too -- its kernel's epilogue got the bf16 kernel's ReLU-on-store and bit
    masks (xmc_conv2d_mx8_bits); until then the mode switched the stored ReLU off for EVERY block, which cost config #5 more
    than its fp8 layers saved.  ``XMC_FP8_RELU_STORED=0``: only the blocks whose first convolution stays on the bf16 kernel."""
    if not _RELU_STORED:
        return False
    if not getattr(ops, "fp8", False) or _FP8_RELU_STORED:
        return True
    return c0 is not None and hasattr(ops, "takes_mx8") and not ops.takes_mx8(c0.cin, c0.taps)


_FP8_RELU_STORED = os.environ.get("XMC_FP8_RELU_STORED", "1") != "0"


class DiscOptimizedBlock:
    """common.py:117-133 -- conv3, relu, conv3, pool; shortcut pool -> conv1 (no leading ReLU)."""

    def __init__(self, ops, arena, path):
        self.ops = ops
        self.c0 = ConvSite(ops, arena, path + "/SpectralConv_0", spectral=True)
        self.c1 = ConvSite(ops, arena, path + "/SpectralConv_1", spectral=True)
        self.c2 = ConvSite(ops, arena, path + "/SpectralConv_2", spectral=True)
        self.c1.phase = "pool"                       # avg_pool(conv3x3(.)): four 2x2 convolutions on the pooled grid
        self.sites = [self.c0, self.c1, self.c2]

    def fwd(self, x):
        """The image has 3 channels: both of its convolutions run on the tap-expanded 32-channel copy."""
        ops = self.ops
        # h1 is stored AFTER its ReLU (_relu_stored): its three readers -- c1's forward, c1's weight gradient and the mask of
        # c1's data gradient -- only ever see relu(h1) / (h1 > 0), and the weight gradient's in-LDS ReLU pass costs 20-28 %
        # of that kernel (tools/relu_cost.py: 311 vs 243 us at 128^2)
        rs = _relu_stored(ops, self.c0)
        h1, xcol = self.c0.fwd_rgb_in(x, emit_bits=True, relu_out=rs)
        xp = ops.pool2(x, 0.25)
        sc, xpcol = self.c2.fwd_rgb_in(xp)
        # emit_bits: the block output is the ReLU mask of the next block's c0.dgrad (h1's bits come from fwd_rgb_in)
        return self.c1.fwd_pool(h1, res=sc, relu_in=not rs, emit_bits=True), (x, h1, xp, xcol, xpcol)

    def bwd(self, tape, dout, lo, hi, wgrad, need_dx):
        """Backward on the batch slice [lo:hi) of the saved activations."""
        x, h1, xp, xcol, xpcol = (_bslice(self.ops, t, lo, hi) for t in tape)
        if wgrad:
            self.c1.wgrad(h1, dout, x_relu=not _relu_stored(self.ops, self.c0), dy_ups=True, alpha=0.25)
            self.c2.wgrad_rgb_in(xpcol, dout)
        dh1 = self.c1.dgrad(dout, ups=True, alpha=0.25, mask=h1)   # d(avgpool) fused as ups * 1/4
        if wgrad:
            self.c0.wgrad_rgb_in(xcol, dh1)
        if not need_dx:
            return None
        dxp = self.c2.dgrad(dout)
        return self.c0.dgrad(dh1, res=dxp, res_ups=True, res_scale=0.25)


class DiscBlock:
    """common.py:58-79 -- relu, conv3, relu, conv3 (+pool); shortcut conv1 (+pool) / identity."""

    def __init__(self, ops, arena, path, cin, cout, downsample):
        self.ops, self.down = ops, downsample
        self.proj = downsample or cin != cout
        self.c0 = ConvSite(ops, arena, path + "/SpectralConv_0", spectral=True)
        self.c1 = ConvSite(ops, arena, path + "/SpectralConv_1", spectral=True)
        if downsample:
            self.c1.phase = "pool"
        self.sites = [self.c0, self.c1]
        if self.proj:
            self.c2 = ConvSite(ops, arena, path + "/SpectralConv_2", spectral=True)
            self.sites.append(self.c2)

    def fwd(self, x):
        ops = self.ops
        # emit_mx8 (config.conv_fp8 only; ignored otherwise): the consumer of h1 (c1) and of the block output (the next
        # block's c0) are 3x3 convolutions with relu_in -- an MX-fp8 producer writes their packets from its epilogue
        # emit_bits: h1 and the block output are ReLU masks of the backward pass (c1.dgrad / the next block's c0.dgrad):
        # written as bits by the producing epilogue, 1/16 of the bytes the data-gradient epilogues wait for
        rs = _relu_stored(ops, self.c0)              # h1 stored after its ReLU (see DiscOptimizedBlock.fwd)
        # round 5: the block INPUT too.  Its readers are c0's forward (relu_in), c0's weight gradient (x_relu: an in-LDS pass
        # over the DMA-staged patch, 20-28 % of that kernel) and the mask of c0's data gradient (x > 0) -- all three see
        # relu(x) only; the raw x feeds the shortcut alone, through the pooling pass, which therefore writes relu(x) on its way
        # (ops.pool2 relu_copy; the 4 x 4 block without pooling: one elementwise pass over 5 MB).  Tape: xr instead of x.
        xr = None
        rx = rs and _RELU_X
        if self.down:
            if rx:
                xp, xr = ops.pool2(x, 0.25, relu_copy=True)
            else:
                xp = ops.pool2(x, 0.25)              # pool(conv1x1(x)) == conv1x1(pool(x))
        elif rx:
            xr = ops.add_relu(x)
            if getattr(x, "bits", None) is not None:
                xr.bits = x.bits
        h1 = self.c0.fwd(x if xr is None else xr, relu_in=xr is None, relu_out=rs, emit_mx8=True, emit_bits=True)
        xt = x if xr is None else xr
        if self.down:
            sc = self.c2.fwd(xp)
            return self.c1.fwd_pool(h1, res=sc, relu_in=not rs, emit_mx8=True, emit_bits=True), (xt, h1, xp)
        sc = self.c2.fwd(x) if self.proj else x
        return self.c1.fwd(h1, relu_in=not rs, res=sc, emit_mx8=True, emit_bits=True), (xt, h1, x if self.proj else None)

    def bwd(self, tape, dout, lo, hi, wgrad):
        """dout: gradient wrt the block output for samples [lo:hi) of the saved activations.  tape = (x or relu(x), h1, the
        shortcut convolution's input)."""
        x, h1, xp = (_bslice(self.ops, t, lo, hi) if t is not None else None for t in tape)
        rs = _relu_stored(self.ops, self.c0)
        rx = rs and _RELU_X                          # the tape holds relu(x)
        if self.down:
            if wgrad:
                self.c1.wgrad(h1, dout, x_relu=not rs, dy_ups=True, alpha=0.25)
                self.c2.wgrad(xp, dout)
            dh1 = self.c1.dgrad(dout, ups=True, alpha=0.25, mask=h1, emit_mx8=False)     # consumers: c0.dgrad ...
            dxp = self.c2.dgrad(dout)                                                    # (dout's readers back to back)
            if wgrad:
                self.c0.wgrad(x, dh1, x_relu=not rx)
            return self.c0.dgrad(dh1, mask=x, res=dxp, res_ups=True, res_scale=0.25, emit_mx8=False)   # ... the previous block's c1.dgrad
        if wgrad:
            self.c1.wgrad(h1, dout, x_relu=not rs)
            if self.proj:
                self.c2.wgrad(xp, dout)
        dh1 = self.c1.dgrad(dout, mask=h1, emit_mx8=False)
        if wgrad:
            self.c0.wgrad(x, dh1, x_relu=not rx)
        dsc = self.c2.dgrad(dout) if self.proj else dout
        return self.c0.dgrad(dh1, mask=x, res=dsc, emit_mx8=False)
