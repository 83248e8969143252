"""Operator table of the XMC-GAN step: thin, allocation-only wrappers that launch the gfx950
kernels of ``libxmcgan_hip.so`` on ``torch.cuda.current_stream()``.

PyTorch is plumbing here: it owns device memory (caching allocator, so a step is
hipGraph-capturable), streams and ``torch.distributed``; every arithmetic op below is a
hand-written HIP kernel reached through the C ABI in ``include/xmcgan_hip.h``.  There is no
fallback implementation -- constructing ``HipOps`` without the library or without a GPU
raises.

Tensor conventions: activations NHWC in ``self.dtype`` (float32 or bfloat16), parameters /
gradients / statistics float32, all tensors contiguous.
"""
from __future__ import annotations

import ctypes as C
import math
import os

import torch

from . import _lib
from ._lib import ConvDesc, WgradDesc, XMC_BF16, XMC_F32, check


def _code(dt):
    if dt == torch.float32:
        return XMC_F32
    if dt == torch.bfloat16:
        return XMC_BF16
    raise TypeError(f"unsupported dtype {dt}")


def _p(t):
    if t is None:
        return None
    assert t.is_contiguous(), "ops take contiguous tensors"
    return C.c_void_p(t.data_ptr())


class PackedWeight:
    """Prepared conv weights in MFMA-fragment order (include/xmcgan_hip.h: xmc_pack_conv_weight)."""
    __slots__ = ("data", "cout", "taps", "cin", "mx8", "phase", "lazy")

    def __init__(self, data, cout, taps, cin):
        self.data, self.cout, self.taps, self.cin = data, cout, taps, cin
        # a phase-only site (data is None): (float32 master, inv_sigma or None, 0 forward | 1 dgrad) -- HipOps.conv makes the
        # plain 3x3 copy from it the first time a launch falls outside the phase kernels' domain
        self.lazy = None
        self.mx8 = None          # (w8, wscale): the MX-fp8 copy, made on first use by HipOps.conv when ops.fp8 is set
        self.phase = None        # ("out" | "in", 16-tap phase weights): xmc_phase_conv_weight, used by ups / pool_out launches


class HipOps:
    """The MI355X backend (the only product backend)."""

    name = "hip-gfx950"

    def __init__(self, dtype=torch.bfloat16, device=None, wgrad_variant=1, stream_conv=None, wgrad_async=None):
        if not torch.cuda.is_available():
            raise _lib.XmcError("HipOps needs a ROCm GPU (torch.cuda.is_available() is False); "
                                "there is no CPU fallback")
        self.lib = _lib.load()
        self.dtype = dtype
        self.code = _code(dtype)
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        # XMC_WGRAD_TUNE: A/B knob for the split-K target / launch order of conv_wgrad_dma.hip (tools/bench_conv.py --wgrad-tunes)
        self.wgrad_variant = wgrad_variant | (int(os.environ.get("XMC_WGRAD_TUNE", "0")) << 4)
        if os.environ.get("XMC_WGRAD_NST3", "0") != "0":                       # A/B: three-stage ring on the 4 x 4 maps too (variant bit 13)
            self.wgrad_variant |= 0x2000
        if os.environ.get("XMC_WGRAD_C96", "1") == "0":                        # A/B: no 96-cout tiles in conv_wgrad_dma (variant bit 11)
            self.wgrad_variant |= 0x800
        # conv3x3 next to a 2x resampling as four 2x2 convolutions (conv_phase_kernel); XMC_PHASE_CONV=0: A/B switch
        self.phase_conv = os.environ.get("XMC_PHASE_CONV", "1") != "0"
        self.phase4 = os.environ.get("XMC_PHASE4", "1") != "0"                 # "out" form: phases as waves (0: as workgroups; A/B)
        self.attn_mfma = os.environ.get("XMC_ATTN_MFMA", "1") != "0"           # attention_for_g on MFMA tiles in the bf16 mode (A/B)
        self.wl_fused = os.environ.get("XMC_WL_FUSED", "1") != "0"             # word_loss fused on the matrix cores in the bf16 mode (A/B)
        self.px128 = os.environ.get("XMC_PHASE_PX128", "1") != "0"            # ... on 128-pixel x 64-cout tiles where they fit (A/B)
        self.mask_bits = os.environ.get("XMC_MASK_BITS", "1") != "0"           # ReLU masks as bits in the conv epilogues (A/B)
        self.compact_pw = os.environ.get("XMC_RESNET_COMPACT", "1") != "0"     # ResNet-50 1x1 layers on the valid corner of their canvases
        self.tile64 = os.environ.get("XMC_TILE64", "1") != "0"                # A/B: 64-cout tiles on unsplit few-tile 3x3 launches
        self.tile32 = os.environ.get("XMC_TILE32", "1") != "0"                # A/B: 32-cout tiles for the <= 32-channel outputs (to-RGB)
        self.pw_variant = int(os.environ.get("XMC_PW_VARIANT", "0"))          # A/B: pointwise kernel variant bits (w_packed 12-15)
        self.no_split_k = os.environ.get("XMC_NO_SPLIT_K", "0") != "0"        # A/B: forward / dgrad convolutions without split-K
        # MX-fp8 mode: the resampling-adjacent layers stay on the bf16 phase kernels (2.25x fewer MFMAs; XMC_FP8_PHASE=0: the fp8
        # 3x3 kernel there too, 2x the MFMA rate).  Round 3 measured the fp8 kernel ahead (C4 53.7 vs 54.5 ms); with round 4's phase
        # kernels it is behind -- round 5, same box, alternated (profiles/r05_c4_vs_c3.txt): C3 bf16 45.2 ms, C4 with fp8 everywhere
        # 46.8 / 47.0, C4 with the bf16 phase kernels 45.5 / 45.7 -> ON.  (Weight gradients are bf16 in either mode.)
        self.fp8_phase = os.environ.get("XMC_FP8_PHASE", "1") != "0"
        # race hunt: 1 = every MX convolution quantises its input itself (producer packets ignored), 2 = the
        # conditional-BatchNorm kernel writes no packets, 4 = the convolution epilogues write none
        self.fp8_debug = int(os.environ.get("XMC_FP8_DEBUG", "0"))
        # smallest channel count that takes the MX-fp8 kernel (the quantisation pass scales with the pixels, the matrix work it saves
        # with the channels: tools/bench_conv.py per layer, profiles/r05_conv_layers_mx_fp8.txt)
        self.fp8_min_cin = int(os.environ.get("XMC_FP8_MIN_CIN", "64"))
        # A/B knobs of the library's launch heuristics: the C side reads no environment; the measurement scripts' XMC_* variables
        # are forwarded HERE through xmc_set_tuning (tools/ab_env_values.sh)
        for env, key in (("XMC_KSPLIT_TARGET", "ksplit_target"), ("XMC_KSPLIT_TARGET_PHASE", "ksplit_target_phase"),
                         ("XMC_KSPLIT_TARGET_PW", "ksplit_target_pw"), ("XMC_TILE64_PCT", "tile64_pct"),
                         ("XMC_WGRAD_TARGET_HI", "wgrad_target_hi"), ("XMC_WGRAD_TARGET_LO", "wgrad_target_lo"),
                         ("XMC_WGRAD_TARGET_PHASE", "wgrad_target_phase"), ("XMC_CBN_RUN", "cbn_run")):
            if os.environ.get(env) is not None:
                check(self.lib.xmc_set_tuning(key.encode(), int(os.environ[env])), f"xmc_set_tuning({key})")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        # per-device handle of the C ABI: validates gfx950 and opts the kernels in to the 160 KiB LDS on this device
        self._handle = C.c_void_p()
        check(self.lib.xmc_create(self._dev_index, C.byref(self._handle)), f"xmc_create(device {self._dev_index})")
        # 3x3 bf16 convolutions on the weight-streaming kernel (prepared weights in MFMA-fragment order);
        # XMC_CONV_STREAM=0 keeps every layer on the LDS-staged kernels (A/B benchmarks)
        self.stream_conv = (os.environ.get("XMC_CONV_STREAM", "1") != "0") if stream_conv is None else stream_conv
        # weight-gradient launches on their own HIP stream: nothing but the optimiser consumes them, so they run
        # beside the data-gradient chain and fill the CUs its small-grid / tail phases leave idle (join_wgrad)
        # (off by default: xmc_gan overlaps the two pullbacks of train_g_d instead and switches this on where it pays)
        self.wgrad_async = (os.environ.get("XMC_WGRAD_ASYNC", "0") != "0") if wgrad_async is None else wgrad_async
        self._wg_stream = None
        self._wg_keep = []
        # deterministic reductions: split-K weight gradients, bias gradients and pooled sums go through caller-owned
        # workspaces and fixed-order second stages instead of float atomics (bit-reproducible gradients);
        # XMC_DETERMINISTIC=0 restores the single-pass atomic variants (A/B benchmarks)
        self.deterministic = os.environ.get("XMC_DETERMINISTIC", "1") != "0"
        # BASELINE config #5 (config.conv_fp8): the 3x3 convolutions (forward + data gradient) multiply MX-fp8 operands
        # -- e4m3 elements, one e8m0 scale per 32 channels, block-scaled MFMA with float32 accumulation; weight
        # gradients, 1x1 / RGB layers, normalisation, attention and every loss stay as in the bf16 mode
        self.fp8 = False
        # round 4: 1 / sigma of the spectrally-normalised layers rides in the convolution's alpha (xmc_conv_desc.alpha_dev), so
        # the prepared weights are a pure cast of W: one batched pass per forward writes every copy AND the first product of
        # the power iteration (xmc_wprep_batched); XMC_FOLD_SIGMA=0: the round-3 path (A/B, and the float32 parity mode)
        self.fold_sigma = (os.environ.get("XMC_FOLD_SIGMA", "1") != "0") and dtype == torch.bfloat16
        # ... and the gradient through sigma + the zeroing of the consumed gradient ride in the Adam kernel
        # (xmc_adam_ema_dev_sn); keep_grads: write the FINAL gradient back instead of zeros (tests / tools that read the
        # gradient arenas after a step) -- the next half step then zero-fills as before
        self.fuse_opt = (os.environ.get("XMC_FUSE_OPT", "1") != "0") and dtype == torch.bfloat16
        self.keep_grads = os.environ.get("XMC_KEEP_GRADS", "0") != "0"
        # round 5: every gradient tensor of a half step has exactly ONE producing launch, so that launch WRITES it
        # (XMC_WGRAD_OVERWRITE, gemm beta = 0) instead of adding into a zeroed arena: the optimiser kernel neither zeroes what it
        # consumed (4 B/param less) nor do the reducing passes read the old value (4 B/param less); ParamArena audits on the
        # first update that every leaf was written.  XMC_FIRST_WRITE=0: the round-4 path (A/B)
        self.first_write = self.fuse_opt and self.deterministic and os.environ.get("XMC_FIRST_WRITE", "1") != "0"
        # round 5: the optimiser kernel emits the prepared weight copies of the batched-preparation tables while it holds the new
        # W (xmc_adam_wprep_tiles): no separate pass re-reads the masters Adam just wrote.  XMC_FUSE_PREP=0: A/B
        # Measured (profiles/r05_ab_fuse_prep.txt, same box): G/D-only 26.05 / 25.88 ms with the separate pass, 26.16 / 26.29 fused --
        # the separate pass runs on the side stream beside the generator's forward pass, the fused copies sit in the optimiser
        # kernels on the critical path.  OFF by default; bit-identical either way (tests/test_gpu_fused_opt.py).
        self.fuse_prep = self.fuse_opt and self.fold_sigma and os.environ.get("XMC_FUSE_PREP", "0") != "0"
        # round 5: the gamma / beta maps of the LOCAL conditional-BatchNorm sites (one fused 1024 -> 4,224 projection at 16 x 16)
        # and their gradients in bf16 -- what the reference's nn.Conv(dtype=bfloat16) produces (layers.py:261-273) -- instead of
        # float32: 242 MB per map less to write and re-read in every pass, and no cast in front of the projection's backward
        self.gb_bf16 = dtype == torch.bfloat16 and os.environ.get("XMC_GB_BF16", "1") != "0"

    def resnet_step_mode(self):
        """may ResNet50Features.forward(reuse_buffers=True) take the training step's launches -- compact pointwise launches into
        persistent buffers, dual-source launches, the fused stem?  (bf16 weight-streaming path only: the float32 parity mode and the
        fp8 mode keep fresh tensors and the reference-shaped launches)"""
        return bool(getattr(self, "compact_pw", False) and self.dtype == torch.bfloat16 and self.stream_conv and not self.fp8)

    def set_fp8_scale_rule(self, rule):
        """MX-fp8 scale rule of every quantiser (xmc_set_tuning("mx8_scale_floor"), process-wide like the other knobs): "next_binade"
        = X one binade above the OCP conversion when the block maximum would saturate e4m3 (the default: lower RMS error, no
        systematic shrink of post-ReLU activations); "ocp_floor" = the OCP MX v1.0 conversion exactly (BASELINE config #5's wording)."""
        if rule not in ("next_binade", "ocp_floor"):
            raise ValueError(f"fp8_scale_rule must be 'next_binade' or 'ocp_floor', not {rule!r}")
        check(self.lib.xmc_set_tuning(b"mx8_scale_floor", 1 if rule == "ocp_floor" else 0), "xmc_set_tuning(mx8_scale_floor)")
        self.fp8_scale_rule = rule

    def __del__(self):
        h = getattr(self, "_handle", None)
        if h is not None and h.value:
            self.lib.xmc_destroy(h)
            self._handle = None

    # ------------------------------------------------------------------ allocation helpers
    def _stream(self):
        # raw hipStream_t of torch's current stream; the C-level getter is ~20x cheaper than
        # torch.cuda.current_stream().cuda_stream and this runs once per kernel launch (~2,000 per step)
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(self._dev_index))

    # ------------------------------------------------------------------ side stream (overlap)
    def side(self, which=0):
        """Context manager: work issued inside runs on a side HIP stream that first waits for
        everything already queued on the current stream.  Pair with ``join_side``.  ``which``: 0 = the step's main side
        stream (prefetched forward passes, the g-stream of train_g_d), 1 = a second one for short chains that must not queue
        behind those (the discriminator's generator-side loss heads)."""
        sides = self.__dict__.setdefault("_sides", {})
        if which not in sides:
            sides[which] = torch.cuda.Stream(device=self.device)
        if which == 0:
            self._side = sides[0]
        sides[which].wait_stream(torch.cuda.current_stream())
        return torch.cuda.stream(sides[which])

    def join_side(self, tensors=(), which=0):
        """Make the current stream wait for the side stream; ``tensors`` produced there are marked
        as used by the current stream (caching-allocator safety)."""
        cur = torch.cuda.current_stream()
        cur.wait_stream(self._sides[which])
        for t in tensors:
            if t is not None:
                t.record_stream(cur)

    def record_event(self):
        """an event on the current stream (inside ``side()``: the side stream); pair with ``wait_event`` on another stream"""
        ev = torch.cuda.Event()
        ev.record()
        return ev

    def wait_event(self, ev):
        torch.cuda.current_stream().wait_event(ev)

    def empty(self, shape, dtype=None):
        return torch.empty(shape, dtype=dtype or self.dtype, device=self.device)

    def begin_pool(self, nbytes=4 << 20):
        """One zero-filled scratch pool per half step: the ~50 small float32 accumulators a step needs (BatchNorm
        sums, loss slots, ...) become slices of ONE memset instead of 50 fill kernels."""
        self._pool = torch.zeros((nbytes // 4,), dtype=torch.float32, device=self.device)
        self._pool_off = 0

    def zeros(self, shape, dtype=torch.float32):
        pool = getattr(self, "_pool", None)
        if pool is not None and dtype == torch.float32:
            n = int(math.prod(shape)) if not isinstance(shape, int) else int(shape)
            end = self._pool_off + ((n + 63) & ~63)              # 256-byte granules
            if 0 < n <= 65536 and end <= pool.numel():
                out = pool[self._pool_off:self._pool_off + n].view(shape)
                self._pool_off = end
                return out
        return torch.zeros(shape, dtype=dtype, device=self.device)

    # ------------------------------------------------------------------------------- convolution
    def can_pool_out(self, x, w, ups=False):
        """fused 2x2 average pooling of the conv output: weight-streaming kernel only -- as four 2x2 convolutions on the
        pooled grid when the weight carries its phase copy, else in the 3x3 kernel's epilogue (rows of >= 32 pixels)"""
        if not (isinstance(w, PackedWeight) and w.taps == 9):
            return False
        if self._phase_ok(w, "in", x.shape[1], x.shape[2], ups, True):
            return True
        return (2 if ups else 1) * x.shape[2] >= 32

    def _conv_pw_dual(self, x, x2, w, bias, *, ks, res, relu_out, valid, emit_bits, compact, out, x2_stride, alpha, res_scale, mask=None,
                      mask_after_res=False):
        """y = epilogue([x | x2(s y, s x)] W^T) on conv_pw_kernel's DUAL instantiation: the down-sampling bottleneck's
        relu(bn3(conv3(h)) + proj_bn(proj_conv(x_in))) (resnet_v1.py:74-86) as ONE reduction over the concatenated channels;
        ``x2_stride`` = -2: x2 at the even pixels, zeros elsewhere -- the same block's data gradient [dh1 | scatter2(g)] [W1^T | Wp^T]^T"""
        n, hi, wi, cin = x.shape
        assert isinstance(w, PackedWeight) and ks == 1 and w.taps == 1 and w.cin == cin + x2.shape[-1] and w.data is not None
        assert compact and valid and out is not None and x.dtype == x2.dtype == self.dtype and x2.shape[0] == n
        assert tuple(out.shape) == (n, hi, wi, w.cout) and out.dtype == self.dtype and out.is_contiguous() and x2.is_contiguous()
        assert res is None or tuple(res.shape) == tuple(out.shape)
        assert mask is None or (tuple(mask.shape) == tuple(out.shape) and mask.dtype == self.dtype)
        self.last_conv_phase = False
        d = ConvDesc(n, hi, wi, cin, w.cout, 1, 0, 0, 0, 0, self.code, float(alpha), float(res_scale), 1 | 64 | ((getattr(self, "pw_variant", 0) & 15) << 12),
                     0, int(relu_out), int(mask_after_res), int(valid), int(valid), None)
        ybits = mbits = None
        if self.mask_bits and w.cout % 16 == 0:
            mbits = getattr(mask, "bits", None) if mask is not None else None
            if emit_bits:
                ybits = torch.empty((n, hi, wi, w.cout // 16), dtype=torch.int16, device=self.device)
        check(self.lib.xmc_conv2d_pw_dual(C.byref(d), _p(x), _p(x2), x2.shape[-1], x2.shape[1], x2.shape[2], int(x2_stride), _p(w.data), _p(bias),
                                          _p(mask), _p(res), _p(out), _p(mbits), _p(ybits), self._stream()), "xmc_conv2d_pw_dual")
        if ybits is not None:
            out.bits = ybits
        return out

    def can_stride2(self, w, hi, wi):
        """may ``conv(..., stride2=True)`` run with this weight on an input of (hi, wi) pixels?  (phase copies of kind
        "s2in" / "s2out" from ``prep_conv_weight(..., phase="s2")``; power-of-two low-resolution grid)"""
        if not isinstance(w, PackedWeight) or w.phase is None or w.phase[0] not in ("s2in", "s2out"):
            return False
        fwd = w.phase[0] == "s2in"
        return self._phase_ok(w, w.phase[0], hi, wi, not fwd, fwd)

    def _phase_flags(self):
        return 1 | 16 | (32 if not self.phase4 else 0) | (128 if not getattr(self, "px128", True) else 0)

    def _phase_ok(self, w, kind, hi, wi, ups, pool_out):
        """may this launch run phase-decomposed (conv_phase_kernel)?  needs the 16-tap copy of the right kind and exactly
        one of ups / pool_out; the fp8 mode keeps its own kernels.  The geometry test is the C side's own
        (xmc_conv2d_phase_supported: one source of truth for the tile / patch limits)."""
        if not self.phase_conv or (self.fp8 and not self.fp8_phase) or w.phase is None or w.phase[0] != kind or bool(ups) == bool(pool_out):
            return False
        d = ConvDesc(1, hi, wi, w.cin, w.cout, 3, int(bool(ups)), 0, 0, 0, self.code, 1.0, 1.0, self._phase_flags(), int(bool(pool_out)), 0, 0, 0, 0)
        return bool(self.lib.xmc_conv2d_phase_supported(C.byref(d)))

    def conv(self, x, w, bias=None, *, ks, ups=False, relu_in=False, mask=None, res=None, res_ups=False,
             res_scale=1.0, alpha=1.0, out_f32=False, pool_out=False, relu_out=False, mask_after_res=False, valid=0,
             emit_mx8=None, stride2=False, emit_bits=False, compact=False, out=None, alpha_dev=None, x2=None, x2_stride=1):
        """xmc_conv2d_nhwc (include/xmcgan_hip.h).  ``stride2`` (see ``can_stride2``): the weight carries the phase copies of
        a stride-2 SAME convolution -- a forward weight gives y (n, hi/2, wi/2, cout) = conv_s2(x), a dgrad weight gives the
        adjoint (n, 2 hi, 2 wi, cout); both run on conv_phase_kernel at the low resolution.  ``emit_bits``: y will serve as
        the ReLU ``mask`` of a later data-gradient launch -- where the kernel can, it also writes (y > 0) as bits
        (``y.bits``, one uint16 per 16 channels), and a launch whose ``mask`` carries ``.bits`` reads those instead of the
        bf16 tensor (``ops.bslice`` keeps them through batch slicing).  ``compact`` (pointwise launches with ``valid``): only
        the valid corner of every canvas is processed -- the margins of ``out`` (a caller-owned, zero-initialised buffer that
        is reused from step to step) are never written.  ``relu_out`` / ``mask_after_res`` / ``valid`` (side of the live
        top-left region; the rest of every image is stored as zero) serve the frozen ResNet-50's canvases.
        ``emit_mx8`` (True / False = the relu_in of the NEXT 3x3 convolution; None = no hint): in the MX-fp8 mode the
        result then carries its fp8 packets (``y.mx8``), written by this launch's epilogue where the kernel can.
        ``alpha_dev`` (float32 device scalar, optional): multiplied into ``alpha`` by the kernel -- 1 / (sigma + eps) of a
        spectrally-normalised layer whose prepared weights are a pure cast of W (``fold_sigma``).
        ``x2`` (compact pointwise launches only, xmc_conv2d_pw_dual): a second source (n, h2, w2, c2) whose pixel
        (x2_stride * y, x2_stride * x) is concatenated behind x's channels -- ``w`` then has cin + c2 input channels."""
        if x2 is not None:
            return self._conv_pw_dual(x, x2, w, bias, ks=ks, res=res, relu_out=relu_out, valid=valid, emit_bits=emit_bits, compact=compact,
                                      out=out, x2_stride=x2_stride, alpha=alpha, res_scale=res_scale, mask=mask, mask_after_res=mask_after_res)
        n, hi, wi, cin = x.shape
        packed = isinstance(w, PackedWeight)
        cout = w.cout if packed else w.shape[0]
        wobj = w
        if stride2:
            # the stride-2 convolution / its adjoint through the "in" / "out" phase kernel: 16 entries per low-resolution
            # pixel (7 of them zero) instead of the 36 of "stride 1, then sub-sample"
            assert packed and ks == 3 and w.phase is not None and w.phase[0] in ("s2in", "s2out") and not (ups or pool_out)
            pool_out, ups = w.phase[0] == "s2in", w.phase[0] == "s2out"
            if pool_out:
                alpha = 4.0 * alpha                      # the kernel's "in" form carries the 1/4 of the average pooling
        if packed:
            assert (w.taps, w.cin) == (ks * ks, cin)
            w = w.data                                   # None: a phase site whose 3x3 copies were never made
        else:
            assert w.shape[1] == ks * ks and w.shape[2] == cin
        assert x.dtype == self.dtype
        ho, wo = (2 * hi, 2 * wi) if ups else (hi, wi)
        if pool_out:
            assert packed and mask is None and not res_ups, "pool_out: see can_pool_out"
            ho, wo = ho // 2, wo // 2                    # shape of y (and of res)
        # conv3x3(upsample2(.)) / avg_pool2(conv3x3(.)) as four 2x2 convolutions on the low-resolution grid (2.25x fewer MFMAs)
        phase = (packed and ks == 3 and not (res_ups or mask_after_res or valid) and (stride2 or not relu_out)
                 and ((ups and res is None and self._phase_ok(wobj, "s2out" if stride2 else "out", hi, wi, True, False))
                      or (pool_out and mask is None and self._phase_ok(wobj, "s2in" if stride2 else "in", hi, wi, False, True))))
        assert phase or not stride2, "stride2: see can_stride2"
        self.last_conv_phase = bool(phase)               # bench.py: this launch executes 4/9 of the 3x3 formulation's MFMAs
        if phase:
            w = wobj.phase[1]
        if w is None and packed and wobj.lazy is not None:
            # a phase-only site reached by a launch outside the phase kernels' domain (a switch toggled after the weights were
            # prepared, relu_out / res / valid set, a 2 x 2 grid): make its plain 3x3 copy now, on THIS stream (a fallback: the
            # product's schedule never takes it, and a copy made here is not ordered against other streams' launches)
            master, inv, idx = wobj.lazy
            made = self.prep_conv_weight(master, inv, True)[idx]
            wobj.data, wobj.mx8 = made.data, made.mx8
            w = wobj.data
        if w is None:
            raise _lib.XmcError("this convolution site has only its phase copies (conv3x3 next to a 2x resampling), but the "
                                "launch is outside the phase kernels' domain")
        assert w.dtype == self.dtype
        if out is not None:
            assert tuple(out.shape) == (n, ho, wo, cout) and out.dtype == (torch.float32 if out_f32 else self.dtype) and out.is_contiguous()
            y = out
        else:
            y = self.empty((n, ho, wo, cout), torch.float32 if out_f32 else self.dtype)
        # ... and on a 3x3 launch: tiles entirely inside the canvas margin are skipped and the margin is NOT zeroed (its consumers are
        # compact pointwise launches that read the valid corner only)
        compact3 = bool(compact and packed and ks == 3 and valid and not phase and not stride2 and not (ups or pool_out) and not self.fp8)
        compact = bool(compact and packed and ks == 1 and valid and out is not None and not self.fp8)
        if mask is not None:
            assert mask.shape == y.shape and mask.dtype == self.dtype
        if res is not None:
            assert res.dtype == self.dtype
            assert tuple(res.shape) == ((n, ho // 2, wo // 2, cout) if res_ups else (n, ho, wo, cout))
        # MX-fp8 where it pays: rows are padded to 64 channels, so a 96-channel input would do 128 channels of work and its
        # (large, 128^2) tensor would pay the quantisation pass on top -- measured 0.74x the bf16 kernel; those stay bf16
        if (self.fp8 and not phase and packed and ks == 3 and self.dtype == torch.bfloat16 and not (mask_after_res or valid)
                and not (relu_out and pool_out)
                and cout % 4 == 0 and cin % 8 == 0 and ((cin % 64 == 0 and cin >= self.fp8_min_cin) or self.fp8 == "all") and self._mx8_patch_fits(ho * (2 if pool_out else 1), wo * (2 if pool_out else 1))):
            return self._conv_mx8(x, wobj, bias, y, ups=ups, relu_in=relu_in, mask=mask, res=res, res_ups=res_ups,
                                  res_scale=res_scale, alpha=alpha, out_f32=out_f32, pool_out=pool_out, emit=emit_mx8, alpha_dev=alpha_dev,
                                  relu_out=relu_out, emit_bits=emit_bits)
        d = ConvDesc(n, hi, wi, cin, cout, ks, int(ups), int(relu_in), int(res_ups), int(out_f32), self.code,
                     float(alpha), float(res_scale), int(packed) | (16 if phase else 0) | (32 if phase and not self.phase4 else 0) | (128 if phase and not getattr(self, "px128", True) else 0) | (64 if compact or compact3 else 0) | (256 if packed and getattr(self, "force_tile128", False) else 0) | (512 if packed and getattr(self, "force_tile96", False) else 0) | (1024 if packed and not self.tile64 else 0) | (2048 if packed and not getattr(self, "tile32", True) else 0) | ((getattr(self, "pw_variant", 0) & 15) << 12 if packed else 0),
                     int(pool_out), int(relu_out), int(mask_after_res), int(valid), int(valid),        # (bit 8: A/B switch, bench_conv.py)
                     alpha_dev.data_ptr() if alpha_dev is not None else None)
        ws_bytes = self.lib.xmc_conv2d_workspace_bytes(C.byref(d)) if packed and not getattr(self, "no_split_k", False) else 0
        ws = self.empty((ws_bytes // 4,), torch.float32) if ws_bytes else None      # split-K scratch (few-tile layers)
        mbits = ybits = None
        if packed and self.mask_bits and cout % 16 == 0:
            mb = getattr(mask, "bits", None) if mask is not None else None
            if mb is not None and not ws_bytes:
                mbits = mb
            if emit_bits and not ws_bytes and not out_f32:
                ybits = torch.empty((n, ho, wo, cout // 16), dtype=torch.int16, device=self.device)
        check(self.lib.xmc_conv2d_nhwc_bits(C.byref(d), _p(x), _p(w), _p(bias), _p(mask), _p(res), _p(y), _p(ws), _p(mbits),
                                            _p(ybits), self._stream()), "xmc_conv2d_nhwc_bits")
        if ybits is not None:
            y.bits = ybits
        return y

    @staticmethod
    def bslice(t, lo, hi):
        """t[lo:hi] along the batch axis, keeping the ReLU-mask bits (``t.bits``) a convolution epilogue attached"""
        s = t[lo:hi]
        b = getattr(t, "bits", None)
        if b is not None:
            s.bits = b[lo:hi]
        return s

    # ------------------------------------------------------------------ MX-fp8 convolution (config.conv_fp8)
    def quantize_mx8(self, x, relu=False):
        """bf16 (..., c) -> x8 uint8 (pixels, cp / 64, 80): OCP MX blocks of 32 channels (e4m3 elements, e8m0 scales), rows
        zero-padded to a multiple of 64 channels; per 64-channel chunk one 80-byte packet = 64 elements + the two scale
        bytes (bytes 64, 65) + pad.  ``relu`` applies max(., 0) first."""
        assert x.dtype == torch.bfloat16 and x.is_contiguous()
        c = x.shape[-1]
        pixels, cp = x.numel() // c, (c + 63) // 64 * 64
        x8 = torch.empty((pixels, cp // 64, 80), dtype=torch.uint8, device=self.device)
        check(self.lib.xmc_mx8_quantize(_p(x), _p(x8), pixels, c, int(relu), self._stream()), "xmc_mx8_quantize")
        return x8

    def pack_mx8(self, w):
        """PackedWeight (bf16 fragment order, 9 taps) -> (w8, wscale) in the MX-fp8 fragment order of xmc_conv2d_mx8"""
        assert isinstance(w, PackedWeight) and w.taps == 9
        nrb, nc64 = (w.cout + 31) // 32, (w.cin + 63) // 64
        w8 = torch.empty((nrb * nc64 * 9 * 2048,), dtype=torch.uint8, device=self.device)
        wsc = torch.zeros((nrb * nc64 * 3 * 256,), dtype=torch.uint8, device=self.device)
        check(self.lib.xmc_mx8_pack_conv_weight(_p(w.data), _p(w8), _p(wsc), w.cout, 9, w.cin, self._stream()),
              "xmc_mx8_pack_conv_weight")
        return w8, wsc

    @staticmethod
    def _mx8_patch_fits(ho, wo):
        """the MX-fp8 kernel stages a 256-pixel tile's patch as 5 vectors per pixel in 2,048 slots (conv_stream_mx8.hip): the
        4x4 maps (16 images per tile, 576 patch pixels) do not fit and stay on the bf16 kernel"""
        wt = min(wo, 64)
        rt = min(256 // wt, ho)
        imgs = 256 // (wt * rt)
        return imgs * (rt + 2) * (wt + 2) * 5 <= 2048

    def takes_mx8(self, cin, taps=9):
        """does a 3x3 convolution with ``cin`` input channels run on the MX-fp8 kernel in this mode?  (the rule of ``conv``)"""
        return bool(self.fp8) and taps == 9 and self.dtype == torch.bfloat16 and (
            (cin % 64 == 0 and cin >= self.fp8_min_cin) or self.fp8 == "all")

    def _with_mx8(self, w):
        """MX-fp8 copy of a freshly prepared weight, made HERE -- on the stream that prepared the bf16 copy, which every
        consumer stream already waits for -- and not lazily at first use: the two pullbacks of train_g_d run the same
        dgrad weights on two streams, and a copy made by one would be read by the other before its kernel ran."""
        if self.fp8 and w.taps == 9 and ((w.cin % 64 == 0 and w.cin >= self.fp8_min_cin) or self.fp8 == "all"):
            w.mx8 = self.pack_mx8(w)
        return w

    def _conv_mx8(self, x, w, bias, y, *, ups, relu_in, mask, res, res_ups, res_scale, alpha, out_f32, pool_out, emit=None, alpha_dev=None,
                  relu_out=False, emit_bits=False):
        n, hi, wi, cin = x.shape
        if w.mx8 is None:                # weights prepared before ops.fp8 was set (tests, benchmarks): single-stream use only
            w.mx8 = self.pack_mx8(w)
        pre = getattr(x, "mx8", None)    # packets written by the producing convolution's epilogue (same relu_in)?
        if self.fp8_debug & 1:
            pre = None
        # (packets of a tensor stored AFTER its ReLU -- tag "relu" -- serve either relu_in: max(., 0) is idempotent)
        x8 = pre[0] if pre is not None and (pre[1] == "relu" or pre[1] == bool(relu_in)) else self.quantize_mx8(x, relu=relu_in)
        d = ConvDesc(n, hi, wi, cin, w.cout, 3, int(ups), 0, int(res_ups), int(out_f32), self.code, float(alpha),
                     float(res_scale), 1, int(pool_out), int(relu_out), 0, 0, 0, alpha_dev.data_ptr() if alpha_dev is not None else None)
        ws_bytes = self.lib.xmc_conv2d_mx8_workspace_bytes(C.byref(d)) if not getattr(self, "no_split_k", False) else 0
        ws = self.empty((ws_bytes // 4,), torch.float32) if ws_bytes else None
        y8 = None
        if (emit is not None and not ws_bytes and not out_f32 and w.cout % 64 == 0 and w.cout >= self.fp8_min_cin and self._mx8_patch_fits(y.shape[1], y.shape[2])
                and not self.fp8_debug & 4):
            y8 = torch.empty((y.numel() // w.cout, w.cout // 64, 80), dtype=torch.uint8, device=self.device)
        mbits = ybits = None             # ReLU masks as bits, as the bf16 kernel's epilogue reads / writes them (HipOps.conv)
        if self.mask_bits and w.cout % 16 == 0 and not ws_bytes:
            mb = getattr(mask, "bits", None) if mask is not None else None
            if mb is not None and not pool_out:
                mbits = mb
            if emit_bits and not out_f32:
                ybits = torch.empty(tuple(y.shape[:-1]) + (w.cout // 16,), dtype=torch.int16, device=self.device)
        check(self.lib.xmc_conv2d_mx8_bits(C.byref(d), _p(x8), _p(w.mx8[0]), _p(w.mx8[1]), _p(bias), _p(mask), _p(res),
                                           _p(y), _p(y8), int(bool(emit)), _p(ws), _p(mbits), _p(ybits), self._stream()),
              "xmc_conv2d_mx8_bits")
        if y8 is not None:
            y.mx8 = (y8, "relu" if relu_out else bool(emit))
        if ybits is not None:
            y.bits = ybits
        return y

    def conv_wgrad(self, x, dy, dw, db=None, *, ks, x_ups=False, x_relu=False, dy_ups=False, alpha=1.0, sync=False,
                   overwrite=False):
        """dw (cout, ks*ks, cin) float32 += alpha * sum_p dy'(p) (x) a(p + tap);
        db (cout,) float32 += alpha * sum_p dy'(p) (fused bias gradient) when given.  ``overwrite``: "=" instead of "+="
        (XMC_WGRAD_OVERWRITE: the first write of a gradient nobody zeroed).
        Unless ``sync``, the launch goes to the weight-gradient stream (``wgrad_async``): dw / db are complete only
        after ``join_wgrad()``."""
        if self.wgrad_async and not sync:
            if self._wg_stream is None:
                self._wg_stream = torch.cuda.Stream(device=self.device)
            self._wg_stream.wait_stream(torch.cuda.current_stream())        # x, dy (and the zeroed dw) are ready
            self._wg_keep.append((x, dy))                                   # their memory must outlive the side launch
            with torch.cuda.stream(self._wg_stream):
                return self.conv_wgrad(x, dy, dw, db, ks=ks, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha,
                                       sync=True, overwrite=overwrite)
        n, hi, wi, cin = x.shape
        cout = dy.shape[-1]
        assert dw.shape == (cout, ks * ks, cin) and dw.dtype == torch.float32
        assert x.dtype == dy.dtype == self.dtype
        d = WgradDesc(n, hi, wi, cin, cout, ks, int(x_ups), int(x_relu), int(dy_ups), self.code,
                      int(self.wgrad_variant) | (0 if self.phase_conv else 256) | (0x1000 if overwrite else 0),
                      float(alpha))                        # bit 8: no phase-decomposed kernel; bit 12: XMC_WGRAD_OVERWRITE
        assert db is None or (db.dtype == torch.float32 and db.numel() == cout)
        ws_bytes = self.lib.xmc_conv2d_wgrad_workspace_bytes(C.byref(d)) if self.deterministic else 0
        if ws_bytes:
            ws = self.empty((ws_bytes // 4,), torch.float32)
            check(self.lib.xmc_conv2d_wgrad_ws(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), _p(ws), ws_bytes,
                                               self._stream()), "xmc_conv2d_wgrad_ws")
        else:
            check(self.lib.xmc_conv2d_wgrad(C.byref(d), _p(x), _p(dy), _p(dw), _p(db), self._stream()),
                  "xmc_conv2d_wgrad")

    def wgrad_is_phase(self, x, dy, *, ks, x_ups=False, x_relu=False, dy_ups=False):
        """does xmc_conv2d_wgrad_ws run this launch phase-decomposed (conv_wgrad_phase.hip: 16 instead of 36 products per
        low-resolution pixel)?  Mirrors xmc_conv2d_wgrad_phase_try's domain; bench.py's FLOP accounting only."""
        n, hi, wi, cin = x.shape
        cout = dy.shape[-1]
        if not (self.phase_conv and self.deterministic and self.dtype == torch.bfloat16 and ks == 3) or cin % 32 or cout % 32:
            return False
        if bool(x_ups) == bool(dy_ups) or (x_ups and x_relu):
            return False
        hv, wv = (hi, wi) if x_ups else (hi // 2, wi // 2)
        if hv < 4 or wv < 4 or hv & (hv - 1) or wv & (wv - 1) or (n * hv * wv) % 64:
            return False
        wt = min(wv, 16)
        rt = min(64 // wt, hv)
        return n % (64 // (wt * rt)) == 0

    def join_wgrad(self):
        """Make the current stream wait for every weight-gradient launch issued so far (before anything reads the
        gradient arenas: spectral-norm gradient fix, all-reduce, Adam)."""
        if self._wg_stream is not None and self._wg_keep:
            torch.cuda.current_stream().wait_stream(self._wg_stream)
        self._wg_keep = []

    def pack_conv_weight(self, w):
        """prepared (cout, taps, cin) weights -> MFMA-fragment order for the weight-streaming kernel"""
        cout, taps, cin = w.shape
        out = self.empty((((cout + 31) // 32) * 32 * taps * cin,))
        check(self.lib.xmc_pack_conv_weight(_p(w), _p(out), cout, taps, cin, self._stream()), "xmc_pack_conv_weight")
        return PackedWeight(out, cout, taps, cin)

    def _packable(self, taps, k):
        """the domain of the kernels on fragment-packed weights (conv_stream.hip: weight-streaming 3x3, pointwise 1x1):
        bf16, reduction channels in 32-chunks"""
        return self.stream_conv and self.dtype == torch.bfloat16 and taps in (1, 9) and k % 32 == 0

    @staticmethod
    def _packed_numel(rows, taps, k):
        return ((rows + 31) // 32) * 32 * taps * k

    def prep_conv_weight(self, w, inv_sigma=None, need_dgrad=True, phase=None):
        """float32 master (cout, taps, cin) -> activation-dtype forward / dgrad copies; each is a PackedWeight
        (MFMA-fragment order, conv_stream.hip) when its shape is in that kernel's domain."""
        cout, taps, cin = w.shape
        pf, pd = self._packable(taps, cin), need_dgrad and self._packable(taps, cout)
        if phase in ("ups", "pool") and need_dgrad and self.skip_plain_copies() and self._phase_only(phase, pf, pd, cout, cin):
            # every launch of this site reads its 16-tap phase copies: the 3x3 copies are not made at all
            wf, wd = PackedWeight(None, cout, taps, cin), PackedWeight(None, cin, taps, cout)
            wf.lazy, wd.lazy = (w, inv_sigma, 0), (w, inv_sigma, 1)
            self.attach_phase_weights(w, inv_sigma, wf, wd, phase)
            return wf, wd
        wf = self.empty((self._packed_numel(cout, taps, cin),) if pf else (cout, taps, cin))
        wd = None
        if need_dgrad:
            wd = self.empty((self._packed_numel(cin, taps, cout),) if pd else (cin, taps, cout))
        check(self.lib.xmc_prep_conv_weight(_p(w), _p(inv_sigma), _p(wf), _p(wd), cout, taps, cin, self.code,
                                            int(pf) | (int(pd) << 1), self._stream()), "xmc_prep_conv_weight")
        wf = self._with_mx8(PackedWeight(wf, cout, taps, cin)) if pf else wf
        wd = self._with_mx8(PackedWeight(wd, cin, taps, cout)) if pd else wd
        self.attach_phase_weights(w, inv_sigma, wf, wd, phase)
        return wf, wd

    def attach_phase_weights(self, w, inv_sigma, wf, wd, phase):
        """``phase`` = "ups" (the layer is conv3x3(upsample2(.))) / "pool" (avg_pool2(conv3x3(.))) / "s2" (a stride-2 SAME
        convolution of an even-sized map, flax padding: the frozen ResNet-50's down-sampling 3x3 layers) / None: give the prepared
        forward / dgrad weights their 16-tap phase copies (xmc_phase_conv_weight) -- the layer's ups / pool_out launches
        and their adjoints then run as four 2x2 convolutions on the low-resolution grid (conv_phase_kernel)."""
        if phase is None or not self.phase_conv or (self.fp8 and not self.fp8_phase) or not isinstance(wf, PackedWeight) or wf.taps != 9:
            return
        cout, cin = wf.cout, wf.cin
        if cout % 32 or cin % 32:
            return
        mode = {"ups": 0, "pool": 1, "s2": 2}[phase]
        pf16 = self.empty((cout * 16 * cin,))
        pd16 = self.empty((cin * 16 * cout,)) if isinstance(wd, PackedWeight) else None
        check(self.lib.xmc_phase_conv_weight(_p(w), _p(inv_sigma), _p(pf16), _p(pd16), cout, cin, mode, self._stream()),
              "xmc_phase_conv_weight")
        wf.phase = (("out", "in", "s2in")[mode], pf16)
        if pd16 is not None:
            wd.phase = (("in", "out", "s2out")[mode], pd16)

    # -------------------------------------------------------------------------------------- GEMM
    def gemm(self, a, b, *, ta=False, tb=False, alpha=1.0, alpha_dev=None, beta=0.0, out=None, fast=False):
        """C = alpha * op(a) @ op(b) + beta * C over the last two dims (float32; 2-D or batched 3-D).
        ``a`` / ``b`` may be arbitrary strided views (no copies are made).  ``fast``: in the bf16 training mode
        the operands are rounded to bf16 inside the kernel and multiplied on the bf16 MFMA (float32 accumulate);
        the float32 parity mode ignores the flag."""
        assert a.dtype == b.dtype == torch.float32
        batched = a.dim() == 3
        if batched:
            assert b.dim() == 3 and a.shape[0] == b.shape[0]
        am, ak = (a.shape[-1], a.shape[-2]) if ta else (a.shape[-2], a.shape[-1])
        bk, bn = (b.shape[-1], b.shape[-2]) if tb else (b.shape[-2], b.shape[-1])
        assert ak == bk, (a.shape, b.shape, ta, tb)
        sam, sak = (a.stride(-1), a.stride(-2)) if ta else (a.stride(-2), a.stride(-1))
        sbk, sbn = (b.stride(-1), b.stride(-2)) if tb else (b.stride(-2), b.stride(-1))
        batch = a.shape[0] if batched else 1
        if out is None:
            assert beta == 0.0
            out = self.empty((batch, am, bn) if batched else (am, bn), torch.float32)
        assert out.dtype == torch.float32 and out.stride(-1) == 1 and out.shape[-2:] == (am, bn)
        bf = fast and self.dtype == torch.bfloat16
        fn = self.lib.xmc_gemm_f32_bf16mfma if bf else self.lib.xmc_gemm_f32
        wsn = self.lib.xmc_gemm_ws_floats(am, bn, ak, batch, int(bf))        # split-K scratch (few tiles, long K)
        ws = self.empty((wsn,), torch.float32) if wsn else None
        check(fn(
            C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()), C.c_void_p(out.data_ptr()), am, bn, ak,
            a.stride(0) if batched else 0, sam, sak, b.stride(0) if batched else 0, sbk, sbn,
            out.stride(0) if batched else 0, out.stride(-2), float(alpha), _p(alpha_dev), float(beta), batch,
            _p(ws), self._stream()), "xmc_gemm_f32")
        return out

    def reduce_mid(self, x, *, relu=False, scale=1.0, out=None, accumulate=False):
        a, r, c = x.shape
        if out is None:
            out = self.empty((a, c), torch.float32)
            accumulate = False
        assert out.dtype == torch.float32 and out.numel() == a * c
        ws = self.empty((self.lib.xmc_reduce_mid_ws_floats(a, r, c),), torch.float32) if self.deterministic else None
        check(self.lib.xmc_reduce_mid_ws(_p(x), _p(out), _p(ws), a, r, c, _code(x.dtype), int(relu), float(scale),
                                         int(accumulate), self._stream()), "xmc_reduce_mid_ws")
        return out

    # -------------------------------------------------------------------------------- batch norm
    def bn_stats(self, x):
        c = x.shape[-1]
        sums = self.zeros((2 * c,))
        check(self.lib.xmc_bn_stats(_p(x), _p(sums), x.numel() // c, c, _code(x.dtype), self._stream()),
              "xmc_bn_stats")
        return sums

    def bn_finalize(self, sums, pixels, run_mean, run_var, update, eps=1e-5, momentum=0.9):
        c = sums.numel() // 2
        mean, rstd = self.empty((c,), torch.float32), self.empty((c,), torch.float32)
        check(self.lib.xmc_bn_finalize(_p(sums), _p(mean), _p(rstd), _p(run_mean), _p(run_var), pixels, c,
                                       eps, momentum, int(update), self._stream()), "xmc_bn_finalize")
        return mean, rstd

    def bn_batch_stats(self, x, run_mean, run_var, update, eps=1e-5, momentum=0.9):
        """Training-mode BatchNorm statistics of x (.., C) -> (mean, rstd); updates the running statistics in place.
        Two-stage partial-row reduction (no atomics: bit-reproducible)."""
        c = x.shape[-1]
        pixels = x.numel() // c
        ws = self.empty((self.lib.xmc_bn_stats_ws_floats(pixels, c),), torch.float32)
        mean, rstd = self.empty((c,), torch.float32), self.empty((c,), torch.float32)
        check(self.lib.xmc_bn_batch_stats(_p(x), _p(ws), _p(mean), _p(rstd), _p(run_mean), _p(run_var), pixels, c,
                                          _code(x.dtype), eps, momentum, int(update), self._stream()),
              "xmc_bn_batch_stats")
        return mean, rstd

    def bn_from_running(self, run_mean, run_var, eps=1e-5):
        c = run_mean.numel()
        mean, rstd = self.empty((c,), torch.float32), self.empty((c,), torch.float32)
        check(self.lib.xmc_bn_from_running(_p(run_mean), _p(run_var), _p(mean), _p(rstd), c, eps,
                                           self._stream()), "xmc_bn_from_running")
        return mean, rstd

    @staticmethod
    def _gb_rows(gb, n, hc, c):
        """gb as (n*hc*hc, 2c) rows with unit inner stride (a channel slice of a wider projection output is fine); float32, or
        bf16 (the fused local projection's output in the bf16 mode: ``gb_bf16``)"""
        g2 = gb.reshape(n * hc * hc, 2 * c) if gb.is_contiguous() else gb
        assert g2.dtype in (torch.float32, torch.bfloat16) and g2.shape == (n * hc * hc, 2 * c) and g2.stride(1) == 1
        return g2, g2.stride(0)

    def cbn_act_fwd(self, x, mean, rstd, gb, hc, relu=True):
        """gb: (n*hc*hc, 2c) float32 or bf16 -- [:, :c] = gamma, [:, c:] = beta (one fused conv / dense output, or a
        column slice of the output of several sites' fused projection: the row stride is passed on)."""
        n, h, w, c = x.shape
        g2, cs = self._gb_rows(gb, n, hc, c)
        y = torch.empty_like(x)
        gp, es = g2.data_ptr(), g2.element_size()
        if (self.fp8 and x.dtype == torch.bfloat16 and c % 64 == 0 and c >= self.fp8_min_cin and self._mx8_patch_fits(2 * h, 2 * w) and not self.fp8_debug & 2
                and g2.dtype == torch.float32):
            # config.conv_fp8: every consumer of this tensor is a 3x3 convolution (GenBlock: conv(a), conv(upsample(a))) --
            # the kernel writes its MX-fp8 packets along with the bf16 tensor (the weight gradient still reads bf16)
            y8 = torch.empty((n * h * w, c // 64, 80), dtype=torch.uint8, device=self.device)
            check(self.lib.xmc_cbn_act_fwd_mx8(_p(x), _p(mean), _p(rstd), C.c_void_p(gp), C.c_void_p(gp + 4 * c), _p(y),
                                               _p(y8), n, h, w, c, hc, cs, int(relu), self._stream()), "xmc_cbn_act_fwd_mx8")
            y.mx8 = (y8, False)
            return y
        check(self.lib.xmc_cbn_act_fwd(_p(x), _p(mean), _p(rstd), C.c_void_p(gp), C.c_void_p(gp + es * c), _p(y), n, h,
                                       w, c, hc, cs, int(relu), _code(x.dtype), _code(g2.dtype), self._stream()), "xmc_cbn_act_fwd")
        return y

    def cbn_act_bwd(self, dy, x, mean, rstd, gb, hc, relu=True, dgb_out=None):
        """-> (dx, dgb) with dgb laid out like gb -- same dtype -- (written into ``dgb_out`` -- same rows / stride rules -- if given)."""
        n, h, w, c = x.shape
        assert dy.dtype == x.dtype and dy.shape == x.shape
        g2, cs = self._gb_rows(gb, n, hc, c)
        if dgb_out is None:
            dgb_out = torch.empty_like(gb) if gb.is_contiguous() else \
                torch.empty((n * hc * hc, 2 * c), dtype=g2.dtype, device=x.device)
        d2, ds = self._gb_rows(dgb_out, n, hc, c)
        assert d2.dtype == g2.dtype, "gamma/beta and their gradients share the dtype"
        assert ds == cs or (gb.is_contiguous() and dgb_out.is_contiguous()), "gamma/beta and their gradients share the row stride"
        code, st, gcode, es = _code(x.dtype), self._stream(), _code(g2.dtype), g2.element_size()
        g_, b_ = C.c_void_p(g2.data_ptr()), C.c_void_p(g2.data_ptr() + es * c)
        dg_, db_ = C.c_void_p(d2.data_ptr()), C.c_void_p(d2.data_ptr() + es * c)
        check(self.lib.xmc_cbn_act_bwd_cells(_p(dy), _p(x), _p(mean), _p(rstd), g_, b_, dg_, db_, n, h, w, c, hc,
                                             cs, int(relu), code, gcode, st), "xmc_cbn_act_bwd_cells")
        s = self.empty((2 * c,), torch.float32)
        ws = self.empty((self.lib.xmc_cbn_bwd_sums_ws_floats(n * hc * hc, c),), torch.float32)
        check(self.lib.xmc_cbn_bwd_sums(g_, dg_, db_, _p(s), _p(ws), n * hc * hc, c, cs, gcode, st), "xmc_cbn_bwd_sums")
        dx = torch.empty_like(x)
        check(self.lib.xmc_cbn_act_bwd_dx(_p(dy), _p(x), _p(mean), _p(rstd), g_, b_, _p(s), _p(dx), n, h, w, c, hc,
                                          cs, int(relu), code, gcode, st), "xmc_cbn_act_bwd_dx")
        return dx, dgb_out

    # --------------------------------------------------------------------------------- pointwise
    def pool2(self, x, scale, res=None, relu_copy=False):
        """y = scale * sum_{2x2} x (+ res).  ``relu_copy``: -> (y, max(x, 0)): the pass holds every element of x, so it also
        writes the ReLU-ed full-resolution copy a down-sampling DiscBlock's first convolution and its weight gradient read
        (xmc_pool2_relu) -- ``x.bits`` (the ReLU-mask bits of the producing epilogue) carry over: (relu(x) > 0) == (x > 0)."""
        n, h, w, c = x.shape
        y = self.empty((n, h // 2, w // 2, c), x.dtype)
        if relu_copy:
            xr = torch.empty_like(x)
            check(self.lib.xmc_pool2_relu(_p(x), _p(res), _p(y), _p(xr), n, h, w, c, float(scale), _code(x.dtype),
                                          self._stream()), "xmc_pool2_relu")
            if getattr(x, "bits", None) is not None:
                xr.bits = x.bits
            return y, xr
        check(self.lib.xmc_pool2(_p(x), _p(res), _p(y), n, h, w, c, float(scale), _code(x.dtype),
                                 self._stream()), "xmc_pool2")
        return y

    def expand_taps(self, x, ks, sign=1):
        """(n,h,w,c<=3) -> (n,h,w,32): channel tap*c+j holds x[pixel + sign*offset(tap)][j] (zero padded)."""
        n, h, w, c = x.shape
        y = self.empty((n, h, w, 32), x.dtype)
        check(self.lib.xmc_expand_taps(_p(x), _p(y), n, h, w, c, ks, sign, _code(x.dtype), self._stream()),
              "xmc_expand_taps")
        return y

    def bcast_relu_bwd(self, dpool, x):
        a, r, c = x.shape
        dx = torch.empty_like(x)
        check(self.lib.xmc_bcast_relu_bwd(_p(dpool), _p(x), _p(dx), a, r, c, _code(x.dtype), self._stream()),
              "xmc_bcast_relu_bwd")
        return dx

    def tanh_out_fwd(self, x):
        y = torch.empty_like(x)
        check(self.lib.xmc_tanh_out_fwd(_p(x), _p(y), x.numel(), _code(x.dtype), self._stream()),
              "xmc_tanh_out_fwd")
        return y

    def tanh_out_bwd(self, dy, y):
        dx = torch.empty_like(y)
        check(self.lib.xmc_tanh_out_bwd(_p(dy), _p(y), _p(dx), y.numel(), _code(y.dtype), self._stream()),
              "xmc_tanh_out_bwd")
        return dx

    def cast(self, x, dtype):
        if x.dtype == dtype:
            return x
        y = torch.empty(x.shape, dtype=dtype, device=x.device)
        check(self.lib.xmc_cast(_p(x), _code(x.dtype), _p(y), _code(dtype), x.numel(), self._stream()), "xmc_cast")
        return y

    def add(self, a, b):
        out = torch.empty_like(a)
        check(self.lib.xmc_add(_p(a), _p(b), _p(out), a.numel(), _code(a.dtype), self._stream()), "xmc_add")
        return out

    def add_into(self, dst, src):
        """dst += src (same dtype, contiguous)"""
        assert dst.dtype == src.dtype and dst.numel() == src.numel()
        check(self.lib.xmc_add(_p(dst), _p(src), _p(dst), dst.numel(), _code(dst.dtype), self._stream()), "xmc_add")
        return dst

    def zeros_act(self, shape):
        """zero-filled tensor in the activation dtype (one memset)"""
        return torch.zeros(shape, dtype=self.dtype, device=self.device)

    # --------------------------------------------------------------------------------- attention
    def _attn_mfma(self, region, b, r, t, e):
        """attention_for_g on the matrix cores (attn_mfma.hip): bf16 mode, inside the kernel's domain; XMC_ATTN_MFMA=0: A/B"""
        return (getattr(self, "attn_mfma", True) and region.dtype == torch.bfloat16
                and bool(self.lib.xmc_attn_g_mfma_supported(b, r, t, e)))

    def attn_g_sliced(self, region, t):
        """may ``attn_g_fwd(..., ctx_out=)`` / ``attn_g_bwd`` with a strided ``dctx`` be used (the MFMA kernels' row-pitch form)?"""
        b, r, e = region.shape
        return self._attn_mfma(region, b, r, t, e)

    def attn_g_fwd(self, region, words_n, max_len, gamma, ctx_out=None):
        """``ctx_out`` (``attn_g_sliced`` only): a (B, R, E) view with unit channel stride and row pitch >= E of a wider tensor --
        the context is written there (no concatenation afterwards)"""
        b, r, e = region.shape
        t = words_n.shape[1]
        attn = self.empty((b, r, t), torch.float32)
        rinv = self.empty((b, r), torch.float32)
        if ctx_out is not None:
            assert self._attn_mfma(region, b, r, t, e) and ctx_out.shape == region.shape and ctx_out.dtype == region.dtype
            ld = ctx_out.stride(1)
            assert ctx_out.stride(2) == 1 and ctx_out.stride(0) == r * ld
            check(self.lib.xmc_attn_g_fwd_mfma_ld(_p(region), _p(words_n), _p(max_len), C.c_void_p(ctx_out.data_ptr()), ld, _p(attn),
                                                  _p(rinv), b, r, t, e, float(gamma), self._stream()), "xmc_attn_g_fwd_mfma_ld")
            return ctx_out, attn, rinv
        ctx = torch.empty_like(region)
        if self._attn_mfma(region, b, r, t, e):
            check(self.lib.xmc_attn_g_fwd_mfma(_p(region), _p(words_n), _p(max_len), _p(ctx), _p(attn), _p(rinv), b, r, t, e,
                                               float(gamma), self._stream()), "xmc_attn_g_fwd_mfma")
            return ctx, attn, rinv
        check(self.lib.xmc_attn_g_fwd(_p(region), _p(words_n), _p(max_len), _p(ctx), _p(attn), _p(rinv), b, r, t,
                                      e, float(gamma), _code(region.dtype), self._stream()), "xmc_attn_g_fwd")
        return ctx, attn, rinv

    def attn_g_bwd(self, dctx, region, words_n, attn, rinv, gamma):
        b, r, e = region.shape
        t = words_n.shape[1]
        dregion = torch.empty_like(region)
        if self._attn_mfma(region, b, r, t, e) and dctx.dtype == torch.bfloat16:
            ld = dctx.stride(1)                      # dctx may be a column slice of a wider tensor (row pitch ld)
            assert dctx.stride(2) == 1 and dctx.stride(0) == r * ld and dctx.shape == region.shape
            check(self.lib.xmc_attn_g_bwd_mfma_ld(C.c_void_p(dctx.data_ptr()), ld, _p(region), _p(words_n), _p(attn), _p(rinv), _p(dregion),
                                                  b, r, t, e, float(gamma), self._stream()), "xmc_attn_g_bwd_mfma_ld")
            return dregion
        dctx = dctx.contiguous()
        check(self.lib.xmc_attn_g_bwd(_p(dctx), _p(region), _p(words_n), _p(attn), _p(rinv), _p(dregion), b, r, t,
                                      e, float(gamma), _code(region.dtype), self._stream()), "xmc_attn_g_bwd")
        return dregion

    def l2norm_fwd(self, x):
        rows, cols = x.shape
        y = self.empty((rows, cols), torch.float32)
        inv = self.empty((rows,), torch.float32)
        check(self.lib.xmc_l2norm_rows_fwd(_p(x), _p(y), _p(inv), rows, cols, _code(x.dtype), self._stream()),
              "xmc_l2norm_rows_fwd")
        return y, inv

    def l2norm_bwd(self, dy, y, inv, out_dtype, out=None):
        rows, cols = y.shape
        dx = self.empty((rows, cols), out_dtype) if out is None else out
        assert dx.dtype == out_dtype and dx.numel() == rows * cols and dx.is_contiguous()
        check(self.lib.xmc_l2norm_rows_bwd(_p(dy), _p(y), _p(inv), _p(dx), rows, cols, _code(out_dtype),
                                           self._stream()), "xmc_l2norm_rows_bwd")
        return dx

    # --------------------------------------------------------------------------------- word loss
    def wl_softmax(self, s, max_len, b, r, t, gamma1):
        alpha = torch.empty_like(s)
        nn = self.empty((b, b * t), torch.float32)
        check(self.lib.xmc_wl_softmax(_p(s), _p(max_len), _p(alpha), _p(nn), b, r, t, float(gamma1),
                                      self._stream()), "xmc_wl_softmax")
        return alpha, nn

    def wl_qdot(self, alpha, h, b, r, t):
        q = self.empty((b, b * t), torch.float32)
        check(self.lib.xmc_wl_qdot(_p(alpha), _p(h), _p(q), b, r, t, self._stream()), "xmc_wl_qdot")
        return q

    def wl_rows(self, nn, q, max_len, b, t, gamma2, gamma3):
        sim_t = self.empty((b, b), torch.float32)
        pi = self.empty((b, b * t), torch.float32)
        check(self.lib.xmc_wl_rows(_p(nn), _p(q), _p(max_len), _p(sim_t), _p(pi), b, t, float(gamma2),
                                   float(gamma3), self._stream()), "xmc_wl_rows")
        return sim_t, pi

    def wl_bwd_cols(self, s, alpha, h, nn, q, pi, dsim_t, b, r, t, gamma1, gamma3):
        """Returns (dS, alpha*dq); dS overwrites ``h``."""
        a_s = torch.empty_like(alpha)
        check(self.lib.xmc_wl_bwd_cols(_p(s), _p(alpha), _p(h), _p(nn), _p(q), _p(pi), _p(dsim_t), _p(a_s), b, r,
                                       t, float(gamma1), float(gamma3), self._stream()), "xmc_wl_bwd_cols")
        return h, a_s

    # ------------------------------------------------------------- word loss, fused on the matrix cores (word_loss_fused.hip)
    def wl_fused_ok(self, image_feat, t):
        """bf16 mode inside the fused kernels' domain (R == 256, E % 64 == 0); XMC_WL_FUSED=0: A/B against the GEMM path"""
        b, r, e = image_feat.shape
        return (getattr(self, "wl_fused", True) and image_feat.dtype == torch.bfloat16
                and bool(self.lib.xmc_wl_fused_supported(b, r, t, e)))

    def wl_prep_words(self, words_n):
        """words_n (B, T, E) float32 normalised -> (w (LDP, E), wT (E, LDP)) bf16, zero-padded to LDP columns"""
        b, t, e = words_n.shape
        ldp = int(self.lib.xmc_wl_fused_ldp(b, t))
        w = self.empty((ldp, e), torch.bfloat16)
        wt = self.empty((e, ldp), torch.bfloat16)
        check(self.lib.xmc_wl_prep_words(_p(words_n), _p(w), _p(wt), b * t, ldp, e, self._stream()), "xmc_wl_prep_words")
        return w, wt

    def wl_prep_regions(self, x):
        """x (B, R, E) bf16 -> l2-normalised rn (B, R, E), its transpose rnT (B, E, R) (bf16) and 1 / |x| (B*R) float32"""
        b, r, e = x.shape
        assert x.is_contiguous()
        rn = self.empty((b, r, e), torch.bfloat16)
        rnt = self.empty((b, e, r), torch.bfloat16)
        rinv = self.empty((b * r,), torch.float32)
        check(self.lib.xmc_wl_prep_regions(_p(x), _p(rn), _p(rnt), _p(rinv), b, r, e, self._stream()), "xmc_wl_prep_regions")
        return rn, rnt, rinv

    def wl_tn_gemm(self, x0, y0, k0, rows_x, rows_y, batch, out_dtype, alpha=1.0, x1=None, y1=None, k1=0, y0_shared=False):
        """out[z][x][y] = alpha * (x0[z][x][:k0] . y0[z][y][:k0] + x1[z][x][:k1] . y1[z][y][:k1]); operands (batch, rows, ld) bf16
        (y0 (rows, ld) when ``y0_shared``)"""
        out = self.empty((batch, rows_x, rows_y), out_dtype)
        sx0, ldx0 = x0.stride(0), x0.stride(1)
        sy0, ldy0 = (0, y0.stride(0)) if y0_shared else (y0.stride(0), y0.stride(1))
        if x1 is None:
            a1 = (None, 0, 0, None, 0, 0, 0)
        else:
            a1 = (_p(x1), x1.stride(0), x1.stride(1), _p(y1), y1.stride(0), y1.stride(1), k1)
        check(self.lib.xmc_wl_tn_gemm(_p(x0), sx0, ldx0, _p(y0), sy0, ldy0, k0, *a1, _p(out), rows_x * rows_y, rows_y,
                                      1 if out_dtype == torch.float32 else 0, float(alpha), rows_x, rows_y, batch,
                                      self._stream()), "xmc_wl_tn_gemm")
        return out

    def wl_cols_fwd(self, rn, w, g, max_len, t, gamma1):
        b, r, e = rn.shape
        nn = self.empty((b, b * t), torch.float32)
        q = self.empty((b, b * t), torch.float32)
        check(self.lib.xmc_wl_cols_fwd(_p(rn), _p(w), _p(g), _p(max_len), _p(nn), _p(q), b, t, e, w.shape[0], float(gamma1),
                                       self._stream()), "xmc_wl_cols_fwd")
        return nn, q

    def wl_cols_bwd(self, rn, w, g, max_len, dsim_t, pi, t, gamma1, gamma3):
        """-> (dS, alpha * dq, alpha), each (B, R, LDP) bf16 with zero padding columns"""
        b, r, e = rn.shape
        ldp = w.shape[0]
        ds = self.empty((b, r, ldp), torch.bfloat16)
        a_s = self.empty((b, r, ldp), torch.bfloat16)
        al = self.empty((b, r, ldp), torch.bfloat16)
        check(self.lib.xmc_wl_cols_bwd(_p(rn), _p(w), _p(g), _p(max_len), _p(dsim_t), _p(pi), _p(ds), _p(a_s), _p(al), b, t, e,
                                       ldp, float(gamma1), float(gamma3), self._stream()), "xmc_wl_cols_bwd")
        return ds, a_s, al

    def l2norm_bwd_bf16y(self, dy, y, inv, out_dtype, out=None):
        rows, cols = y.shape
        dx = self.empty((rows, cols), out_dtype) if out is None else out
        assert dx.dtype == out_dtype and dx.numel() == rows * cols and dx.is_contiguous() and y.dtype == torch.bfloat16
        check(self.lib.xmc_l2norm_rows_bwd_bf16y(_p(dy), _p(y), _p(inv), _p(dx), rows, cols, _code(out_dtype), self._stream()),
              "xmc_l2norm_rows_bwd_bf16y")
        return dx

    # ------------------------------------------------------------------------------ scalar losses
    def xent_sym(self, logits, weight, loss_acc, want_grad=True, stats=None):
        """stats (2,) float32, optional: receives {accuracy, entropy} of get_statistics."""
        b = logits.shape[0]
        dl = torch.empty_like(logits) if want_grad else None
        check(self.lib.xmc_xent_sym(_p(logits), b, float(weight), _p(loss_acc), _p(dl), _p(stats), self._stream()),
              "xmc_xent_sym")
        return dl

    # contrastive_loss in two launches per direction (losses.hip cl_*; XMC_CL_FUSED=0: the l2norm + GEMM + xent chain, A/B)
    def cl_fused_ok(self, a, b):
        return (getattr(self, "cl_fused", None) if getattr(self, "cl_fused", None) is not None else
                os.environ.get("XMC_CL_FUSED", "1") != "0") and a.dtype == b.dtype == torch.float32 and a.shape == b.shape \
            and a.shape[1] <= 2048 and a.is_contiguous() and b.is_contiguous()

    def cl_logits(self, a, b, inv_t):
        n, d = a.shape
        logits = self.empty((n, n), torch.float32)
        ainv, binv = self.empty((n,), torch.float32), self.empty((n,), torch.float32)
        check(self.lib.xmc_cl_logits(_p(a), _p(b), _p(logits), _p(ainv), _p(binv), n, d, float(inv_t), self._stream()), "xmc_cl_logits")
        return logits, ainv, binv

    def cl_bwd(self, dl, x, y, xinv, yinv, inv_t, trans, out=None):
        """the pullback of the logits onto x (see xmc_cl_bwd); ``out``: ADDED to it, else a fresh tensor"""
        n, d = x.shape
        acc = out is not None
        if out is None:
            out = self.empty((n, d), torch.float32)
        assert out.dtype == torch.float32 and out.shape == (n, d) and out.is_contiguous()
        check(self.lib.xmc_cl_bwd(_p(dl), _p(x), _p(y), _p(xinv), _p(yinv), _p(out), n, d, float(inv_t), int(trans), int(acc),
                                  self._stream()), "xmc_cl_bwd")
        return out

    def loss_assemble(self, loss_vec, hinge):
        """-> (4,) float32 {d_loss, g_loss, c_loss_d, c_loss_g} in one launch (xmc_loss_assemble)"""
        out = self.empty((4,), torch.float32)
        check(self.lib.xmc_loss_assemble(_p(loss_vec), _p(hinge), _p(out), self._stream()), "xmc_loss_assemble")
        return out

    def hinge(self, logit, b, d_loss_acc, g_loss_acc):
        dld = self.empty((2 * b,), torch.float32)
        dlg = self.empty((2 * b,), torch.float32)
        check(self.lib.xmc_hinge(_p(logit), b, _p(d_loss_acc), _p(g_loss_acc), _p(dld), _p(dlg), self._stream()),
              "xmc_hinge")
        return dld, dlg

    def proj_head_fwd(self, pool, w, inv_sigma, bias, emb):
        n2, c = pool.shape
        out = self.empty((n2,), torch.float32)
        check(self.lib.xmc_proj_head_fwd(_p(pool), _p(w), _p(inv_sigma), _p(bias), _p(emb), _p(out), n2,
                                         emb.shape[0], c, self._stream()), "xmc_proj_head_fwd")
        return out

    def proj_head_bwd(self, dout, pool, w, inv_sigma, emb, want_demb):
        n2, c = pool.shape
        dpool = self.empty((n2, c), torch.float32)
        demb = self.empty(tuple(emb.shape), torch.float32) if want_demb else None
        check(self.lib.xmc_proj_head_bwd(_p(dout), _p(pool), _p(w), _p(inv_sigma), _p(emb), _p(dpool), _p(demb), n2,
                                         emb.shape[0], c, 0, self._stream()), "xmc_proj_head_bwd")
        return dpool, demb

    # ----------------------------------------------------------------------------- spectral norm
    def spectral_power_iter(self, w2d, u0, u_axis, eps=1e-10):
        rows, cols = w2d.shape
        nu, nv = (rows, cols) if u_axis == 0 else (cols, rows)
        assert u0.numel() == nu
        u_new = self.empty((1, nu), torch.float32)
        v = self.empty((nv,), torch.float32)
        scal = self.empty((2,), torch.float32)
        tmp = self.empty((rows + cols + 4,), torch.float32)
        check(self.lib.xmc_spectral_power_iter(_p(w2d), _p(u0), _p(u_new), _p(v), _p(scal), _p(tmp), rows, cols,
                                               u_axis, eps, self._stream()), "xmc_spectral_power_iter")
        return u_new, v, scal

    def spectral_grad_fix(self, g2d, w2d, u, v, scal, u_axis):
        rows, cols = w2d.shape
        tmp = self.empty((4,), torch.float32)
        check(self.lib.xmc_spectral_grad_fix(_p(g2d), _p(w2d), _p(u), _p(v), _p(scal), _p(tmp), rows, cols, u_axis,
                                             self._stream()), "xmc_spectral_grad_fix")

    # ------------------------------------------------------------------ batched spectral norm
    def sn_bank_create(self, entries, keep_uv=False):
        """entries: list of dicts (w_off, rows, cols, u_axis, taps, is_conv).  Builds the two device
        descriptor tables (prep / grad-fix prefix) of include/xmcgan_hip.h::xmc_sn_entry.  ``keep_uv``: the entries are a
        subset of another bank's and keep that bank's ``u_off`` / ``v_off`` (they share its u / v buffers)."""
        from ._lib import SnEntry
        n = len(entries)
        tabs = ((SnEntry * n)(), (SnEntry * n)())
        u_off = v_off = blk_a = blk_b = blk_p = blk_d = 0
        wf_off = wd_off = 0
        for i, e in enumerate(entries):
            rows, cols = e["rows"], e["cols"]
            nu, nv = (rows, cols) if e["u_axis"] == 0 else (cols, rows)
            if keep_uv:
                u_off, v_off = e["u_off"], e["v_off"]
            cin = cols // e["taps"]
            pf = bool(e["is_conv"]) and self._packable(e["taps"], cin)
            pd = bool(e["is_conv"]) and self._packable(e["taps"], rows)
            for t, bp in ((tabs[0], blk_p), (tabs[1], blk_d)):
                t[i] = SnEntry(e["w_off"], rows, cols, e["u_axis"], u_off, v_off, blk_a, blk_b, e["taps"],
                               int(e["is_conv"]), wf_off, wd_off, bp,
                               int(pf) | (int(pd) << 1) | (4 if self._phase_only(e.get("phase"), pf, pd, rows, cin) else 0))
            e.update(u_off=u_off, v_off=v_off, nu=nu, nv=nv, wf_off=wf_off, wd_off=wd_off, pf=pf, pd=pd)
            u_off += (nu + 3) & ~3               # 16-byte aligned slices: the matvec / fix kernels read them as float4
            v_off += (nv + 3) & ~3
            blk_a += (rows + 3) // 4             # "rows" pass: 4 rows per workgroup (spectral_batched.hip ROWS_PER_WG)
            blk_b += (cols + 31) // 32              # "cols" pass: one workgroup per 32 columns, all rows (COLS_PER_WG)
            if e["is_conv"]:
                blk_p += e["taps"] * ((cin + 31) // 32) * ((rows + 31) // 32)
                e["nf"] = self._packed_numel(rows, e["taps"], cin) if pf else rows * cols
                e["nd"] = self._packed_numel(cin, e["taps"], rows) if pd else rows * cols
                wf_off += e["nf"]
                wd_off += e["nd"]
            blk_d += (rows * cols + 65535) // 65536
        dev = [torch.frombuffer(bytearray(bytes(t)), dtype=torch.uint8).to(self.device) for t in tabs]
        return dict(n=n, entries=entries, tab_prep=dev[0], tab_fix=dev[1], nu=u_off, nv=v_off, blocks_a=blk_a,
                    blocks_b=blk_b, blocks_p=blk_p, blocks_d=blk_d, wtotal=wf_off, wdtotal=wd_off)

    def sn_bank_power_iter(self, bank, params, u0_flat, eps=1e-10):
        u_new = self.empty((bank["nu"],), torch.float32)
        u_raw = self.empty((bank["nu"],), torch.float32)
        v = self.empty((bank["nv"],), torch.float32)
        scal = self.empty((2 * bank["n"],), torch.float32)
        check(self.lib.xmc_sn_batched_power_iter(_p(bank["tab_prep"]), bank["n"], _p(params), _p(u0_flat), _p(u_new),
                                                 _p(v), _p(u_raw), _p(scal), bank["blocks_a"], bank["blocks_b"],
                                                 bank["nu"], bank["nv"], eps, self._stream()),
              "xmc_sn_batched_power_iter")
        return u_new, v, scal

    def _phase_only(self, phase, pf, pd, cout, cin):
        """static part of "this site's launches read only its 16-tap phase copies" (attach_phase_weights' domain)"""
        return bool(phase) and pf and pd and cout % 32 == 0 and cin % 32 == 0

    def skip_plain_copies(self):
        """dynamic part: the phase kernels are on and the MX-fp8 mode (which converts the 3x3 copies) is off"""
        # (round 5: ... or the fp8 mode leaves the resampling-adjacent sites on the bf16 phase kernels -- fp8_phase, the default)
        return self.phase_conv and (not self.fp8 or self.fp8_phase) and self.dtype == torch.bfloat16

    def sn_bank_prep(self, bank, params, scal, need_dgrad=True):
        wf = self.empty((bank["wtotal"],))
        wd = self.empty((bank["wdtotal"],)) if need_dgrad else None
        skip = 256 if self.skip_plain_copies() and need_dgrad else 0       # phase sites: their 3x3 copies are never read
        bank["skipped"] = bool(skip)
        check(self.lib.xmc_sn_batched_prep(_p(bank["tab_prep"]), bank["n"], _p(params), _p(scal), _p(wf), _p(wd),
                                           bank["blocks_p"], self.code | skip, self._stream()), "xmc_sn_batched_prep")
        return wf, wd

    def sn_bank_weights(self, bank, i, wf, wd):
        """The prepared forward / dgrad weights of conv entry ``i`` as ``conv`` takes them."""
        e = bank["entries"][i]
        cout, taps = e["rows"], e["taps"]
        cin = e["cols"] // taps
        f = wf[e["wf_off"]:e["wf_off"] + e["nf"]]
        d = wd[e["wd_off"]:e["wd_off"] + e["nd"]] if wd is not None else None
        if bank.get("skipped") and self._phase_only(e.get("phase"), e["pf"], e["pd"], cout, cin):
            # the batched pass left these copies unwritten: the site's launches read its 16-tap copies only
            # (attach_phase_weights); anything else that reaches for .data gets None and fails loudly
            return PackedWeight(None, cout, taps, cin), PackedWeight(None, cin, taps, cout)
        f = self._with_mx8(PackedWeight(f, cout, taps, cin)) if e["pf"] else f.view(cout, taps, cin)
        if d is not None:
            d = self._with_mx8(PackedWeight(d, cin, taps, cout)) if e["pd"] else d.view(cin, taps, cout)
        return f, d

    def sn_bank_grad_fix(self, bank, params, grads, u, v, scal):
        dots = self.empty((bank["blocks_d"],), torch.float32)      # one partial <G, W> per 64K-element chunk
        check(self.lib.xmc_sn_batched_grad_fix(_p(bank["tab_fix"]), bank["n"], _p(params), _p(grads), _p(u), _p(v),
                                               _p(scal), _p(dots), bank["blocks_d"], self._stream()),
              "xmc_sn_batched_grad_fix")

    # ------------------------------------------------- batched weight preparation (fold_sigma)
    def wprep_create(self, entries):
        """entries: dicts (w_off, cout, cin, taps, phase = None | "ups" | "pool", spectral, u_off, v_off) of the packable
        convolution weights of one arena -> the device table of xmc_wprep_batched + buffer sizes.  A phase site whose
        launches read only its 16-tap copies gets no plain 3x3 copies (``_phase_only``)."""
        from ._lib import WprepEntry
        n = len(entries)
        assert 0 < n <= 64
        tab = (WprepEntry * n)()
        wf = wd = pf = pd = part = blk = blkc = 0
        for i, e in enumerate(entries):
            cout, cin, taps = e["cout"], e["cin"], e["taps"]
            assert cout % 32 == 0 and cin % 32 == 0 and taps in (1, 9)
            phase = e.get("phase") if (taps == 9 and self.phase_conv and not (self.fp8 and not self.fp8_phase)) else None
            plain = not (phase and self.skip_plain_copies())
            flags = ((3 if plain else 0) | ({None: 0, "ups": 1, "pool": 2}[phase] << 2) | (16 if e.get("spectral") else 0)
                     | (int(e.get("site", 0)) << 8))          # bits 8..: index in the spectral bank (xmc_adam_wprep_tiles)
            nw, nph = cout * taps * cin, cout * 16 * cin
            tab[i] = WprepEntry(e["w_off"], wf, wd, pf, pd, part, cout, cin, taps, blk, flags, e.get("u_off", 0), e.get("v_off", 0), blkc)
            e.update(wf_off=wf, wd_off=wd, pf_off=pf, pd_off=pd, plain=plain, phase_eff=phase, nw=nw, nph=nph)
            if plain:
                wf += nw
                wd += nw
            if phase:
                pf += nph
                pd += nph
            if e.get("spectral"):
                part += (cout // 32) * taps * cin
                blkc += (taps * cin + 255) // 256
            blk += (cout // 32) * (cin // 32)
        dev = torch.frombuffer(bytearray(bytes(tab)), dtype=torch.uint8).to(self.device)
        return dict(n=n, entries=entries, tab=dev, blocks=blk, blocks_c=blkc, wf=wf, wd=wd, pf=pf, pd=pd, part=part)

    def wprep_alloc(self, wp):
        """(bufs, part) of ``wprep_run`` / ``adam_wprep``: the four copy buffers and the partial-row buffer"""
        return ([self.empty((max(wp[k], 8),)) for k in ("wf", "wd", "pf", "pd")], self.empty((max(wp["part"], 4),), torch.float32))

    def wprep_run(self, wp, params, u0=None, out=None):
        """one pass over the masters of ``wp``'s weights -> their prepared copies (and the W^T u0 partials of the spectral ones);
        ``out`` = (bufs, part) from ``wprep_alloc``: written in place (persistent buffers of the fused optimiser path)"""
        bufs, part = out if out is not None else self.wprep_alloc(wp)
        check(self.lib.xmc_wprep_batched(_p(wp["tab"]), wp["n"], _p(params), _p(u0), *[_p(b) for b in bufs], _p(part), wp["blocks"],
                                         self._stream()), "xmc_wprep_batched")
        return bufs, part

    def wprep_weights(self, wp, i, bufs):
        """the prepared forward / dgrad PackedWeights of entry ``i`` (with their phase copies attached)"""
        e = wp["entries"][i]
        cout, cin, taps = e["cout"], e["cin"], e["taps"]
        wf_b, wd_b, pf_b, pd_b = bufs
        if e["plain"]:
            f = self._with_mx8(PackedWeight(wf_b[e["wf_off"]:e["wf_off"] + e["nw"]], cout, taps, cin))
            d = self._with_mx8(PackedWeight(wd_b[e["wd_off"]:e["wd_off"] + e["nw"]], cin, taps, cout))
        else:
            f, d = PackedWeight(None, cout, taps, cin), PackedWeight(None, cin, taps, cout)
        if e["phase_eff"]:
            mode = {"ups": 0, "pool": 1}[e["phase_eff"]]
            f.phase = (("out", "in")[mode], pf_b[e["pf_off"]:e["pf_off"] + e["nph"]])
            d.phase = (("in", "out")[mode], pd_b[e["pd_off"]:e["pd_off"] + e["nph"]])
        return f, d

    def sn_bank_power_iter_fused(self, bank, irr, wp, params, u0_flat, part, eps=1e-10):
        """sn_bank_power_iter with the first product of ``wp``'s weights taken from ``part`` (wprep_run); ``irr``: the bank of
        the remaining weights (or None)"""
        u_new = self.empty((bank["nu"],), torch.float32)
        u_raw = self.empty((bank["nu"],), torch.float32)
        v = self.empty((bank["nv"],), torch.float32)
        scal = self.empty((2 * bank["n"],), torch.float32)
        check(self.lib.xmc_sn_power_iter_fused(
            _p(bank["tab_prep"]), bank["n"], bank["blocks_a"], bank["blocks_b"],
            _p(irr["tab_prep"]) if irr else None, irr["n"] if irr else 0, irr["blocks_a"] if irr else 0, irr["blocks_b"] if irr else 0,
            _p(wp["tab"]), wp["n"], wp["blocks_c"], _p(params), _p(u0_flat), _p(part), _p(u_new), _p(v), _p(u_raw), _p(scal), eps,
            self._stream()), "xmc_sn_power_iter_fused")
        return u_new, v, scal

    def sn_bank_dot(self, bank, params, grads, scal):
        """kvec[i] = <G_i, W_i> / (sigma_i + eps): first half of the gradient through sigma (the rest: adam_ema_dev_sn)"""
        dots = self.empty((bank["blocks_d"],), torch.float32)
        kvec = self.empty((bank["n"],), torch.float32)
        check(self.lib.xmc_sn_batched_dot(_p(bank["tab_fix"]), bank["n"], _p(params), _p(grads), _p(scal), _p(dots), _p(kvec),
                                          bank["blocks_d"], self._stream()), "xmc_sn_batched_dot")
        return kvec

    def sn_bank_map(self, bank, arena_size, align=64):
        """one int16 per ``align`` arena elements: the bank entry that owns them, or -1 (xmc_adam_ema_dev_sn)"""
        assert arena_size % align == 0
        m = torch.full((arena_size // align,), -1, dtype=torch.int16)
        for i, e in enumerate(bank["entries"]):
            assert e["w_off"] % align == 0
            m[e["w_off"] // align:(e["w_off"] + e["rows"] * e["cols"] + align - 1) // align] = i
        return m.to(self.device)

    def wprep_skip_map(self, wp, arena_size, base=None, align=64):
        """the int16-per-64-elements map of xmc_adam_ema_dev_sn with -2 ("xmc_adam_wprep_tiles updates this tensor") on the
        weights of ``wp``; ``base``: a spectral map (``sn_bank_map``) to start from, else all -1"""
        m = base.cpu().clone() if base is not None else torch.full((arena_size // align,), -1, dtype=torch.int16)
        for e in wp["entries"]:
            assert e["w_off"] % align == 0 and (e["cout"] * e["taps"] * e["cin"]) % align == 0
            m[e["w_off"] // align:(e["w_off"] + e["cout"] * e["taps"] * e["cin"]) // align] = -2
        return m.to(self.device)

    def adam_wprep(self, wp, out, p, g, m, v, ema, step_state, *, lr, beta1, beta2, eps=1e-8, grad_scale=1.0, ema_decay=0.0,
                   zero_grads=True, fix=None):
        """xmc_adam_wprep_tiles: the Adam (+ EMA) update of ``wp``'s weights fused with their preparation into ``out`` = (bufs,
        part); ``step_state`` already advanced by ``adam_ema_dev_sn`` on the same arena (whose map skipped these tensors).
        ``fix`` = (kvec, scal, u, v) for a spectral arena."""
        bufs, part = out
        fa = tuple(_p(t) for t in fix) if fix is not None else (None, None, None, None)
        check(self.lib.xmc_adam_wprep_tiles(_p(wp["tab"]), wp["n"], wp["blocks"], _p(p), _p(g), _p(m), _p(v), _p(ema), lr, beta1, beta2,
                                            eps, _p(step_state), grad_scale, ema_decay, 2 if self.keep_grads else int(bool(zero_grads)),
                                            *fa, *[_p(b) for b in bufs], _p(part), self._stream()), "xmc_adam_wprep_tiles")

    def adam_ema_dev_sn(self, p, g, m, v, ema, step_state, *, lr, beta1, beta2, eps=1e-8, grad_scale=1.0, ema_decay=0.0,
                        zero_grads=True, fix=None, skip_map=None):
        """adam_ema_dev that zeroes the consumed gradient in place and, with ``fix`` = (map, bank, kvec, scal, u, v), applies the
        gradient through sigma on the fly; ``skip_map`` (no ``fix``): tensors marked -2 are left alone (``adam_wprep``)"""
        assert step_state.dtype == torch.float32 and step_state.numel() >= 4
        if fix is not None:
            mp, bank, kvec, scal, u, vv = fix
            args = (_p(mp), _p(bank["tab_fix"]), bank["n"], _p(kvec), _p(scal), _p(u), _p(vv))
        elif skip_map is not None:
            args = (_p(skip_map), None, 0, None, None, None, None)
        else:
            args = (None, None, 0, None, None, None, None)
        check(self.lib.xmc_adam_ema_dev_sn(_p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), lr, beta1, beta2, eps, _p(step_state),
                                           grad_scale, ema_decay, 2 if self.keep_grads else int(bool(zero_grads)), *args, self._stream()),
              "xmc_adam_ema_dev_sn")
        return not self.keep_grads and bool(zero_grads)        # True: the gradient arena is all zeros again

    # ---------------------------------------------------------------------------------- optimiser
    def adam_ema(self, p, g, m, v, ema, *, lr, beta1, beta2, step, eps=1e-8, grad_scale=1.0, ema_decay=0.0):
        c1, c2 = 1.0 - beta1 ** step, 1.0 - beta2 ** step
        check(self.lib.xmc_adam_ema(_p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), lr, beta1, beta2, eps, c1, c2,
                                    grad_scale, ema_decay, self._stream()), "xmc_adam_ema")

    def adam_ema_dev(self, p, g, m, v, ema, step_state, *, lr, beta1, beta2, eps=1e-8, grad_scale=1.0, ema_decay=0.0):
        """Adam (+EMA) with the step counter / bias corrections in device memory (``step_state``, 4 float32 slots;
        advanced by one per call) -- the form a captured hipGraph can replay."""
        assert step_state.dtype == torch.float32 and step_state.numel() >= 4
        check(self.lib.xmc_adam_ema_dev(_p(p), _p(g), _p(m), _p(v), _p(ema), p.numel(), lr, beta1, beta2, eps,
                                        _p(step_state), grad_scale, ema_decay, self._stream()), "xmc_adam_ema_dev")

    # ------------------------------------------------------- frozen ResNet-50 feature path (canvases)
    def resize_to_canvas(self, x, hd, hc):
        """(n, hs, ws, c) -> bilinear (half-pixel centres) to hd x hd inside a zero-margin (n, hc, hc, c) canvas"""
        n, hs, ws, c = x.shape
        y = self.empty((n, hc, hc, c), x.dtype)
        check(self.lib.xmc_resize_bilinear(_p(x), _p(y), n, hs, ws, c, hd, hd, hc, hc, 0, _code(x.dtype), self._stream()),
              "xmc_resize_bilinear")
        return y

    def resize_to_canvas_bwd(self, dy, hs, hd):
        n, hc, _, c = dy.shape
        dx = self.empty((n, hs, hs, c), dy.dtype)
        check(self.lib.xmc_resize_bilinear(_p(dy), _p(dx), n, hs, hs, c, hd, hd, hc, hc, 1, _code(dy.dtype), self._stream()),
              "xmc_resize_bilinear")
        return dx

    def stem_im2col(self, x, hv, ho, kp=160):
        """image canvas (n, hc, hc, 3), valid hv -> col canvas (n, ho, ho, kp) of the 7x7 stride-2 SAME stem"""
        n, hc, _, _ = x.shape
        col = self.empty((n, ho, ho, kp), x.dtype)
        check(self.lib.xmc_stem_im2col(_p(x), _p(col), n, hc, hc, hv, hv, ho, ho, kp, 0, _code(x.dtype), self._stream()),
              "xmc_stem_im2col")
        return col

    def stem_conv(self, x, wfrag, bias, hv, hov, out):
        """image canvas (n, hc, hc, 3) with valid rows ``hv`` -> ``out`` (n, ho, ho, 64) (valid corner ``hov`` written; margins
        untouched): the 7x7 stride-2 stem as one implicit-GEMM launch (xmc_stem_conv7x7s2), ``wfrag`` from ``pack_stem_weight``"""
        n, hc, wc, c = x.shape
        assert c == 3 and x.dtype == torch.bfloat16 and out.dtype == torch.bfloat16 and out.shape[0] == n and out.shape[3] == 64
        check(self.lib.xmc_stem_conv7x7s2(_p(x), _p(wfrag), _p(bias), _p(out), n, hc, wc, hv, out.shape[1], out.shape[2], hov, hov,
                                          self._stream()), "xmc_stem_conv7x7s2")
        return out

    def pack_stem_weight(self, w):
        """folded stem weights (64, 49, 3) float32 (tap = ky * 7 + kx) -> the fragment order of xmc_stem_conv7x7s2 (bf16, device)"""
        import numpy as np
        w = np.asarray(w, np.float32).reshape(64, 7, 21)                       # [cout][ky][kx * 3 + ch]
        wk = np.zeros((64, 8, 24), np.float32)
        wk[:, :7, :21] = w
        wk = wk.reshape(64, 192)[:, :176]                                       # k' = ky * 24 + j
        frag = np.zeros((2, 11, 64, 8), np.float32)
        lane = np.arange(64)
        for cb in range(2):
            for ks in range(11):
                k0 = ks * 16 + (lane >> 5) * 8
                frag[cb, ks] = np.stack([wk[cb * 32 + (l & 31), k0[l]:k0[l] + 8] for l in range(64)])
        return torch.as_tensor(frag).to(self.device).to(torch.bfloat16).contiguous()

    def stem_dgrad(self, ds, wfrag, hov, hc):
        """ds canvas (n, ho, ho, 64), valid ``hov`` -> d image canvas (n, hc, hc, 3) (valid corner 2 hov written): the stem's data
        gradient as one launch (xmc_stem_conv7x7s2_dgrad), ``wfrag`` from ``pack_stem_dgrad_weight``"""
        n, ho, wo, c = ds.shape
        assert c == 64 and ds.dtype == torch.bfloat16 and ds.is_contiguous()
        dx = self.empty((n, hc, hc, 3), ds.dtype)
        check(self.lib.xmc_stem_conv7x7s2_dgrad(_p(ds), _p(wfrag), _p(dx), n, ho, wo, hov, hov, hc, hc, self._stream()),
              "xmc_stem_conv7x7s2_dgrad")
        return dx

    def pack_stem_dgrad_weight(self, w):
        """folded stem weights (64, 49, 3) float32 -> the fragment order of xmc_stem_conv7x7s2_dgrad (bf16, device)"""
        import numpy as np
        w = np.asarray(w, np.float32).reshape(64, 7, 7, 3)                       # [co][ky][kx][c]
        wd = np.zeros((32, 16, 64), np.float32)                                  # [row r][tap t * 4 + u][co]
        for py in range(2):
            for px in range(2):
                for t in range(4):
                    for u in range(4):
                        ky, kx = 2 * t + py, 2 * u + px
                        if ky <= 6 and kx <= 6:
                            r0 = (2 * py + px) * 3
                            wd[r0:r0 + 3, t * 4 + u, :] = w[:, ky, kx, :].T
        frag = np.zeros((2, 16, 2, 64, 8), np.float32)
        for half in range(2):
            for tap in range(16):
                for s in range(2):
                    for l in range(64):
                        co0 = half * 32 + s * 16 + (l >> 5) * 8
                        frag[half, tap, s, l] = wd[l & 31, tap, co0:co0 + 8]
        return torch.as_tensor(frag).to(self.device).to(torch.bfloat16).contiguous()

    def stem_col2im(self, dcol, hc, hv):
        n, ho, _, kp = dcol.shape
        dx = self.empty((n, hc, hc, 3), dcol.dtype)
        check(self.lib.xmc_stem_im2col(_p(dcol), _p(dx), n, hc, hc, hv, hv, ho, ho, kp, 1, _code(dcol.dtype), self._stream()),
              "xmc_stem_im2col")
        return dx

    def maxpool3x3s2(self, x, hv):
        """-> (y, idx): idx (uint8) names each window's first maximum, for maxpool3x3s2_bwd"""
        n, hc, wc, c = x.shape
        y = self.empty((n, hc // 2, wc // 2, c), x.dtype)
        idx = torch.empty((n, hc // 2, wc // 2, c), dtype=torch.uint8, device=self.device)
        check(self.lib.xmc_maxpool3x3s2(_p(x), _p(y), _p(idx), n, hc, wc, c, hv, hv, _code(x.dtype), self._stream()),
              "xmc_maxpool3x3s2")
        return y, idx

    def maxpool3x3s2_bwd(self, dy, idx, hv):
        n, ho, wo, c = dy.shape
        dx = self.empty((n, 2 * ho, 2 * wo, c), dy.dtype)
        check(self.lib.xmc_maxpool3x3s2_bwd(_p(dy), _p(idx), _p(dx), n, 2 * ho, 2 * wo, c, hv, hv, _code(dy.dtype),
                                            self._stream()), "xmc_maxpool3x3s2_bwd")
        return dx

    def zero_margin_(self, x, hv):
        n, hc, wc, c = x.shape
        check(self.lib.xmc_zero_margin(_p(x), n, hc, wc, c, hv, hv, _code(x.dtype), self._stream()), "xmc_zero_margin")
        return x

    def subsample2(self, x, off):
        n, hc, wc, c = x.shape
        y = self.empty((n, hc // 2, wc // 2, c), x.dtype)
        check(self.lib.xmc_subsample2(_p(x), _p(y), n, hc, wc, c, off, 0, _code(x.dtype), self._stream()), "xmc_subsample2")
        return y

    def subsample2_bwd(self, dy, off):
        n, ho, wo, c = dy.shape
        dx = self.empty((n, 2 * ho, 2 * wo, c), dy.dtype)
        check(self.lib.xmc_subsample2(_p(dx), _p(dy), n, 2 * ho, 2 * wo, c, off, 1, _code(dy.dtype), self._stream()),
              "xmc_subsample2")
        return dx

    def add_relu(self, a, b=None):
        out = torch.empty_like(a)
        check(self.lib.xmc_add_relu(_p(a), _p(b), _p(out), a.numel(), _code(a.dtype), self._stream()), "xmc_add_relu")
        return out

    def relu_bwd(self, dy, out, dy2=None):
        g = torch.empty_like(out)
        check(self.lib.xmc_relu_bwd(_p(dy), _p(dy2), _p(out), _p(g), out.numel(), _code(out.dtype), self._stream()),
              "xmc_relu_bwd")
        return g

    def probe_layouts(self):
        out = self.zeros((2 * 64 * 16 + 64 * 4,))
        check(_lib.load_probe().xmc_probe_layouts(_p(out), self._stream()), "xmc_probe_layouts")      # libxmc_probe.so: diagnostics
        return out
