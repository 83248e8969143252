"""MX-fp8 convolution path (BASELINE config #5) on the MI355X: operand layout of the block-scaled MFMA, the MX quantiser
(OCP MX v1.0: e4m3 elements, e8m0 scale per 32 channels), xmc_conv2d_mx8 forward / data-gradient use against float64
``F.conv2d``, and the training step with ``config.conv_fp8`` against the float32 oracle."""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def e4m3_decode_table():
    """OCP e4m3fn: 1-4-3, bias 7, no infinities, 0x7f / 0xff = NaN"""
    t = np.zeros(256, np.float64)
    for b in range(256):
        s, e, m = b >> 7, (b >> 3) & 0xF, b & 7
        v = (m / 8.0) * 2.0 ** -6 if e == 0 else (1 + m / 8.0) * 2.0 ** (e - 7)
        if e == 15 and m == 7:
            v = np.nan
        t[b] = -v if s else v
    return t


def lossless_mx(shape_blocks, gen, zero_frac=0.1):
    """float32 array (..., 32 * nblocks) whose MX-fp8 quantisation is EXACT: every value is a 4-bit-significand number
    within 2^-4 of its 32-block's largest magnitude; block magnitudes span 2^-20 .. 2^12."""
    *lead, nb = shape_blocks
    k = gen.integers(-20, 13, size=(*lead, nb, 1))
    m = gen.integers(0, 7, size=(*lead, nb, 32))          # <= 1.75: 1.875 * 2^8 = 480 would saturate at e4m3's 448
    r = gen.integers(-4, 1, size=(*lead, nb, 32))
    sgn = gen.choice([-1.0, 1.0], size=(*lead, nb, 32))
    v = sgn * (1 + m / 8.0) * 2.0 ** (k + r)
    v[gen.random(v.shape) < zero_frac] = 0.0
    return v.reshape(*lead, nb * 32).astype(np.float32)


def test_scaled_mfma_operand_layout():
    """xmc_mx8_probe: D = (A * 2^(sa - 127)) (B * 2^(sb - 127))^T with A, B given as plain [32][64] byte matrices and one
    scale byte per (row, 32-wide K block) -- pins lane -> (row = lane % 32, K block = lane / 32) and the per-lane scale."""
    from xmcgan_image_generation_amd import _lib
    lib = _lib.load()
    gen = np.random.default_rng(0)
    tab = e4m3_decode_table()
    a8 = gen.integers(0, 256, size=(32, 64), dtype=np.uint8)
    b8 = gen.integers(0, 256, size=(32, 64), dtype=np.uint8)
    a8[(a8 & 0x7F) == 0x7F] = 0x38                       # no NaN encodings
    b8[(b8 & 0x7F) == 0x7F] = 0xB8
    sa = gen.integers(118, 136, size=(32, 2), dtype=np.uint8)
    sb = gen.integers(118, 136, size=(32, 2), dtype=np.uint8)
    av = tab[a8] * np.repeat(2.0 ** (sa.astype(np.float64) - 127), 32, axis=1)
    bv = tab[b8] * np.repeat(2.0 ** (sb.astype(np.float64) - 127), 32, axis=1)
    want = av @ bv.T
    dev = [torch.from_numpy(t).cuda() for t in (a8, sa, b8, sb)]
    d = torch.zeros((32, 32), dtype=torch.float32, device="cuda")
    _lib.check(lib.xmc_mx8_probe(*[C.c_void_p(t.data_ptr()) for t in dev], C.c_void_p(d.data_ptr()), None), "probe")
    got = d.cpu().numpy().astype(np.float64)
    # the matrix pipe aligns the 64 products of an instruction before adding them (not an exact float32 fma chain): the
    # error is measured against sum |a| |b| of each entry; measured 1.9e-4, a wrong lane / K-block / scale mapping gives O(1)
    bound = np.abs(av) @ np.abs(bv).T
    ratio = float((np.abs(got - want) / bound).max())
    print("scaled MFMA vs exact: max |error| / sum |a||b| =", ratio)
    assert ratio < 1e-3, ratio


@pytest.mark.parametrize("rule", ["next_binade", "ocp_floor"])
def test_mx8_quantizer_scale_rules(rule):
    """The MX quantiser under both scale rules (config.fp8_scale_rule): "ocp_floor" is the OCP MX v1.0 conversion to the letter
    -- X = 2^(floor(log2 amax) - emax), emax(e4m3) = 8, elements RNE with saturation at 448 -- "next_binade" (the build's default,
    a deliberate deviation: commit 9f9ee5c) takes X one binade higher exactly when amax / X would exceed 448, so nothing saturates."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    ops.set_fp8_scale_rule(rule)
    try:
        gen = torch.Generator().manual_seed(1)
        tab = e4m3_decode_table()
        clipped = 0
        for c in (96, 64, 200):
            x = torch.randn((37, 5, c), generator=gen) * torch.exp2(torch.randint(-18, 10, (37, 5, 1), generator=gen).float())
            x[3] = 0.0                                                          # all-zero blocks
            x = x.bfloat16()
            for relu in (False, True):
                pk = ops.quantize_mx8(x.cuda(), relu=relu).cpu().numpy()
                cp = (c + 63) // 64 * 64
                assert pk.shape == (185, cp // 64, 80)
                x8 = pk[:, :, :64].reshape(185, cp)                                  # elements of the packets
                s = pk[:, :, 64:66].reshape(185, cp // 32)                           # their two scale bytes
                ref = x.double().reshape(185, c)
                if relu:
                    ref = ref.clamp_min(0)
                ref = F.pad(ref, (0, cp - c)).numpy()
                blocks = np.abs(ref).reshape(185, cp // 32, 32).max(-1)
                with np.errstate(divide="ignore"):
                    ex = np.floor(np.log2(np.maximum(blocks, 1e-300)))
                    if rule == "next_binade":
                        ex = ex + (blocks / 2.0 ** ex > 1.75)                       # no saturating block maximum
                    want_s = np.where(blocks > 0, ex - 8 + 127, 0).clip(0, 254)
                assert np.array_equal(s, want_s.astype(np.uint8)), (c, relu)
                x_unit = np.repeat(2.0 ** (s.astype(np.float64) - 127), 32, axis=1)
                deq = tab[x8] * x_unit
                # e4m3: 3 mantissa bits -> relative error <= 2^-4 for normal elements; elements below 2^-6 of the scale unit are
                # subnormal (absolute error <= 2^-10 X).  next_binade: amax / X in (224, 448], nothing saturates; ocp_floor: amax / X in
                # [256, 512), elements above 448 X saturate to 448 X (the specification's clamp)
                if rule == "next_binade":
                    assert (np.abs(ref) <= 448 * x_unit).all()
                sat = np.clip(ref, -448 * x_unit, 448 * x_unit)
                clipped += int((sat != ref).sum())
                err = np.abs(deq - sat)
                assert (err <= np.maximum(2.0 ** -4 * np.abs(sat), 2.0 ** -10 * x_unit) + 1e-300).all(), (c, relu, err.max())
        assert (clipped > 0) == (rule == "ocp_floor"), clipped       # the floor rule really clips block maxima on Gaussian data
    finally:
        ops.set_fp8_scale_rule("next_binade")


def test_train_step_fp8_ocp_floor_rule_switch():
    """config.fp8_scale_rule = "ocp_floor" reaches every quantiser of the step (weights, conditional-BatchNorm packets, convolution
    epilogues, stand-alone passes): the step's losses differ from the default rule's, stay within the fp8 bar of the float32 oracle,
    and the default rule is back afterwards (the knob is process-wide)."""
    from tests.test_gpu_step import _c1_b8_oracle
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    o = _c1_b8_oracle()
    tb = {k: torch.as_tensor(v).cuda() for k, v in o["batch"].items()}
    got = {}
    try:
        for rule in ("next_binade", "ocp_floor"):
            cfg = o["cfg"].copy()
            cfg.dtype = "bfloat16"
            cfg.conv_fp8 = True
            cfg.fp8_scale_rule = rule
            gen, disc, state = train_utils.create_train_state(cfg, 0)
            assert gen(train=True).ops.fp8_scale_rule == rule
            state = train_utils.load_flax_params(state, *o["init"])
            state, m = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
            got[rule] = {k: float(v) for k, v in m.items()}
            del state, gen, disc
    finally:
        from xmcgan_image_generation_amd import _lib
        _lib.check(_lib.load().xmc_set_tuning(b"mx8_scale_floor", -1), "reset")
    ref = o["ref_metrics"]
    scale = max(abs(float(ref[k])) for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"))
    print("fp8 scale rules:", got)
    assert got["next_binade"] != got["ocp_floor"]
    for rule in got:
        for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
            r = abs(got[rule][k] - float(ref[k])) / scale
            assert np.isfinite(got[rule][k]) and r < (1e-1 if k in ("d_loss", "g_loss") else 1e-2), (rule, k, got[rule][k], float(ref[k]))


CASES = [
    # n, h, cin, cout, flags
    (4, 16, 64, 128, {}),
    (2, 32, 96, 96, {}),                                   # ragged: cin padded to 128, cout tile 3 of 4 blocks
    (2, 16, 192, 64, dict(ups=True)),
    (3, 8, 128, 160, dict(res=True, mask=True, bias=True, alpha=0.5)),
    (2, 64, 64, 96, dict(pool_out=True, bias=True)),
    (2, 8, 1024, 256, dict(split_k=True)),
    (2, 16, 128, 128, dict(res_ups=True, ups=True, relu_in=True)),
]


@pytest.mark.parametrize("n,h,cin,cout,fl", CASES)
def test_conv_mx8_exact_on_lossless_operands(n, h, cin, cout, fl):
    """Operands whose MX quantisation is exact (4-bit significands within 2^-4 of their block maximum, block magnitudes
    2^-20 .. 2^12): xmc_conv2d_mx8 must then equal the float64 convolution up to float32 accumulation -- taps, halo,
    upsampling gather, channel padding, per-block scales on both operands, split-K and every epilogue option."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    ops.fp8 = "all"                                        # also the padded-row case (cin = 96) the step leaves to the bf16 kernel
    gen = np.random.default_rng(n * 1000 + h + cin)
    cp = (cin + 31) // 32 * 32
    x = torch.from_numpy(lossless_mx((n, h, h, cp // 32), gen))[..., :cin].contiguous()
    # weight blocks run along cin for each (cout, tap); keep |w| small so that products stay well inside float32
    w = torch.from_numpy(lossless_mx((cout, 9, cp // 32), gen))[..., :cin].contiguous() * 2.0 ** -8
    if fl.get("relu_in"):
        x = x.abs() * torch.sign(torch.randn(x.shape))     # real sign mix: the ReLU must zero the negatives
    xb, wb = x.bfloat16(), w.bfloat16()
    assert torch.equal(xb.float(), x) and torch.equal(wb.float(), w)          # 4-bit significands: exact in bf16 too
    ups = bool(fl.get("ups"))
    ho = 2 * h if ups else h
    wf, _ = ops.prep_conv_weight(w.cuda(), None, False)
    assert hasattr(wf, "taps"), "fragment-packed weights expected"
    bias = torch.randn(cout).cuda() if fl.get("bias") else None
    oh = ho // 2 if fl.get("pool_out") else ho
    res = None
    if fl.get("res"):
        res = torch.randn((n, oh, oh, cout)).bfloat16().cuda()
    if fl.get("res_ups"):
        res = torch.randn((n, oh // 2, oh // 2, cout)).bfloat16().cuda()
    mask = (torch.randn((n, oh, oh, cout)) > 0).to(torch.bfloat16).cuda() if fl.get("mask") else None
    if not fl.get("split_k"):
        ops.no_split_k = True
    y = ops.conv(xb.cuda(), wf, bias, ks=3, ups=ups, relu_in=bool(fl.get("relu_in")), mask=mask, res=res,
                 res_ups=bool(fl.get("res_ups")), alpha=fl.get("alpha", 1.0), out_f32=True, pool_out=bool(fl.get("pool_out")))
    xin = x.double().clamp_min(0) if fl.get("relu_in") else x.double()
    xin = xin.permute(0, 3, 1, 2)
    if ups:
        xin = xin.repeat_interleave(2, 2).repeat_interleave(2, 3)
    ref = F.conv2d(xin, w.double().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), padding=1) * fl.get("alpha", 1.0)
    if bias is not None:
        ref = ref + bias.double().cpu().view(1, -1, 1, 1)
    if fl.get("pool_out"):
        ref = F.avg_pool2d(ref, 2)
        if bias is not None:                               # the kernel adds the bias after the pooling scale: same thing
            pass
    ref = ref.permute(0, 2, 3, 1)
    if mask is not None:
        ref = ref * (mask.double().cpu() > 0)
    if res is not None:
        r = res.double().cpu()
        if fl.get("res_ups"):
            r = r.repeat_interleave(2, 1).repeat_interleave(2, 2)
        ref = ref + r
    got = y.double().cpu()
    # error bar: the same convolution of |x|, |w| (the matrix pipe's product alignment + float32 accumulation are relative
    # to the magnitude of the terms, and the block magnitudes span 2^32 here)
    mag = F.conv2d(xin.abs(), w.double().abs().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2), padding=1) * abs(fl.get("alpha", 1.0))
    if fl.get("pool_out"):
        mag = F.avg_pool2d(mag, 2)
    mag = mag.permute(0, 2, 3, 1) + ref.abs()
    ratio = float(((got - ref).abs() / mag.clamp_min(1e-30)).max())
    print("conv_mx8 lossless case", (n, h, cin, cout, fl), "max |error| / magnitude:", ratio)
    assert ratio < 2e-4, ratio                            # measured 1-3e-5 (bf16 residual / float32 output rounding included)


@pytest.mark.parametrize("relu,pool,mask", [(True, False, False), (False, False, True), (True, True, False)])
def test_conv_mx8_epilogue_emits_the_next_layers_packets(relu, pool, mask):
    """``emit_mx8``: the MX-fp8 convolution's epilogue writes its bf16 output AND that output's packets for the next
    convolution (the consumer's relu_in folded in) -- byte for byte what the separate quantisation pass writes."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    ops.fp8 = True
    ops.no_split_k = True
    g = torch.Generator().manual_seed(11)
    n, h, cin, cout = 3, 32, 128, 192
    x = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
    w = (torch.randn((cout, 9, cin), generator=g) * 0.05).cuda()
    wf, _ = ops.prep_conv_weight(w, None, False)
    oh = h // 2 if pool else h
    res = torch.randn((n, oh, oh, cout), generator=g).bfloat16().cuda()
    m = (torch.randn((n, oh, oh, cout), generator=g) > 0).to(torch.bfloat16).cuda() if mask else None
    y = ops.conv(x, wf, torch.randn(cout, generator=g).cuda(), ks=3, relu_in=True, res=res, mask=m, pool_out=pool, emit_mx8=relu)
    assert getattr(y, "mx8", None) is not None and y.mx8[1] == relu
    want = ops.quantize_mx8(y, relu=relu)
    got = y.mx8[0]
    assert got.shape == want.shape
    assert torch.equal(got[:, :, :66], want[:, :, :66])                  # 64 elements + 2 scale bytes of every packet
    # and the consumer takes them: same result as quantising again
    w2 = (torch.randn((64, 9, cout), generator=g) * 0.05).cuda()
    wf2, _ = ops.prep_conv_weight(w2, None, False)
    z1 = ops.conv(y, wf2, None, ks=3, relu_in=relu)
    y2 = y.clone()                                                       # no packets attached
    z2 = ops.conv(y2, wf2, None, ks=3, relu_in=relu)
    assert torch.equal(z1, z2)


@pytest.mark.parametrize("split_k", [False, True])
def test_conv_mx8_relu_on_store_and_bit_masks(split_k):
    """round 5 (xmc_conv2d_mx8_bits): the MX-fp8 kernel's epilogue stores max(., 0), writes (y > 0) as bits and reads its ReLU
    mask as bits -- exactly the plain launch followed by those operations; the packets of a ReLU-stored output serve a
    consumer with either relu_in."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    ops.fp8 = True
    ops.no_split_k = not split_k
    g = torch.Generator().manual_seed(5)
    n, h, cin, cout = (56, 8, 768, 192) if split_k else (3, 32, 128, 192)
    x = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
    w = (torch.randn((cout, 9, cin), generator=g) * 0.05).cuda()
    wf, _ = ops.prep_conv_weight(w, None, False)
    bias = torch.randn(cout, generator=g).cuda()
    plain = ops.conv(x, wf, bias, ks=3, relu_in=True)
    y = ops.conv(x, wf, bias, ks=3, relu_in=True, relu_out=True, emit_mx8=True, emit_bits=True)
    assert torch.equal(y, torch.clamp_min(plain.float(), 0).bfloat16())
    if not split_k:
        bits = y.bits.view(n, h, h, cout // 16).to(torch.int32) & 0xffff
        want = ((y.float() > 0).view(n, h, h, cout // 16, 16).to(torch.int32) << torch.arange(16, device="cuda", dtype=torch.int32)).sum(-1)
        assert torch.equal(bits, want)
        assert y.mx8[1] == "relu" and torch.equal(y.mx8[0][:, :, :66], ops.quantize_mx8(y, relu=False)[:, :, :66])
        # the mask as bits == the mask as a bf16 tensor
        dy = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
        wd = (torch.randn((cout, 9, cin), generator=g) * 0.05).cuda()
        wdf, _ = ops.prep_conv_weight(wd, None, False)
        a = ops.conv(dy, wdf, None, ks=3, mask=y)                       # y carries .bits
        yc = y.clone()                                                  # ... and this copy does not
        b = ops.conv(dy, wdf, None, ks=3, mask=yc)
        assert torch.equal(a, b)
        # a consumer with relu_in = False takes the packets of the ReLU-stored tensor
        w2 = (torch.randn((64, 9, cout), generator=g) * 0.05).cuda()
        wf2, _ = ops.prep_conv_weight(w2, None, False)
        for r in (False, True):
            assert torch.equal(ops.conv(y, wf2, None, ks=3, relu_in=r), ops.conv(yc, wf2, None, ks=3, relu_in=r))


def test_cbn_act_emits_packets():
    """the conditional-BatchNorm + ReLU kernel writes the packets of its output when config.conv_fp8 is on: equal to the
    separate quantisation pass byte for byte"""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(2)
    n, h, c, hc = 3, 16, 192, 1
    x = torch.randn((n, h, h, c), generator=g).bfloat16().cuda()
    mean, rstd = torch.randn(c, generator=g).cuda() * 0.1, (torch.rand(c, generator=g) + 0.5).cuda()
    gb = (torch.randn((n * hc * hc, 2 * c), generator=g) * 0.3).cuda()
    y0 = ops.cbn_act_fwd(x, mean, rstd, gb, hc, relu=True)
    assert getattr(y0, "mx8", None) is None
    ops.fp8 = True
    y1 = ops.cbn_act_fwd(x, mean, rstd, gb, hc, relu=True)
    assert torch.equal(y0, y1) and y1.mx8[1] is False
    want = ops.quantize_mx8(y1, relu=False)
    assert torch.equal(y1.mx8[0][:, :, :66], want[:, :, :66])


def test_conv_mx8_accuracy_on_gaussian_data_and_dgrad_adjoint():
    """Generic data: the MX-fp8 convolution against the exact one (norm-relative error of the output, e4m3 has 3 mantissa
    bits) and, through the prepared dgrad weights, the adjoint identity <dy, conv(x, W)> ~ <x, dgrad(dy, W)>."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    n, h, cin, cout = 4, 32, 192, 96
    x = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
    dy = (torch.randn((n, h, h, cout), generator=g) * 1e-4).bfloat16().cuda()          # gradient-sized values
    w = (torch.randn((cout, 9, cin), generator=g) * 0.03).cuda()
    wf, wd = ops.prep_conv_weight(w, None, True)
    y16 = ops.conv(x, wf, None, ks=3, out_f32=True)
    dx16 = ops.conv(dy, wd, None, ks=3, out_f32=True)
    ops.fp8 = True
    y8 = ops.conv(x, wf, None, ks=3, out_f32=True)
    dx8 = ops.conv(dy, wd, None, ks=3, out_f32=True)
    for name, a, b in (("fwd", y8, y16), ("dgrad", dx8, dx16)):
        rel = float((a - b).norm() / b.norm())
        print(f"MX-fp8 vs bf16 {name}: norm-relative difference {rel:.3e}")
        assert rel < 6e-2, (name, rel)
    lhs = float((y8.double() * dy.double()).sum())
    rhs = float((dx8.double() * x.double()).sum())
    assert abs(lhs - rhs) <= 5e-2 * (abs(lhs) + float((y8.double() * dy.double()).abs().sum()) * 1e-2), (lhs, rhs)


def test_train_step_conv_fp8_vs_fp32_oracle():
    """config.conv_fp8 (BASELINE config #5) at the C1 network, per-device batch 8, against the float32 oracle.  SURVEY 8(d):
    the reduced-precision modes are REPORTED against the 2e-2 bar, not gated on it; measured here 0.5e-2 .. 2.1e-2 of the
    loss scale on d_loss (the hinge term of a random-init discriminator at batch 8 amplifies the ~4 % per-convolution
    fp8 noise), < 2e-3 on the contrastive losses.  The hinge terms are a DRAW of that noise, not a bias: three builds of
    round 3 that differ only in the summation order of float32 reductions elsewhere in the step (bit-identical in the bf16
    and float32 modes to 1e-6) gave 0.2e-2, 2.7e-2 and 5.1e-2 on g_loss -- one flipped fp8 rounding early in D re-draws
    everything downstream.  tools/poison_check.py --fp8 shows the step is deterministic and reads no uninitialised memory.
    Gate: 1e-1 on the hinge losses, 1e-2 on the contrastive ones.  Second step finite."""
    from tests.test_gpu_step import _c1_b8_oracle
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    o = _c1_b8_oracle()
    cfg = o["cfg"].copy()
    cfg.dtype = "bfloat16"
    cfg.conv_fp8 = True
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    assert gen(train=True).ops.fp8
    state = train_utils.load_flax_params(state, *o["init"])
    tb = {k: torch.as_tensor(v).cuda() for k, v in o["batch"].items()}
    state, m = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    ref = o["ref_metrics"]
    scale = max(abs(float(ref[k])) for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"))
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = abs(float(m[k]) - float(ref[k])) / scale
        print("conv_fp8 C1 b8", k, float(m[k]), float(ref[k]), r)
        assert np.isfinite(float(m[k])) and r < (1e-1 if k in ("d_loss", "g_loss") else 1e-2), (k, float(m[k]), float(ref[k]))
    state, m2 = train_utils.train_step(1, state, tb, xmc_gan, gen, disc, cfg, {})
    assert all(np.isfinite(float(v)) for v in m2.values())
    assert bool(torch.isfinite(state.g_optimizer.arena.params).all()) and bool(torch.isfinite(state.d_optimizer.arena.params).all())


def _run_steps(cfg, init, batches, nsteps=1):
    """-> list of per-step metric dicts of ``nsteps`` train_steps from ``init`` over ``batches`` (cycled)"""
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    state = train_utils.load_flax_params(state, *init)
    out = []
    for s in range(nsteps):
        state, m = train_utils.train_step(s, state, batches[s % len(batches)], xmc_gan, gen, disc, cfg, {})
        out.append({k: float(v) for k, v in m.items()})
    fin = bool(torch.isfinite(state.g_optimizer.arena.params).all()) and bool(torch.isfinite(state.d_optimizer.arena.params).all())
    del state, gen, disc
    torch.cuda.empty_cache()
    return out, fin


def test_c4_workload_256px_fp8_small_batch_vs_fp32_mode_and_full_size():
    """BASELINE config #5's OWN workload: 256 px, gf = df = 96, MX-fp8 3x3 convolutions.
    (1) per-device batch 4 against the product's float32 parity mode on the same batch and parameters (the oracle cannot
    run the 256 px network in test time; the float32 mode is itself held to the oracle at 256 px / dims 16 by
    test_train_step_fp32_256px_small): contrastive losses within 2e-2 of the loss scale, hinge losses within 1e-1.
    (2) the FULL-SIZE step, per-device batch 32 (64 images through D): finite, and two runs from the same state are
    bit-identical (the MX path has no order-dependent accumulation)."""
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd.configs import coco_xmc
    res = {}
    for mode in ("float32", "fp8"):
        cfg = coco_xmc.get_c4_config()
        cfg.pretrained_image_contrastive = False
        cfg.batch_size = 4
        if mode == "float32":
            cfg.dtype, cfg.conv_fp8 = "float32", False
        init = (*syn.init_generator(cfg, seed=42, bias_scale=0.05), *syn.init_discriminator(cfg, seed=43, bias_scale=0.05))
        tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=4).items()}
        assert tb["image"].shape == (8, 256, 256, 3)
        res[mode], fin = _run_steps(cfg, init, [tb])
        assert fin
    m32, m8 = res["float32"][0], res["fp8"][0]
    scale = max(abs(m32[k]) for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"))
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = abs(m8[k] - m32[k]) / scale
        print("C4 256 px b4 fp8 vs float32 mode", k, m8[k], m32[k], r)
        assert np.isfinite(m8[k]) and r < (1e-1 if k in ("d_loss", "g_loss") else 2e-2), (k, m8[k], m32[k])
    cfg = coco_xmc.get_c4_config()
    cfg.pretrained_image_contrastive = False
    init = (*syn.init_generator(cfg, seed=42, bias_scale=0.05), *syn.init_discriminator(cfg, seed=43, bias_scale=0.05))
    tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=cfg.batch_size).items()}
    assert tb["image"].shape == (64, 256, 256, 3)
    a, fin_a = _run_steps(cfg, init, [tb], nsteps=2)
    b, fin_b = _run_steps(cfg, init, [tb], nsteps=2)
    print("C4 full size (256 px, B = 32, fp8): two steps", a)
    assert fin_a and fin_b and all(np.isfinite(v) for m in a for v in m.values())
    assert a == b, "the full-size MX-fp8 step is not bit-reproducible"


def test_fp8_accuracy_is_unbiased_over_seeds_and_steps():
    """Is the MX-fp8 error of the hinge losses a BIAS or a draw?  C1 network, per-device batch 8:
    the fp8 run against the bf16 run of the same product (same initial state, same batches) over 5 data seeds, and over a
    10-step trajectory of one seed.  MEASURED (round 4, MI355X): the contrastive losses and d_loss are unbiased (|mean| <= 1.2e-2
    of the loss scale), but g_loss = -mean(fake logit) + ... is BIASED: +1.4 ... +8.9 % on all five seeds, mean +4.8 % -- the
    fp8 rounding of the generated images' path through D shifts the fake logits one way.  SURVEY 8(d)'s reduced-precision bar
    (2e-2) is therefore MISSED by the fp8 mode on g_loss; the gates below hold d_loss / c_loss_* to it on the mean, g_loss to
    1e-1, every single draw to 1.5e-1, and DESIGN.md section 10 reports the bias (config #5 is not recommended for training).
    Round 5: with the MX scale chosen so that no block maximum saturates (conv_stream_mx8.hip mx_scale_byte) the g_loss bias is
    +1.5 ... +4.1 %, mean +2.8 % (profiles/r05_fp8_bias_after_nonsaturating_scale.txt) -- smaller, still one-sided, still over 2e-2."""
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd.configs import coco_xmc
    keys = ("d_loss", "g_loss", "c_loss_d", "c_loss_g")

    def cfg_of(fp8):
        cfg = coco_xmc.get_c1_config()
        cfg.pretrained_image_contrastive = False
        cfg.batch_size = 8
        cfg.conv_fp8 = fp8
        return cfg
    cfg = cfg_of(False)
    init = (*syn.init_generator(cfg, seed=42, bias_scale=0.05), *syn.init_discriminator(cfg, seed=43, bias_scale=0.05))
    signed = {k: [] for k in keys}
    for seed in range(5):
        tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=8, rank=seed).items()}
        m16, _ = _run_steps(cfg_of(False), init, [tb])
        m8, _ = _run_steps(cfg_of(True), init, [tb])
        scale = max(abs(m16[0][k]) for k in keys)
        for k in keys:
            signed[k].append((m8[0][k] - m16[0][k]) / scale)
    for k in keys:
        mean, worst = float(np.mean(signed[k])), float(np.max(np.abs(signed[k])))
        print(f"fp8 vs bf16 over 5 data seeds, {k}: signed relative differences {np.round(signed[k], 4).tolist()} mean {mean:+.4f} worst {worst:.4f}")
        assert abs(mean) < (1e-1 if k == "g_loss" else 2e-2) and worst < 1.5e-1, (k, signed[k])
    batches = [{k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=8, rank=10 + s).items()} for s in range(10)]
    t16, f16 = _run_steps(cfg_of(False), init, batches, nsteps=10)
    t8, f8 = _run_steps(cfg_of(True), init, batches, nsteps=10)
    assert f16 and f8
    scale = max(abs(t16[0][k]) for k in keys)
    for k in ("d_loss", "g_loss"):
        diffs = [abs(a[k] - b[k]) / scale for a, b in zip(t8, t16)]
        print(f"fp8 vs bf16 over a 10-step trajectory, {k}: |difference| / scale per step {np.round(diffs, 4).tolist()} mean {np.mean(diffs):.4f}")
        assert np.mean(diffs) < 1e-1, (k, diffs)


def test_conv_mx8_with_residual_is_bit_stable_beside_a_weight_gradient_launch():
    """Round 4's "fp8 stream race", at the kernel: the MX-fp8 convolution with a residual (the discriminator's c0 data gradient:
    mask + upsampled residual + device alpha) must give the same bytes alone on the GPU and beside a weight-gradient launch on
    another stream.  Before csrc/common.h made the residual an explicit fma, ~0.01 % of the outputs lost their residual term
    whenever the two shared the CUs (25 launches of 25; tools/mx8_concurrency2.py) -- the bf16 kernel never did."""
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    ops.fp8 = True
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    side = torch.cuda.Stream()
    nx = torch.randn((32, 64, 64, 192), generator=g).to(dt).cuda()
    ndw = torch.zeros((192, 9, 192), device="cuda")
    ndb = torch.zeros((192,), device="cuda")
    n, h, cin, cout = 32, 64, 384, 192
    x = torch.randn((n, h, h, cin), generator=g).to(dt).cuda()
    w = (torch.randn((cout, 9, cin), generator=g) / (9 * cin) ** 0.5).cuda()
    wf, _ = ops.prep_conv_weight(w)
    mask = torch.randn((n, h, h, cout), generator=g).to(dt).cuda()
    res = torch.randn((n, h // 2, h // 2, cout), generator=g).to(dt).cuda()
    res_full = torch.randn((n, h, h, cout), generator=g).to(dt).cuda()
    alpha_dev = torch.full((1,), 0.37, device="cuda")
    for kw in (dict(res=res_full, res_scale=0.25),
               dict(mask=mask, res=res, res_ups=True, res_scale=0.25, alpha_dev=alpha_dev)):
        torch.cuda.synchronize()
        ref = ops.conv(x, wf, None, ks=3, **kw).clone()
        torch.cuda.synchronize()
        for _ in range(10):
            with torch.cuda.stream(side):
                for _ in range(8):
                    ops.conv_wgrad(nx, nx, ndw, ndb, ks=3, x_relu=True, sync=True)
            y = ops.conv(x, wf, None, ks=3, **kw)
            torch.cuda.synchronize()
            assert torch.equal(y, ref), sorted(kw)


@pytest.mark.usefixtures("keep_grads")
def test_fp8_x_cond_fan_in_in_place_keeps_no_stale_packets(monkeypatch):
    """Discriminator.backward_d adds the x_cond branch's gradient IN PLACE on the real half's rows of dx
    (XMC_XC_REAL_HALF=1, the default); a packet twin emitted with the old dx must not survive that, or a packet-reading
    consumer (every 3x3 data gradient when the fp8 mode runs without the bf16 phase kernels: XMC_FP8_PHASE=0) drops the
    real-word-loss gradient of all earlier D layers.  The in-place path against the fresh-tensor path (XMC_XC_REAL_HALF=0):
    D's gradient agrees to fp8 noise in every block in front of the fan-in."""
    from tests.test_gpu_step import _c1_b8_oracle
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.nets import xmc_net
    monkeypatch.setenv("XMC_FP8_PHASE", "0")
    o = _c1_b8_oracle()
    tb = {k: torch.as_tensor(v).cuda() for k, v in o["batch"].items()}
    grads = {}
    for real_half in (False, True):
        monkeypatch.setattr(xmc_net, "_XC_REAL_HALF", real_half)
        cfg = o["cfg"].copy()
        cfg.dtype = "bfloat16"
        cfg.conv_fp8 = True
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        assert not gen(train=True).ops.fp8_phase
        state = train_utils.load_flax_params(state, *o["init"])
        state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
        a = state.d_optimizer.arena
        tree = a.tree(a.grads)
        grads[real_half] = {k: torch.cat([t.detach().double().cpu().reshape(-1) for _, t in syn.tree_leaves(tree[k])])
                            for k in ("DiscOptimizedBlock_0", "DiscBlock_0", "DiscBlock_1")}
        del state, gen, disc
        torch.cuda.empty_cache()
    for k in grads[True]:
        a, b = grads[True][k], grads[False][k]
        r = float((a - b).norm() / b.norm())
        print("x_cond fan-in, fp8 without phase kernels:", k, "in-place vs fresh norm-relative difference", r)
        assert r < 5e-2, (k, r)
