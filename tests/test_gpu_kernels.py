"""Per-kernel parity of the gfx950 kernels (through the C ABI) against float64 torch-CPU
restatements of the same math on the same (dtype-rounded) inputs.

Tolerances: float32 kernels accumulate in fp32 (MFMA 32x32x2 f32 == fmaf chain) -> 2e-4 of
the output scale; bf16 kernels take bf16 inputs, accumulate fp32 and round the output to
bf16 once -> 2^-8 relative to the output scale (1e-2 used).
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DT = [torch.float32, torch.bfloat16]


def _ops(dtype, variant=0):
    from xmcgan_image_generation_amd.ops import HipOps
    return HipOps(dtype=dtype, wgrad_variant=variant, stream_conv=False)


def _tol(dtype):
    return 2e-4 if dtype == torch.float32 else 1.2e-2


def _close(got, ref, dtype, what="", scale=None):
    got = got.detach().double().cpu()
    ref = ref.detach().double().cpu()
    assert got.shape == ref.shape, (what, got.shape, ref.shape)
    s = float(ref.abs().max()) if scale is None else scale
    err = float((got - ref).abs().max())
    assert math.isfinite(err) and err <= _tol(dtype) * max(s, 1e-6), (what, err, s)


def _rnd(shape, dtype, gen, scale=1.0):
    """random tensor, rounded to dtype; returns (device tensor, float64 cpu copy)"""
    x = (torch.randn(shape, generator=gen, dtype=torch.float32) * scale).to(dtype)
    return x.cuda(), x.double()


def _ref_conv(x, w_master, bias, ks, ups=False, relu_in=False):
    """x NHWC f64, w_master (cout, taps, cin) f64 -> NHWC f64"""
    cout, taps, cin = w_master.shape
    if relu_in:
        x = torch.relu(x)
    if ups:
        x = x.repeat_interleave(2, 1).repeat_interleave(2, 2)
    w = w_master.reshape(cout, ks, ks, cin).permute(0, 3, 1, 2)
    y = F.conv2d(x.permute(0, 3, 1, 2), w, bias, padding=ks // 2)
    return y.permute(0, 2, 3, 1)


def test_probe_layouts():
    out = _ops(torch.float32).probe_layouts().cpu().numpy()
    drow = out[:1024].reshape(64, 16)
    dcol = out[1024:2048].reshape(64, 16)
    tr = out[2048:].reshape(64, 4)
    lane = np.arange(64)[:, None]
    reg = np.arange(16)[None, :]
    exp_row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5) + 1
    exp_col = (lane & 31) + 1 + 0 * reg
    assert np.array_equal(drow, exp_row), "MFMA 32x32 C/D row map differs from the kernels' assumption"
    assert np.array_equal(dcol, exp_col), "MFMA 32x32 C/D col map differs from the kernels' assumption"
    e = np.arange(4)[None, :]
    exp_tr = ((lane >> 5) * 8 + e) * 256 + (lane & 31)
    print("tr16 probe lane0..3:", tr[:4].tolist(), "lane16:", tr[16].tolist(), "lane32:", tr[32].tolist())
    assert np.array_equal(tr, exp_tr), "ds_read_b64_tr_b16 lane map differs from the wgrad kernel's assumption"


CONV_CASES = [
    # n, h, cin, cout, ks, ups, relu_in, extras
    (2, 8, 16, 32, 3, False, False, {}),
    (2, 8, 32, 24, 3, True, True, {}),
    (3, 16, 48, 136, 3, False, True, dict(bias=True, alpha=0.25)),
    (2, 16, 64, 40, 1, False, False, dict(bias=True)),
    (2, 4, 40, 16, 1, True, False, dict(bias=True)),
    (2, 32, 3, 16, 3, False, False, dict(bias=True)),             # RGB input: scalar gather
    (2, 16, 24, 3, 3, False, True, dict(bias=True)),              # RGB output: scalar store
    (1, 8, 16, 16, 3, False, False, dict(mask=True, res=True, res_scale=0.5)),
    (2, 8, 16, 8, 3, False, False, dict(mask=True, res=True, res_ups=True, res_scale=0.25, bias=True)),
    (2, 4, 264, 200, 3, False, False, dict(bias=True)),           # multi n-tile, K > 1 chunk
    (1, 8, 16, 16, 3, False, False, dict(out_f32=True, bias=True)),
    # shapes eligible for the LDS-staged im2col (patch) kernel in bf16: every tile geometry
    (1, 128, 32, 96, 3, False, False, dict(bias=True)),           # Wo = 128: one row segment per tile
    (1, 256, 32, 32, 3, False, True, dict()),                     # Wo = 256: x0 = 0 / 128
    (1, 64, 64, 32, 3, True, False, dict(bias=True)),             # fused upsample to 128
    (2, 64, 32, 40, 3, False, False, dict()),                     # 2 rows per tile
    (4, 32, 64, 96, 3, False, True, dict(bias=True, alpha=0.25)),
    (8, 16, 32, 128, 3, False, False, dict(mask=True, res=True, res_scale=0.5)),
    (8, 8, 96, 136, 3, False, False, dict(bias=True)),            # 2 images per tile, 2 n-tiles
    (16, 4, 64, 64, 3, False, False, dict(bias=True)),            # 8 images per tile
    (8, 4, 32, 32, 3, True, True, dict(res=True, res_ups=True, res_scale=0.25, mask=True)),
    (4, 128, 32, 192, 3, False, True, dict(bias=True, mask=True)),     # 256-pixel tiles x 96-wide cout tiles (6 waves)
    (8, 64, 32, 128, 3, False, False, dict(bias=True, res=True)),       # 256-pixel tiles x 128-wide (8 waves)
    (4, 64, 64, 96, 1, True, False, dict(bias=True)),                   # 1x1, upsample, 96-wide, 256-pixel tiles
    (2, 16, 1024, 96, 1, False, False, dict(bias=True, out_f32=True)),
    (8, 4, 64, 32, 1, True, False, dict(bias=True)),
    # the frozen ResNet-50's epilogue options: ReLU on the output, mask applied after the residual, zeroed canvas margin
    (3, 16, 64, 64, 3, False, True, dict(bias=True, res=True, relu_out=True, valid=14)),
    (3, 16, 64, 256, 1, False, True, dict(bias=True, res=True, relu_out=True, valid=14)),
    (3, 32, 128, 64, 1, False, False, dict(mask=True, res=True, mask_after_res=True)),
    (2, 8, 40, 24, 3, False, False, dict(mask=True, res=True, mask_after_res=True, relu_out=True, valid=7, bias=True)),
    (3, 64, 32, 32, 3, False, False, dict(mask=True, valid=56)),
]


STREAM_CASES = [c for c in CONV_CASES if c[4] == 3 and c[2] % 32 == 0] + [
    (3, 64, 32, 72, 3, False, False, dict(bias=True)),            # partial last image group / cout block
    (5, 8, 64, 3, 3, False, True, dict(bias=True)),               # N not a multiple of the images per tile, cout 3
    (1, 256, 32, 160, 3, False, False, dict(mask=True)),
    # few tiles x many chunks: split-K through the workspace + finishing kernel
    (8, 4, 256, 64, 3, False, False, dict(bias=True)),
    (16, 4, 512, 160, 3, False, True, dict(bias=True, mask=True, res=True, res_scale=0.5, alpha=0.25)),
    (4, 4, 384, 96, 3, True, False, dict(res=True, res_ups=True, res_scale=0.25, bias=True)),
    (6, 8, 320, 40, 3, False, False, dict(out_f32=True, bias=True)),
    (16, 8, 512, 512, 3, False, True, dict(bias=True, res=True, relu_out=True, valid=7)),             # split-K + ResNet epilogue
    (16, 8, 512, 512, 3, False, False, dict(mask=True, res=True, mask_after_res=True, valid=7)),
    # round 4: 128-pixel x 96-cout tiles (three workgroups per CU) for the many-tile 96 / 192-cout launches (>= 98,304 pixels)
    (7, 128, 96, 96, 3, False, True, dict(bias=True, mask=True, res=True, res_ups=True, res_scale=0.25)),
    (26, 64, 192, 96, 3, False, False, dict(bias=True, res=True, res_scale=0.5, alpha=0.25)),
    (25, 64, 96, 192, 3, False, True, dict(bias=True, relu_out=True)),
    (3, 256, 32, 96, 3, False, False, dict(mask=True)),
]


@pytest.mark.parametrize("case", STREAM_CASES)
def test_conv_stream_packed(case):
    """weight-streaming kernel (fragment-packed weights) against the same float64 reference"""
    _run_conv_case(torch.bfloat16, case, packed=True)


PW_CASES = [
    # pointwise kernel (conv_stream.hip: conv_pw_kernel) on packed 1x1 weights: n, h, cin, cout, ks, ups, relu_in, extras
    (4, 16, 64, 128, 1, False, False, dict(bias=True)),                       # KC = 64, one tile row
    (3, 16, 96, 40, 1, False, True, dict(bias=True)),                         # KC = 32 (96 = 3 x 32), ragged cout, partial last tile
    (5, 8, 32, 3, 1, False, False, dict(bias=True)),                          # one chunk, cout 3, M = 320
    (4, 64, 64, 96, 1, True, False, dict(bias=True)),                         # fused upsample
    (2, 16, 1024, 96, 1, False, False, dict(bias=True, out_f32=True)),        # the local-cBN projection's shape class
    (8, 16, 192, 256, 1, False, False, dict(mask=True, res=True, res_scale=0.5, alpha=0.25)),
    (8, 8, 128, 64, 1, True, True, dict(res=True, res_ups=True, res_scale=0.25, mask=True)),
    (16, 8, 2048, 512, 1, False, False, dict(bias=True)),                     # few tiles x 32 chunks: split-K
    (16, 8, 1024, 160, 1, False, True, dict(bias=True, mask=True, res=True, mask_after_res=True, valid=7)),   # split-K + epilogue flags
    (7, 16, 256, 1024, 1, False, True, dict(bias=True, res=True, relu_out=True, valid=14)),
    (3, 64, 64, 256, 1, False, False, dict(bias=True, res=True, relu_out=True, valid=56)),
    (2, 32, 160, 64, 1, False, False, dict(bias=True, valid=28)),             # 160 = 5 x 32 (the stem's im2col width)
]


@pytest.mark.parametrize("case", PW_CASES)
def test_conv_pointwise_packed(case):
    _run_conv_case(torch.bfloat16, case, packed=True)


@pytest.mark.parametrize("case", PW_CASES[:8])
def test_conv_pointwise_dgrad_layout(case):
    """the same kernel on the dgrad-layout weights (cin <-> cout swapped) against autograd"""
    n, h, cin, cout, ks, ups, relu_in, ex = case
    if ups:
        pytest.skip("dgrad of an upsampling layer goes through the pooling adjoint")
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(5)
    if cout % 32:
        pytest.skip("dgrad reduces over cout: packed only for multiples of 32")
    w32 = torch.randn((cout, 1, cin), generator=g) / math.sqrt(cin)
    wf, wd = ops.prep_conv_weight(w32.cuda())
    dy, dyr = _rnd((n, h, h, cout), torch.bfloat16, g)
    dx = ops.conv(dy, wd, None, ks=1)
    xr = torch.zeros((n, h, h, cin), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(xr, w32.bfloat16().double(), None, 1)
    (ref,) = torch.autograd.grad(y, xr, dyr)
    _close(dx, ref, torch.bfloat16, f"pw dgrad {case}")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fwd(dtype, case):
    _run_conv_case(dtype, case, packed=False)


def _run_conv_case(dtype, case, packed):
    n, h, cin, cout, ks, ups, relu_in, ex = case
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(hash(case[:7]) % 1000)
    x, xr = _rnd((n, h, h, cin), dtype, g)
    w32 = torch.randn((cout, ks * ks, cin), generator=g) / math.sqrt(ks * ks * cin)
    wf, wd = ops.prep_conv_weight(w32.cuda())
    wr = wf.double().cpu()
    ho = 2 * h if ups else h
    bias = torch.randn(cout, generator=g) if ex.get("bias") else None
    mask, maskr = _rnd((n, ho, ho, cout), dtype, g) if ex.get("mask") else (None, None)
    res, resr = (None, None)
    if ex.get("res"):
        rs = (n, ho // 2, ho // 2, cout) if ex.get("res_ups") else (n, ho, ho, cout)
        res, resr = _rnd(rs, dtype, g)
    alpha, res_scale = ex.get("alpha", 1.0), ex.get("res_scale", 1.0)
    if packed:
        wf = ops.pack_conv_weight(wf)
    y = ops.conv(x, wf, bias.cuda() if bias is not None else None, ks=ks, ups=ups, relu_in=relu_in, mask=mask,
                 res=res, res_ups=ex.get("res_ups", False), res_scale=res_scale, alpha=alpha,
                 out_f32=ex.get("out_f32", False), relu_out=ex.get("relu_out", False),
                 mask_after_res=ex.get("mask_after_res", False), valid=ex.get("valid", 0))
    ref = alpha * _ref_conv(xr, wr, None, ks, ups, relu_in)
    if bias is not None:
        ref = ref + bias.double()
    if maskr is not None and not ex.get("mask_after_res"):
        ref = torch.where(maskr > 0, ref, torch.zeros_like(ref))
    if resr is not None:
        rr = resr.repeat_interleave(2, 1).repeat_interleave(2, 2) if ex.get("res_ups") else resr
        ref = ref + res_scale * rr
    if maskr is not None and ex.get("mask_after_res"):
        ref = torch.where(maskr > 0, ref, torch.zeros_like(ref))
    if ex.get("relu_out"):
        ref = torch.relu(ref)
    if ex.get("valid"):
        ref[:, ex["valid"]:] = 0
        ref[:, :, ex["valid"]:] = 0
    assert y.dtype == (torch.float32 if ex.get("out_f32") or dtype == torch.float32 else torch.bfloat16)
    _close(y, ref, dtype, f"conv {case}")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [(2, 8, 16, 32, 3), (2, 16, 40, 24, 3), (2, 8, 24, 48, 1), (2, 16, 3, 16, 3)])
def test_conv_dgrad(dtype, case):
    """dgrad = the forward kernel on the dgrad-layout weights (flipped taps, swapped channels)."""
    n, h, cin, cout, ks = case
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(7)
    w32 = torch.randn((cout, ks * ks, cin), generator=g) / math.sqrt(ks * ks * cin)
    wf, wd = ops.prep_conv_weight(w32.cuda())
    dy, dyr = _rnd((n, h, h, cout), dtype, g)
    dx = ops.conv(dy, wd, None, ks=ks)
    xr = torch.zeros((n, h, h, cin), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(xr, wf.double().cpu(), None, ks)
    (ref,) = torch.autograd.grad(y, xr, dyr)
    _close(dx, ref, dtype, f"dgrad {case}")


WG_CASES = [
    # n, h(x), cin, cout, ks, x_ups, x_relu, dy_ups, alpha
    (2, 8, 16, 32, 3, False, False, False, 1.0),
    (2, 8, 32, 24, 3, True, True, False, 1.0),
    (3, 16, 48, 136, 3, False, True, True, 0.25),
    (2, 16, 64, 40, 1, False, False, False, 1.0),
    (2, 32, 3, 16, 3, False, False, False, 1.0),
    (2, 16, 24, 3, 3, False, True, False, 1.0),
    (2, 4, 264, 200, 3, False, False, False, 1.0),
    (4, 32, 16, 16, 3, False, False, False, 1.0),     # several split-K blocks
    # eligible for the patch wgrad kernel (bf16, variant 1): every tile geometry
    (1, 128, 32, 96, 3, False, False, False, 1.0),    # 64-pixel row segments, x0 = 0 / 64
    (1, 64, 64, 40, 3, True, False, False, 1.0),      # fused upsample of x
    (2, 64, 32, 32, 3, False, True, True, 0.25),      # dy at half resolution
    (4, 32, 64, 96, 3, False, True, False, 1.0),      # 2 rows per tile
    (8, 16, 32, 136, 3, False, False, True, 0.25),    # 4 rows per tile, dy_ups, 2 cout tiles
    (8, 8, 96, 64, 3, False, False, False, 1.0),      # one image per tile
    (16, 4, 64, 64, 3, False, False, False, 1.0),     # 4 images per tile
    (8, 4, 32, 32, 3, True, True, False, 1.0),
    (2, 16, 1024, 96, 1, False, False, False, 1.0),
    (8, 8, 64, 32, 1, False, False, False, 1.0),
    # round 5: 96-cout workgroup tiles of the LDS-DMA kernel (16-pixel-wide tiles, Cout = 96 / 192 / 288)
    (2, 32, 96, 192, 3, False, True, False, 1.0),     # two 96-cout tiles x three cin slabs, ReLU on x
    (1, 64, 64, 288, 3, False, False, False, 0.5),    # three tiles
    (4, 16, 32, 96, 3, False, False, False, 1.0),     # 4 rows per tile
]


@pytest.mark.parametrize("dtype,variant", [(torch.float32, 0), (torch.bfloat16, 0), (torch.bfloat16, 1)])
@pytest.mark.parametrize("case", WG_CASES)
def test_conv_wgrad(dtype, variant, case):
    n, h, cin, cout, ks, x_ups, x_relu, dy_ups, alpha = case
    ops = _ops(dtype, variant)
    g = torch.Generator().manual_seed(11)
    x, xr = _rnd((n, h, h, cin), dtype, g)
    ho = 2 * h if x_ups else h
    hd = ho // 2 if dy_ups else ho
    dy, dyr = _rnd((n, hd, hd, cout), dtype, g)
    dw = torch.zeros((cout, ks * ks, cin), device="cuda")
    db = torch.zeros((cout,), device="cuda")
    ops.conv_wgrad(x, dy, dw, db, ks=ks, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha)
    ops.conv_wgrad(x, dy, dw, db, ks=ks, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha)   # accumulates
    wr = torch.zeros((cout, ks * ks, cin), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(xr, wr, None, ks, x_ups, x_relu)
    cot = dyr.repeat_interleave(2, 1).repeat_interleave(2, 2) if dy_ups else dyr
    (ref,) = torch.autograd.grad(y, wr, cot)
    tol_dt = torch.float32 if dtype == torch.float32 else torch.bfloat16
    _close(dw, 2 * alpha * ref, torch.float32 if dtype == torch.float32 else tol_dt, f"wgrad {case} v{variant}",
           scale=float((2 * alpha * ref).abs().max()))
    bref = 2 * alpha * cot.sum((0, 1, 2))
    _close(db, bref, torch.float32, f"bias grad {case} v{variant}", scale=float(bref.abs().max()) * 4)


@pytest.mark.parametrize("case", [(2, 64, 96, 96, False), (2, 64, 96, 192, True), (4, 32, 192, 192, False), (1, 128, 32, 288, False)])
def test_conv_wgrad_96_cout_tiles_equal_128_cout_tiles_and_first_write(case):
    """conv_wgrad_dma_kernel<3, 2, 18, 1, C96>: the 96-cout wave mapping (7 / 7 / 7 / 6 products per k-step) against the 128-cout
    one (variant bit 11) on the same operands -- every (cout, tap, cin) entry sums the same pixel tiles in the same order and
    the split count is the same, so the results are BIT-identical; and XMC_WGRAD_OVERWRITE into a poisoned buffer equals the
    accumulation into zeros (first-write gradients, round 5)"""
    n, h, cin, cout, x_relu = case
    ops = _ops(torch.bfloat16, 1)
    g = torch.Generator().manual_seed(5)
    x, _ = _rnd((n, h, h, cin), torch.bfloat16, g)
    dy, _ = _rnd((n, h, h, cout), torch.bfloat16, g)
    outs = []
    for off in (0, 0x800):
        ops.wgrad_variant = 1 | off
        dw = torch.zeros((cout, 9, cin), device="cuda")
        db = torch.zeros((cout,), device="cuda")
        ops.conv_wgrad(x, dy, dw, db, ks=3, x_relu=x_relu, alpha=0.5, sync=True)
        outs.append((dw, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert float(outs[0][0].abs().max()) > 0
    ops.wgrad_variant = 1
    dw = torch.full((cout, 9, cin), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    ops.conv_wgrad(x, dy, dw, db, ks=3, x_relu=x_relu, alpha=0.5, sync=True, overwrite=True)
    assert torch.equal(dw, outs[0][0]) and torch.equal(db, outs[0][1])


@pytest.mark.parametrize("case", [("ups", 2, 16, 32, 96), ("pool", 2, 16, 64, 64), ("pool", 8, 4, 64, 64), ("1x1", 2, 16, 1024, 96),
                                  ("generic", 2, 8, 16, 24)])
def test_conv_wgrad_first_write_every_kernel_path(case):
    """XMC_WGRAD_OVERWRITE on every weight-gradient kernel path (phase-decomposed single / several splits, pointwise, the generic
    kernel's clear-then-accumulate fallback): a poisoned dw / db buffer ends up equal to the accumulation into zeros"""
    kind, n, h, cin, cout = case
    ops = _ops(torch.bfloat16, 1)
    g = torch.Generator().manual_seed(9)
    kw = dict(ks=3)
    if kind == "ups":
        x, _ = _rnd((n, h, h, cin), torch.bfloat16, g)
        dy, _ = _rnd((n, 2 * h, 2 * h, cout), torch.bfloat16, g)
        kw.update(x_ups=True)
    elif kind == "pool":
        x, _ = _rnd((n, 2 * h, 2 * h, cin), torch.bfloat16, g)
        dy, _ = _rnd((n, h, h, cout), torch.bfloat16, g)
        kw.update(dy_ups=True, alpha=0.25)
    else:
        x, _ = _rnd((n, h, h, cin), torch.bfloat16, g)
        dy, _ = _rnd((n, h, h, cout), torch.bfloat16, g)
        kw.update(ks=1 if kind == "1x1" else 3)
    taps = kw["ks"] ** 2
    dw0 = torch.zeros((cout, taps, cin), device="cuda")
    db0 = torch.zeros((cout,), device="cuda")
    ops.conv_wgrad(x, dy, dw0, db0, sync=True, **kw)
    dw = torch.full((cout, taps, cin), float("nan"), device="cuda")
    db = torch.full((cout,), float("nan"), device="cuda")
    ops.conv_wgrad(x, dy, dw, db, sync=True, overwrite=True, **kw)
    assert torch.equal(dw, dw0) and torch.equal(db, db0), kind


WGP_CASES = [
    # form, n, V side, cin, cout, x_relu
    ("ups", 4, 4, 64, 64, False), ("ups", 2, 8, 96, 64, False), ("ups", 2, 16, 32, 96, False), ("ups", 1, 64, 32, 64, False),
    ("ups", 3, 32, 64, 160, False),
    ("pool", 4, 4, 64, 64, True), ("pool", 2, 8, 96, 64, True), ("pool", 2, 16, 32, 96, False), ("pool", 1, 64, 32, 64, True),
    ("pool", 3, 32, 64, 160, True),
]


@pytest.mark.parametrize("case", WGP_CASES)
def test_conv_wgrad_phase(case):
    """Weight / bias gradient next to a 2x resampling as 16 (phase, tap) products per low-resolution pixel
    (conv_wgrad_phase.hip) against the float64 3x3 formulation and against the 3x3 kernel."""
    form, n, v, cin, cout, x_relu = case
    dtype = torch.bfloat16
    ops = _ops(dtype, 1)
    assert ops.deterministic and ops.phase_conv
    g = torch.Generator().manual_seed(59)
    x_ups, dy_ups, alpha = (True, False, 1.0) if form == "ups" else (False, True, 0.25)
    hx = v if form == "ups" else 2 * v
    hd = 2 * v if form == "ups" else v
    x, xr = _rnd((n, hx, hx, cin), dtype, g)
    dy, dyr = _rnd((n, hd, hd, cout), dtype, g)
    outs = []
    for phase in (True, False):
        ops.phase_conv = phase
        dw = torch.zeros((cout, 9, cin), device="cuda")
        db = torch.zeros((cout,), device="cuda")
        ops.conv_wgrad(x, dy, dw, db, ks=3, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha, sync=True)
        ops.conv_wgrad(x, dy, dw, db, ks=3, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha, sync=True)   # accumulates
        outs.append((dw, db))
    wr = torch.zeros((cout, 9, cin), dtype=torch.float64, requires_grad=True)
    y = _ref_conv(xr, wr, None, 3, x_ups, x_relu)
    cot = dyr.repeat_interleave(2, 1).repeat_interleave(2, 2) if dy_ups else dyr
    (ref,) = torch.autograd.grad(y, wr, cot)
    bref = 2 * alpha * cot.sum((0, 1, 2))
    for (dw, db), what in zip(outs, ("phase", "3x3")):
        _close(dw, 2 * alpha * ref, dtype, f"wgrad {what} {case}", scale=float((2 * alpha * ref).abs().max()))
        _close(db, bref, torch.float32, f"bias grad {what} {case}", scale=float(bref.abs().max()) * 4)
    # float32 accumulation of the same bf16 products in a different order: the two kernels agree far below the bf16 tolerance
    _close(outs[0][0], outs[1][0].double(), torch.float32, f"phase vs 3x3 {case}", scale=10 * float(ref.abs().max()))
    # bit-reproducible
    dw2 = torch.zeros((cout, 9, cin), device="cuda")
    ops.phase_conv = True
    ops.conv_wgrad(x, dy, dw2, None, ks=3, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha, sync=True)
    ops.conv_wgrad(x, dy, dw2, None, ks=3, x_ups=x_ups, x_relu=x_relu, dy_ups=dy_ups, alpha=alpha, sync=True)
    assert torch.equal(dw2, outs[0][0])


def test_prep_conv_weight_layouts():
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(3)
    w = torch.randn((40, 9, 24), generator=g)
    inv = torch.tensor([0.5])
    wf, wd = ops.prep_conv_weight(w.cuda(), inv.cuda())
    assert torch.equal(wf.cpu(), w * 0.5)
    exp = (w * 0.5).flip(1).permute(2, 1, 0).contiguous()
    assert torch.equal(wd.cpu(), exp)


@pytest.mark.parametrize("shape", [(70, 50, 33, False, False), (130, 140, 64, False, True), (17, 256, 768, True, False),
                                   (256, 17, 100, True, True), (300, 5, 1, False, False),
                                   # split-K shapes whose k-tile count does not divide into the picked split count: no split may
                                   # be empty (an empty split left uninitialised workspace in the sum; round 3)
                                   (56, 56, 1000, False, True), (56, 56, 1008, False, True), (60, 64, 1336, False, False),
                                   (56, 56, 768, False, True), (8, 8, 300, False, True)])
def test_gemm(shape):
    m, n, k, ta, tb = shape
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(5)
    a = torch.randn((k, m) if ta else (m, k), generator=g)
    b = torch.randn((n, k) if tb else (k, n), generator=g)
    c0 = torch.randn((m, n), generator=g)
    out = c0.clone().cuda()
    dev = torch.tensor([0.5]).cuda()
    ops.gemm(a.cuda(), b.cuda(), ta=ta, tb=tb, alpha=2.0, alpha_dev=dev, beta=1.0, out=out)
    ref = (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double()) + c0.double()
    _close(out, ref, torch.float32, f"gemm {shape}")


def test_gemm_batched_strided():
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(6)
    a = torch.randn((3, 40, 96), generator=g).cuda()
    big = torch.randn((3, 96, 80), generator=g).cuda()
    b = big[:, :, 8:72]                       # strided view, no copy
    out = ops.gemm(a, b)
    _close(out, a.double().cpu() @ b.double().cpu(), torch.float32, "bgemm")
    out2 = ops.gemm(a, a, tb=True)
    _close(out2, a.double().cpu() @ a.double().cpu().transpose(1, 2), torch.float32, "bgemm nt")


@pytest.mark.parametrize("dtype", DT)
def test_reduce_mid_and_bcast(dtype):
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(8)
    for (a, r, c) in [(1, 1000, 3), (4, 16, 96), (2, 300, 24), (1, 5000, 1536), (1, 56, 24576), (2, 9, 4100)]:
        x, xr = _rnd((a, r, c), dtype, g)
        y = ops.reduce_mid(x, relu=True, scale=0.5)
        _close(y, 0.5 * torch.relu(xr).sum(1), torch.float32, f"reduce {a,r,c}")
        y2 = ops.reduce_mid(x, out=y, accumulate=True)
        _close(y2, 0.5 * torch.relu(xr).sum(1) + xr.sum(1), torch.float32, "reduce acc")
    x, xr = _rnd((4, 16, 96), dtype, g)
    dp = torch.randn((4, 96), generator=g)
    dx = ops.bcast_relu_bwd(dp.cuda(), x)
    ref = torch.where(xr > 0, dp.double()[:, None, :].expand(-1, 16, -1), torch.zeros(()).double())
    _close(dx, ref, dtype, "bcast_relu_bwd")


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("geo", [(3, 8, 16, 1), (2, 16, 24, 4), (2, 16, 40, 16), (4, 4, 1536, 1), (2, 32, 64, 4)])   # first / last: 8 x 8 cells, second / fourth: 4 x 4 -> the row-per-thread kernels
def test_cbn(dtype, geo):
    n, h, c, hc = geo
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(9)
    x, xr = _rnd((n, h, h, c), dtype, g, 2.0)
    gamma = torch.randn((n, hc, hc, c), generator=g) * 0.3
    beta = torch.randn((n, hc, hc, c), generator=g) * 0.3
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
    sums = ops.bn_stats(x)
    mean, rstd = ops.bn_finalize(sums, n * h * h, rm, rv, True)
    xr = xr.requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    m_ref = xr.mean((0, 1, 2))
    v_ref = (xr * xr).mean((0, 1, 2)) - m_ref ** 2
    _close(mean, m_ref, torch.float32, "bn mean", scale=float(xr.abs().max()))
    _close(rstd, torch.rsqrt(v_ref + 1e-5), torch.float32, "bn rstd")
    _close(rm, 0.1 * m_ref, torch.float32, "running mean", scale=1.0)
    _close(rv, 0.9 + 0.1 * v_ref, torch.float32, "running var", scale=4.0)
    # the atomic-free two-stage path the step uses: same numbers, and bit-identical from run to run
    rm2, rv2 = torch.zeros(c).cuda(), torch.ones(c).cuda()
    mean2, rstd2 = ops.bn_batch_stats(x, rm2, rv2, True)
    _close(mean2, m_ref, torch.float32, "bn mean (two-stage)", scale=float(xr.abs().max()))
    _close(rstd2, torch.rsqrt(v_ref + 1e-5), torch.float32, "bn rstd (two-stage)")
    _close(rv2, 0.9 + 0.1 * v_ref, torch.float32, "running var (two-stage)", scale=4.0)
    mean3, rstd3 = ops.bn_batch_stats(x, torch.zeros(c).cuda(), torch.ones(c).cuda(), True)
    assert torch.equal(mean2, mean3) and torch.equal(rstd2, rstd3)
    f = h // hc
    up = lambda t: t.repeat_interleave(f, 1).repeat_interleave(f, 2)
    y_ref = torch.relu((xr - m_ref) * torch.rsqrt(v_ref + 1e-5) * (up(gr) + 1) + up(br))
    gb = torch.cat([gamma, beta], dim=-1).contiguous().cuda()         # fused (cells, 2C) gamma|beta layout
    y = ops.cbn_act_fwd(x, mean, rstd, gb, hc)
    _close(y, y_ref, dtype, "cbn fwd")
    dy, dyr = _rnd((n, h, h, c), dtype, g)
    dx, dgb = ops.cbn_act_bwd(dy, x, mean, rstd, gb, hc)
    dg, db = dgb[..., :c], dgb[..., c:]
    rx, rg, rb = torch.autograd.grad(y_ref, (xr, gr, br), dyr)
    _close(dx, rx, dtype, "cbn dx")
    _close(dg, rg, torch.float32, "cbn dgamma", scale=float(rg.abs().max()) * (1 if dtype == torch.float32 else 40))
    _close(db, rb, torch.float32, "cbn dbeta", scale=float(rb.abs().max()) * (1 if dtype == torch.float32 else 40))


@pytest.mark.parametrize("geo", [(3, 8, 16, 1), (2, 16, 24, 4), (2, 16, 40, 16), (2, 16, 768, 16), (2, 32, 64, 4), (2, 64, 96, 16)])
def test_cbn_bf16_gamma_beta(geo):
    """gamma / beta maps (and their gradients) in bf16 -- the LocalConditionalBatchNorm dtype flow of the reference's bf16 mode
    (layers.py:261-273), ops.gb_bf16 -- against the float32-map kernels fed with the SAME bf16-rounded values: forward bit-equal
    (the arithmetic after the conversion is the same), dgamma / dbeta equal to the float32 kernels' outputs rounded to bf16, dx
    within the effect of that rounding on the two per-channel sums; every cell geometry (pixel-per-thread, run, row, split)."""
    n, h, c, hc = geo
    dtype = torch.bfloat16
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(19)
    x, _ = _rnd((n, h, h, c), dtype, g, 2.0)
    wide = 2 * c + 64                                                          # a column slice of a wider projection output
    full = (torch.randn((n * hc * hc, wide), generator=g) * 0.3).to(dtype).cuda()
    gb16 = full[:, 64:64 + 2 * c]
    gb32 = gb16.float().contiguous()
    rm, rv = torch.zeros(c).cuda(), torch.ones(c).cuda()
    mean, rstd = ops.bn_batch_stats(x, rm, rv, True)
    y16 = ops.cbn_act_fwd(x, mean, rstd, gb16, hc)
    y32 = ops.cbn_act_fwd(x, mean, rstd, gb32, hc)
    assert torch.equal(y16, y32)
    dy, _ = _rnd((n, h, h, c), dtype, g)
    dfull = torch.zeros_like(full)
    dx16, dgb16 = ops.cbn_act_bwd(dy, x, mean, rstd, gb16, hc, dgb_out=dfull[:, 64:64 + 2 * c])
    dx32, dgb32 = ops.cbn_act_bwd(dy, x, mean, rstd, gb32, hc)
    assert dgb16.dtype == torch.bfloat16 and torch.equal(dgb16, dgb32.to(dtype))
    assert not dfull[:, :64].any() and not dfull[:, 64 + 2 * c:].any()        # nothing written outside the slice
    err = float((dx16.float() - dx32.float()).abs().max()) / float(dx32.float().abs().max())
    assert err < 2e-2, err


@pytest.mark.parametrize("dtype", DT)
def test_pointwise(dtype):
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(10)
    for c in (3, 16, 96):
        x, xr = _rnd((2, 8, 8, c), dtype, g)
        r, rr = _rnd((2, 4, 4, c), dtype, g)
        y = ops.pool2(x, 0.25, res=r)
        ref = F.avg_pool2d(xr.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1) + rr
        _close(y, ref, dtype, "pool2")
        # the same pass also emitting relu(x) at full resolution (xmc_pool2_relu): same pooled result bit for bit, and the
        # ReLU-ed copy exactly max(x, 0) in the storage dtype
        y2, xrl = ops.pool2(x, 0.25, res=r, relu_copy=True)
        assert torch.equal(y2, y) and torch.equal(xrl, torch.relu(x))
    x, xr = _rnd((1000,), dtype, g)
    y = ops.tanh_out_fwd(x)
    _close(y, (torch.tanh(xr) + 1) / 2, dtype, "tanh fwd")
    dy, dyr = _rnd((1000,), dtype, g)
    yr = y.double().cpu()
    _close(ops.tanh_out_bwd(dy, y), dyr * 0.5 * (1 - (2 * yr - 1) ** 2), dtype, "tanh bwd")
    _close(ops.add(x, dy), xr + dyr, dtype, "add")
    z = ops.cast(x, torch.float32)
    assert torch.equal(z.cpu(), x.float().cpu())
    assert torch.equal(ops.cast(z, torch.bfloat16).cpu(), z.cpu().to(torch.bfloat16))


@pytest.mark.parametrize("dtype", DT)
def test_attention_for_g(dtype):
    from oracle import torch_ref as R
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(12)
    b, r, t, e = 3, 64, 17, 768
    region, rr = _rnd((b, r, e), dtype, g)
    words = torch.randn((b, t, e), generator=g)
    max_len = torch.tensor([[4.0], [17.0], [9.0]])
    wn, _ = ops.l2norm_fwd(words.reshape(b * t, e).cuda())
    _close(wn, R.l2n(words.double()).reshape(b * t, e), torch.float32, "l2norm fwd")
    ctx, attn, rinv = ops.attn_g_fwd(region, wn.view(b, t, e), max_len.cuda().view(-1), 15.0)
    rr = rr.requires_grad_(True)
    mask = (torch.arange(t, dtype=torch.float64)[None, :] >= max_len.double()).double()[:, None, :].expand(-1, r, -1)
    ctx_ref, attn_ref = R.attention_for_g(rr, words.double(), 15.0, mask)
    _close(attn, attn_ref, torch.float32, "attn probs", scale=1.0)
    _close(ctx, ctx_ref, dtype, "attn ctx")
    assert torch.equal(attn.argmax(-1).cpu(), attn_ref.argmax(-1)), "attention indices must be bit-exact"
    dctx, dcr = _rnd((b, r, e), dtype, g)
    dreg = ops.attn_g_bwd(dctx, region, wn.view(b, t, e), attn, rinv, 15.0)
    (ref,) = torch.autograd.grad(ctx_ref, rr, dcr)
    _close(dreg, ref, dtype, "attn bwd")


# (56, 256) / (32, 1024): attention_for_g as the benchmarked C1 step and the 256 px C3 step run it (per-GPU batches of
# BASELINE configs #2 / #4; the conditioning map is 16 x 16 at 128 px, 32 x 32 at 256 px)
@pytest.mark.parametrize("case", [(3, 256), (2, 128), (5, 1024), (56, 256), (32, 1024)])
def test_attention_for_g_on_mfma(case):
    """attention_for_g on the matrix cores (attn_mfma.hip, the bf16 mode's kernel) against the oracle: forward probabilities,
    context and the data gradient.  The kernel multiplies bf16 words (one more rounding than the VALU kernel, which reads the
    float32 normalised words): compared (a) tightly against the float64 oracle on the SAME bf16-rounded words, (b) against the
    oracle on the exact words within the reduced-precision bar; and against the VALU kernel on the same inputs."""
    from oracle import torch_ref as R
    b, r = case
    t, e = 17, 768
    dtype = torch.bfloat16
    ops = _ops(dtype)
    assert ops._attn_mfma(torch.empty((1,), dtype=dtype), b, r, t, e)
    g = torch.Generator().manual_seed(112 + r)
    region, rr = _rnd((b, r, e), dtype, g)
    words = torch.randn((b, t, e), generator=g)
    max_len = torch.tensor([[4.0], [17.0], [9.0], [1.0], [12.0]])[:b]
    if b > 5:                                        # full-size batches: U{1..17}, the five hand-picked lengths in front
        max_len = torch.cat([max_len, torch.randint(1, 18, (b - 5, 1), generator=g).float()])
    wn, _ = ops.l2norm_fwd(words.reshape(b * t, e).cuda())
    ctx, attn, rinv = ops.attn_g_fwd(region, wn.view(b, t, e), max_len.cuda().view(-1), 15.0)
    ops.attn_mfma = False
    ctx_v, attn_v, rinv_v = ops.attn_g_fwd(region, wn.view(b, t, e), max_len.cuda().view(-1), 15.0)
    ops.attn_mfma = True
    mask = (torch.arange(t, dtype=torch.float64)[None, :] >= max_len.double()).double()[:, None, :].expand(-1, r, -1)

    def oracle(words_hat):                           # R.attention_for_g normalises its words: feed it the given unit rows
        rq = rr.clone().requires_grad_(True)
        c, a = R.attention_for_g(rq, words_hat, 15.0, mask)
        return rq, c, a
    wn_bf = wn.view(b, t, e).bfloat16().double().cpu()       # what the kernel multiplies (|w^| = 1 up to 2^-9)
    rq, ctx_ref, attn_ref = oracle(wn_bf)
    _close(rinv, 1.0 / rr.norm(dim=-1), torch.float32, "rinv", scale=float((1.0 / rr.norm(dim=-1)).max()))
    # R.attention_for_g re-normalises the bf16-rounded words (norm 1 +- 2^-9): same bar as the kernel tolerance
    _close(attn, attn_ref, dtype, "attn probs (bf16 words)", scale=1.0)
    _close(ctx, ctx_ref, dtype, "attn ctx (bf16 words)")
    _, ctx_x, attn_x = oracle(words.double())
    print("MFMA attention vs exact-word oracle: probs", float((attn.double().cpu() - attn_x).abs().max()),
          "ctx", float((ctx.double().cpu() - ctx_x.detach()).abs().max()) / float(ctx_x.abs().max()),
          "| vs VALU kernel: probs", float((attn - attn_v).abs().max()), "argmax agreement",
          float((attn.argmax(-1) == attn_v.argmax(-1)).float().mean()))
    assert float((attn.double().cpu() - attn_x).abs().max()) < 4e-2 and float((attn.argmax(-1) == attn_v.argmax(-1)).float().mean()) > 0.97
    dctx, dcr = _rnd((b, r, e), dtype, g)
    dreg = ops.attn_g_bwd(dctx, region, wn.view(b, t, e), attn, rinv, 15.0)
    (ref,) = torch.autograd.grad(ctx_ref, rq, dcr)
    _close(dreg, ref, dtype, "attn bwd (bf16 words)", scale=2.0 * float(ref.abs().max()))
    ops.attn_mfma = False
    dreg_v = ops.attn_g_bwd(dctx, region, wn.view(b, t, e), attn_v, rinv_v, 15.0)
    rel = float((dreg.float() - dreg_v.float()).norm() / dreg_v.float().norm())
    print("MFMA attention backward vs VALU kernel: norm-relative difference", rel)
    assert rel < 3e-2, rel


def test_l2norm_bwd():
    from oracle import torch_ref as R
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(13)
    x = torch.randn((37, 200), generator=g)
    x[5] = 0.0
    y, inv = ops.l2norm_fwd(x.cuda())
    xr = x.double().requires_grad_(True)
    yr = R.l2n(xr)
    _close(y, yr, torch.float32, "l2n")
    dy = torch.randn((37, 200), generator=g)
    dx = ops.l2norm_bwd(dy.cuda(), y, inv, torch.float32)
    (ref,) = torch.autograd.grad(yr, xr, dy.double())
    _close(dx, ref, torch.float32, "l2n bwd")


def test_xent_hinge_proj():
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(14)
    for b in (4, 56):
        L = torch.randn((b, b), generator=g) * 3
        acc = torch.zeros(1).cuda()
        stats = torch.zeros(2).cuda()
        dL = ops.xent_sym(L.cuda(), 1.0, acc, True, stats)
        from oracle import np_spec as S
        a1, e1 = S.get_statistics(L.double().numpy(), np.eye(b))
        a2, e2 = S.get_statistics(L.double().numpy().T, np.eye(b))
        assert abs(float(stats[0]) - 0.5 * (a1 + a2)) < 1e-6 and abs(float(stats[1]) - 0.5 * (e1 + e2)) < 1e-4
        Lr = L.double().requires_grad_(True)
        loss = -torch.diagonal(torch.log_softmax(Lr, 1)).mean() - torch.diagonal(torch.log_softmax(Lr.t(), 1)).mean()
        (ref,) = torch.autograd.grad(loss, Lr)
        _close(acc, loss.reshape(1), torch.float32, "xent loss")
        _close(dL, ref, torch.float32, "xent grad")
        logit = torch.randn(2 * b, generator=g) * 2
        dacc, gacc = torch.zeros(1).cuda(), torch.zeros(1).cuda()
        dld, dlg = ops.hinge(logit.cuda(), b, dacc, gacc)
        lr = logit.double().requires_grad_(True)
        dl = (torch.relu(1 - lr[:b]) + torch.relu(1 + lr[b:])).mean()
        gl = -lr[b:].mean()
        _close(dacc, dl.reshape(1), torch.float32, "hinge d")
        _close(gacc, gl.reshape(1), torch.float32, "hinge g")
        _close(dld, torch.autograd.grad(dl, lr)[0], torch.float32, "hinge dd", scale=1.0)
        _close(dlg, torch.autograd.grad(gl, lr)[0], torch.float32, "hinge dg", scale=1.0)
    n2, b, c = 8, 4, 96
    pool = torch.randn((n2, c), generator=g)
    w = torch.randn(c, generator=g)
    emb = torch.randn((b, c), generator=g)
    inv = torch.tensor([0.7])
    bias = torch.tensor([0.3])
    out = ops.proj_head_fwd(pool.cuda(), w.cuda(), inv.cuda(), bias.cuda(), emb.cuda())
    pr, er = pool.double().requires_grad_(True), emb.double().requires_grad_(True)
    ref = (pr * (w.double() * 0.7 + er.repeat(2, 1))).sum(1) + 0.3
    _close(out, ref, torch.float32, "proj fwd")
    dout = torch.randn(n2, generator=g)
    dpool, demb = ops.proj_head_bwd(dout.cuda(), pool.cuda(), w.cuda(), inv.cuda(), emb.cuda(), True)
    rp, re = torch.autograd.grad(ref, (pr, er), dout.double())
    _close(dpool, rp, torch.float32, "proj dpool")
    _close(demb, re, torch.float32, "proj demb")


@pytest.mark.parametrize("b,d", [(4, 96), (56, 1536), (32, 1000), (9, 2048)])
def test_contrastive_loss_fused_vs_float64_and_gemm_chain(b, d):
    """contrastive_loss (attention_lib.py:46-79) in two launches per direction (losses.hip cl_logits / cl_bwd, round 5) against
    float64 autograd of the restatement, and against the l2norm + GEMM + xent chain it replaces: loss, logits, both gradients
    (plain and accumulated into an existing tensor), a zero row (the 1e-12 clamp of l2_normalize)"""
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd.libml import attention_lib as A
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(100 + b)
    a = torch.randn((b, d), generator=g) * 3
    bb = torch.randn((b, d), generator=g) * 0.5
    a[1] = 0.0
    ar, br = a.double().requires_grad_(True), bb.double().requires_grad_(True)
    ref_loss = R.contrastive_loss(ar, br)[0]
    ga, gb = torch.autograd.grad(ref_loss, (ar, br))
    outs = {}
    for fused in (True, False):
        ops.cl_fused = fused
        acc = torch.zeros(1).cuda()
        tape = A.contrastive_loss_fwd(ops, a.cuda(), bb.cuda(), acc)
        assert bool(tape.get("fused")) == fused
        da, db = A.contrastive_loss_bwd(ops, tape)
        base_a, base_b = torch.randn((b, d), generator=g).cuda(), torch.randn((b, d), generator=g).cuda()
        sa, sb = base_a.clone(), base_b.clone()
        A.contrastive_loss_bwd(ops, tape, add_a=sa, add_b=sb)
        outs[fused] = (float(acc), tape["logits"].clone(), da.clone(), db.clone())
        _close(acc, ref_loss.reshape(1), torch.float32, f"contrastive loss fused={fused}")
        rows = [i for i in range(b) if i != 1]                   # (the all-zero row: the clamp's gradient, not the oracle's NaN-free limit)
        _close(da[rows], ga[rows], torch.float32, f"contrastive da fused={fused}", scale=float(ga[rows].abs().max()))
        _close(db, gb, torch.float32, f"contrastive db fused={fused}", scale=float(gb.abs().max()))
        for got, base, d_ in ((sa, base_a, da), (sb, base_b, db)):               # (the kernel may contract "old + v" into one fma)
            assert float((got - (base + d_)).abs().max()) <= 4e-7 * float(base.abs().max() + d_.abs().max())
    ops.cl_fused = None
    assert abs(outs[True][0] - outs[False][0]) <= 1e-5 * max(1.0, abs(outs[False][0]))
    for k in (1, 2, 3):
        sc = float(outs[False][k].abs().max())
        assert float((outs[True][k] - outs[False][k]).abs().max()) <= 2e-5 * sc + 1e-9, k


@pytest.mark.parametrize("u_axis", [0, 1])
def test_spectral(u_axis):
    from oracle import np_spec as S
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(15)
    rows, cols = (48, 9 * 40) if u_axis == 0 else (300, 64)
    w = torch.randn((rows, cols), generator=g) / 10
    nu = rows if u_axis == 0 else cols
    u0 = torch.randn((1, nu), generator=g) * 0.01
    u_new, v, scal = ops.spectral_power_iter(w.cuda(), u0.cuda(), u_axis)
    k2d = w.double().numpy().T if u_axis == 0 else w.double().numpy()      # (K, Cout) reference view
    _, u_ref, sigma = S.spectral_normalize(k2d, u0.double().numpy())
    _close(u_new, torch.from_numpy(u_ref), torch.float32, "u_new", scale=1.0)
    _close(scal[:1], torch.tensor([sigma]), torch.float32, "sigma")
    _close(scal[1:], torch.tensor([1.0 / (sigma + 1e-10)]), torch.float32, "inv sigma")
    # gradient through sigma: W_bar = W / (v W u^T + eps) with u, v constant
    gbar = torch.randn((rows, cols), generator=g)
    wr = w.double().requires_grad_(True)
    ur, vr = u_new.double().cpu(), v.double().cpu()
    k = wr.t() if u_axis == 0 else wr
    sig = (vr.reshape(1, -1) @ k @ ur.reshape(-1, 1))[0, 0]
    wbar = wr / (sig + 1e-10)
    (ref,) = torch.autograd.grad(wbar, wr, gbar.double())
    gdev = gbar.clone().cuda()
    ops.spectral_grad_fix(gdev, w.cuda(), u_new, v, scal, u_axis)
    _close(gdev, ref, torch.float32, "sn grad fix")


def test_adam_ema():
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(16)
    n = 1000
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    m, v, ema = torch.zeros(n), torch.zeros(n), p.clone()
    pd, md, vd, ed = p.clone().cuda(), m.cuda(), v.cuda(), ema.clone().cuda()
    pr, mr, vr, er = p.double(), m.double(), v.double(), ema.double()
    for step in (1, 2, 3):
        ops.adam_ema(pd, gr.cuda(), md, vd, ed, lr=1e-2, beta1=0.5, beta2=0.999, step=step, grad_scale=0.5,
                     ema_decay=0.999)
        gg = gr.double() * 0.5
        mr = 0.5 * mr + 0.5 * gg
        vr = 0.999 * vr + 0.001 * gg * gg
        pr = pr - 1e-2 * (mr / (1 - 0.5 ** step)) / (torch.sqrt(vr / (1 - 0.999 ** step)) + 1e-8)
        er = er * 0.999 + 0.001 * pr
    _close(pd, pr, torch.float32, "adam p")
    _close(ed, er, torch.float32, "adam ema")


def test_spectral_bank_matches_per_weight_path():
    """Batched spectral norm (one descriptor table) vs the per-weight kernels and vs oracle/np_spec."""
    from oracle import np_spec as S
    ops = _ops(torch.bfloat16)
    g = torch.Generator().manual_seed(21)
    shapes = [(96, 27, 0, 9), (192, 864, 0, 9), (768, 384, 0, 1), (1536, 1, 1, 1), (768, 1536, 1, 1), (40, 360, 0, 9)]
    sizes = [r * c for r, c, _, _ in shapes]
    offs = np.cumsum([0] + [((n + 63) // 64) * 64 for n in sizes])
    params = torch.zeros(int(offs[-1]))
    grads = torch.zeros(int(offs[-1]))
    entries, u0s = [], []
    for (r, c, ax, taps), off in zip(shapes, offs[:-1]):
        params[off:off + r * c] = torch.randn(r * c, generator=g) / 10
        grads[off:off + r * c] = torch.randn(r * c, generator=g)
        entries.append(dict(w_off=int(off), rows=r, cols=c, u_axis=ax, taps=taps, is_conv=(ax == 0)))
        u0s.append(torch.randn(r if ax == 0 else c, generator=g) * 0.01)
    bank = ops.sn_bank_create(entries)
    pd, gd = params.cuda(), grads.clone().cuda()
    u0_flat = torch.zeros(bank["nu"])                   # the bank's flat layout: every slice starts 16-byte aligned
    for e, u0 in zip(bank["entries"], u0s):
        u0_flat[e["u_off"]:e["u_off"] + e["nu"]] = u0
    u_new, v, scal = ops.sn_bank_power_iter(bank, pd, u0_flat.cuda())
    wf, wd = ops.sn_bank_prep(bank, pd, scal, True)
    ops.sn_bank_grad_fix(bank, pd, gd, u_new, v, scal)
    for i, (e, u0) in enumerate(zip(bank["entries"], u0s)):
        r, c, ax = e["rows"], e["cols"], e["u_axis"]
        w = params[e["w_off"]:e["w_off"] + r * c].view(r, c)
        k2d = w.double().numpy().T if ax == 0 else w.double().numpy()
        _, u_ref, sigma = S.spectral_normalize(k2d, u0.double().numpy().reshape(1, -1))
        got_u = u_new[e["u_off"]:e["u_off"] + e["nu"]]
        _close(got_u, torch.from_numpy(u_ref).reshape(-1), torch.float32, f"bank u {i}", scale=1.0)
        _close(scal[2 * i:2 * i + 1], torch.tensor([sigma]), torch.float32, f"bank sigma {i}")
        # per-weight path on the same data
        u1, v1, s1 = ops.spectral_power_iter(w.cuda(), u0.cuda().view(1, -1), ax)
        _close(v[e["v_off"]:e["v_off"] + e["nv"]], v1, torch.float32, f"bank v {i}", scale=1.0)
        g1 = grads[e["w_off"]:e["w_off"] + r * c].view(r, c).clone().cuda()
        ops.spectral_grad_fix(g1, w.cuda(), u1, v1, s1, ax)
        _close(gd[e["w_off"]:e["w_off"] + r * c].view(r, c), g1, torch.float32, f"bank grad fix {i}")
        if e["is_conv"]:
            f1, d1 = ops.prep_conv_weight(w.cuda().view(r, e["taps"], -1), s1[1:2])
            n = r * c
            # sigma comes out of float atomics (summation order differs between the two paths): the bf16
            # copies may differ by one ulp on a few elements
            _close(wf[e["wf_off"]:e["wf_off"] + n].view_as(f1), f1, torch.bfloat16, f"bank wf {i}")
            _close(wd[e["wf_off"]:e["wf_off"] + n].view_as(d1), d1, torch.bfloat16, f"bank wd {i}")


@pytest.mark.parametrize("dtype", DT)
def test_expand_taps_and_rgb_paths(dtype):
    """expand_taps is a pure gather (bit-exact); the RGB conv / wgrad built on it must equal the direct ones."""
    ops = _ops(dtype, variant=1)
    g = torch.Generator().manual_seed(21)
    for (n, h, w, c) in ((2, 16, 16, 3), (1, 8, 24, 3), (2, 16, 16, 1), (3, 128, 128, 3), (1, 8, 20, 3), (2, 5, 8, 3)):
        x, xr = _rnd((n, h, w, c), dtype, g)
        for ks in (1, 3):
            for sign in (1, -1):
                y = ops.expand_taps(x, ks, sign).cpu()
                half = ks // 2
                xp = F.pad(x.cpu(), (0, 0, half, half, half, half))
                ref = torch.zeros((n, h, w, 32), dtype=dtype)
                for tap in range(ks * ks):
                    dy, dx = sign * (tap // ks - half), sign * (tap % ks - half)
                    ref[..., tap * c:(tap + 1) * c] = xp[:, half + dy:half + dy + h, half + dx:half + dx + w, :]
                assert torch.equal(y, ref), (n, h, w, c, ks, sign)
    # conv with cin = 3 through the expanded 1x1 path == conv2d
    n, h, w, cout = 2, 64, 64, 96
    x, xr = _rnd((n, h, w, 3), dtype, g)
    wt, wr = _rnd((cout, 9, 3), dtype, g, 0.2)
    w32 = torch.zeros((cout, 1, 32), dtype=dtype, device="cuda")
    w32[:, 0, :27] = wt.reshape(cout, 27)
    xcol = ops.expand_taps(x, 3, 1)
    y = ops.conv(xcol, w32, None, ks=1)
    ref = F.conv2d(xr.permute(0, 3, 1, 2), wr.view(cout, 3, 3, 3).permute(0, 3, 1, 2), padding=1).permute(0, 2, 3, 1)
    _close(y, ref, dtype, "rgb-in conv")
    dy, dyr = _rnd((n, h, w, cout), dtype, g)
    dw32 = torch.zeros((cout, 1, 32), dtype=torch.float32, device="cuda")
    db = torch.zeros((cout,), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(xcol, dy, dw32, db, ks=1)
    xu = F.unfold(xr.permute(0, 3, 1, 2), 3, padding=1).view(n, 3, 9, h * w)          # [n][c][tap][q]
    ref_dw = torch.einsum("nctq,nqo->otc", xu, dyr.view(n, h * w, cout))
    _close(dw32[:, 0, :27].view(cout, 9, 3), ref_dw, dtype, "rgb-in wgrad")
    _close(db, dyr.sum((0, 1, 2)), dtype, "rgb-in db")
    # wgrad with cout = 3 through the expanded dy
    cin = 96
    a, ar = _rnd((n, h, w, cin), dtype, g)
    d3, d3r = _rnd((n, h, w, 3), dtype, g)
    dyx = ops.expand_taps(d3, 3, -1)
    dw = torch.zeros((32, 1, cin), dtype=torch.float32, device="cuda")
    ops.conv_wgrad(a, dyx, dw, None, ks=1)
    au = F.unfold(ar.permute(0, 3, 1, 2), 3, padding=1).view(n, cin, 9, h * w)
    ref_dw = torch.einsum("nctq,nqo->otc", au, d3r.view(n, h * w, 3))
    _close(dw[:27, 0, :].view(9, 3, cin).permute(1, 0, 2), ref_dw, dtype, "rgb-out wgrad")


def test_prep_conv_weight_packed_matches_pack_of_plain():
    """the prep kernels' fragment-order output == xmc_pack_conv_weight of their plain output (bit-exact),
    for the per-weight path and the batched spectral bank"""
    from xmcgan_image_generation_amd.ops import HipOps, PackedWeight
    plain, packed = HipOps(dtype=torch.bfloat16, stream_conv=False), HipOps(dtype=torch.bfloat16, stream_conv=True)
    g = torch.Generator().manual_seed(5)
    for cout, cin in ((64, 32), (40, 96), (96, 40), (3, 64), (160, 192)):
        w = torch.randn((cout, 9, cin), generator=g).cuda()
        inv = torch.tensor([0.37], device="cuda")
        wf, wd = plain.prep_conv_weight(w, inv)
        pf, pd = packed.prep_conv_weight(w, inv)
        for a, b, k in ((wf, pf, cin), (wd, pd, cout)):
            if k % 32 == 0:
                assert isinstance(b, PackedWeight)
                assert torch.equal(plain.pack_conv_weight(a).data, b.data), (cout, cin)
            else:
                assert torch.equal(a, b)
    # batched bank
    shapes = [(64, 9, 32), (96, 9, 64), (48, 1, 32), (3, 9, 32)]
    params = torch.randn(sum(a * b * c for a, b, c in shapes), generator=g).cuda()
    outs = []
    for ops in (plain, packed):
        entries, off = [], 0
        for (a, b, c) in shapes:
            entries.append(dict(w_off=off, rows=a, cols=b * c, u_axis=0, taps=b, is_conv=True))
            off += a * b * c
        bank = ops.sn_bank_create(entries)
        u0 = torch.randn(bank["nu"], generator=g).cuda() if not outs else outs[0][3]
        _, _, scal = ops.sn_bank_power_iter(bank, params, u0)
        wf, wd = ops.sn_bank_prep(bank, params, scal)
        outs.append((bank, wf, wd, u0))
    for i in range(len(shapes)):
        fa, da = plain.sn_bank_weights(outs[0][0], i, outs[0][1], outs[0][2])
        fb, db = packed.sn_bank_weights(outs[1][0], i, outs[1][1], outs[1][2])
        for a, b in ((fa, fb), (da, db)):
            if isinstance(b, PackedWeight):
                assert torch.equal(plain.pack_conv_weight(a).data, b.data), shapes[i]
            else:
                assert torch.equal(a, b), shapes[i]


@pytest.mark.parametrize("shape", [(70, 50, 33, False, False), (130, 140, 64, False, True), (300, 952, 768, False, True),
                                   (256, 768, 952, False, False), (200, 136, 100, True, False), (65, 65, 40, True, True),
                                   (56, 56, 1000, False, True), (56, 60, 2104, False, False), (64, 64, 96, False, True)])
def test_gemm_bf16_mfma(shape):
    """fast=True in the bf16 mode: operands rounded to bf16 inside the kernel, float32 accumulation -> equals the
    float64 product of the bf16-rounded operands to float32 round-off"""
    m, n, k, ta, tb = shape
    ops = _ops(torch.bfloat16)
    g = torch.Generator().manual_seed(15)
    a = torch.randn((k, m) if ta else (m, k), generator=g)
    b = torch.randn((n, k) if tb else (k, n), generator=g)
    c0 = torch.randn((m, n), generator=g)
    out = c0.clone().cuda()
    ops.gemm(a.cuda(), b.cuda(), ta=ta, tb=tb, alpha=2.0, beta=1.0, out=out, fast=True)
    ar, br = a.bfloat16().double(), b.bfloat16().double()
    ref = 2.0 * (ar.t() if ta else ar) @ (br.t() if tb else br) + c0.double()
    _close(out, ref, torch.float32, f"gemm bf16mfma {shape}")
    # batched, strided view, accumulate
    a3 = torch.randn((3, 40, 96), generator=g).cuda()
    big = torch.randn((3, 96, 80), generator=g).cuda()
    b3 = big[:, :, 8:72]
    out3 = ops.gemm(a3, b3, fast=True)
    _close(out3, a3.bfloat16().double().cpu() @ b3.bfloat16().double().cpu(), torch.float32, "bgemm bf16mfma")
    # the float32 mode ignores the flag (exact path)
    o32 = _ops(torch.float32).gemm(a.cuda(), b.cuda(), ta=ta, tb=tb, fast=True)
    _close(o32, (a.double().t() if ta else a.double()) @ (b.double().t() if tb else b.double()), torch.float32, "fast ignored")


@pytest.mark.parametrize("case", [(2, 64, 32, 96, False, True), (1, 128, 64, 40, False, False), (3, 32, 64, 136, True, True),
                                  (2, 16, 32, 64, True, False), (5, 32, 96, 3, False, True),
                                  # 96-cout tile (2-block waves): 32-wide rows, 64-wide rows (2 x 32 patches per wave)
                                  (2, 32, 64, 192, False, True), (1, 64, 96, 96, True, False), (3, 32, 32, 192, True, True)])
def test_conv_stream_pool_out(case):
    """fused 2x2 average pooling (+ residual at the pooled resolution) in the weight-streaming kernel's epilogue"""
    n, h, cin, cout, ups, relu_in = case
    dtype = torch.bfloat16
    ops = _ops(dtype)
    g = torch.Generator().manual_seed(31)
    x, xr = _rnd((n, h, h, cin), dtype, g)
    w32 = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    wf, _ = ops.prep_conv_weight(w32.cuda())
    wr = wf.double().cpu()
    bias = torch.randn(cout, generator=g)
    ho = 2 * h if ups else h
    res, resr = _rnd((n, ho // 2, ho // 2, cout), dtype, g)
    pw = ops.pack_conv_weight(wf)
    assert ops.can_pool_out(x, pw, ups)
    y = ops.conv(x, pw, bias.cuda(), ks=3, ups=ups, relu_in=relu_in, res=res, res_scale=0.5, alpha=0.5, pool_out=True)
    full = 0.5 * _ref_conv(xr, wr, None, 3, ups, relu_in) + bias.double()
    ref = F.avg_pool2d(full.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1) + 0.5 * resr
    assert y.shape == ref.shape
    _close(y, ref, dtype, f"pool_out {case}")


@pytest.mark.parametrize("case", [("ups", 3, 4, 64, 128), ("ups", 2, 8, 512, 128), ("ups", 5, 16, 96, 192), ("ups", 2, 32, 32, 96),
                                  ("ups", 1, 64, 64, 64), ("pool", 3, 8, 64, 128), ("pool", 2, 16, 512, 64), ("pool", 5, 32, 96, 192),
                                  ("pool", 2, 64, 32, 96), ("pool", 1, 128, 32, 128), ("pool", 18, 8, 128, 32)])
@pytest.mark.parametrize("exact", [True, False])
def test_conv_phase(case, exact):
    """conv3x3(upsample2(.)) / avg_pool2(conv3x3(.)) and their data gradients as four 2x2 convolutions on the low-resolution
    grid (conv_phase_kernel + xmc_phase_conv_weight), against the float64 3x3 formulation.  ``exact``: weights on a 1/16 grid,
    whose tap sums are exact in bf16 -- the tight kernel tolerance applies; otherwise N(0, 1/K) weights and the tolerance of one
    more bf16 rounding of the weights."""
    kind, n, h, cin, cout = case
    dtype = torch.bfloat16
    ops = _ops(dtype)
    ops.stream_conv = True                              # prepared weights in fragment order (the product default)
    g = torch.Generator().manual_seed(53)
    if exact:
        w32 = torch.randint(-4, 5, (cout, 9, cin), generator=g).float() / 16.0
    else:
        w32 = torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)
    wf, wd = ops.prep_conv_weight(w32.cuda(), None, True, phase=kind)
    assert wf.phase is not None and wd.phase is not None and wf.phase[0] == ("out" if kind == "ups" else "in")
    wr = (w32.bfloat16() if exact else w32).double()
    bias = torch.randn(cout, generator=g)
    x, xr = _rnd((n, h, h, cin), dtype, g)
    xr.requires_grad_(True)
    wscale = 1.0 if exact else 4.0                      # summed taps are rounded to bf16 once more
    if kind == "ups":
        assert ops._phase_ok(wf, "out", h, h, True, False) and ops._phase_ok(wd, "in", 2 * h, 2 * h, False, True)
        m, mr = _rnd((n, 2 * h, 2 * h, cout), dtype, g)
        y = ops.conv(x, wf, bias.cuda(), ks=3, ups=True, mask=m, alpha=0.5)
        full = _ref_conv(xr, wr, None, 3, True, False)
        ref = (0.5 * full + bias.double()) * (mr > 0)
        _close(y, ref, dtype, f"phase out {case}", scale=wscale * float(ref.detach().abs().max()))
        dy, dyr = _rnd((n, 2 * h, 2 * h, cout), dtype, g)
        dx = ops.conv(dy, wd, None, ks=3, pool_out=True, alpha=4.0)          # ConvSite.dgrad_sumpool
        (dxr,) = torch.autograd.grad(full, xr, dyr)
        _close(dx, dxr, dtype, f"phase out dgrad {case}", scale=wscale * float(dxr.abs().max()))
    else:
        assert ops.can_pool_out(x, wf) and ops._phase_ok(wf, "in", h, h, False, True) and ops._phase_ok(wd, "out", h // 2, h // 2, True, False)
        res, resr = _rnd((n, h // 2, h // 2, cout), dtype, g)
        y = ops.conv(x, wf, bias.cuda(), ks=3, relu_in=True, res=res, res_scale=0.5, pool_out=True)
        pooled = F.avg_pool2d(_ref_conv(xr, wr, None, 3, False, True).permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)
        ref = pooled + bias.double() + 0.5 * resr
        _close(y, ref, dtype, f"phase in {case}", scale=wscale * float(ref.detach().abs().max()))
        dp, dpr = _rnd((n, h // 2, h // 2, cout), dtype, g)
        dh = ops.conv(dp, wd, None, ks=3, ups=True, alpha=0.25, mask=x)      # DiscBlock.bwd: d(avg_pool) fused as ups * 1/4
        (dhr,) = torch.autograd.grad(pooled, xr, dpr)                         # includes the ReLU mask (x > 0)
        _close(dh, dhr, dtype, f"phase in dgrad {case}", scale=wscale * float(dhr.abs().max()))
    # the same launches without the phase copies: both formulations agree with each other
    assert wf.data is None and wd.data is None          # a phase site: its 3x3 copies are not made at all
    ops.phase_conv = False
    wf3, _ = ops.prep_conv_weight(w32.cuda(), None, True)
    if kind == "ups":
        y2 = ops.conv(x, wf3, bias.cuda(), ks=3, ups=True, mask=m, alpha=0.5)
    else:
        y2 = ops.pool2(ops.conv(x, wf3, bias.cuda(), ks=3, relu_in=True), 0.25, res=res * 0.5)
    # a launch of the phase-only site OUTSIDE the phase kernels' domain (no resampling here): the plain 3x3 copy is made on
    # first use from the master the weight remembers and gives exactly the 3x3 kernel's result; a weight that
    # carries neither data nor a master still fails loudly
    y_plain = ops.conv(x, wf, bias.cuda(), ks=3)
    assert wf.data is not None and torch.equal(y_plain, ops.conv(x, wf3, bias.cuda(), ks=3))
    from xmcgan_image_generation_amd.ops import PackedWeight
    with pytest.raises(Exception):
        ops.conv(x, PackedWeight(None, cout, 9, cin), bias.cuda(), ks=3)
    _close(y, y2.double(), dtype, f"phase vs 3x3 {case}", scale=wscale * float(y2.abs().max()))


@pytest.mark.parametrize("case", [(3, 8, 64, 64), (2, 16, 512, 128), (2, 64, 128, 96), (1, 32, 256, 256)])
def test_conv_stride2_phase(case):
    """stride-2 SAME 3x3 convolution (flax padding (0, 1) on an even-sized map: y[o] = sum_r w[r] x[2o + r]) and its adjoint on
    conv_phase_kernel (16 entries per low-resolution pixel, 7 of them zero) against float64."""
    n, h, cin, cout = case
    dtype = torch.bfloat16
    ops = _ops(dtype)
    ops.stream_conv = True
    g = torch.Generator().manual_seed(61)
    w32 = torch.randint(-4, 5, (cout, 9, cin), generator=g).float() / 16.0        # exact in bf16
    wf, wd = ops.prep_conv_weight(w32.cuda(), None, True, phase="s2")
    assert ops.can_stride2(wf, h, h) and ops.can_stride2(wd, h // 2, h // 2)
    bias = torch.randn(cout, generator=g)
    x, xr = _rnd((n, h, h, cin), dtype, g)
    xr.requires_grad_(True)
    y = ops.conv(x, wf, bias.cuda(), ks=3, relu_out=True, stride2=True)
    wk = w32.double().reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    lin = F.conv2d(F.pad(xr.permute(0, 3, 1, 2), (0, 1, 0, 1)), wk, None, stride=2).permute(0, 2, 3, 1)
    ref = torch.relu(lin + bias.double())
    assert y.shape == ref.shape
    _close(y, ref, dtype, f"stride-2 fwd {case}", scale=float(ref.detach().abs().max()))
    # the same layer the old way: stride 1, then the odd positions
    y1 = ops.conv(x, wf, bias.cuda(), ks=3, relu_out=True)[:, 1::2, 1::2, :]
    _close(y, y1.double(), dtype, f"stride-2 vs subsampled stride-1 {case}", scale=float(ref.detach().abs().max()))
    dy, dyr = _rnd((n, h // 2, h // 2, cout), dtype, g)
    m, mr = _rnd((n, h, h, cin), dtype, g)
    dx = ops.conv(dy, wd, None, ks=3, mask=m, stride2=True)
    (dxr,) = torch.autograd.grad(lin, xr, dyr)
    dxr = dxr * (mr > 0)
    _close(dx, dxr, dtype, f"stride-2 adjoint {case}", scale=float(dxr.abs().max()))


@pytest.mark.parametrize("case", [("plain", 3, 16, 64, 96), ("plain", 1, 32, 256, 64), ("plain", 3, 64, 32, 48), ("after_res", 1, 16, 64, 256),
                                  ("ups", 3, 16, 96, 192), ("ups", 3, 4, 64, 64)])
def test_conv_mask_bits(case):
    """ReLU masks as bits: a convolution's epilogue writes (y > 0) as one uint16 per 16 channels (``y.bits``), and a launch
    whose mask carries bits gives the same result, bit for bit, as with the bf16 mask tensor."""
    kind, ks, h, cin, cout = case
    dtype = torch.bfloat16
    ops = _ops(dtype)
    ops.stream_conv = True
    g = torch.Generator().manual_seed(67)
    taps = ks * ks
    w32 = torch.randn((cout, taps, cin), generator=g) / math.sqrt(taps * cin)
    bias = torch.randn(cout, generator=g).cuda()
    hm = 2 * h if kind == "ups" else h                     # resolution of the masked output
    # producer: a tensor of the consumer's output shape, with bits
    xm, _ = _rnd((2, hm, hm, cin), dtype, g)
    wm, _ = ops.prep_conv_weight(w32.cuda(), None, True)
    m = ops.conv(xm, wm, bias, ks=ks, relu_out=(kind == "after_res"), emit_bits=True)
    assert hasattr(m, "bits") and tuple(m.bits.shape) == tuple(m.shape[:-1]) + (cout // 16,)
    bits = m.bits.cpu().numpy().view(np.uint16)
    want = (m.float().cpu().numpy() > 0).reshape(bits.shape + (16,))
    got = ((bits[..., None] >> np.arange(16, dtype=np.uint16)) & 1).astype(bool)
    assert np.array_equal(got, want)
    assert torch.equal(ops.bslice(m, 1, 2).bits, m.bits[1:2])
    plain = torch.empty_like(m).copy_(m)                   # the same mask without the bits
    x, _ = _rnd((2, h, h, cin), dtype, g)
    if kind == "ups":
        wf, wd = ops.prep_conv_weight(torch.randn((cin, 9, cout), generator=g).cuda() / math.sqrt(9 * cin), None, True, phase="pool")
        a = ops.conv(x, wd, None, ks=3, ups=True, alpha=0.25, mask=m)            # DiscBlock.bwd's c1.dgrad
        b = ops.conv(x, wd, None, ks=3, ups=True, alpha=0.25, mask=plain)
    else:
        res, _ = _rnd((2, h, h, cout), dtype, g)
        after = kind == "after_res"
        a = ops.conv(x, wm, None, ks=ks, mask=m, res=res, mask_after_res=after)
        b = ops.conv(x, wm, None, ks=ks, mask=plain, res=res, mask_after_res=after)
    assert torch.equal(a, b)


@pytest.mark.parametrize("case", [(3, 64, 56, 64, 256), (5, 32, 28, 128, 64), (2, 16, 14, 256, 96), (9, 8, 7, 512, 128), (40, 128, 112, 32, 64)])
def test_conv_pointwise_compact(case):
    """pointwise kernel, compact mode: only the valid corner of every canvas is walked -- the valid region equals the
    whole-canvas launch bit for bit, the margin of the (caller-owned) output is never written."""
    n, s, hv, cin, cout = case
    dtype = torch.bfloat16
    ops = _ops(dtype)
    ops.stream_conv = True
    g = torch.Generator().manual_seed(71)
    w32 = torch.randn((cout, 1, cin), generator=g) / math.sqrt(cin)
    wf, _ = ops.prep_conv_weight(w32.cuda(), None, True)
    bias = torch.randn(cout, generator=g).cuda()
    x, _ = _rnd((n, s, s, cin), dtype, g)
    res, _ = _rnd((n, s, s, cout), dtype, g)
    m, _ = _rnd((n, s, s, cout), dtype, g)
    full = ops.conv(x, wf, bias, ks=1, res=res, mask=m, mask_after_res=True, relu_out=True, valid=hv, emit_bits=True)
    out = torch.full((n, s, s, cout), 7.0, dtype=dtype, device="cuda")
    y = ops.conv(x, wf, bias, ks=1, res=res, mask=m, mask_after_res=True, relu_out=True, valid=hv, emit_bits=True, compact=True, out=out)
    assert y is out
    assert torch.equal(y[:, :hv, :hv], full[:, :hv, :hv])
    if hasattr(full, "bits"):                          # (not on the split-K route)
        assert torch.equal(y.bits[:, :hv, :hv], full.bits[:, :hv, :hv])
    margin = torch.ones((n, s, s), dtype=torch.bool, device="cuda")
    margin[:, :hv, :hv] = False
    # (a launch that takes the split-K route is not compacted: its finishing kernel zeroes the margin as before)
    assert bool((y[margin] == 7.0).all()) or bool((y[margin] == 0).all()), "the compact launch wrote into the margin"
    assert bool((full[margin] == 0).all())


def test_adam_ema_device_step_counter():
    """xmc_adam_ema_dev: the step counter / bias corrections live in device memory (hipGraph replay)."""
    ops = _ops(torch.float32)
    g = torch.Generator().manual_seed(17)
    n = 4096
    p, gr = torch.randn(n, generator=g), torch.randn(n, generator=g)
    pd, md, vd, ed = p.clone().cuda(), torch.zeros(n).cuda(), torch.zeros(n).cuda(), p.clone().cuda()
    state = torch.zeros(4, device="cuda")
    pr, mr, vr, er = p.double(), torch.zeros(n).double(), torch.zeros(n).double(), p.double()
    for step in (1, 2, 3, 4):
        ops.adam_ema_dev(pd, gr.cuda(), md, vd, ed, state, lr=4e-4, beta1=0.5, beta2=0.999, grad_scale=0.25,
                         ema_decay=0.999)
        gg = gr.double() * 0.25
        mr = 0.5 * mr + 0.5 * gg
        vr = 0.999 * vr + 0.001 * gg * gg
        pr = pr - 4e-4 * (mr / (1 - 0.5 ** step)) / (torch.sqrt(vr / (1 - 0.999 ** step)) + 1e-8)
        er = er * 0.999 + 0.001 * pr
        assert int(state.view(torch.int32)[0]) == step
        assert abs(float(state[2]) * (1 - 0.999 ** step) - 1.0) < 1e-6          # 1 / (1 - beta2^t), computed in double
    upd, upd_ref = (pd.double().cpu() - p.double()), (pr - p.double())
    # the parameter itself is float32 (|p| ~ 1 -> 6e-8 per rounding, 4 roundings), the update is ~ lr per step
    assert float((upd - upd_ref).abs().max()) <= 4e-7 * float(p.abs().max()) + 1e-4 * float(upd_ref.abs().max())
    _close(ed, er, torch.float32, "adam ema")


def _word_loss_case(b, r, t, e, max_len, seed, dtype, ops):
    """image features (b, r, e) in the activation dtype, words (b, t, e) float32 -> tape of word_loss_fwd"""
    from xmcgan_image_generation_amd.libml import attention_lib as A
    g = torch.Generator().manual_seed(seed)
    feat, feat64 = _rnd((b, r, e), dtype, g)
    words = torch.randn((b, t, e), generator=g)
    ml = torch.tensor(max_len, dtype=torch.float32).view(b, 1)
    wn = A.normalize_words(ops, words.cuda())
    loss = torch.zeros(1, device="cuda")
    stats = torch.zeros(2, device="cuda")
    tape = A.word_loss_fwd(ops, feat, wn, ml.cuda(), loss, stats=stats)
    return feat, feat64, words, ml, tape, loss, stats


@pytest.mark.parametrize("dtype", DT)
@pytest.mark.parametrize("case", [
    dict(b=3, r=64, t=17, e=64, max_len=[4, 17, 9]),
    dict(b=4, r=256, t=17, e=768, max_len=[1, 17, 1, 12]),       # max_len edge rows: one word / no masked word
    dict(b=2, r=16, t=5, e=32, max_len=[5, 5]),
    dict(b=5, r=32, t=17, e=96, max_len=[17, 17, 17, 17, 17]),
])
def test_word_loss_kernels_vs_spec(dtype, case):
    """wl_softmax / wl_qdot / wl_rows / wl_bwd_cols + the three GEMMs against the float64 NumPy specification of
    attention_lib.word_loss (reference attention_lib.py:130-191) and its autograd gradient: loss, the B x B similarity
    matrix, accuracy / entropy statistics, d loss / d image features."""
    from oracle import np_spec as S
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd.libml import attention_lib as A
    ops = _ops(dtype)
    b, r, t, e, max_len = case["b"], case["r"], case["t"], case["e"], case["max_len"]
    feat, feat64, words, ml, tape, loss, stats = _word_loss_case(b, r, t, e, max_len, 31 + b, dtype, ops)
    ref_loss, ref_acc, ref_ent, ref_sims = S.word_loss(feat64.numpy(), words.double().numpy(), ml.double().numpy(),
                                                       return_logits=True)
    # bf16 mode multiplies on the bf16 MFMA (operands rounded to bf16 inside the GEMMs): similarities are O(50)
    tol = 2e-4 if dtype == torch.float32 else 2e-2
    got_sim = tape["sim_t"].double().cpu().numpy().T             # kernel: [caption i, image j]; spec sims_n: [image, caption]
    assert np.abs(got_sim - ref_sims).max() <= tol * np.abs(ref_sims).max(), np.abs(got_sim - ref_sims).max()
    assert abs(float(loss) - ref_loss) <= tol * max(1.0, abs(ref_loss)), (float(loss), ref_loss)
    if dtype == torch.float32:
        assert abs(float(stats[0]) - ref_acc) < 1e-6 and abs(float(stats[1]) - ref_ent) <= 1e-3 * max(1.0, ref_ent)
    # gradient wrt the image features vs autograd through the torch restatement
    x = feat64.clone().requires_grad_(True)
    l_ref, sim_ref = R.word_loss(x, words.double(), ml.double())
    assert np.abs(sim_ref.detach().numpy() - ref_sims).max() <= 1e-9 * np.abs(ref_sims).max()      # the two oracles agree
    (gref,) = torch.autograd.grad(l_ref, x)
    dx = A.word_loss_bwd(ops, tape).double().cpu()
    err = float((dx - gref).norm() / gref.norm())
    assert err <= (1e-3 if dtype == torch.float32 else 4e-2), err


def test_word_loss_all_equal_features_is_two_ln_b():
    """Known answer: identical images and identical captions -> every similarity equal -> 2 ln B."""
    from xmcgan_image_generation_amd.libml import attention_lib as A
    ops = _ops(torch.float32)
    b, r, t, e = 4, 32, 17, 64
    g = torch.Generator().manual_seed(5)
    feat = torch.randn((1, r, e), generator=g).expand(b, -1, -1).contiguous().cuda()
    words = torch.randn((1, t, e), generator=g).expand(b, -1, -1).contiguous().cuda()
    ml = torch.full((b, 1), 9.0).cuda()
    loss = torch.zeros(1, device="cuda")
    A.word_loss_fwd(ops, feat, A.normalize_words(ops, words), ml, loss)
    assert abs(float(loss) - 2 * math.log(b)) < 1e-4
