"""Size-independent properties of the convolution kernels at the FULL C1 layer sizes (BASELINE.json configs[1]:
per-device batch 56, 112 images through D), where no CPU oracle finishes in test time:

* adjoint identities  <dy, conv(x, W)> == <W, wgrad(x, dy)> == <x, dgrad(dy, W)>  tie the weight-streaming forward /
  data-gradient kernel, the LDS-DMA weight-gradient kernel and the fragment-order / dgrad weight layouts to
  each other (float32 outputs, inner products in float64);
* exact homogeneity  conv(2 x) == 2 conv(x)  (a power-of-two scale commutes with every rounding) -- bitwise;
* batch-permutation equivariance  conv(x[perm]) == conv(x)[perm]  -- bitwise: a pixel's result may not depend on
  which tile / workgroup / split computed it.
"""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu

# (n, h, cin, cout, ups): the 128^2 and 64^2 96/192-channel layers, a mid layer, and a split-K 4x4 layer
LAYERS = [(112, 128, 96, 96, False), (56, 64, 192, 96, True), (112, 32, 192, 384, False), (112, 4, 1536, 1536, False)]


def _ops():
    from xmcgan_image_generation_amd.ops import HipOps
    return HipOps(dtype=torch.bfloat16)


@pytest.mark.parametrize("layer", LAYERS)
def test_adjoint_identities_full_size(layer):
    n, h, cin, cout, ups = layer
    ops = _ops()
    g = torch.Generator().manual_seed(3)
    ho = 2 * h if ups else h
    x = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
    dy = torch.randn((n, ho, ho, cout), generator=g).bfloat16().cuda()
    w = (torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)).cuda()
    wf, wd = ops.prep_conv_weight(w)
    y = ops.conv(x, wf, None, ks=3, ups=ups, out_f32=True)                      # weight-streaming forward
    dw = torch.zeros_like(w)
    ops.conv_wgrad(x, dy, dw, None, ks=3, x_ups=ups)                            # LDS-DMA weight gradient
    # the kernels see the bf16-rounded weights: pair dW with the same rounded weights
    from xmcgan_image_generation_amd.ops import HipOps
    wr = HipOps(dtype=torch.bfloat16, stream_conv=False).prep_conv_weight(w)[0].double()
    a = float((dy.double() * y.double()).sum())
    b = float((wr * dw.double()).sum())
    print(layer, "<dy,conv>", a, "<W,wgrad>", b)
    scale = float(dy.double().norm() * y.double().norm())
    assert abs(a - b) <= 2e-4 * scale / math.sqrt(y.numel()) * 30 + 1e-6 * abs(a), (a, b)
    if not ups:
        dx = ops.conv(dy, wd, None, ks=3, out_f32=True)                         # data gradient (streaming kernel)
        c = float((x.double() * dx.double()).sum())
        print(layer, "<x,dgrad>", c)
        assert abs(a - c) <= 2e-4 * scale / math.sqrt(y.numel()) * 30 + 1e-6 * abs(a), (a, c)


@pytest.mark.parametrize("layer", LAYERS)
def test_scaling_and_permutation_bitwise_full_size(layer):
    n, h, cin, cout, ups = layer
    ops = _ops()
    g = torch.Generator().manual_seed(4)
    x = torch.randn((n, h, h, cin), generator=g).bfloat16().cuda()
    w = (torch.randn((cout, 9, cin), generator=g) / math.sqrt(9 * cin)).cuda()
    bias = None
    wf, _ = ops.prep_conv_weight(w)
    y = ops.conv(x, wf, bias, ks=3, ups=ups, relu_in=True)
    y2 = ops.conv(x * 2, wf, bias, ks=3, ups=ups, relu_in=True)
    assert torch.equal(y2.float(), 2 * y.float()), "conv(2x) != 2 conv(x)"
    perm = torch.randperm(n, generator=g).cuda()
    yp = ops.conv(x[perm].contiguous(), wf, bias, ks=3, ups=ups, relu_in=True)
    assert torch.equal(yp, y[perm]), "conv is not batch-permutation equivariant bit for bit"


@pytest.mark.parametrize("mode", [["--pretrained"], ["--fp8"], ["--dtype", "float32"]])
def test_step_reads_no_uninitialised_memory(mode):
    """tools/poison_check.py: two C1-network steps (batch 4) with every torch.empty / empty_like allocation filled with 0xFF
    bytes (NaN in float32, bf16, e4m3 and e8m0) give the same losses, bit for bit, as with the allocator's leftovers --
    no kernel reads memory nobody wrote (workspaces of empty splits, packet padding, canvas margins)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "poison_check.py"), "--batch", "4"] + mode, cwd=root,
                         env=dict(os.environ, PYTHONPATH=root), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "poison check OK" in out.stdout, out.stdout[-3000:] + out.stderr[-2000:]
