"""SURVEY.md 8(f) N1 -- the frozen ResNet-50 image-contrastive term -- on the CPU mock operator table: the
canvas / folded-BatchNorm / sub-sampled-stride formulation of ``utils/pretrained_model_utils.py`` and the hand-written
data gradient against the oracle's plain NHWC ResNet-50 (``oracle/torch_ref.resnet50``), plus the train_g_d wiring."""
import math

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests.cpu_ops import CpuOps
from xmcgan_image_generation_amd import synthetic as syn
from xmcgan_image_generation_amd import train_utils, xmc_gan
from xmcgan_image_generation_amd.configs import coco_xmc
from xmcgan_image_generation_amd.nets import xmc_net
from xmcgan_image_generation_amd.utils import pretrained_model_utils as P
from xmcgan_image_generation_amd.utils import resnet_v1 as RV


def test_resnet50_parameter_count_and_tree():
    """25,557,032 = the torchvision / Flax ResNet-50 v1 count; tree names follow resnet_v1.py"""
    p, s = RV.init_resnet50(0)
    assert RV.count_params(p) == 25_557_032
    assert set(p) == {"init_conv", "init_bn", "stage1", "stage2", "stage3", "stage4", "head"}
    assert [len(p[f"stage{i}"]) for i in (1, 2, 3, 4)] == [3, 4, 6, 3]
    assert p["stage2"]["block1"]["proj_conv"]["kernel"].shape == (1, 1, 256, 512)
    assert "proj_conv" not in p["stage2"]["block2"]
    assert float(np.abs(p["head"]["kernel"]).max()) == 0.0          # zero-initialised head (resnet_v1.py:168-171)
    assert s["stage4"]["block3"]["bn3"]["var"].shape == (2048,)


@pytest.fixture(scope="module")
def net_and_ref():
    p, s = RV.init_resnet50(1, head_scale=0.05, randomize_bn=True)
    net = P.ResNet50Features(CpuOps(torch.float32), p, s)
    g = torch.Generator().manual_seed(0)
    x = torch.rand((3, 32, 32, 3), generator=g) * 2 - 1
    dl = torch.randn((3, 1000), generator=g)
    return p, s, net, x, dl


def test_forward_matches_oracle(net_and_ref):
    p, s, net, x, _ = net_and_ref
    pool_ref, ref = R.get_pretrained_embs(R.to_torch(p, torch.float64), R.to_torch(s, torch.float64), x.double())
    pool, logits = P.get_pretrained_embs(None, net, x)
    assert pool.shape == (3, 7, 7, 2048) and logits.shape == (3, 1000)
    assert float((logits.double() - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert float((pool.double() - pool_ref).abs().max()) <= 2e-5 * float(pool_ref.abs().max())


def test_backward_matches_oracle_on_a_batch_slice(net_and_ref):
    p, s, net, x, dl = net_and_ref
    xr = x.clone().requires_grad_(True)
    _, ref = R.get_pretrained_embs(R.to_torch(p, torch.float32), R.to_torch(s, torch.float32), xr)
    (gref,) = torch.autograd.grad(ref, xr, dl)
    logits, tape = net.forward(x)
    dimg = net.backward(tape, dl[1:3].contiguous(), 1, 3)
    assert dimg.shape == (2, 32, 32, 3)
    # float32 ReLU / max-pool decisions flip between two evaluation orders of a random-weight network: the float32 and
    # float64 oracles differ from each other by 3e-3 here; a wiring error would be O(1)
    rel = float((dimg - gref[1:3]).norm() / gref[1:3].norm())
    assert rel < 1e-2, rel


@pytest.mark.parametrize("bias_nudge", [0.0, 1.0])
def test_zero_head_gives_two_ln_b_and_no_gradient(bias_nudge):
    """model.init's zero head kernel: every output row is the bias -> all rows equal -> both cross-entropies are ln B and
    nothing depends on the images.  With the zero bias of model.init the rows are 0 and l2_normalize's
    clamp (rsqrt(max(|x|^2, 1e-12)), attention_lib.py:30-33) keeps them 0; a non-zero bias gives equal non-zero rows."""
    ops = CpuOps(torch.float32)
    p, s = P.get_pretrained_model(checkpoint_path=None)
    model = P.ImageModel({"params": p, "batch_stats": s})
    g = torch.Generator().manual_seed(3)
    real, fake = torch.rand((4, 16, 16, 3), generator=g), torch.rand((4, 16, 16, 3), generator=g)
    model.bind(ops).head_b += bias_nudge
    loss, pull = xmc_gan.calculate_contrastive_loss_on_pretrained(model, model.state, real, fake, ops=ops)
    assert abs(float(loss[0]) - 2 * math.log(4)) < 1e-5
    assert float(pull().abs().max()) == 0.0


def test_create_additional_data_surface(tmp_path):
    cfg = coco_xmc.get_test_config()
    assert xmc_gan.create_additional_data(cfg) == {}                 # flag off: xmc_gan.py:43-55
    p, s = RV.init_resnet50(5, head_scale=0.1)
    ckpt = tmp_path / "resnet_pretrained.npy"
    np.save(ckpt, {"params": p, "batch_stats": s}, allow_pickle=True)   # the reference's checkpoint format (:93-98)
    cfg.pretrained_image_contrastive = True
    cfg.pretrained_model_path = str(ckpt)
    ad = xmc_gan.create_additional_data(cfg)
    assert set(ad) == {"image_model", "image_model_state"}
    np.testing.assert_array_equal(ad["image_model_state"]["params"]["head"]["kernel"], p["head"]["kernel"])
    with pytest.raises(ValueError):
        P.get_pretrained_model("vgg16")
    cfg.pretrained_model_path = str(tmp_path / "missing.npy")       # the reference's np.load fails hard too
    with pytest.raises(FileNotFoundError):
        xmc_gan.create_additional_data(cfg)


@pytest.fixture(scope="module")
def stepped_with_resnet():
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.pretrained_image_contrastive = True
    cfg.d_step_per_g_step = 1
    rp, rs = RV.init_resnet50(7, head_scale=0.2, randomize_bn=True)
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = syn.make_batch(cfg, per_device_batch=2)
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        state = train_utils.load_flax_params(state, gp, gs, dp, ds)
        tb = {k: torch.as_tensor(v) for k, v in batch.items()}
        st = {"params": rp, "batch_stats": rs}
        ad = {"image_model": P.ImageModel(st), "image_model_state": st}
        new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, ad)
        ref_state = R.make_state(gp, gs, dp, ds, torch.float32, resnet=(rp, rs))
        ref_new, ref_metrics, dbg = R.train_step(ref_state, R.batch_to_torch(batch), cfg, return_debug=True)
    finally:
        xmc_net.set_ops_factory(None)
    return cfg, new_state, metrics, ref_metrics, dbg


def test_train_step_with_pretrained_term_metrics(stepped_with_resnet):
    _, _, metrics, ref_metrics, _ = stepped_with_resnet
    assert float(ref_metrics["c_loss_g_pretrained"]) > 0.1           # the term is live
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g", "c_loss_g_pretrained"):
        assert abs(float(metrics[k]) - float(ref_metrics[k])) <= 3e-4 * max(1.0, abs(float(ref_metrics[k]))), k


def test_train_step_with_pretrained_term_gradients(stepped_with_resnet):
    """the ResNet term only reaches the GENERATOR's gradient (through the generated images)"""
    _, new_state, _, _, dbg = stepped_with_resnet
    for which, opt, tol in (("d_grad", new_state.d_optimizer, 2e-3), ("g_grad", new_state.g_optimizer, 1e-2)):
        got = opt.arena.tree(opt.arena.grads)
        ref_leaves = R.leaves(dbg[which])
        rms = (sum(float(b.double().pow(2).sum()) for _, b in ref_leaves) / sum(b.numel() for _, b in ref_leaves)) ** 0.5
        for (p1, a), (p2, b) in zip(syn.tree_leaves(got), ref_leaves):
            assert p1 == p2
            err = float((a.double() - b.double()).norm())
            r = err / max(float(b.double().norm()), 1e-2 * rms * b.numel() ** 0.5)
            assert r < tol, (which, p1, r)


@pytest.mark.parametrize("n_in", [16, 100, 128, 224, 256, 300])
def test_oracle_resize_matrix_against_torch_antialiased_bilinear(n_in):
    """the oracle's restatement of jax.image.resize's weight matrix, pinned against an independent implementation
    of the same filter (torch's anti-aliased bilinear = PIL's): identical when enlarging AND when shrinking"""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(n_in)
    x = torch.rand((2, n_in, n_in, 3), generator=g, dtype=torch.float64)
    w = R.resize_weight_matrix(n_in, 224)
    got = torch.einsum("nhwc,hy,wx->nyxc", x, w, w)
    ref = F.interpolate(x.permute(0, 3, 1, 2), size=(224, 224), mode="bilinear", align_corners=False,
                        antialias=True).permute(0, 2, 3, 1)
    assert float((got - ref).abs().max()) < 1e-12
    assert float((w.sum(0) - 1).abs().max()) < 1e-12              # every output sample is a convex combination
    if n_in <= 224:                                               # enlarging: no anti-aliasing = plain bilinear
        plain = F.interpolate(x.permute(0, 3, 1, 2), size=(224, 224), mode="bilinear", align_corners=False).permute(0, 2, 3, 1)
        assert float((got - plain).abs().max()) < 1e-12


@pytest.mark.parametrize("size", [128, 256])
def test_output_shapes_with_random_init(size):
    """the reference's own test of this path (pretrained_model_utils_test.py:28-36): pool (B, 7, 7, 2048) and outputs
    (B, 1000) from 128 px and 256 px images (256 -> 224 takes the anti-aliased resize)"""
    p, s = P.get_pretrained_model(checkpoint_path=None)
    model = P.ImageModel({"params": p, "batch_stats": s})
    images = torch.rand((1, size, size, 3), generator=torch.Generator().manual_seed(size))
    pool, outputs = P.get_pretrained_embs(model.state, model, images, ops=CpuOps(torch.float32))
    assert tuple(pool.shape) == (1, 7, 7, 2048) and tuple(outputs.shape) == (1, 1000)
    pool_ref, out_ref = R.get_pretrained_embs(R.to_torch(p, torch.float64), R.to_torch(s, torch.float64), images.double())
    assert float((pool.double() - pool_ref).abs().max()) <= 2e-5 * max(float(pool_ref.abs().max()), 1e-6)
    with pytest.raises(ValueError):
        P.get_pretrained_embs(model.state, model, images[0], ops=CpuOps(torch.float32))


def test_step_mode_launch_forms_equal_the_reference_shaped_ones_on_the_mock():
    """ResNet50Features's HOST logic for the training step's launch forms (round 6) on the CPU mock, float32: compact pointwise
    launches into reused buffers, the projection shortcut folded into the block's last 1x1 ([h | x(s y, s x)] [W3 | Wp]^T, b3 + bp), the
    blocks' data gradient over [dh1 | scatter2(g)] with [W1^T | Wp^T], the fused stem and its data gradient -- against the
    reference-shaped launches (separate projection, sub-sampling copies, im2col stem): same logits, same image gradient, twice in a
    row (the second pass reuses the first one's buffers)."""
    from tests.cpu_ops import CpuOps, CpuOpsStepMode
    from xmcgan_image_generation_amd.utils import pretrained_model_utils as P
    from xmcgan_image_generation_amd.utils import resnet_v1 as RV
    p, s = RV.init_resnet50(3, head_scale=0.2, randomize_bn=True)
    g = torch.Generator().manual_seed(0)
    dl = torch.randn((2, 1000), generator=g)
    ref_net, step_net = P.ResNet50Features(CpuOps(torch.float32), p, s), P.ResNet50Features(CpuOpsStepMode(torch.float32), p, s)
    assert all(("c3p" in b) == (b["proj"] is not None) for b in step_net.blocks) and step_net.stem_frag is not None
    for rep in range(2):
        x = torch.rand((3, 128, 128, 3), generator=g) * 2 - 1
        ref_logits, ref_tape = ref_net.forward(x)
        got_logits, got_tape = step_net.forward(x, reuse_buffers=True)
        assert got_tape["compact"] and not ref_tape["compact"]
        scale = float(ref_logits.abs().max())
        assert float((got_logits - ref_logits).abs().max()) <= 2e-5 * scale, rep
        ref_g = ref_net.backward(ref_tape, dl, 1, 3)
        got_g = step_net.backward(got_tape, dl, 1, 3)
        # the folded forward differs from the separate one by float32 rounding (1e-5 of the logits): where that flips a max-pool arg-max or
        # a ReLU sign next to zero, that pixel's gradient is rerouted -- a few per cent of the elements, 2e-3 of the norm
        assert float((got_g - ref_g).norm() / ref_g.norm()) < 1e-2, rep
        # ... so the data-gradient forms are held to the reference-shaped ones on the SAME tape: the step-mode forward's, pulled back once
        # with the dual-source launches and the fused stem gradient, once without them
        keep = [(b.pop("c1pd", None)) for b in step_net.blocks]
        dfrag, step_net.stem_dfrag = step_net.stem_dfrag, None
        sep_g = step_net.backward(got_tape, dl, 1, 3).clone()
        for b, k in zip(step_net.blocks, keep):
            if k is not None:
                b["c1pd"] = k
        step_net.stem_dfrag = dfrag
        assert float((got_g - sep_g).abs().max()) <= 2e-5 * float(sep_g.abs().max()), rep


def test_stem_weight_packers_on_the_host():
    """HipOps.pack_stem_weight / pack_stem_dgrad_weight (host-side NumPy, no GPU needed): every fragment element is the weight the
    header documents (include/xmcgan_hip.h: xmc_stem_conv7x7s2, xmc_stem_conv7x7s2_dgrad), zeros in the pad slots"""
    import itertools
    from xmcgan_image_generation_amd.ops import HipOps

    class Dev:
        device = "cpu"
    w = np.random.default_rng(0).standard_normal((64, 49, 3)).astype(np.float32)
    wb = torch.as_tensor(w).bfloat16().float().numpy()
    f = HipOps.pack_stem_weight(Dev(), w).float().numpy()                        # [cout / 32][k-step][lane][8]
    for cb, ks, l, e in itertools.product(range(2), range(11), range(64), range(8)):
        k = ks * 16 + (l >> 5) * 8 + e
        ky, j = divmod(k, 24)
        want = wb[cb * 32 + (l & 31), ky * 7 + j // 3, j % 3] if (ky < 7 and j < 21) else 0.0
        assert f[cb, ks, l, e] == want
    d = HipOps.pack_stem_dgrad_weight(Dev(), w).float().numpy()                  # [channel half][tap][k-step][lane][8]
    for half, tap, s_, l, e in itertools.product(range(2), range(16), range(2), range(64), range(8)):
        r, co = l & 31, half * 32 + s_ * 16 + (l >> 5) * 8 + e
        t, u = tap >> 2, tap & 3
        want = 0.0
        if r < 12:
            q, c = divmod(r, 3)
            ky, kx = 2 * t + (q >> 1), 2 * u + (q & 1)
            if ky <= 6 and kx <= 6:
                want = wb[co, ky * 7 + kx, c]
        assert d[half, tap, s_, l, e] == want
