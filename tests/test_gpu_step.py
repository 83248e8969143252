"""Whole-step parity on the MI355X: ``train_step`` through the C ABI vs the oracle's
``train_step`` (oracle/torch_ref.py) on identical synthetic batches and parameters.

Bar (BASELINE.json north_star / SURVEY.md 8(d)): float32 mode -- G/D losses and all contrastive
logits within 1e-3 relative, attention argmax indices identical, gradients within 2e-3 of the
oracle's (norm-relative, floored at the round-off level); bf16 mode -- losses within 2e-2 of the
float32 oracle (reported, loose gate 5e-2).
"""
import re

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(cfg, b, seed=0):
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils
    gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
    dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
    batch = syn.make_batch(cfg, per_device_batch=b)
    gen, disc, state = train_utils.create_train_state(cfg, seed)
    state = train_utils.load_flax_params(state, gp, gs, dp, ds)
    ref_state = R.make_state(gp, gs, dp, ds, torch.float32)
    return gen, disc, state, ref_state, batch


def _rel_scalar(a, b):
    return abs(float(a) - float(b)) / max(abs(float(b)), 1e-6)


def _check_grads(got_tree, ref_leaves, tol, tag):
    from xmcgan_image_generation_amd import synthetic as syn
    rms = (sum(float(b.double().pow(2).sum()) for _, b in ref_leaves) / sum(b.numel() for _, b in ref_leaves)) ** 0.5
    worst, worst_p = 0.0, None
    for (p1, a), (p2, b) in zip(syn.tree_leaves(got_tree), ref_leaves):
        assert p1 == p2
        a = a.detach().double().cpu()
        err = float((a - b.double()).norm())
        zero = 1e-3 * rms * b.numel() ** 0.5
        if float(a.norm()) < zero and float(b.double().norm()) < zero:
            continue        # analytically-zero gradient (a bias that only feeds BatchNorms): pure round-off on both sides
        # every reduction of the step is deterministic now (no float atomics), so there is no run-to-run noise to
        # allow for: the error is measured against the leaf's own norm
        r = err / float(b.double().norm())
        if r > worst:
            worst, worst_p = r, p1
    print(f"{tag}: worst norm-relative gradient error {worst:.3e} at {worst_p}")
    assert worst < tol, (tag, worst_p, worst)


@pytest.mark.usefixtures("keep_grads")
@pytest.mark.parametrize("b", [2, 4])
def test_train_step_fp32_tiny(b):
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = b
    gen, disc, state, ref_state, batch = _setup(cfg, b)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    ref_new, ref_metrics, dbg = R.train_step(ref_state, R.batch_to_torch(batch), cfg, return_debug=True)
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = _rel_scalar(metrics[k], ref_metrics[k])
        print(k, float(metrics[k]), float(ref_metrics[k]), r)
        assert r < 1e-3, (k, float(metrics[k]), float(ref_metrics[k]))
    # contrastive logits and attention indices of the train_g_d half
    aux, daux = dbg["aux"], disc(train=True).last_aux
    for k in ("fake_sentence_logits", "real_sentence_logits", "image_contrastive_logits"):
        ref = aux[k][0].detach()
        got = daux[k].cpu()
        assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), k
    for k, rk in (("fake_word_sim_t", "fake_word_sim"), ("real_word_sim_t", "real_word_sim")):
        ref = aux[rk].detach().t()
        got = daux[k].cpu()
        assert float((got - ref).abs().max()) <= 1e-3 * float(ref.abs().max()), k
    attn = gen(train=True).last_attn.cpu()
    assert torch.equal(attn.argmax(-1), aux["attn"].argmax(-1)), "attention indices must be identical"
    _check_grads(new_state.d_optimizer.arena.tree(new_state.d_optimizer.arena.grads), R.leaves(dbg["d_grad"]),
                 2e-3, "d_grad")
    # G: the worst leaf is always a conditioning bias (Dense_0/bias of a cBN site, or a conv bias in front of one: sums
    # with heavy cancellation); the step is deterministic, measured 1.4e-3 (B = 2) / 1.9e-3 (B = 4) on MI355X
    _check_grads(new_state.g_optimizer.arena.tree(new_state.g_optimizer.arena.grads), R.leaves(dbg["g_grad"]),
                 2.5e-3, "g_grad")
    assert new_state.step == 1 and new_state.d_optimizer.state["step"] == 2
    _post_step_check(new_state, ref_new, dbg, 1e-3, f"tiny b{b}")
    for (p1, a), (p2, bb) in zip(_leaves(new_state.generator_state["batch_stats"]),
                                 R.leaves(ref_new["generator_state"])):
        assert p1 == p2
        assert float((a.cpu() - bb).abs().max()) <= 1e-4 * max(1.0, float(bb.abs().max())), p1


def _leaves(tree):
    from xmcgan_image_generation_amd import synthetic as syn
    return syn.tree_leaves(tree)


def test_train_step_bf16_tiny_losses():
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.dtype = "bfloat16"
    gen, disc, state, ref_state, batch = _setup(cfg, 4)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    cfg32 = cfg.copy()
    _, ref_metrics = R.train_step(ref_state, R.batch_to_torch(batch), cfg32)
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = _rel_scalar(metrics[k], ref_metrics[k])
        print("bf16", k, float(metrics[k]), float(ref_metrics[k]), r)
        assert np.isfinite(float(metrics[k])) and r < 5e-2, k
    flat = new_state.g_optimizer.arena.params
    assert bool(torch.isfinite(flat).all())


_NOISE_FLOOR = 1e-3      # the same floor _check_grads uses for "analytically zero"


def _noise_leaves(ref_grad_leaves):
    """Leaves whose true gradient is ZERO (conv / dense biases that only feed a BatchNorm): both sides hold float32
    round-off there, and Adam turns round-off into +-lr steps (m / sqrt(v) = sign(g)) -- not comparable."""
    rms = (sum(float(b.double().pow(2).sum()) for _, b in ref_grad_leaves) / sum(b.numel() for _, b in ref_grad_leaves)) ** 0.5
    return {p for p, b in ref_grad_leaves if float(b.double().norm()) < _NOISE_FLOOR * rms * b.numel() ** 0.5}


def _post_step_check(new_state, ref_new, dbg, tol_param, tag):
    """Post-step parameters, EMA, spectral-norm u0 and BatchNorm running statistics vs the oracle's new state
    ("post-step params within 1e-3 rel"); per-leaf norm-relative error."""
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import synthetic as syn
    worst = {}
    skip = {"g_params": _noise_leaves(R.leaves(dbg["g_grad"])), "d_params": _noise_leaves(R.leaves(dbg["d_grad"]))}
    skip["ema_params"] = skip["g_params"]
    for name, arena, ref_tree, buf in (("g_params", new_state.g_optimizer.arena, ref_new["g_params"], None),
                                       ("d_params", new_state.d_optimizer.arena, ref_new["d_params"], None),
                                       ("ema_params", new_state.g_optimizer.arena, ref_new["ema_params"], new_state.ema_buffer)):
        got_leaves = syn.tree_leaves(arena.tree(buf) if buf is not None else arena.tree())
        w, wp = 0.0, None
        num = den = 0.0
        for (p1, a), (p2, b) in zip(got_leaves, R.leaves(ref_tree)):
            assert p1 == p2
            if p1 in skip[name]:
                continue
            a, b = a.detach().double().cpu(), b.double()
            r = float((a - b).norm() / max(float(b.norm()), 1e-12)) if float(b.norm()) > 1e-6 else float((a - b).norm())
            if r > w:
                w, wp = r, p1
            num += float((a - b).pow(2).sum())
            den += float(b.pow(2).sum())
        worst[name] = (w, wp, (num / den) ** 0.5)
        print(f"{tag} {name}: worst leaf norm-relative error {w:.3e} at {wp}; whole-tree {worst[name][2]:.3e}")
        assert w < tol_param, (tag, name, wp, w)
    got_sn = dict(syn.tree_leaves(new_state.discriminator_state["spectral_norm_stats"]))
    ref_sn = dict(R.leaves(ref_new["discriminator_state"]))
    assert set(got_sn) == set(ref_sn)
    for p1, b in ref_sn.items():
        # u0 after the step's two power iterations (the second one on the UPDATED weights): same scale of error as
        # the parameters it was iterated on, amplified by the spectral gap
        assert float((got_sn[p1].cpu().reshape(b.shape) - b).norm() / b.norm()) <= 5 * tol_param, (tag, "u0", p1)
    got_bn = dict(syn.tree_leaves(new_state.generator_state["batch_stats"]))
    ref_bn = dict(R.leaves(ref_new["generator_state"]))
    assert set(got_bn) == set(ref_bn)
    for p1, b in ref_bn.items():
        assert float((got_bn[p1].cpu().reshape(b.shape) - b).abs().max()) <= max(1e-3, tol_param) * max(1.0, float(b.abs().max())), (tag, "batch_stats", p1)
    return worst


_C1B8 = {}


def _c1_b8_oracle():
    """ONE oracle train_step at the C1 network (gf = df = 96, z = 128, 128 px) with per-device batch 8 (16 images
    through D) -- the size bench.py's cpu_baseline times in ~10 s -- shared by the float32 and the bf16 test."""
    if not _C1B8:
        from oracle import torch_ref as R
        from xmcgan_image_generation_amd import synthetic as syn
        from xmcgan_image_generation_amd.configs import coco_xmc
        cfg = coco_xmc.get_c1_config()
        cfg.pretrained_image_contrastive = False      # the ResNet-50 term: tests/test_gpu_resnet.py
        cfg.dtype = "float32"
        cfg.batch_size = 8
        cfg.ema = True
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = syn.make_batch(cfg, per_device_batch=8)
        ref_state = R.make_state(gp, gs, dp, ds, torch.float32)
        ref_new, ref_metrics, dbg = R.train_step(ref_state, R.batch_to_torch(batch), cfg, return_debug=True)
        _C1B8.update(cfg=cfg, init=(gp, gs, dp, ds), batch=batch, ref_new=ref_new, ref_metrics=ref_metrics, dbg=dbg)
    return _C1B8


def _run_c1_b8(dtype):
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    o = _c1_b8_oracle()
    cfg = o["cfg"].copy()
    cfg.dtype = dtype
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    state = train_utils.load_flax_params(state, *o["init"])
    tb = {k: torch.as_tensor(v).cuda() for k, v in o["batch"].items()}
    new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    return o, gen, disc, new_state, metrics


def _check_logits(daux, aux, tol, tag):
    """the three B x B contrastive logit matrices and both word-similarity matrices of the train_g_d half"""
    for k in ("fake_sentence_logits", "real_sentence_logits", "image_contrastive_logits"):
        ref, got = aux[k][0].detach(), daux[k].cpu()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        print(tag, k, "max error / max |logit|:", err)
        assert err <= tol, (tag, k, err)
    for k, rk in (("fake_word_sim_t", "fake_word_sim"), ("real_word_sim_t", "real_word_sim")):
        ref, got = aux[rk].detach().t(), daux[k].cpu()
        err = float((got - ref).abs().max()) / float(ref.abs().max())
        print(tag, k, "max error / max |sim|:", err)
        assert err <= tol, (tag, k, err)


@pytest.mark.usefixtures("keep_grads")
def test_train_step_fp32_c1_shapes_batch8():
    """C1 network at per-device batch 8, float32 parity mode, vs the oracle: losses, all B x B contrastive logits,
    word similarities, attention indices, gradients and the post-step state (SURVEY.md 8(d) parity bar)."""
    from oracle import torch_ref as R
    o, gen, disc, new_state, metrics = _run_c1_b8("float32")
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = _rel_scalar(metrics[k], o["ref_metrics"][k])
        print("c1 b8", k, float(metrics[k]), float(o["ref_metrics"][k]), r)
        assert r < 1e-3, k
    _check_logits(disc(train=True).last_aux, o["dbg"]["aux"], 1e-3, "c1 b8 fp32")
    attn = gen(train=True).last_attn.cpu()
    assert torch.equal(attn.argmax(-1), o["dbg"]["aux"]["attn"].argmax(-1)), "attention indices must be identical"
    _check_grads(new_state.d_optimizer.arena.tree(new_state.d_optimizer.arena.grads), R.leaves(o["dbg"]["d_grad"]),
                 5e-3, "c1 b8 d_grad")
    _check_grads(new_state.g_optimizer.arena.tree(new_state.g_optimizer.arena.grads), R.leaves(o["dbg"]["g_grad"]),
                 5e-3, "c1 b8 g_grad")
    _post_step_check(new_state, o["ref_new"], o["dbg"], 1e-3, "c1 b8 fp32")


@pytest.mark.usefixtures("keep_grads")
def test_train_step_bf16_c1_shapes_batch8_vs_oracle():
    """The bf16 product path (weight-streaming conv, LDS-DMA wgrad, bf16-MFMA word_loss products) at the C1 network,
    per-device batch 8, against the float32 ORACLE: losses within 2e-2, the B x B logit matrices and word-similarity
    matrices within 8e-2 of their scale (measured 2.9e-2 .. 4.2e-2 on the sentence logits from run to run: 1536-long
    bf16 features times 1/0.1; SURVEY 8(d) has the bf16 logits "reported, not gated" -- the gate only catches blunders),
    post-step parameters within 2.5e-2 per leaf (measured 1.2e-2: bf16 rounding moves the sign of the smallest
    gradients, and one Adam step is lr * sign(g))."""
    o, gen, disc, new_state, metrics = _run_c1_b8("bfloat16")
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = _rel_scalar(metrics[k], o["ref_metrics"][k])
        print("c1 b8 bf16", k, float(metrics[k]), float(o["ref_metrics"][k]), r)
        assert r < 2e-2, k
    _check_logits(disc(train=True).last_aux, o["dbg"]["aux"], 8e-2, "c1 b8 bf16")
    attn = gen(train=True).last_attn.cpu()
    same = float((attn.argmax(-1) == o["dbg"]["aux"]["attn"].argmax(-1)).float().mean())
    print("c1 b8 bf16: attention argmax agreement with the float32 oracle:", same)
    assert same > 0.97
    _post_step_check(new_state, o["ref_new"], o["dbg"], 2.5e-2, "c1 b8 bf16")


def test_eval_step_and_determinism():
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.dtype = "bfloat16"
    gen, disc, state, _, batch = _setup(cfg, 2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    half = {k: v[:2] for k, v in tb.items()}
    img, ema = train_utils.eval_step(0, state, half, gen, cfg)
    assert img.shape == (2, 128, 128, 3) and torch.equal(img, ema)        # ema == params before any step
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    img2, _ = train_utils.eval_step(0, state, half, gen, cfg)
    assert torch.equal(img, img2)


@pytest.mark.usefixtures("keep_grads")
def test_train_step_fp32_256px_small():
    """256 px topology (6 generator stages, 6 discriminator blocks, attention at the 16x16 stage) at small
    width: float32 parity of the whole step vs the oracle."""
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.image_size = 256
    cfg.batch_size = 2
    gen, disc, state, ref_state, batch = _setup(cfg, 2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    _, ref_metrics, dbg = R.train_step(ref_state, R.batch_to_torch(batch), cfg, return_debug=True)
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = _rel_scalar(metrics[k], ref_metrics[k])
        print("256px", k, float(metrics[k]), float(ref_metrics[k]), r)
        assert r < 1e-3, k
    _check_grads(new_state.d_optimizer.arena.tree(new_state.d_optimizer.arena.grads), R.leaves(dbg["d_grad"]),
                 3e-3, "256px d_grad")
    _check_grads(new_state.g_optimizer.arena.tree(new_state.g_optimizer.arena.grads), R.leaves(dbg["g_grad"]),
                 3e-3, "256px g_grad")


def test_train_step_bf16_256px_runs():
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.image_size = 256
    cfg.dtype = "bfloat16"
    cfg.batch_size = 2
    gen, disc, state, _, batch = _setup(cfg, 2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    _, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    assert all(np.isfinite(float(v)) for v in metrics.values())


def _bench_additional():
    """the frozen ResNet-50 exactly as bench.py builds it (random init, non-zero head: no network for the checkpoint)"""
    from xmcgan_image_generation_amd.utils import pretrained_model_utils, resnet_v1
    rp, rs = resnet_v1.init_resnet50(seed=7, head_scale=0.05)
    st = {"params": rp, "batch_stats": rs}
    return {"image_model": pretrained_model_utils.ImageModel(st), "image_model_state": st}


@pytest.mark.usefixtures("keep_grads")
def test_benchmarked_workload_c1_b56_resnet_on():
    """THE workload bench.py times (BASELINE config #2 with the reference-default objective): C1 -- 128 px, gf = df = 96,
    per-device batch 56 (112 images through D), bf16, EMA off, frozen ResNet-50 image-contrastive term ON.  The oracle
    cannot run this size in test time, so the checks are size-independent properties of the product itself:
      (1) the bf16 step (weight-streaming conv, pointwise kernel, LDS-DMA wgrad, bf16-MFMA word_loss products, split-K
          paths) against the float32 parity mode (exact-fp32 MFMA everywhere; parity-tested against the oracle at small
          batch) on the same batch and parameters: every loss incl. c_loss_g_pretrained within 2e-2 of the loss scale,
          gradient arenas cosine > 0.99 (measured 0.998 / 0.999) and norm-relative difference < 1e-1;
      (2) hipGraph replay == eager, BIT for bit: a second step replayed from the captured graph gives the same
          metrics and the same updated G and D parameter arenas as a second eager step from the same state."""
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    out = {}
    for dt in ("float32", "bfloat16"):
        cfg = coco_xmc.get_c1_config()
        assert cfg.batch_size == 56 and not cfg.get("ema", True)
        cfg.pretrained_image_contrastive = True
        cfg.dtype = dt
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=cfg.batch_size).items()}
        assert batch["image"].shape[0] == 112
        runs = []
        for mode in (("eager",) if dt == "float32" else ("eager", "graph")):
            additional = _bench_additional()
            gen, disc, state = train_utils.create_train_state(cfg, 0)
            state = train_utils.load_flax_params(state, gp, gs, dp, ds)
            state, metrics = train_utils.train_step(0, state, batch, xmc_gan, gen, disc, cfg, additional)
            first = ({k: float(v) for k, v in metrics.items()}, state.g_optimizer.arena.grads.clone(),
                     state.d_optimizer.arena.grads.clone())
            if dt == "bfloat16":
                if mode == "graph":
                    graphed = train_utils.GraphedTrainStep(state, batch, xmc_gan, gen, disc, cfg, additional)
                    state, m2 = graphed(graphed.state)
                else:
                    state, m2 = train_utils.train_step(1, state, batch, xmc_gan, gen, disc, cfg, additional)
                torch.cuda.synchronize()
                runs.append(({k: float(v) for k, v in m2.items()}, state.g_optimizer.arena.params.clone(),
                             state.d_optimizer.arena.params.clone()))
                assert all(np.isfinite(v) for v in runs[-1][0].values())
                if mode == "graph":
                    del graphed
            out.setdefault(dt, first)
            del state, gen, disc, additional
            torch.cuda.empty_cache()
        if dt == "bfloat16":
            (me, ge, de), (mg, gg, dg) = runs
            assert me == mg, ("graph replay vs eager metrics", me, mg)
            assert torch.equal(ge, gg) and torch.equal(de, dg), "graph replay vs eager: updated parameters differ"
    m32, g32, d32 = out["float32"]
    m16, g16, d16 = out["bfloat16"]
    scale = max(abs(m32[k]) for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"))
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g", "c_loss_g_pretrained"):
        r = abs(m16[k] - m32[k]) / scale
        print("benchmarked workload", k, m16[k], m32[k], r)
        assert r < 2e-2, (k, m16[k], m32[k])
    assert m32["c_loss_g_pretrained"] > 0.0
    for name, a, b in (("g_grad", g16, g32), ("d_grad", d16, d32)):
        r = float((a - b).norm() / b.norm())
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        print("benchmarked workload", name, "norm-relative difference bf16 vs fp32:", r, "cosine:", cos)
        assert r < 1e-1 and cos > 0.99, (name, r, cos)


def test_benchmarked_workload_product_optimiser_mode_equals_the_test_sessions():
    """The tests that read a gradient arena run with XMC_KEEP_GRADS=1 (the ``keep_grads`` fixture: the optimiser kernel writes the
    final gradient back); bench.py, every user and the rest of this session run the DEFAULT mode (the optimiser consumes the
    gradient arena in place).
    At the benchmarked size (C1, B = 56, bf16, EMA off, ResNet-50 term on) two steps in each mode from the same state:
    metrics of both steps, both parameter arenas and D's Adam moments equal BIT for bit."""
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_c1_config()
    cfg.pretrained_image_contrastive = True
    cfg.dtype = "bfloat16"
    gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
    dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
    batches = [{k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=cfg.batch_size, rank=s).items()}
               for s in range(2)]
    out = {}
    for keep in (True, False):
        additional = _bench_additional()
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        ops = gen(train=True).ops
        assert disc(train=True).ops is ops and ops.fuse_opt
        ops.keep_grads = keep
        state = train_utils.load_flax_params(state, gp, gs, dp, ds)
        ms = []
        for s in range(2):
            state, m = train_utils.train_step(s, state, batches[s], xmc_gan, gen, disc, cfg, additional)
            ms.append({k: float(v) for k, v in m.items()})
        torch.cuda.synchronize()
        out[keep] = (ms, state.g_optimizer.arena.params.clone(), state.d_optimizer.arena.params.clone(),
                     state.d_optimizer.arena.m.clone(), state.d_optimizer.arena.v.clone())
        del state, gen, disc, additional
        torch.cuda.empty_cache()
    assert out[True][0] == out[False][0], (out[True][0], out[False][0])
    assert all(np.isfinite(v) for m in out[False][0] for v in m.values())
    for name, a, b in zip(("g params", "d params", "d m", "d v"), out[True][1:], out[False][1:]):
        assert torch.equal(a, b), (name, float((a - b).abs().max()))


def test_checkpoint_and_sampling_on_device(tmp_path):
    """N2 / N3 on the HIP backend: flax-layout checkpoint round trip of device arenas, sampling grids."""
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.utils import checkpoint
    cfg = coco_xmc.get_test_config()
    cfg.dtype = "bfloat16"
    gen, disc, state, _, batch = _setup(cfg, 2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    path = str(tmp_path / "ckpt.flax")
    checkpoint.save(path, state)
    gen2, disc2, other = train_utils.create_train_state(cfg, 99)
    other = checkpoint.restore(path, other)
    for a, b in ((state.g_optimizer.arena, other.g_optimizer.arena), (state.d_optimizer.arena, other.d_optimizer.arena)):
        assert torch.equal(a.params, b.params) and torch.equal(a.m, b.m) and torch.equal(a.v, b.v)
    half = {k: v[:2] for k, v in tb.items()}
    img_a, _ = train_utils.eval_step(3, state, half, gen, cfg)
    img_b, _ = train_utils.eval_step(3, other, half, gen2, cfg)
    assert torch.equal(img_a, img_b)                       # restored state generates the same images
    out = train_utils.generate_batch(3, other, {k: v[:4] for k, v in tb.items()}, gen2, cfg)
    assert out["generated_image_batch"].shape == (1, 256, 256, 3) and out["generated_image_batch"].dtype == torch.float32


def test_eval_step_matches_oracle_generator_eval_mode():
    """N2: eval_step = G(train=False) with RUNNING BatchNorm statistics, once with the parameters and
    once with the EMA parameters (train_utils.py:245-281, eval_metrics.py:90-124), against the oracle's generator in
    eval mode on the oracle's own post-step state -- after one training step, so running statistics, parameters and
    EMA all differ from their initial values."""
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    gen, disc, state, ref_state, batch = _setup(cfg, 2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    ref_new, _ = R.train_step(ref_state, R.batch_to_torch(batch), cfg)
    half = {k: v[:2] for k, v in tb.items()}
    img, ema_img = train_utils.eval_step(0, state, half, gen, cfg)
    rb = R.batch_to_torch({k: v[:2] for k, v in batch.items()})
    ref, _, _ = R.generator(ref_new["g_params"], ref_new["generator_state"], rb, rb["z"], cfg, False)
    ref_e, _, _ = R.generator(ref_new["ema_params"], ref_new["generator_state"], rb, rb["z"], cfg, False)
    err, err_e = float((img.cpu() - ref).abs().max()), float((ema_img.cpu() - ref_e).abs().max())
    print("eval_step vs oracle: max |image error|", err, "EMA", err_e)
    assert err < 2e-3 and err_e < 2e-3                       # images live in [0, 1]
    assert float((img.cpu() - ema_img.cpu()).abs().max()) > 1e-6        # the two generators really differ


@pytest.mark.usefixtures("keep_grads")
def test_c3_full_size_properties():
    """BASELINE config #4's per-GPU workload -- 256 px, gf = df = 96, per-device batch 32 (64 images through D) --
    at full size.  The oracle cannot run it in test time, so: (1) the bf16 step against the product's own float32
    parity mode on the same batch and parameters (losses 2e-2, gradient arenas cosine > 0.99), (2) the adjoint
    identities <dy, conv(x, W)> = <W, wgrad(x, dy)> = <x, dgrad(dy, W)> on the 256^2 layer shapes that only this
    config has."""
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.ops import HipOps
    out = {}
    for dt in ("float32", "bfloat16"):
        cfg = coco_xmc.get_c3_config()
        cfg.pretrained_image_contrastive = False      # the ResNet-50 term: tests/test_gpu_resnet.py
        cfg.dtype = dt
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=cfg.batch_size).items()}
        assert batch["image"].shape == (64, 256, 256, 3)
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        state = train_utils.load_flax_params(state, gp, gs, dp, ds)
        state, metrics = train_utils.train_step(0, state, batch, xmc_gan, gen, disc, cfg, {})
        out[dt] = ({k: float(v) for k, v in metrics.items()}, state.g_optimizer.arena.grads.clone(),
                   state.d_optimizer.arena.grads.clone())
        del state, gen, disc, batch
        torch.cuda.empty_cache()
    m32, g32, d32 = out["float32"]
    m16, g16, d16 = out["bfloat16"]
    scale = max(abs(m32[k]) for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"))   # g_loss = hinge_g + c_loss_g cancels
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        r = abs(m16[k] - m32[k]) / scale
        print("full C3", k, m16[k], m32[k], r)
        assert np.isfinite(m16[k]) and r < 2e-2, (k, m16[k], m32[k])
    for name, a, b in (("g_grad", g16, g32), ("d_grad", d16, d32)):
        cos = float((a * b).sum() / (a.norm() * b.norm()))
        print("full C3", name, "cosine bf16 vs fp32:", cos)
        assert cos > 0.99, (name, cos)
    # adjoint identities on the 256^2 layers (G 128>256 192>96 up, G 256 96>96, D 256 96>96)
    ops = HipOps(dtype=torch.bfloat16)
    g = torch.Generator().manual_seed(3)
    for n, hi, cin, cout, ups in ((8, 128, 192, 96, True), (8, 256, 96, 96, False)):
        ho = 2 * hi if ups else hi
        x = (torch.randn((n, hi, hi, cin), generator=g) * 0.5).to(torch.bfloat16).cuda()
        dy = (torch.randn((n, ho, ho, cout), generator=g) * 0.5).to(torch.bfloat16).cuda()
        w = (torch.randn((cout, 9, cin), generator=g) * 0.05).cuda()
        wf, wd = ops.prep_conv_weight(w, None, True)
        y = ops.conv(x, wf, None, ks=3, ups=ups, out_f32=True)
        a = float((y.double() * dy.double()).sum())
        dw = torch.zeros_like(w)
        ops.conv_wgrad(x, dy, dw, None, ks=3, x_ups=ups, sync=True)
        wq = w.to(torch.bfloat16).double()                               # the conv multiplies by the bf16-rounded weights
        b = float((dw.double() * wq).sum())
        dx = ops.conv(dy, wd, None, ks=3, out_f32=True)
        if ups:
            dx = ops.pool2(dx, 1.0)                                      # adjoint of the nearest upsample: 2x2 SUM
        c = float((dx.double() * x.double()).sum())
        print("adjoint", (n, hi, cin, cout, ups), a, b, c)
        assert abs(a - b) <= 2e-3 * abs(a) and abs(a - c) <= 1e-2 * abs(a), (a, b, c)


@pytest.mark.parametrize("dtype", ["float32", "bfloat16"])
def test_train_step_is_bit_reproducible(dtype):
    """No float atomics anywhere in the step (split-K weight gradients, bias gradients, BatchNorm statistics,
    spectral-norm matvecs, split-K GEMMs all reduce through workspaces in a fixed order): two runs of train_step
    from the same state on the same batch give BIT-IDENTICAL losses, gradients and updated parameters."""
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    outs = []
    for _ in range(2):
        cfg = coco_xmc.get_test_config()
        cfg.dtype = dtype
        cfg.df_dim = cfg.gf_dim = 32                 # wide enough for split-K paths and the tiled bf16 kernels
        cfg.batch_size = 4
        gen, disc, state, _, batch = _setup(cfg, 4)
        tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
        state, m = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
        state, m = train_utils.train_step(1, state, tb, xmc_gan, gen, disc, cfg, {})
        outs.append(({k: float(v) for k, v in m.items()}, state.g_optimizer.arena.grads.clone(), state.d_optimizer.arena.grads.clone(),
                     state.g_optimizer.arena.params.clone(), state.d_optimizer.arena.params.clone()))
        del state, gen, disc
    assert outs[0][0] == outs[1][0], (outs[0][0], outs[1][0])
    for k, name in enumerate(("g_grad", "d_grad", "g_params", "d_params"), start=1):
        same = torch.equal(outs[0][k], outs[1][k])
        if not same:
            diff = (outs[0][k] != outs[1][k]).float().mean()
            print(dtype, name, "fraction of differing elements:", float(diff))
        assert same, name


def test_independent_flax_encoder_checkpoint_into_device_arenas_vs_oracle():
    """N3 on the GPU against something OTHER than the product's own writer: a full C0-shaped
    reference-layout TrainState (Flax names, HWIO conv kernels, Adam ``grad_ema`` / ``grad_sq_ema``, int32 step counters,
    batch_stats, spectral_norm_stats, ema_params) is encoded by the INDEPENDENT from-the-spec encoder of
    tests/golden/make_flax_fixture.py (no code shared with utils/checkpoint.py or the msgpack library), restored into
    device arenas, and then (1) every arena leaf equals the encoder's input bit for bit through the Flax-layout views,
    (2) ``eval_step`` on the restored parameters / EMA parameters / running statistics equals the ORACLE's eval-mode
    generator on the same NumPy weights."""
    import importlib.util
    import os
    from oracle import torch_ref as R
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.utils import checkpoint
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    spec = importlib.util.spec_from_file_location("make_flax_fixture", os.path.join(here, "make_flax_fixture.py"))
    enc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(enc)

    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    rng = np.random.default_rng(77)
    gp, gs = syn.init_generator(cfg, seed=142, bias_scale=0.05)
    dp, ds = syn.init_discriminator(cfg, seed=143, bias_scale=0.05)
    ema = syn.tree_map(lambda a: (a + 0.01 * rng.standard_normal(a.shape)).astype(np.float32), gp)
    gs = syn.tree_map(lambda a: (np.abs(a) + 0.1 * rng.random(a.shape)).astype(np.float32), gs)      # non-trivial running stats

    def adam(params):
        return syn.tree_map(lambda a: {"grad_ema": (1e-3 * rng.standard_normal(a.shape)).astype(np.float32),
                                       "grad_sq_ema": (1e-6 * rng.random(a.shape)).astype(np.float32)}, params)
    tree = {"step": 7,
            "g_optimizer": {"target": gp, "state": {"step": np.asarray(7, np.int32), "param_states": adam(gp)}},
            "d_optimizer": {"target": dp, "state": {"step": np.asarray(14, np.int32), "param_states": adam(dp)}},
            "generator_state": {"batch_stats": gs}, "discriminator_state": {"spectral_norm_stats": ds}, "ema_params": ema}
    data = enc.enc(tree)                                                       # ~25 MB, built in memory
    gen, disc, state = train_utils.create_train_state(cfg, 5)
    state = checkpoint.from_bytes(state, data)
    assert int(state.step) == 7 and state.g_optimizer.arena.opt_step == 7 and state.d_optimizer.arena.opt_step == 14
    for opt, params, key in ((state.g_optimizer, gp, "g_optimizer"), (state.d_optimizer, dp, "d_optimizer")):
        a = opt.arena
        ps = tree[key]["state"]["param_states"]
        for path, leaf in syn.tree_leaves(params):
            assert np.array_equal(a.flax_view(path).cpu().numpy(), leaf), path
            st = ps
            for k in path.split("/"):
                st = st[k]
            assert np.array_equal(a.flax_view(path, a.m).cpu().numpy(), st["grad_ema"]), path
            assert np.array_equal(a.flax_view(path, a.v).cpu().numpy(), st["grad_sq_ema"]), path
    for path, leaf in syn.tree_leaves(ema):
        assert np.array_equal(state.g_optimizer.arena.flax_view(path, state.ema_buffer).cpu().numpy(), leaf), path
    batch = syn.make_batch(cfg, per_device_batch=2)
    tb = {k: torch.as_tensor(v).cuda() for k, v in batch.items()}
    half = {k: v[:2] for k, v in tb.items()}
    img, ema_img = train_utils.eval_step(0, state, half, gen, cfg)
    rb = R.batch_to_torch({k: v[:2] for k, v in batch.items()})
    ref_state = R.make_state(gp, gs, dp, ds, torch.float32)
    ref, _, _ = R.generator(ref_state["g_params"], ref_state["generator_state"], rb, rb["z"], cfg, False)
    ref_e, _, _ = R.generator(R.make_state(ema, gs, dp, ds, torch.float32)["g_params"], ref_state["generator_state"], rb, rb["z"], cfg, False)
    err, err_e = float((img.cpu() - ref).abs().max()), float((ema_img.cpu() - ref_e).abs().max())
    print("restored (independent encoder) eval_step vs oracle: max |image error|", err, "EMA", err_e)
    assert err < 2e-3 and err_e < 2e-3
    assert float((img.cpu() - ema_img.cpu()).abs().max()) > 1e-6
