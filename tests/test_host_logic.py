"""Host logic (explicit backward schedule, arenas, state handling) vs the oracle, on the CPU mock
operator table (tests/cpu_ops.py).  Any wiring error in the hand-written backward shows up as an
O(1) gradient mismatch here, without a GPU."""
import os
import re

import numpy as np
import pytest
import torch

from oracle import torch_ref as R
from tests.cpu_ops import CpuOps
from xmcgan_image_generation_amd import synthetic as syn
from xmcgan_image_generation_amd import train_utils, xmc_gan
from xmcgan_image_generation_amd.configs import coco_xmc
from xmcgan_image_generation_amd.nets import xmc_net


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / max(float(b.norm()), 1e-12))


@pytest.fixture(scope="module")
def stepped():
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = syn.make_batch(cfg, per_device_batch=2)
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        state = train_utils.load_flax_params(state, gp, gs, dp, ds)
        tb = {k: torch.as_tensor(v) for k, v in batch.items()}
        new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
        ref_state = R.make_state(gp, gs, dp, ds, torch.float32)
        ref_new, ref_metrics, dbg = R.train_step(ref_state, R.batch_to_torch(batch), cfg, return_debug=True)
        eval_imgs = train_utils.eval_step(0, new_state, {k: v[:2] for k, v in tb.items()}, gen, cfg)
    finally:
        xmc_net.set_ops_factory(None)
    return cfg, new_state, metrics, ref_new, ref_metrics, dbg, eval_imgs


def test_metrics_match(stepped):
    _, _, metrics, _, ref_metrics, _, _ = stepped
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        assert abs(float(metrics[k]) - float(ref_metrics[k])) <= 2e-4 * max(1.0, abs(float(ref_metrics[k]))), k


def test_gradients_match(stepped):
    _, new_state, _, _, _, dbg, _ = stepped
    for which, opt in (("d_grad", new_state.d_optimizer), ("g_grad", new_state.g_optimizer)):
        got = opt.arena.tree(opt.arena.grads)
        ref_leaves = R.leaves(dbg[which])
        # leaves whose true gradient is analytically zero (biases that only feed BatchNorms) hold
        # round-off noise on both sides: measure errors against a floor of 1e-2 x the network RMS
        rms = (sum(float(b.double().pow(2).sum()) for _, b in ref_leaves) / sum(b.numel() for _, b in ref_leaves)) ** 0.5
        worst = 0.0
        for (p1, a), (p2, b) in zip(syn.tree_leaves(got), ref_leaves):
            assert p1 == p2
            err = float((a.double() - b.double()).norm())
            r = err / max(float(b.double().norm()), 1e-2 * rms * b.numel() ** 0.5)
            worst = max(worst, r)
            assert r < 2e-3, (which, p1, r, float(a.norm()), float(b.norm()))
        print(which, "worst rel err", worst)


def test_post_step_params_and_state(stepped):
    _, new_state, _, ref_new, _, _, _ = stepped
    assert new_state.step == 1
    assert new_state.d_optimizer.state["step"] == 2 and new_state.g_optimizer.state["step"] == 1
    # Adam's bias-corrected first steps move every element by ~lr * sign(g): elements whose gradient
    # is round-off noise (see test_gradients_match) may legitimately differ by up to 2 * lr per step.
    cfg = stepped[0]
    for tree, ref, tol in ((new_state.d_optimizer.target, ref_new["d_params"], 4.2 * cfg.d_lr),
                           (new_state.g_optimizer.target, ref_new["g_params"], 2.1 * cfg.g_lr),
                           (new_state.ema_params, ref_new["ema_params"], 2.1e-3 * cfg.g_lr)):
        for (p1, a), (p2, b) in zip(syn.tree_leaves(tree), R.leaves(ref)):
            assert p1 == p2 and a.shape == b.shape
            assert float((a - b).abs().max()) <= tol + 1e-6, p1
            assert _rel(a, b) < 2e-2, p1
    for (p1, a), (p2, b) in zip(syn.tree_leaves(new_state.generator_state["batch_stats"]),
                                R.leaves(ref_new["generator_state"])):
        assert p1 == p2 and _rel(a, b) < 1e-4, p1
    got = dict(syn.tree_leaves(new_state.discriminator_state["spectral_norm_stats"]))
    for p, b in R.leaves(ref_new["discriminator_state"]):
        assert _rel(got[p], b) < 1e-4, p


def test_eval_step_images(stepped):
    cfg, new_state, _, ref_new, _, _, (img, ema_img) = stepped
    assert img.shape == (2, cfg.image_size, cfg.image_size, 3) and ema_img.shape == img.shape
    assert float(img.min()) >= 0.0 and float(img.max()) <= 1.0
    batch = syn.make_batch(cfg, per_device_batch=2)
    tb = R.batch_to_torch({k: v[:2] for k, v in batch.items()})
    ref, _, _ = R.generator(ref_new["g_params"], ref_new["generator_state"], tb, tb["z"], cfg, False)
    assert float((img - ref).abs().max()) < 2e-3
    ref_e, _, _ = R.generator(ref_new["ema_params"], ref_new["generator_state"], tb, tb["z"], cfg, False)
    assert float((ema_img - ref_e).abs().max()) < 2e-3


def test_flax_style_apply_surface():
    """Generator/Discriminator keep the reference's init/apply surface and tree names."""
    cfg = coco_xmc.get_test_config()
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        g = xmc_net.Generator(cfg, train=False)
        gv = g.init(0, None)
        assert set(gv) == {"params", "batch_stats"}
        assert gv["params"]["GenBlock_1"]["Conv_0"]["kernel"].shape == (3, 3, 256, 128)
        assert "LocalConditionalBatchNorm_0" in gv["params"] and "Conv_1" in gv["params"]
        d = xmc_net.Discriminator(cfg, train=False)
        dv = d.init(1, None)
        assert set(dv) == {"params", "spectral_norm_stats"}
        assert dv["spectral_norm_stats"]["DiscBlock_4"]["SpectralConv_1"]["u0"].shape == (1, 256)
        assert "SpectralConv_2" not in dv["params"]["DiscBlock_4"]
        batch = {k: torch.as_tensor(v) for k, v in syn.make_batch(cfg, per_device_batch=1).items()}
        img, new = g.apply(gv, (batch, batch["z"]), mutable=["batch_stats"])
        assert img.shape == (2, 128, 128, 3) and "batch_stats" in new
        (logit, stats), newd = d.apply(dv, (torch.cat([batch["image"], img]), batch), mutable=["spectral_norm_stats"])
        assert logit.shape == (4, 1) and len(stats) == 15
    finally:
        xmc_net.set_ops_factory(None)


def test_apply_statistics_match_numpy_spec():
    """The 15-key statistic_dict of Discriminator.apply (losses, accuracies, entropies) vs oracle/np_spec."""
    from oracle import np_spec as S
    cfg = coco_xmc.get_test_config()
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        batch = syn.make_batch(cfg, per_device_batch=2)
        half = {k: v[:2] for k, v in batch.items()}
        f64 = lambda t: syn.tree_map(lambda a: a.astype(np.float64), t)
        img, _ = S.generator(f64(gp), f64(gs), half, half["z"], cfg, True)
        allimg = np.concatenate([half["image"].astype(np.float64), img])
        (logit, ref), new_sn = S.discriminator(f64(dp), f64(ds), allimg, half, cfg)
        d = xmc_net.Discriminator(cfg, train=True)
        dv = d.init(1, None)
        dv["params"].arena.load_flax(dp)
        dv["spectral_norm_stats"] = xmc_net._tree_to_dev(d.ops, ds)
        tb = {k: torch.as_tensor(v) for k, v in half.items()}
        (got_logit, got), new_vars = d.apply(dv, (torch.as_tensor(allimg, dtype=torch.float32), tb),
                                             mutable=["spectral_norm_stats"])
        assert set(got) == set(ref)
        for k, v in ref.items():
            assert abs(float(got[k]) - float(v)) <= 2e-3 * max(1.0, abs(float(v))), (k, float(got[k]), float(v))
        assert np.allclose(got_logit.numpy(), logit, rtol=1e-3, atol=1e-3)
        for path, u in syn.tree_leaves(new_sn):
            got_u = dict(syn.tree_leaves(new_vars["spectral_norm_stats"]))[path]
            assert np.allclose(got_u.numpy(), u, rtol=1e-4, atol=1e-6), path
    finally:
        xmc_net.set_ops_factory(None)


@pytest.mark.parametrize("flags", [dict(word_contrastive=False), dict(sentence_contrastive=False),
                                   dict(image_contrastive=False),
                                   dict(word_contrastive=False, sentence_contrastive=False, image_contrastive=False)])
def test_contrastive_head_flags_are_honoured(flags):
    """word_contrastive / sentence_contrastive / image_contrastive (reference xmc_net.py:105-125): a disabled head
    contributes 0 to the losses, receives no gradient, and the x_cond 1x1 conv does not exist without the word head."""
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    cfg.update(flags)
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
        dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
        assert ("SpectralConv_0" in dp) == cfg.word_contrastive
        batch = syn.make_batch(cfg, per_device_batch=2)
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        state = train_utils.load_flax_params(state, gp, gs, dp, ds)
        tb = {k: torch.as_tensor(v) for k, v in batch.items()}
        new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
    finally:
        xmc_net.set_ops_factory(None)
    ref_new, ref_metrics, dbg = R.train_step(R.make_state(gp, gs, dp, ds, torch.float32), R.batch_to_torch(batch), cfg,
                                             return_debug=True)
    for k in ("d_loss", "g_loss", "c_loss_d", "c_loss_g"):
        assert abs(float(metrics[k]) - float(ref_metrics[k])) <= 2e-4 * max(1.0, abs(float(ref_metrics[k]))), k
    for which, opt in (("d_grad", new_state.d_optimizer), ("g_grad", new_state.g_optimizer)):
        got = torch.cat([a.reshape(-1) for _, a in syn.tree_leaves(opt.arena.tree(opt.arena.grads))])
        ref = torch.cat([b.reshape(-1) for _, b in R.leaves(dbg[which])])
        assert _rel(got, ref) < 2e-3, (which, _rel(got, ref))


@pytest.mark.parametrize("bad", [dict(g_spectral_norm=True), dict(d_spectral_norm=False), dict(batch_norm_group_size=2),
                                 dict(image_size=64), dict(architecture="dcgan")])
def test_unsupported_config_values_are_rejected(bad):
    cfg = coco_xmc.get_test_config()
    cfg.update(bad)
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        with pytest.raises(ValueError):
            train_utils.create_train_state(cfg, 0)
    finally:
        xmc_net.set_ops_factory(None)


def test_split_input_dict_rejects_uneven_batches_and_z_fallback():
    with pytest.raises(ValueError):
        train_utils.split_input_dict({"a": torch.zeros(5, 3)}, 2)
    parts = train_utils.split_input_dict({"a": torch.arange(8).view(4, 2)}, 2)
    assert parts[0]["a"].data_ptr() == parts[1]["a"].data_ptr() - 4 * 8       # zero-copy views
    # a batch without "z": drawn from rng like the reference (xmc_gan.py:132-136,225-229)
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        batch = {k: torch.as_tensor(v) for k, v in syn.make_batch(cfg, per_device_batch=2).items() if k != "z"}
        _, m1 = train_utils.train_step(7, state, batch, xmc_gan, gen, disc, cfg, {})
        assert all(np.isfinite(float(v)) for v in m1.values())
    finally:
        xmc_net.set_ops_factory(None)


def test_train_step_gives_each_half_step_its_own_rng():
    """reference train_utils.py:121 splits the key per half step; a z-less batch must not draw the same noise twice"""
    import types
    from xmcgan_image_generation_amd import train_utils
    from xmcgan_image_generation_amd.configs import coco_xmc
    seen = []
    fake = types.SimpleNamespace(
        train_d=lambda rng, state, batch, g, d, cfg, **kw: (seen.append(("d", rng)), state)[1],
        train_g_d=lambda rng, state, batch, g, d, cfg, add, **kw: (seen.append(("g", rng)), (state, {}))[1])
    cfg = coco_xmc.get_test_config()
    batch = {"sentence_embedding": torch.zeros((4, 768))}
    train_utils.train_step(5, None, batch, fake, None, None, cfg, {})
    train_utils.train_step(6, None, batch, fake, None, None, cfg, {})
    rngs = [r for _, r in seen]
    assert [k for k, _ in seen] == ["d", "g", "d", "g"] and len(set(rngs)) == 4, seen


def test_generator_forward_draws_z_from_its_rng_on_the_ops_device():
    from tests.cpu_ops import CpuOps
    from xmcgan_image_generation_amd import synthetic as syn
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    from xmcgan_image_generation_amd.nets import xmc_net
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        cfg = coco_xmc.get_test_config()
        gen, disc, state = train_utils.create_train_state(cfg, 0)
        b = {k: torch.as_tensor(v) for k, v in syn.make_batch(cfg, per_device_batch=1).items() if k != "z"}
        half = {k: v[:2] for k, v in b.items()}
        img = [xmc_gan._generator_forward(r, cfg, state, half, gen(train=True), False)[0] for r in (3, 3, 4)]
        assert torch.equal(img[0], img[1]) and not torch.equal(img[0], img[2])
        with pytest.raises(ValueError, match="'z'"):
            train_utils.GraphedTrainStep(state, b, xmc_gan, gen, disc, cfg, {})
    finally:
        xmc_net.set_ops_factory(None)


def test_gradsync_bf16_transport_single_process_gloo():
    """optional bf16 gradient transport: the arena comes back as the bf16-rounded sum (world 1: the rounded gradients)"""
    import os
    import torch.distributed as dist
    from xmcgan_image_generation_amd.dp import GradSync
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581")
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        g = torch.randn(1000)
        want = g.to(torch.bfloat16).float()
        sync = GradSync(bucket_elems=256, transport="bf16")
        assert sync.all_reduce(g[:600], "g") == 1.0
        sync.all_reduce(g[600:], "g", append=True)
        sync.wait("g")
        assert torch.equal(g, want)
        g32 = torch.randn(100)
        keep = g32.clone()
        s32 = GradSync(transport="float32")
        s32.all_reduce(g32, "d")
        s32.wait("d")
        assert torch.equal(g32, keep)
        with pytest.raises(ValueError):
            GradSync(transport="fp8")
    finally:
        dist.destroy_process_group()


def test_prefetched_generator_forward_is_matched_by_object_identity_not_by_address():
    """the prefetched generator forward is keyed on the batch's conditioning OBJECTS and their in-place version
    counters -- a different batch that happens to sit at the same address (the caching allocator recycles them) or a batch
    modified in place does not match."""
    a = {k: torch.zeros((4, 3)) for k in ("sentence_embedding", "embedding", "max_len", "z")}
    ident = xmc_gan._batch_identity(a)
    assert xmc_gan._same_batch(ident, a)
    b = {k: v.clone() for k, v in a.items()}                 # equal values, other objects
    assert not xmc_gan._same_batch(ident, b)
    views = {k: v.view_as(v) for k, v in a.items()}          # same storage (same data_ptr), other tensor objects
    assert all(views[k].data_ptr() == a[k].data_ptr() for k in a) and not xmc_gan._same_batch(ident, views)
    a["z"].add_(1.0)                                         # modified in place after the prefetch
    assert not xmc_gan._same_batch(ident, a)
    c = {k: v for k, v in a.items() if k != "z"}             # a z-less batch is another batch
    assert not xmc_gan._same_batch(xmc_gan._batch_identity(a), c)


def test_discriminator_exchange_is_bucketed_only_when_the_sigma_term_follows_it():
    """xmc_gan._d_bucketer: slices of D's gradient arena go to the all-reduce from inside the backward pass only when the
    gradient through sigma is applied AFTER the exchange (fix_args given); the slices it issues are appended to one tag."""
    class Sync:
        world = 2

        def __init__(self):
            self.calls = []

        def all_reduce(self, t, tag, append=False):
            self.calls.append((t.numel(), tag, append))
            return 0.5

    class Arena:
        grads = torch.zeros((100,))
    kw, sent = xmc_gan._d_bucketer(None, Arena(), ("fix",))
    assert kw == {} and not sent()
    kw, sent = xmc_gan._d_bucketer(Sync(), Arena(), None)
    assert kw == {} and not sent()
    s = Sync()
    kw, sent = xmc_gan._d_bucketer(s, Arena(), ("fix",))
    kw["on_ready"](60, 100)
    kw["on_ready"](20, 60)
    kw["on_ready"](0, 20)
    kw["on_ready"](100, 100)                                 # empty slice: nothing issued
    assert sent() and s.calls == [(40, "d", False), (40, "d", True), (20, "d", True)]


def test_discriminator_arena_follows_the_tree_order_the_bucketed_exchange_assumes():
    """round-4 advisor finding: backward_d's on_ready slices assume the arena layout [DiscOptimizedBlock_0, DiscBlock_0 .. n,
    SpectralDense_0, SpectralDense_1, SpectralConv_0]; the discriminator checks it at build time (``bucket_order_ok``) and falls
    back to one exchange at the end otherwise"""
    from xmcgan_image_generation_amd.libml.layers import ParamArena
    cfg = coco_xmc.get_test_config()
    xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
    try:
        d = xmc_net.Discriminator(cfg, train=True)
        dv = d.init(1, None)
        d._bind(dv["params"])
        assert d.bucket_order_ok
        # a tree in another order (alphabetical puts SpectralConv_0 in front of the dense heads): detected
        shapes = d.shapes()[0]
        resorted = {k: shapes[k] for k in sorted(shapes)}
        arena = ParamArena(d.ops, resorted, with_opt=False)
        d._build(arena)
        assert not d.bucket_order_ok
    finally:
        xmc_net.set_ops_factory(None)


def test_global_cbn_projections_fused_layout_and_unfused_path(monkeypatch):
    """round 5 (nets/common.py FusedGlobalGB): the four global conditional-BatchNorm sites' merged gamma | beta Dense kernels are
    stored transposed, back to back, in FRONT of every other tensor (their fused gradient is the last thing G's backward pass
    produces: the last slice of the replicas' exchange), the Flax-named members are transposed slice views of them, and the
    per-site path (``XMC_GLOBAL_GB_FUSED=0``: DenseSite on the transposed storage) computes the same step."""
    from xmcgan_image_generation_amd.libml.layers import ParamArena
    cfg = coco_xmc.get_test_config()
    cfg.batch_size = 2
    ops = CpuOps(torch.float32)
    shapes = syn.generator_shapes(cfg)[0]
    arena = ParamArena(ops, shapes, with_opt=False)
    sites = [f"GenBlock_{i}/ConditionalBatchNorm_{j}/GB" for i in range(2) for j in range(2)]
    ko = [arena.offset(s + "/kernel") for s in sites]
    sz = [int(np.prod(arena.merged[s + "/kernel"][1])) for s in sites]
    assert ko[0] == 0 and all(ko[i] + sz[i] == ko[i + 1] for i in range(3))                   # head of the arena, contiguous
    bo = [arena.offset(s + "/bias") for s in sites]
    assert bo[0] == ko[3] + sz[3] and all(bo[i] + arena.merged[sites[i] + "/bias"][1][0] == bo[i + 1] for i in range(3))
    assert arena.prefix_offset("GenBlock_1") > bo[3]                                          # the cut of the exchange ignores them
    assert arena.prefix_offset("GenBlock_0") > bo[3] and arena.prefix_offset("Dense_0") > bo[3]
    gp, _ = syn.init_generator(cfg, seed=5, bias_scale=0.05)
    arena.load_flax(gp)
    for (p1, a), (p2, b) in zip(syn.tree_leaves(arena.tree()), syn.tree_leaves(gp)):            # Flax-layout round trip
        assert p1 == p2 and torch.equal(a, torch.as_tensor(np.asarray(b), dtype=torch.float32)), p1
    k0 = arena.view("GenBlock_0/ConditionalBatchNorm_0/Dense_0/kernel")
    mk = arena.view("GenBlock_0/ConditionalBatchNorm_0/GB/kernel")
    assert mk.shape == (2 * k0.shape[1], k0.shape[0]) and torch.equal(mk[:k0.shape[1]].t(), k0)

    def step(fused):
        monkeypatch.setenv("XMC_GLOBAL_GB_FUSED", "1" if fused else "0")
        xmc_net.set_ops_factory(lambda dtype: CpuOps(dtype))
        try:
            gp, gs = syn.init_generator(cfg, seed=42, bias_scale=0.05)
            dp, ds = syn.init_discriminator(cfg, seed=43, bias_scale=0.05)
            gen, disc, state = train_utils.create_train_state(cfg, 0)
            state = train_utils.load_flax_params(state, gp, gs, dp, ds)
            tb = {k: torch.as_tensor(v) for k, v in syn.make_batch(cfg, per_device_batch=2).items()}
            new_state, metrics = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, {})
            assert gen(train=True).global_gb.ok == fused
            return new_state, metrics
        finally:
            xmc_net.set_ops_factory(None)
    (sa, ma), (sb, mb) = step(True), step(False)
    for k in ("d_loss", "g_loss"):
        assert abs(float(ma[k]) - float(mb[k])) <= 1e-5 * max(1.0, abs(float(mb[k])))
    ga, gb = sa.g_optimizer.arena, sb.g_optimizer.arena
    for (p1, a), (p2, b) in zip(syn.tree_leaves(ga.tree(ga.grads)), syn.tree_leaves(gb.tree(gb.grads))):
        assert p1 == p2 and _rel(a, b) < 2e-4, (p1, _rel(a, b))


def test_graph_family_time_tool_on_a_synthetic_trace(tmp_path):
    """tools/graph_family_time.py (bench.py's roofline.in_replayed_graph): families by kernel name, per-step division, and the union of
    the kernel intervals as GPU-busy wall time (overlapping kernels on two streams are not counted twice)"""
    import json
    import sqlite3
    import subprocess
    import sys
    db = str(tmp_path / "t.db")
    con = sqlite3.connect(db)
    con.execute("create table kernels (name text, start integer, end integer)")
    rows = [("void (anonymous namespace)::conv_stream_kernel<3, 2, 4, 2, 4>(SArgs)", 0, 1_000_000),
            ("void (anonymous namespace)::conv_wgrad_dma_kernel<3, 2, 18, 1, false, 3>(WArgs)", 500_000, 2_000_000),     # overlaps the first
            ("void (anonymous namespace)::conv_pw_kernel<32, 3, 128, true>(SArgs)", 3_000_000, 3_500_000),
            ("void (anonymous namespace)::stem_conv_kernel(...)", 3_500_000, 3_600_000),
            ("void at::native::vectorized_elementwise_kernel<4, at::native::FillFunctor<float>>(...)", 4_000_000, 4_010_000),
            ("some_unknown_kernel(int)", 5_000_000, 5_100_000)]
    con.executemany("insert into kernels values (?, ?, ?)", rows)
    con.commit()
    con.close()
    out = str(tmp_path / "fam.json")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run([sys.executable, os.path.join(root, "tools", "graph_family_time.py"), db, "2", out], check=True, capture_output=True)
    d = json.load(open(out))
    f = d["families"]
    assert f["conv3x3_fwd_dgrad"] == {"launches_per_step": 0.5, "ms_per_step": 0.5}
    assert f["wgrad"]["ms_per_step"] == 0.75 and f["conv_pointwise_1x1"] == {"launches_per_step": 1.0, "ms_per_step": 0.3}
    assert f["torch"]["launches_per_step"] == 0.5 and f["other"]["launches_per_step"] == 0.5
    assert abs(d["sum_kernel_ms_per_step"] - (1.0 + 1.5 + 0.5 + 0.1 + 0.01 + 0.1) / 2) < 2e-3     # (each family is rounded to 1 us)
    assert abs(d["gpu_busy_wall_ms_per_step"] - (2.0 + 0.6 + 0.01 + 0.1) / 2) < 1e-3   # [0, 2] and [3, 3.6] ms merged


def test_bench_reads_the_newest_committed_profile_summaries():
    """bench.py attaches two NOT-measured-in-this-run objects from profiles/: the PMC traffic of the dominant kernels and the per-family
    kernel time inside the replayed graph -- each names its source file, newest round first"""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(root, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    fam = b._in_graph_families()
    assert fam is not None and fam["source"].startswith("profiles/r") and "conv3x3_fwd_dgrad" in fam["families"]
    traffic, src = b._pmc_traffic(("conv_stream_kernel", "conv_phase4_kernel", "conv_phase_kernel"))
    assert traffic and traffic > 1e7 and src.startswith("profiles/r") and src >= "profiles/r06"
