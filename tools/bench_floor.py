#!/usr/bin/env python
"""The batch-independent part of the C1 step in isolation: spectral normalisation + weight preparation of D and G, the
gradient through sigma, Adam (+ gradient zeroing) -- round-4 path (1 / sigma in the convolutions' alpha: xmc_wprep_batched,
xmc_sn_power_iter_fused, xmc_sn_batched_dot, xmc_adam_ema_dev_sn) against the round-3 path (xmc_sn_batched_power_iter / _prep /
_grad_fix, xmc_phase_conv_weight, per-site xmc_prep_conv_weight, fill + xmc_adam_ema_dev), each piece timed alone with HIP
events (interleaved rounds) and as bytes / time against the ~5 TB/s a streaming kernel reaches on this chip.
usage (GPU box): PYTHONPATH=. python tools/bench_floor.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def timed(fn, n=5):
    fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    from xmcgan_image_generation_amd import train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_c1_config()
    cfg.pretrained_image_contrastive = False
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    g, d = gen(train=True), disc(train=True)
    ops = d.ops
    dparams, sn = state.d_optimizer.target, state.discriminator_state["spectral_norm_stats"]
    da, ga = d._bind(dparams), g._bind(state.g_optimizer.target)
    sn = d.flat_sn_stats(dparams, sn)
    u0 = sn.flat
    torch.manual_seed(0)
    da.grads.normal_(0, 1e-3)
    ga.grads.normal_(0, 1e-3)
    mb = lambda a: a.size * 4 / 1e6
    print(f"D arena {mb(da):.0f} MB, G arena {mb(ga):.0f} MB (float32)")
    rows = []

    def row(name, us, mbytes):
        rows.append((name, us, mbytes))
        print(f"{name:58s} {us:8.1f} us   {mbytes:8.0f} MB   {mbytes / us if us else 0:5.2f} TB/s")
    # ---- round-4 pieces
    ops.fold_sigma = ops.fuse_opt = True
    d.prepare(dparams, sn)
    wp = d.wp
    copies = (wp["wf"] + wp["wd"] + wp["pf"] + wp["pd"]) * 2 / 1e6
    part_mb = wp["part"] * 4 / 1e6
    res = {}
    res["wprep"] = lambda: ops.wprep_run(wp, da.params, u0)
    bufs, part = ops.wprep_run(wp, da.params, u0)
    res["iter_fused"] = lambda: ops.sn_bank_power_iter_fused(d.bank, d.irr, wp, da.params, u0, part)
    u_new, v, scal = ops.sn_bank_power_iter_fused(d.bank, d.irr, wp, da.params, u0, part)
    res["dot"] = lambda: ops.sn_bank_dot(d.bank, da.params, da.grads, scal)
    kvec = ops.sn_bank_dot(d.bank, da.params, da.grads, scal)
    step = torch.zeros((4,), dtype=torch.float32, device="cuda")
    res["adam_sn_D"] = lambda: ops.adam_ema_dev_sn(da.params, da.grads, da.m, da.v, None, step, lr=0.0, beta1=0.5, beta2=0.999,
                                                   fix=(d.sn_map, d.bank, kvec, scal, u_new, v), zero_grads=False)
    res["adam_G"] = lambda: ops.adam_ema_dev_sn(ga.params, ga.grads, ga.m, ga.v, None, step, lr=0.0, beta1=0.5, beta2=0.999, zero_grads=False)
    gwp = g.wp
    res["wprep_G"] = lambda: ops.wprep_run(gwp, ga.params)
    # ---- round-3 pieces
    res["iter_legacy"] = lambda: ops.sn_bank_power_iter(d.bank, da.params, u0)
    res["prep_legacy"] = lambda: ops.sn_bank_prep(d.bank, da.params, scal, True)
    phase_sites = [s for s in d.conv_sites if s.phase]
    gkeep = da.grads.clone()
    res["fix_legacy"] = lambda: ops.sn_bank_grad_fix(d.bank, da.params, gkeep, u_new, v, scal)
    res["adam_legacy_D"] = lambda: ops.adam_ema_dev(da.params, da.grads, da.m, da.v, None, step, lr=0.0, beta1=0.5, beta2=0.999)
    res["fill_D"] = lambda: gkeep.zero_()

    def phase_legacy():
        from xmcgan_image_generation_amd.ops import PackedWeight
        for s in phase_sites:
            wf, wd = PackedWeight(None, s.cout, 9, s.cin), PackedWeight(None, s.cin, 9, s.cout)
            ops.attach_phase_weights(s.w, scal[1:2], wf, wd, s.phase)
    res["phase_weight_legacy"] = phase_legacy
    only = os.environ.get("XMC_FLOOR_ONLY")          # e.g. "iter_fused": time (or trace) that piece alone
    if only:
        for r in range(10):
            res[only]()
        torch.cuda.synchronize()
        print(only, f"{timed(res[only], 10):.1f} us")
        return
    t = {k: 0.0 for k in res}
    for r in range(3):                                # interleaved rounds
        for k, fn in res.items():
            t[k] += timed(fn) / 3
    A = mb(da)
    print("---- round 4 (per D half step; G once per step)")
    row("xmc_wprep_batched (D): read W, write copies + partials", t["wprep"], A + copies + part_mb)
    row("xmc_sn_power_iter_fused (D): read partials + W once", t["iter_fused"], A + part_mb)
    row("xmc_sn_batched_dot (D): read G, W", t["dot"], 2 * A)
    row("xmc_adam_ema_dev_sn (D, sigma term): read p g m v, write p m v (+ g)", t["adam_sn_D"], 7 * A)
    row("xmc_adam_ema_dev_sn (G)", t["adam_G"], 7 * mb(ga))
    row("xmc_wprep_batched (G)", t["wprep_G"], 0)
    new_d = t["wprep"] + t["iter_fused"] + t["dot"] + t["adam_sn_D"]
    print("---- round 3")
    row("xmc_sn_batched_power_iter (D): W twice", t["iter_legacy"], 2 * A)
    row("xmc_sn_batched_prep (D): read W, write copies", t["prep_legacy"], A + copies)
    row("xmc_phase_conv_weight x phase sites (D)", t["phase_weight_legacy"], 0)
    row("xmc_sn_batched_grad_fix (D): dot + fix", t["fix_legacy"], 5 * A)
    row("xmc_adam_ema_dev (D)", t["adam_legacy_D"], 7 * A)
    row("gradient fill (D)", t["fill_D"], A)
    old_d = t["iter_legacy"] + t["prep_legacy"] + t["phase_weight_legacy"] + t["fix_legacy"] + t["adam_legacy_D"] + t["fill_D"]
    print(f"D per half step: round 4 {new_d / 1e3:.3f} ms, round 3 {old_d / 1e3:.3f} ms; per step (two half steps) {2 * new_d / 1e3:.3f} vs {2 * old_d / 1e3:.3f} ms")


if __name__ == "__main__":
    main()
