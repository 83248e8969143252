"""Single-process emulation of two data-parallel replicas with the ORACLE (test infrastructure): each replica's
oracle gradients are averaged (lax.pmean, reference xmc_gan.py:170-171,251) and the mean is applied on both.
Shared by tests/test_dist_gloo.py (CPU mock operator table) and tests/test_gpu_dp.py (HIP backend)."""


def reference_two_replicas(cfg, gp, gs, dp_, ds, batches, resnet=None):
    """batches[r] = numpy batch dict of replica r (leading dim 2 * per-device batch) -> ([state_r], [metrics_r], {"d": mean D gradient, "g": mean G gradient} of train_g_d)"""
    from oracle import torch_ref as R
    state = [R.make_state(gp, gs, dp_, ds, resnet=resnet) for _ in range(2)]
    halves = [R._split(R.batch_to_torch(batches[r]), 2) for r in range(2)]
    avg = lambda a, b: R.tree_map(lambda x, y: 0.5 * (x + y), a, b)
    # train_d
    g = [R.train_d(state[r], halves[r][0], cfg)[1]["d_grad"] for r in range(2)]
    mean_d = avg(g[0], g[1])
    state = [R.train_d(state[r], halves[r][0], cfg, grad_hook=lambda tag, _g: mean_d)[0] for r in range(2)]
    # train_g_d
    dbg = [R.train_g_d(state[r], halves[r][1], cfg)[2] for r in range(2)]
    mean = {"d": avg(dbg[0]["d_grad"], dbg[1]["d_grad"]), "g": avg(dbg[0]["g_grad"], dbg[1]["g_grad"])}
    out = [R.train_g_d(state[r], halves[r][1], cfg, grad_hook=lambda tag, _g: mean[tag]) for r in range(2)]
    return [o[0] for o in out], [o[1] for o in out], mean
