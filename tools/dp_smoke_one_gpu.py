#!/usr/bin/env python
"""Two data-parallel ranks on ONE GPU over gloo (RCCL refuses two ranks per device): exercises dp.GradSync and the
deferred discriminator update with the HIP backend.  Ranks must end bit-identical on parameters.
DP_DTYPE=float32 DP_DUMP=<dir>: one float32 step from the synthetic init (seeds 42 / 43) on per-rank batches; rank 0
writes its post-step parameter trees and metrics to <dir>/dp_rank0.pt for tests/test_gpu_dp.py, which compares them
with the averaged-oracle emulation (lax.pmean, reference xmc_gan.py:170-171,251).
usage: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29544 tools/dp_smoke_one_gpu.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from xmcgan_image_generation_amd import dp, synthetic, train_utils, xmc_gan  # noqa: E402
from xmcgan_image_generation_amd.configs import coco_xmc  # noqa: E402


def main():
    backend = os.environ.get("DP_BACKEND", "gloo")           # "nccl" (= RCCL) needs one GPU per rank: world size 1 here
    torch.cuda.set_device(0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    cfg = coco_xmc.get_test_config()
    cfg.dtype = os.environ.get("DP_DTYPE", "bfloat16")
    cfg.batch_size = 2
    dump = os.environ.get("DP_DUMP")
    additional = {}
    if os.environ.get("DP_PRETRAINED", "1") != "0":          # the reference default: frozen ResNet-50 term in g_loss
        from xmcgan_image_generation_amd.utils import pretrained_model_utils, resnet_v1
        cfg.pretrained_image_contrastive = True
        rp, rs = resnet_v1.init_resnet50(7, head_scale=0.2)
        st = {"params": rp, "batch_stats": rs}
        additional = {"image_model": pretrained_model_utils.ImageModel(st), "image_model_state": st}
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    if dump:             # the initial state the oracle emulation starts from (non-zero biases)
        gp, gs = synthetic.init_generator(cfg, seed=42, bias_scale=0.05)
        dp_, ds = synthetic.init_discriminator(cfg, seed=43, bias_scale=0.05)
        state = train_utils.load_flax_params(state, gp, gs, dp_, ds)
    sync = dp.GradSync()
    for step in range(1 if dump else 2):
        kw = {} if dump else {"seed": 100 + step}
        batch = {k: torch.as_tensor(v).cuda() for k, v in
                 synthetic.make_batch(cfg, per_device_batch=2, rank=rank, **kw).items()}
        state, metrics = train_utils.train_step(step, state, batch, xmc_gan, gen, disc, cfg, additional, grad_sync=sync)
    torch.cuda.synchronize()
    for name, a in (("g", state.g_optimizer.arena.params), ("d", state.d_optimizer.arena.params)):
        mine = a.detach().clone() if backend == "nccl" else a.detach().cpu()     # RCCL gathers device tensors only
        others = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(others, mine)
        same = all(torch.equal(o, others[0]) for o in others)
        if rank == 0:
            print(f"{name}: finite={bool(torch.isfinite(mine).all())} identical_across_ranks={same}")
        assert same and bool(torch.isfinite(mine).all())
    if rank == 0 and dump:
        torch.save(dict(g_tree={p: t.detach().cpu().clone() for p, t in synthetic.tree_leaves(state.g_optimizer.target)},
                        d_tree={p: t.detach().cpu().clone() for p, t in synthetic.tree_leaves(state.d_optimizer.target)},
                        metrics={k: float(v) for k, v in metrics.items()}), os.path.join(dump, "dp_rank0.pt"))
    if rank == 0:
        assert not additional or float(metrics["c_loss_g_pretrained"]) > 0.0
        print("dp smoke OK", {k: round(float(v), 4) for k, v in metrics.items()})
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
