#!/usr/bin/env python
"""What do RCCL's copy kernels cost the step when they share the chip with it?  ("bound RCCL's CU
contention by timing the step with a copy kernel pinned to 8 / 16 / 32 CUs on the side stream")

A stand-in for the collective's kernels -- `xmc_load_path_probe`, persistent workgroups that stream memory through registers
(what an all-reduce's send / receive loops do on a 1-GPU box that has no peer) -- runs on a side stream with k workgroups
for the whole duration of several replays of the captured C1 training step; the step's time is measured with events on its
own stream.  k = 0 is the undisturbed step.  The slowdown per k goes into tools/model_scaling.py as the contention penalty
(RCCL on xGMI rings typically runs 8-32 channels = workgroups of 256-512 threads).
usage (GPU box): PYTHONPATH=. python tools/cu_contention.py [--ks 0,8,16,32,64]"""
import argparse
import ctypes as C
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ks", default="0,1,2,4,8,16,32")
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--pretrained", default="on")
    ap.add_argument("--kind", default="copy", choices=["copy", "mfma"], help="copy: streams memory through registers; mfma: matrix "
                    "instructions on registers only (no memory traffic at all)")
    ap.add_argument("--policy", type=int, default=0, help="cache policy of the stand-in's loads: 0 default, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt")
    ap.add_argument("--buf-mib", type=float, default=256.0, help="bytes the stand-in streams over, cyclically (256 MiB = the Infinity Cache's size)")
    a = ap.parse_args()
    from xmcgan_image_generation_amd import _lib, synthetic as syn, train_utils, xmc_gan
    from xmcgan_image_generation_amd.configs import coco_xmc
    cfg = coco_xmc.get_c1_config()
    cfg.pretrained_image_contrastive = a.pretrained == "on"
    ad = {}
    if cfg.pretrained_image_contrastive:
        from xmcgan_image_generation_amd.utils import pretrained_model_utils, resnet_v1
        rp, rs = resnet_v1.init_resnet50(seed=7, head_scale=0.05)
        st = {"params": rp, "batch_stats": rs}
        ad = {"image_model": pretrained_model_utils.ImageModel(st), "image_model_state": st}
    gen, disc, state = train_utils.create_train_state(cfg, 0)
    tb = {k: torch.as_tensor(v).cuda() for k, v in syn.make_batch(cfg, per_device_batch=cfg.batch_size).items()}
    state, _ = train_utils.train_step(0, state, tb, xmc_gan, gen, disc, cfg, ad)
    torch.cuda.synchronize()
    graphed = train_utils.GraphedTrainStep(state, tb, xmc_gan, gen, disc, cfg, ad)
    state = graphed.state
    for _ in range(3):
        state, _ = graphed(state)
    torch.cuda.synchronize()
    lib = _lib.load_probe()
    src = torch.empty((int(a.buf_mib * (1 << 20)),), dtype=torch.uint8, device="cuda").random_(0, 255)      # 256 MiB: streams from HBM / MALL, not L2
    out = torch.zeros((1 << 16,), dtype=torch.float32, device="cuda")
    side = torch.cuda.Stream()

    def probe(k, iters):
        if a.kind == "mfma":
            _lib.check(lib.xmc_mfma_rate_probe(0, k, iters, C.c_void_p(out.data_ptr()), C.c_void_p(side.cuda_stream)), "xmc_mfma_rate_probe")
            return
        _lib.check(lib.xmc_load_path_probe(0 | (3 << 4) | (a.policy << 8), k, iters, C.c_void_p(src.data_ptr()), src.numel(), C.c_void_p(out.data_ptr()),
                                           C.c_void_p(side.cuda_stream)), "xmc_load_path_probe")
    # calibrate the probe: time per iteration of one workgroup set
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with torch.cuda.stream(side):
        e0.record(side); probe(16, 20000); e1.record(side)
    torch.cuda.synchronize()
    per_iter_ms = e0.elapsed_time(e1) / 20000
    gbs = 16 * 24576 / (per_iter_ms * 1e-3) / 1e9
    print(f"stand-in kind {a.kind}, load policy {a.policy}, buffer {a.buf_mib:g} MiB, {per_iter_ms * 1e3:.2f} us per iteration")
    print(f"stand-in copy kernel: 16 workgroups move {gbs:.0f} GB/s ({gbs / 16:.1f} GB/s per workgroup)")
    res = {}
    base = None
    for k in [int(v) for v in a.ks.split(",")]:
        ts = []
        for rep in range(3):
            torch.cuda.synchronize()
            s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if k > 0:
                need_ms = (a.steps + 2) * 40.0
                probe(k, int(need_ms / per_iter_ms))
            s0.record()
            for _ in range(a.steps):
                state, _ = graphed(state)
            s1.record()
            torch.cuda.synchronize()
            ts.append(s0.elapsed_time(s1) / a.steps)
        t = sorted(ts)[1]
        base = t if k == 0 else base
        res[k] = t
        print(f"copy kernel on {k:3d} workgroups (~{k * gbs / 16:5.0f} GB/s of reads): step {t:7.3f} ms" + (f"  +{100 * (t / base - 1):.1f} %" if base and k else ""))
    print(json.dumps({"what": "C1 step (graph replay) beside a k-workgroup streaming kernel on a second stream", "ms_per_step": res,
                      "copy_GBps_per_workgroup": gbs / 16}))


if __name__ == "__main__":
    main()
