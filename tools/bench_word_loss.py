"""word_loss at the benchmarked shape (B = 56, R = 256, T = 17, E = 768, bf16): the fused path (csrc/word_loss_fused.hip)
against the GEMM + column-kernel path, forward and backward, and each fused launch on its own.
    python tools/bench_word_loss.py [--batch 56] [--iters 30]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def timed(fn, iters, warm=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(iters):
        fn()
    ev[1].record()
    torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) * 1e3 / iters          # us


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=56)
    ap.add_argument("--iters", type=int, default=30)
    a = ap.parse_args()
    from xmcgan_image_generation_amd.libml import attention_lib as A
    from xmcgan_image_generation_amd.ops import HipOps
    ops = HipOps(dtype=torch.bfloat16, stream_conv=False)
    b, r, t, e = a.batch, 256, 17, 768
    g = torch.Generator().manual_seed(0)
    feat = torch.randn((b, r, e), generator=g).to(torch.bfloat16).cuda()
    words = torch.randn((b, t, e), generator=g).cuda()
    ml = torch.randint(3, t + 1, (b, 1), generator=g).float().cuda()
    wn = A.normalize_words(ops, words)
    loss = torch.zeros(1, device="cuda")
    res = {}
    for fused in (False, True, False, True):
        ops.wl_fused = fused
        tape = A.word_loss_fwd(ops, feat, wn, ml, loss)
        f = timed(lambda: A.word_loss_fwd(ops, feat, wn, ml, loss), a.iters)
        bw = timed(lambda: A.word_loss_bwd(ops, tape), a.iters)
        res.setdefault(fused, []).append((f, bw))
        print(f"{'fused' if fused else 'gemm '} path: forward {f:7.1f} us   backward {bw:7.1f} us", flush=True)
    # each fused launch on its own
    ops.wl_fused = True
    w, wt = ops.wl_prep_words(wn)
    rn, rnt, rinv = ops.wl_prep_regions(feat)
    gm = ops.wl_tn_gemm(rn, rn, e, r, r, b, torch.bfloat16)
    nn, q = ops.wl_cols_fwd(rn, w, gm, ml.view(-1), t, 5.0)
    sim_t, pi = ops.wl_rows(nn, q, ml.view(-1), b, t, 5.0, 50.0)
    dsim = ops.xent_sym(sim_t, 1.0, loss, True, None)
    ds, a_s, al = ops.wl_cols_bwd(rn, w, gm, ml.view(-1), dsim, pi, t, 5.0, 50.0)
    ldp = w.shape[0]
    dg2 = ops.wl_tn_gemm(a_s, al, ldp, r, r, b, torch.bfloat16, alpha=2.0)
    drn = ops.wl_tn_gemm(ds, wt, ldp, r, e, b, torch.float32, x1=dg2, y1=rnt, k1=r, y0_shared=True)
    gf = lambda flop, us: f"{flop / us * 1e-6:7.1f} TF/s"
    rows = [
        ("wl_prep_words", lambda: ops.wl_prep_words(wn), None),
        ("wl_prep_regions", lambda: ops.wl_prep_regions(feat), None),
        ("wl_tn_gemm  G = R^ R^^T", lambda: ops.wl_tn_gemm(rn, rn, e, r, r, b, torch.bfloat16), 2.0 * b * r * r * e),
        ("wl_cols_fwd", lambda: ops.wl_cols_fwd(rn, w, gm, ml.view(-1), t, 5.0), 2.0 * b * r * ldp * (e + r)),
        ("wl_rows", lambda: ops.wl_rows(nn, q, ml.view(-1), b, t, 5.0, 50.0), None),
        ("xent_sym", lambda: ops.xent_sym(sim_t, 1.0, loss, True, None), None),
        ("wl_cols_bwd", lambda: ops.wl_cols_bwd(rn, w, gm, ml.view(-1), dsim, pi, t, 5.0, 50.0), 2.0 * b * r * ldp * (e + r)),
        ("wl_tn_gemm  dG = 2 (alpha dq) alpha^T", lambda: ops.wl_tn_gemm(a_s, al, ldp, r, r, b, torch.bfloat16, alpha=2.0),
         2.0 * b * r * r * ldp),
        ("wl_tn_gemm  dR^ = [dS | dG] [W^T | R^T]^T", lambda: ops.wl_tn_gemm(ds, wt, ldp, r, e, b, torch.float32, x1=dg2, y1=rnt,
                                                                              k1=r, y0_shared=True), 2.0 * b * r * e * (ldp + r)),
        ("l2norm_bwd_bf16y", lambda: ops.l2norm_bwd_bf16y(drn.view(b * r, e), rn.view(b * r, e), rinv, torch.bfloat16), None),
    ]
    for name, fn, flop in rows:
        us = timed(fn, a.iters)
        print(f"  {name:44s} {us:7.1f} us   {gf(flop, us) if flop else ''}", flush=True)


if __name__ == "__main__":
    main()
