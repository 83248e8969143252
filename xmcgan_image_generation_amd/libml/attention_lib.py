"""Attention and contrastive losses of XMC-GAN on the HIP operator table (forward + explicit
backward).  Mirrors ``xmcgan/libml/attention_lib.py`` of the reference:

* ``attention_for_g``  -- :194-219, one fused kernel per direction (wave per region);
* ``contrastive_loss`` -- :46-79, l2-normalise + one small fp32 GEMM + fused symmetric CE;
* ``word_loss``        -- :130-191.  The reference loops ``attention()`` (:105-127) over all
  (caption i, image j) pairs and materialises a (B, B, T, E) context tensor.  Here the same
  quantity is computed from three GEMMs and column reductions (DESIGN.md "word_loss"):

      S[(j,r),(i,t)] = <R^_j[r], W^_i[t]>              (B*R x B*T x E GEMM)
      G_j            = R^_j R^_j^T                      (B batched R x R x E GEMM)
      alpha          = softmax_r(gamma1 * S + mask)
      cos(word, ctx) = (sum_r alpha_r S_r) / sqrt(alpha^T G_j alpha)

  because ctx = sum_r alpha_r R^_j[r], <W_i[t], ctx> = |W_i[t]| sum_r alpha_r S_r and
  |ctx|^2 = alpha^T G_j alpha -- the word norm cancels in the cosine.  Word embeddings are
  data (no gradient); only region features receive gradients.
"""
from __future__ import annotations

import torch

LARGE_NUM = 1e9


def normalize_words(ops, words):
    """l2_normalize(word_feat, -1) (attention_lib.py:30-33,118,209) -> float32 (B, T, E)."""
    b, t, e = words.shape
    wn, _ = ops.l2norm_fwd(words.reshape(b * t, e).contiguous())
    return wn.view(b, t, e)


# ------------------------------------------------------------------------------ attention_for_g
def attention_for_g_fwd(ops, region, words_n, max_len, gamma, ctx_out=None):
    """region (B, R, E) activation dtype -> (ctx (B, R, E), tape).  ``ctx_out``: see ops.attn_g_fwd"""
    kw = {"ctx_out": ctx_out} if ctx_out is not None else {}
    ctx, attn, rinv = ops.attn_g_fwd(region, words_n, max_len.reshape(-1).contiguous(), gamma, **kw)
    return ctx, (region, words_n, attn, rinv, gamma)


def attention_for_g_bwd(ops, tape, dctx):
    region, words_n, attn, rinv, gamma = tape
    return ops.attn_g_bwd(dctx, region, words_n, attn, rinv, gamma)


# ----------------------------------------------------------------------------- contrastive_loss
def contrastive_loss_fwd(ops, a, b, loss_acc, temperature=0.1, want_grad=True, stats=None):
    """a, b: (B, D) float32.  Adds the loss into ``loss_acc`` (1-element float32); ``stats`` (2,) optionally
    receives (accuracy, entropy) of get_statistics (:36-43); returns tape."""
    if hasattr(ops, "cl_fused_ok") and ops.cl_fused_ok(a, b):
        # round 5: normalisation + logits in ONE launch (no normalised copies, no GEMM + split-K reduction), then the fused CE
        logits, ainv, binv = ops.cl_logits(a, b, 1.0 / temperature)
        dlogits = ops.xent_sym(logits, 1.0, loss_acc, want_grad, stats)
        return dict(fused=True, a=a, b=b, ainv=ainv, binv=binv, dlogits=dlogits, logits=logits, t=temperature)
    an, ainv = ops.l2norm_fwd(a)
    bn, binv = ops.l2norm_fwd(b)
    logits = ops.gemm(an, bn, tb=True, alpha=1.0 / temperature)        # logits_img2cond (:64-65)
    dlogits = ops.xent_sym(logits, 1.0, loss_acc, want_grad, stats)    # both directions (:68-78)
    return dict(an=an, ainv=ainv, bn=bn, binv=binv, dlogits=dlogits, logits=logits, t=temperature)


def contrastive_loss_bwd(ops, tape, want_a=True, want_b=True, add_a=None, add_b=None):
    """-> (da, db) float32 (None where not requested).  ``add_a`` / ``add_b``: ADD the term into this (B, D) tensor instead
    (returned in place of da / db)."""
    dl, t = tape["dlogits"], tape["t"]
    da = db = None
    if tape.get("fused"):
        if want_a:
            da = ops.cl_bwd(dl, tape["a"], tape["b"], tape["ainv"], tape["binv"], 1.0 / t, False, out=add_a)
        if want_b:
            db = ops.cl_bwd(dl, tape["b"], tape["a"], tape["binv"], tape["ainv"], 1.0 / t, True, out=add_b)
        return da, db
    if want_a:
        dan = ops.gemm(dl, tape["bn"], alpha=1.0 / t)
        da = ops.l2norm_bwd(dan, tape["an"], tape["ainv"], torch.float32)
        if add_a is not None:
            da = ops.add_into(add_a, da)
    if want_b:
        dbn = ops.gemm(dl, tape["an"], ta=True, alpha=1.0 / t)
        db = ops.l2norm_bwd(dbn, tape["bn"], tape["binv"], torch.float32)
        if add_b is not None:
            db = ops.add_into(add_b, db)
    return da, db


# ------------------------------------------------------------------------------------ word_loss
def prepare_words(ops, words_n, image_feat_dtype, r):
    """bf16 copies of the normalised words for the fused path (shared by the real / fake calls of one D forward), or None"""
    b, t, e = words_n.shape
    if (getattr(ops, "wl_fused", False) and image_feat_dtype == torch.bfloat16 and hasattr(ops, "wl_prep_words")
            and bool(ops.lib.xmc_wl_fused_supported(b, r, t, e))):
        return ops.wl_prep_words(words_n)
    return None


def _word_loss_fwd_fused(ops, image_feat, words_n, ml, loss_acc, gamma1, gamma2, gamma3, want_grad, stats, wprep):
    """word_loss_fused.hip: S / alpha / H stay inside one kernel per direction; same tape contract as the GEMM path."""
    b, r, e = image_feat.shape
    t = words_n.shape[1]
    w, wt = wprep if wprep is not None else ops.wl_prep_words(words_n)
    rn, rnt, rinv = ops.wl_prep_regions(image_feat.contiguous())
    g = ops.wl_tn_gemm(rn, rn, e, r, r, b, torch.bfloat16)                          # G_j = R^_j R^_j^T
    nn, q = ops.wl_cols_fwd(rn, w, g, ml, t, gamma1)
    sim_t, pi = ops.wl_rows(nn, q, ml, b, t, gamma2, gamma3)
    dsim = ops.xent_sym(sim_t, 1.0, loss_acc, want_grad, stats)
    return dict(fused=True, rn=rn, rnt=rnt, rinv=rinv, g=g, w=w, wt=wt, ml=ml, pi=pi, dsim=dsim, sim_t=sim_t, nn=nn, q=q,
                dims=(b, r, t, e), g1=gamma1, g3=gamma3, dtype=image_feat.dtype)


def _word_loss_bwd_fused(ops, tape, out=None):
    b, r, t, e = tape["dims"]
    rn, w, wt, g = tape["rn"], tape["w"], tape["wt"], tape["g"]
    ldp = w.shape[0]
    ds, a_s, al = ops.wl_cols_bwd(rn, w, g, tape["ml"], tape["dsim"], tape["pi"], t, tape["g1"], tape["g3"])
    dg2 = ops.wl_tn_gemm(a_s, al, ldp, r, r, b, torch.bfloat16, alpha=2.0)          # 2 sum_c dq alpha alpha^T
    drn = ops.wl_tn_gemm(ds, wt, ldp, r, e, b, torch.float32, x1=dg2, y1=tape["rnt"], k1=r, y0_shared=True)
    dx = ops.l2norm_bwd_bf16y(drn.view(b * r, e), rn.view(b * r, e), tape["rinv"], tape["dtype"], out=out)
    return dx.view(b, r, e)


def word_loss_fwd(ops, image_feat, words_n, max_len, loss_acc, gamma1=5.0, gamma2=5.0, gamma3=50.0,
                  want_grad=True, stats=None, wprep=None):
    """image_feat (B, R, E) activation dtype; words_n (B, T, E) float32 normalised; ``wprep``: prepare_words(...) of the
    same words (optional)."""
    b, r, e = image_feat.shape
    t = words_n.shape[1]
    ml = max_len.reshape(-1).contiguous()
    fused = getattr(ops, "wl_fused_ok", None)
    if fused is not None and fused(image_feat, t):
        return _word_loss_fwd_fused(ops, image_feat, words_n, ml, loss_acc, gamma1, gamma2, gamma3, want_grad, stats, wprep)
    rn, rinv = ops.l2norm_fwd(image_feat.reshape(b * r, e))
    s = ops.gemm(rn, words_n.view(b * t, e), tb=True, fast=True)       # (B*R, B*T)
    rn3 = rn.view(b, r, e)
    g = ops.gemm(rn3, rn3, tb=True, fast=True)                         # (B, R, R)
    alpha, nn = ops.wl_softmax(s, ml, b, r, t, gamma1)
    h = ops.gemm(g, alpha.view(b, r, b * t), fast=True)                # (B, R, B*T)
    q = ops.wl_qdot(alpha, h, b, r, t)
    sim_t, pi = ops.wl_rows(nn, q, ml, b, t, gamma2, gamma3)           # sim_t[caption i, image j]
    dsim = ops.xent_sym(sim_t, 1.0, loss_acc, want_grad, stats)        # (:175-190)
    return dict(rn=rn, rinv=rinv, s=s, alpha=alpha, h=h, nn=nn, q=q, pi=pi, dsim=dsim, sim_t=sim_t,
                words_n=words_n, dims=(b, r, t, e), g1=gamma1, g3=gamma3, dtype=image_feat.dtype)


def word_loss_bwd(ops, tape, out=None):
    """-> d image_feat (B, R, E) in the activation dtype (written into ``out`` when given)."""
    if tape.get("fused"):
        return _word_loss_bwd_fused(ops, tape, out)
    b, r, t, e = tape["dims"]
    ds, a_s = ops.wl_bwd_cols(tape["s"], tape["alpha"], tape["h"], tape["nn"], tape["q"], tape["pi"],
                              tape["dsim"], b, r, t, tape["g1"], tape["g3"])
    dg = ops.gemm(a_s.view(b, r, b * t), tape["alpha"].view(b, r, b * t), tb=True, fast=True)   # sum dq alpha alpha^T
    drn = ops.gemm(ds.view(b * r, b * t), tape["words_n"].view(b * t, e), fast=True)                     # (B*R, E)
    ops.gemm(dg, tape["rn"].view(b, r, e), alpha=2.0, beta=1.0, out=drn.view(b, r, e), fast=True)
    dx = ops.l2norm_bwd(drn, tape["rn"], tape["rinv"], tape["dtype"], out=out)
    return dx.view(b, r, e)
