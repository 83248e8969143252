"""PyTorch-CPU restatement of the reference ``train_step`` (forward + autograd + Adam + EMA).

TEST INFRASTRUCTURE ONLY -- never imported by ``xmcgan_image_generation_amd``; used by
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg (kind "port").

PARITY UNPINNED (see oracle/np_spec.py header): the reference cannot run here.  This file
is written independently of ``np_spec.py`` (different conv/attention formulations) so the
two cross-check each other; gradients come from torch autograd and are cross-checked by
float64 finite differences in tests/test_oracle.py.

Layouts follow the reference: NHWC activations, HWIO conv kernels, (in,out) dense kernels,
nested-dict parameter trees with Flax auto-names.  File:line citations are relative to
/root/reference.
"""
from __future__ import annotations

import math

import torch
import torch.nn.functional as tf
import torch.nn.functional as F

G_CHANNELS = {128: [16, 8, 4, 2, 1], 256: [16, 8, 8, 4, 2, 1]}
D_CHANNELS = {128: ([2, 4, 8, 16, 16], [True, True, True, True, False]),
              256: ([2, 4, 8, 8, 16, 16], [True, True, True, True, True, False])}


# ------------------------------------------------------------------------------ tree helpers
def to_torch(tree, dtype=torch.float32, requires_grad=False):
    if isinstance(tree, dict):
        return {k: to_torch(v, dtype, requires_grad) for k, v in tree.items()}
    t = torch.as_tensor(tree).to(dtype).clone()
    return t.requires_grad_(requires_grad)


def leaves(tree, prefix=""):
    out = []
    for k, v in tree.items():
        p = f"{prefix}/{k}" if prefix else k
        out.extend(leaves(v, p) if isinstance(v, dict) else [(p, v)])
    return out


def tree_map(fn, tree, *rest):
    if isinstance(tree, dict):
        return {k: tree_map(fn, v, *[r[k] for r in rest]) for k, v in tree.items()}
    return fn(tree, *rest)


# ----------------------------------------------------------------------------------- layers
def conv(x, p):
    """flax nn.Conv / lax.conv_general_dilated NHWC-HWIO stride 1 SAME (layers.py:224-233)."""
    return conv_k(x, p["kernel"], p["bias"])


def conv_k(x, kernel, bias):
    k = kernel.shape[0]
    y = tf.conv2d(x.permute(0, 3, 1, 2), kernel.permute(3, 2, 0, 1), bias, padding=k // 2)
    return y.permute(0, 2, 3, 1)


def dense(x, p):
    return x @ p["kernel"] + p["bias"]


# Finite-difference tests must hold the stop_gradient'ed (u, v) fixed while W is perturbed:
# {"mode": "record"|"replay", "uv": {u0.data_ptr(): (u, v)}}; None in normal use.
_UV_FREEZE = None


def sn_kernel(kernel, u0, eps=1e-10):
    """One power-iteration step (layers.py:92-101, :209-220).  Returns (W/sigma, new u)."""
    w = kernel.reshape(-1, kernel.shape[-1])
    with torch.no_grad():                                   # stop_gradient on u, v
        v = u0 @ w.t()
        v = v * torch.rsqrt((v * v).sum() + eps)
        u = v @ w
        u = u * torch.rsqrt((u * u).sum() + eps)
        if _UV_FREEZE is not None:
            if _UV_FREEZE["mode"] == "record":
                _UV_FREEZE["uv"][u0.data_ptr()] = (u.clone(), v.clone())
            else:
                u, v = _UV_FREEZE["uv"][u0.data_ptr()]
    sigma = (v @ w @ u.t())[0, 0]                           # gradient flows through sigma
    return (w / (sigma + eps)).reshape(kernel.shape), u


def sconv(x, p, st, new, name):
    k, u = sn_kernel(p[name]["kernel"], st[name]["u0"])
    new[name] = {"u0": u}
    return conv_k(x, k, p[name]["bias"])


def sdense(x, p, st, new, name):
    k, u = sn_kernel(p[name]["kernel"], st[name]["u0"])
    new[name] = {"u0": u}
    return x @ k + p[name]["bias"]


def batch_norm(x, st, train, momentum=0.9, eps=1e-5):
    """flax 0.3.3 BatchNorm, no scale/bias (xmc_net.py:192-201): biased var = E[x^2]-E[x]^2."""
    if train:
        mean = x.mean(dim=(0, 1, 2))
        var = (x * x).mean(dim=(0, 1, 2)) - mean * mean
        new = {"mean": (momentum * st["mean"] + (1 - momentum) * mean).detach(),
               "var": (momentum * st["var"] + (1 - momentum) * var).detach()}
    else:
        mean, var, new = st["mean"], st["var"], st
    return (x - mean) * torch.rsqrt(var + eps), new


def cbn(x, emb, p, st, train):
    """ConditionalBatchNorm (layers.py:244-258)."""
    g = dense(emb, p["Dense_0"])[:, None, None, :]
    b = dense(emb, p["Dense_1"])[:, None, None, :]
    xh, new = batch_norm(x, st["BatchNorm_0"], train)
    return xh * (g + 1.0) + b, {"BatchNorm_0": new}


def lcbn(x, emb, p, st, train):
    """LocalConditionalBatchNorm (layers.py:261-273)."""
    g = conv(emb, p["Conv_0"])
    b = conv(emb, p["Conv_1"])
    xh, new = batch_norm(x, st["BatchNorm_0"], train)
    return xh * (g + 1.0) + b, {"BatchNorm_0": new}


def upsample(x):
    """common.py:48-51 nearest x2."""
    return x.repeat_interleave(2, dim=1).repeat_interleave(2, dim=2)


def dsample(x):
    """common.py:23-55: 2x2/s2 mean."""
    return tf.avg_pool2d(x.permute(0, 3, 1, 2), 2).permute(0, 2, 3, 1)


# ------------------------------------------------------------------------------ attention_lib
def l2n(x, eps=1e-12):
    """attention_lib.py:30-33."""
    return x * torch.rsqrt(torch.clamp((x * x).sum(-1, keepdim=True), min=eps))


def xent_rows(logits):
    """mean over rows of -log_softmax(logits)[i, i]  (losses.py:47-51 with one-hot eye)."""
    return -torch.diagonal(torch.log_softmax(logits, dim=-1)).mean()


def contrastive_loss(a, b, temperature=0.1):
    """attention_lib.py:46-79."""
    a, b = l2n(a), l2n(b)
    l1 = a @ b.t() / temperature
    l2 = b @ a.t() / temperature
    return xent_rows(l1) + xent_rows(l2), (l1, l2)


def word_loss(img, words, max_len, g1=5.0, g2=5.0, g3=50.0):
    """attention_lib.py:130-191, batched over (caption i, image j) with einsum."""
    bsz, rnum, _ = img.shape
    tlen = words.shape[1]
    r = l2n(img)                                             # (J,R,E)
    w = l2n(words)                                           # (I,T,E)
    s = torch.einsum("jre,ite->ijrt", r, w) * g1             # (I,J,R,T)
    mask = (torch.arange(tlen, dtype=img.dtype)[None, :] >= max_len.reshape(-1, 1)).to(img.dtype)
    s = s + mask[:, None, None, :] * (-1e9)
    alpha = torch.softmax(s, dim=2)                          # over regions
    ctx = torch.einsum("ijrt,jre->ijte", alpha, r)           # (I,J,T,E)
    wi = words[:, None, :, :].expand(-1, bsz, -1, -1)        # un-normalised words
    cos = (wi * ctx).sum(-1) / (wi.norm(dim=-1) * ctx.norm(dim=-1))
    row = cos * g2 + mask[:, None, :] * (-1e9)
    sim_t = torch.logsumexp(row, dim=-1) / g2 * g3           # [caption i, image j]
    sim = sim_t.t()
    return xent_rows(sim) + xent_rows(sim_t), sim


def attention_for_g(region, words, gamma, mask):
    """attention_lib.py:194-219."""
    r, w = l2n(region), l2n(words)
    s = r @ w.transpose(1, 2) * gamma + mask * (-1e9)
    attn = torch.softmax(s, dim=-1)
    return attn @ w, attn


# -------------------------------------------------------------------------------------- nets
def gen_block(x, cond, p, st, train, norm, c0, c1):
    x0 = x
    new = {}
    x, new[c0] = norm(x, cond[0], p[c0], st[c0], train)
    x = conv(upsample(torch.relu(x)), p["Conv_0"])
    x, new[c1] = norm(x, cond[1], p[c1], st[c1], train)
    x = conv(torch.relu(x), p["Conv_1"])
    return x + conv(upsample(x0), p["Conv_2"]), new


def generator(params, bstats, cond_dict, z, cfg, train):
    """xmc_net.py:160-248 -> (image, new_batch_stats, aux)."""
    dt = z.dtype
    sent = cond_dict["sentence_embedding"].to(dt)
    words = cond_dict["embedding"].to(dt)
    max_len = cond_dict["max_len"].to(dt)
    gf, chans = cfg["gf_dim"], G_CHANNELS[cfg["image_size"]]
    bsz, edim = z.shape[0], words.shape[-1]
    new = {}
    gcond = torch.cat([dense(sent, params["Dense_0"]), z], dim=-1)
    x = dense(z, params["Dense_1"]).reshape(-1, 4, 4, gf * 16)
    for i in range(2):
        nm = f"GenBlock_{i}"
        x, new[nm] = gen_block(x, (gcond, gcond), params[nm], bstats[nm], train, cbn,
                               "ConditionalBatchNorm_0", "ConditionalBatchNorm_1")
    xc = conv(x, params["Conv_0"])
    ss = xc.shape[1]
    xc = xc.reshape(bsz, ss * ss, edim)
    mask = (torch.arange(words.shape[1], dtype=dt)[None, :] >= max_len).to(dt)
    mask = mask[:, None, :].expand(-1, ss * ss, -1)
    ctx, attn = attention_for_g(xc, words, float(cfg["gamma_for_g"]), mask)
    scond = torch.cat([ctx.reshape(bsz, ss, ss, edim),
                       gcond.reshape(bsz, 1, 1, -1).expand(-1, ss, ss, -1)], dim=-1)
    for i in range(2, len(chans)):
        nm = f"GenSpatialBlock_{i - 2}"
        up = upsample(scond)
        x, new[nm] = gen_block(x, (scond, up), params[nm], bstats[nm], train, lcbn,
                               "LocalConditionalBatchNorm_0", "LocalConditionalBatchNorm_1")
        scond = up
    nm = "LocalConditionalBatchNorm_0"
    x, new[nm] = lcbn(x, scond, params[nm], bstats[nm], train)
    x = conv(torch.relu(x), params["Conv_1"])
    return (torch.tanh(x) + 1.0) / 2.0, new, {"attn": attn}


def discriminator(params, sn, images, cond_dict, cfg):
    """xmc_net.py:45-142 -> ((logit, stats), new_sn_stats, aux)."""
    dt = images.dtype
    sent = cond_dict["sentence_embedding"].to(dt)
    words = cond_dict["embedding"].to(dt)
    max_len = cond_dict["max_len"].to(dt)
    df = cfg["df_dim"]
    chans, downs = D_CHANNELS[cfg["image_size"]]
    new = {}
    # DiscOptimizedBlock (common.py:117-133)
    p, st = params["DiscOptimizedBlock_0"], sn["DiscOptimizedBlock_0"]
    nb = {}
    h = sconv(images, p, st, nb, "SpectralConv_0")
    h = sconv(torch.relu(h), p, st, nb, "SpectralConv_1")
    x = dsample(h) + sconv(dsample(images), p, st, nb, "SpectralConv_2")
    new["DiscOptimizedBlock_0"] = nb
    x_cond = None
    for i, (c, d) in enumerate(zip(chans, downs)):           # DiscBlock (common.py:58-79)
        nm = f"DiscBlock_{i}"
        p, st, nb = params[nm], sn[nm], {}
        x0 = x
        h = sconv(torch.relu(x), p, st, nb, "SpectralConv_0")
        h = sconv(torch.relu(h), p, st, nb, "SpectralConv_1")
        if d or x0.shape[-1] != df * c:
            x0 = sconv(x0, p, st, nb, "SpectralConv_2")
        if d:
            h, x0 = dsample(h), dsample(x0)
        x = x0 + h
        new[nm] = nb
        if x.shape[1] == cfg["cond_size"]:
            x_cond = x
    x_pool = torch.relu(x).sum(dim=(1, 2))
    out = sdense(x_pool, params, sn, new, "SpectralDense_0")
    sent_cond = sdense(sent, params, sn, new, "SpectralDense_1")
    emb = sent_cond.repeat(x_pool.shape[0] // sent_cond.shape[0], 1)
    out = out + (x_pool * emb).sum(dim=1, keepdim=True)
    real_feat, fake_feat = x_pool.chunk(2)
    zero = torch.zeros((), dtype=dt)
    fs = rs = fw = rw = ic = zero                             # xmc_net.py:60-64: disabled heads contribute 0
    fs_l = rs_l = ic_l = fw_sim = rw_sim = None
    if cfg.get("sentence_contrastive", True):                # :105-111
        fs, fs_l = contrastive_loss(fake_feat, sent_cond)
        rs, rs_l = contrastive_loss(real_feat, sent_cond)
    if cfg.get("word_contrastive", True):                    # :112-121
        xc = sconv(x_cond, params, sn, new, "SpectralConv_0")
        xc = xc.reshape(-1, cfg["cond_size"] ** 2, words.shape[-1])
        real_xc, fake_xc = xc.chunk(2)
        fw, fw_sim = word_loss(fake_xc, words, max_len)
        rw, rw_sim = word_loss(real_xc, words, max_len)
    if cfg.get("image_contrastive", True):                   # :122-125
        ic, ic_l = contrastive_loss(fake_feat, real_feat)
    stats = dict(fake_word_loss=fw, real_word_loss=rw, fake_sentence_loss=fs,
                 real_sentence_loss=rs, image_contrastive_loss=ic)
    aux = dict(fake_sentence_logits=fs_l, real_sentence_logits=rs_l,
               image_contrastive_logits=ic_l, fake_word_sim=fw_sim, real_word_sim=rw_sim,
               x_pool=x_pool)
    return (out, stats), new, aux


# ---------------------------------------------------- frozen ResNet-50 feature path (xmc_gan.py:74-90)
def _same_pads(size, k, stride):
    """flax / XLA "SAME": out = ceil(in / s); total padding split low = total // 2, high = rest"""
    out = -(-size // stride)
    total = max((out - 1) * stride + k - size, 0)
    return total // 2, total - total // 2


def conv_same(x, kernel, stride=1):
    """flax nn.Conv(use_bias=False, padding="SAME") on NHWC with an HWIO kernel (resnet_v1.py:25-26,148-154)"""
    kh, kw = kernel.shape[0], kernel.shape[1]
    pt, pb = _same_pads(x.shape[1], kh, stride)
    pl, pr = _same_pads(x.shape[2], kw, stride)
    xp = F.pad(x.permute(0, 3, 1, 2), (pl, pr, pt, pb))
    return F.conv2d(xp, kernel.permute(3, 2, 0, 1), None, stride=stride).permute(0, 2, 3, 1)


def bn_eval(x, p, st, eps=1e-5):
    """flax nn.BatchNorm(use_running_average=True) with scale and bias (resnet_v1.py:144-146)"""
    return (x - st["mean"]) * torch.rsqrt(st["var"] + eps) * p["scale"] + p["bias"]


def resnet50(params, bstats, x):
    """ResNet.__call__(train=False) of resnet_v1.py:129-172 -> (pool (N, 7, 7, 2048), logits (N, classes)).
    Note: the root block has NO ReLU between init_bn and the max-pool (resnet_v1.py:155-156)."""
    x = bn_eval(conv_same(x, params["init_conv"]["kernel"], 2), params["init_bn"], bstats["init_bn"])
    pt, pb = _same_pads(x.shape[1], 3, 2)
    xp = F.pad(x.permute(0, 3, 1, 2), (pt, pb, pt, pb), value=float("-inf"))
    x = F.max_pool2d(xp, kernel_size=3, stride=2).permute(0, 2, 3, 1)            # nn.max_pool "SAME" (:156)
    for i in range(4):
        sp, ss = params[f"stage{i + 1}"], bstats[f"stage{i + 1}"]
        for k in range(len(sp)):
            bp, bs = sp[f"block{k + 1}"], ss[f"block{k + 1}"]
            stride = 2 if (i > 0 and k == 0) else 1
            res = x
            y = torch.relu(bn_eval(conv_same(x, bp["conv1"]["kernel"]), bp["bn1"], bs["bn1"]))       # :60-66
            y = torch.relu(bn_eval(conv_same(y, bp["conv2"]["kernel"], stride), bp["bn2"], bs["bn2"]))
            y = bn_eval(conv_same(y, bp["conv3"]["kernel"]), bp["bn3"], bs["bn3"])
            if "proj_conv" in bp:                                                                    # :78-82
                res = bn_eval(conv_same(res, bp["proj_conv"]["kernel"], stride), bp["proj_bn"], bs["proj_bn"])
            x = torch.relu(res + y)
    pool = x
    out = pool.mean(dim=(1, 2)) @ params["head"]["kernel"] + params["head"]["bias"]                  # :166-171
    return pool, out


RESNET_IMG_SIZE = 224


def resize_weight_matrix(n_in, n_out, dtype=torch.float64):
    """jax.image.resize(method="bilinear", antialias=True [the default]) along one axis, as the (n_in, n_out) matrix
    jax builds in ``jax/_src/image/scale.py:compute_weight_mat`` (jax is a dependency of the reference that is absent
    here and not pinned in requirements.txt -- flax 0.3.3 implies jax 0.2.x; this restates its published algorithm):
    sample positions on half-pixel centres, a triangle kernel widened by max(n_in / n_out, 1) (anti-aliasing when
    shrinking), weights normalised over the input taps; columns whose sample falls outside the input are zero."""
    inv_scale = n_in / n_out
    kernel_scale = max(inv_scale, 1.0)
    sample_f = (torch.arange(n_out, dtype=dtype) + 0.5) * inv_scale - 0.5
    x = (sample_f[None, :] - torch.arange(n_in, dtype=dtype)[:, None]).abs() / kernel_scale
    w = torch.clamp(1.0 - x, min=0.0)
    total = w.sum(0, keepdim=True)
    w = torch.where(total.abs() > 1000.0 * torch.finfo(torch.float32).eps, w / total, torch.zeros_like(w))
    inside = (sample_f >= -0.5) & (sample_f <= n_in - 0.5)
    return torch.where(inside[None, :], w, torch.zeros_like(w))


def get_pretrained_embs(params, bstats, images):
    """pretrained_model_utils.py:102-127: jax.image.resize to 224 x 224 (unless already 224) + ResNet-50"""
    if images.shape[1] != RESNET_IMG_SIZE and images.shape[2] != RESNET_IMG_SIZE:
        wy = resize_weight_matrix(images.shape[1], RESNET_IMG_SIZE, images.dtype)
        wx = resize_weight_matrix(images.shape[2], RESNET_IMG_SIZE, images.dtype)
        images = torch.einsum("nhwc,hy,wx->nyxc", images, wy, wx)
    return resnet50(params, bstats, images)


def contrastive_loss_on_pretrained(resnet, real_images, fake_images):
    """calculate_contrastive_loss_on_pretrained (xmc_gan.py:74-90); ``resnet`` = (params, batch_stats)"""
    _, real_out = get_pretrained_embs(resnet[0], resnet[1], real_images)
    _, fake_out = get_pretrained_embs(resnet[0], resnet[1], fake_images)
    return contrastive_loss(real_out, fake_out)[0]


def hinge_loss(real, fake):
    """losses.py:30-35."""
    return (torch.relu(1.0 - real) + torch.relu(1.0 + fake)).mean(), -fake.mean()


# ------------------------------------------------------------------------------ optimisation
def adam_init(params):
    return {"step": 0,
            "m": tree_map(torch.zeros_like, params),
            "v": tree_map(torch.zeros_like, params)}


def adam_apply(params, grads, opt, lr, b1, b2, eps=1e-8):
    """flax.optim.Adam.apply_gradient (flax 0.3.3), weight_decay=0."""
    t = opt["step"] + 1
    new_m = tree_map(lambda m, g: b1 * m + (1 - b1) * g, opt["m"], grads)
    new_v = tree_map(lambda v, g: b2 * v + (1 - b2) * g * g, opt["v"], grads)
    c1, c2 = 1 - b1 ** t, 1 - b2 ** t

    def upd(p, m, v):
        return (p - lr * (m / c1) / (torch.sqrt(v / c2) + eps)).detach()
    new_p = tree_map(upd, params, new_m, new_v)
    return new_p, {"step": t, "m": new_m, "v": new_v}


def make_state(g_params, g_bstats, d_params, d_sn, dtype=torch.float32, resnet=None):
    """TrainState (train_utils.py:42-50) as a plain dict; ``resnet`` = (params, batch_stats) of the frozen ResNet-50
    (``additional_data`` of xmc_gan.py:43-55) when pretrained_image_contrastive is on."""
    gp, dp = to_torch(g_params, dtype), to_torch(d_params, dtype)
    extra = {} if resnet is None else {"resnet": (to_torch(resnet[0], dtype), to_torch(resnet[1], dtype))}
    return dict(**extra, step=0, g_params=gp, d_params=dp, g_opt=adam_init(gp), d_opt=adam_init(dp),
                generator_state=to_torch(g_bstats, dtype),
                discriminator_state=to_torch(d_sn, dtype),
                ema_params=tree_map(lambda t: t.clone(), gp))


def _split(batch, n):
    """train_utils.split_input_dict (train_utils.py:69-88)."""
    return [{k: v.chunk(n)[i] for k, v in batch.items()} for i in range(n)]


def _losses(gp, dp, state, batch, cfg):
    """loss_fn of xmc_gan.py:124-160 (and :220-243)."""
    img, new_g, gaux = generator(gp, state["generator_state"], batch, batch["z"], cfg, True)
    allimg = torch.cat([batch["image"].to(img.dtype), img])
    (logit, rd), new_d, daux = discriminator(dp, state["discriminator_state"], allimg, batch, cfg)
    real, fake = logit.chunk(2)
    hd, hg = hinge_loss(real, fake)
    c_d = rd["real_word_loss"] + rd["real_sentence_loss"]
    c_g = rd["fake_word_loss"] + rd["fake_sentence_loss"] + rd["image_contrastive_loss"]
    daux.update(gaux)
    daux.update(image=img, logit=logit)
    c_pre = torch.zeros((), dtype=img.dtype)
    if cfg.get("pretrained_image_contrastive", False):                       # xmc_gan.py:149-152
        c_pre = contrastive_loss_on_pretrained(state["resnet"], batch["image"].to(img.dtype), img)
    daux.update(c_loss_g_pretrained=c_pre)
    return hd + c_d, hg + c_g + c_pre, c_d, c_g, new_g, new_d, daux


def _req(tree):
    return tree_map(lambda t: t.detach().clone().requires_grad_(True), tree)


def train_d(state, batch, cfg, grad_hook=None):
    """xmc_gan.py:194-256."""
    gp, dp = state["g_params"], _req(state["d_params"])
    d_loss, _, _, _, _, new_d, _ = _losses(gp, dp, state, batch, cfg)
    dl = [t for _, t in leaves(dp)]
    grads = torch.autograd.grad(d_loss, dl)
    it = iter(grads)
    d_grad = tree_map(lambda t: next(it), dp)
    if grad_hook is not None:
        d_grad = grad_hook("d", d_grad)
    new_dp, new_opt = adam_apply(state["d_params"], d_grad, state["d_opt"], cfg["d_lr"],
                                 cfg["beta1"], cfg["beta2"])
    out = dict(state)
    out.update(d_params=new_dp, d_opt=new_opt,
               discriminator_state=tree_map(lambda t: t.detach(), new_d))
    return out, {"d_grad": d_grad, "d_loss": d_loss.detach()}


def train_g_d(state, batch, cfg, grad_hook=None):
    """xmc_gan.py:93-191 (pretrained_image_contrastive=False)."""
    gp, dp = _req(state["g_params"]), _req(state["d_params"])
    d_loss, g_loss, c_d, c_g, new_g, new_d, aux = _losses(gp, dp, state, batch, cfg)
    dl = [t for _, t in leaves(dp)]
    gl = [t for _, t in leaves(gp)]
    dg = torch.autograd.grad(d_loss, dl, retain_graph=True)           # pullback (1, 0)
    gg = torch.autograd.grad(g_loss, gl)                              # pullback (0, 1)
    it = iter(dg)
    d_grad = tree_map(lambda t: next(it), dp)
    it = iter(gg)
    g_grad = tree_map(lambda t: next(it), gp)
    if grad_hook is not None:
        d_grad, g_grad = grad_hook("d", d_grad), grad_hook("g", g_grad)
    new_dp, new_dopt = adam_apply(state["d_params"], d_grad, state["d_opt"], cfg["d_lr"],
                                  cfg["beta1"], cfg["beta2"])
    new_gp, new_gopt = adam_apply(state["g_params"], g_grad, state["g_opt"], cfg["g_lr"],
                                  cfg["beta1"], cfg["beta2"])
    decay = cfg["polyak_decay"]
    ema = tree_map(lambda e, p: e * decay + (1 - decay) * p, state["ema_params"], new_gp)
    out = dict(state)
    out.update(step=state["step"] + 1, d_params=new_dp, d_opt=new_dopt, g_params=new_gp,
               g_opt=new_gopt, generator_state=tree_map(lambda t: t.detach(), new_g),
               discriminator_state=tree_map(lambda t: t.detach(), new_d), ema_params=ema)
    metrics = dict(d_loss=d_loss.detach(), g_loss=g_loss.detach(), c_loss_d=c_d.detach(),
                   c_loss_g=c_g.detach(), c_loss_g_pretrained=aux["c_loss_g_pretrained"].detach())
    dbg = dict(d_grad=d_grad, g_grad=g_grad, aux=aux)
    return out, metrics, dbg


def train_step(state, batch, cfg, grad_hook=None, return_debug=False):
    """train_utils.train_step (train_utils.py:91-130): train_d on half 0, train_g_d on half 1."""
    n = cfg["d_step_per_g_step"]
    parts = _split(batch, n)
    dbg_d = None
    for i in range(n - 1):
        state, dbg_d = train_d(state, parts[i], cfg, grad_hook)
    state, metrics, dbg = train_g_d(state, parts[-1], cfg, grad_hook)
    if return_debug:
        dbg["train_d"] = dbg_d
        return state, metrics, dbg
    return state, metrics


def batch_to_torch(batch, dtype=torch.float32):
    return {k: torch.as_tensor(v).to(dtype) for k, v in batch.items()}


def two_ln_b(b):
    """Known answer: contrastive loss with all-equal logits = 2 ln B."""
    return 2.0 * math.log(b)
