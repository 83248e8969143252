/* libxmcgan_hip.so -- C ABI of the MI355X-native (gfx950) XMC-GAN G+D training-step kernels.
 *
 * The reference (google-research/xmcgan_image_generation) has no FFI: every device op is
 * emitted by XLA from jax.numpy / flax.linen calls.  This header therefore declares the
 * operator boundary a maintainer would bind from xmcgan/libml/layers.py,
 * xmcgan/libml/attention_lib.py, xmcgan/libml/losses.py, xmcgan/nets/common.py and
 * xmcgan/xmc_gan.py (file:line cited per entry point; paths relative to the reference
 * root).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is DEVICE memory owned by the caller
 *    (kernels never allocate, so a whole step is hipGraph-capturable);
 *  - activations are NHWC, `dtype` = XMC_F32 or XMC_BF16; parameters, gradients,
 *    statistics and losses are always float32;
 *  - conv weights are "prepared" copies in the activation dtype:
 *      forward  layout [Cout][kh*kw][Cin]            (K contiguous per output channel)
 *      dgrad    layout [Cin][kh*kw flipped][Cout]
 *    produced by xmc_prep_conv_weight from the float32 master [Cout][kh*kw][Cin];
 *  - all launches are asynchronous on `stream` (a hipStream_t passed as void*), no hidden
 *    synchronisation; no environment variables are read and the only process-wide state is a set of
 *    idempotent per-device "LDS opt-in done" flags and the explicit tuning table of xmc_set_tuning
 *    (re-entrant, thread-safe, several GPUs per process);
 *  - return 0 on success, XMC_EINVAL for bad shape/dtype/alignment, -(1000+hipError_t) for
 *    HIP launch errors.  No exceptions, no abort.
 */
#ifndef XMCGAN_HIP_H_
#define XMCGAN_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XMC_OK 0
#define XMC_EINVAL (-22)

#define XMC_F32 0
#define XMC_BF16 1

#define XMC_ABI_VERSION 22
int xmc_abi_version(void);

/* Launch-heuristic knobs -- split-K workgroup targets and tile-selection thresholds whose defaults were A/B'd inside the
 * training step.  Process-wide, relaxed atomics; meant for repeating those A/Bs without a rebuild, not
 * for production use (the defaults are the product).  value 0 restores a target's default ("cbn_run": -1).  Keys:
 * "ksplit_target", "ksplit_target_phase", "ksplit_target_pw", "tile64_pct", "wgrad_target_hi", "wgrad_target_lo",
 * "wgrad_target_phase", "cbn_run", and one numerics switch: "mx8_scale_floor" (MX-fp8 quantisers, BASELINE config #5): 0 / -1 = the
 * build's default rule -- X = 2^(floor(log2 amax) - 8), one binade higher when the block maximum would saturate e4m3 (amax / X > 448)
 * -- 1 = the OCP MX v1.0 conversion exactly (plain floor rule; the largest element of ~1/5 of the blocks clips by up to 12.5 %).
 * Read at launch time by xmc_mx8_quantize, xmc_mx8_pack_conv_weight, xmc_cbn_act_fwd_mx8 and the packet-emitting epilogue of
 * xmc_conv2d_mx8(_bits).  Unknown key: XMC_EINVAL.  The library itself reads NO environment variable. */
int xmc_set_tuning(const char* key, int32_t value);
int xmc_get_tuning(const char* key, int32_t* value);

/* ------------------------------------------------------------------------------ per-device handle
 * xmc_create validates `device` (gfx950 only), performs the per-device kernel setup (opt-in to the
 * 160 KiB LDS of the convolution / word-loss kernels) and returns an opaque handle; xmc_destroy frees
 * it.  Launch entry points do not take the handle: the only process-wide state behind them are
 * idempotent per-device flags set lock-free, so calls are re-entrant and thread-safe with or without a
 * handle.  Create one per GPU before capturing a hipGraph or sharing a GPU between host threads. */
int xmc_create(int32_t device, void** handle);
int xmc_destroy(void* handle);
int xmc_handle_device(void* handle);     /* -> the device index the handle was created for */

/* ------------------------------------------------------------------ convolution (K1, K2, K4, K5)
 * Implicit-GEMM NHWC convolution, stride 1, SAME, ks in {1,3}; replaces
 * lax.conv_general_dilated at xmcgan/libml/layers.py:224-233 and flax nn.Conv in
 * xmcgan/nets/common.py:71-75,127-132,153-159,179-185, xmcgan/nets/xmc_net.py:114,220,245.
 *   a   = relu_in ? relu(x) : x ; a = ups ? nearest_upsample2(a) : a      (common.py:48-51)
 *   v   = alpha * conv(a, w) + bias
 *   v   = mask && !mask_after_res ? (mask > 0 ? v : 0) : v                (ReLU backward)
 *   v   = v + res_scale * (res_ups ? nearest_upsample2(res) : res)        (residual / unpool)
 *   v   = mask && mask_after_res ? (mask > 0 ? v : 0) : v                 (ReLU backward of a post-activation sum)
 *   v   = relu_out ? max(v, 0) : v                                        (post-activation residual, resnet_v1.py:86)
 *   y   = valid_h && (oy >= valid_h || ox >= valid_w) ? 0 : v             (canvas margin of the ResNet-50 path)
 * The same entry point computes dgrad when given the dgrad-layout weights.
 * Output spatial dims must be powers of two (4..256 in every XMC-GAN layer). */
typedef struct {
    int32_t n, hi, wi, cin;   /* x is (n, hi, wi, cin) */
    int32_t cout;             /* y is (n, ho, wo, cout), ho = ups ? 2*hi : hi */
    int32_t ks;               /* 1 or 3 */
    int32_t ups;              /* nearest 2x upsample of x fused into the gather */
    int32_t relu_in;          /* relu applied to x on load */
    int32_t res_ups;          /* res is (n, ho/2, wo/2, cout) and is nearest-upsampled */
    int32_t out_f32;          /* store y as float32 even when dtype == XMC_BF16 */
    int32_t dtype;            /* dtype of x, w, mask, res (and y unless out_f32) */
    float alpha;              /* scale on the convolution result */
    float res_scale;
    int32_t w_packed;         /* bit 0: w is in MFMA-fragment order (xmc_pack_conv_weight): bf16, cin % 32 == 0, ks == 3;
                                 bit 4: w holds the 16-tap PHASE weights of xmc_phase_conv_weight -- only with ks == 3 and
                                 exactly one of ups / pool_out: the launch runs as four 2x2 convolutions on the low-resolution
                                 grid (16 instead of 36 multiply-adds per low-resolution pixel; `ups`: no res, `pool_out`: no mask);
                                 bit 5: with bit 4 and `ups`: the one-phase-per-workgroup form (A/B hook of tools/);
                                 bit 6: COMPACT pointwise launch (ks == 1, bf16, valid_h == valid_w = v > 0, no split-K): the
                                 workgroups cover only the v x v valid pixels of each ho x wo canvas (ResNet's 112/56/28/14/7
                                 maps on 128/64/32/16/8 canvases: 1.31x fewer pixels); margin pixels of y are NOT written --
                                 the caller keeps y in a buffer whose margins are zero once and stay zero;
                                 bit 7: with bit 4 and `ups`: keep the 64-pixel x 128-cout tiles where the 128-pixel x 64-cout
                                 ones would be chosen (A/B hook); bits 8-15: kernel A/B hooks of tools/ (0 = the shipped choice) */
    int32_t pool_out;         /* y = avg_pool2x2(v) + res_scale * res, y and res at (ho/2, wo/2): fused pooling of
                                 DiscBlock / DiscOptimizedBlock (common.py:76-78,131); w_packed, wo >= 32, no mask */
    int32_t relu_out;         /* ReLU on the result (after the residual) */
    int32_t mask_after_res;   /* the mask applies to v + res instead of to v */
    int32_t valid_h, valid_w; /* 0: every output pixel is live; else pixels outside the top-left valid_h x valid_w
                                 region of each image are stored as zero (no pool_out) */
    const float* alpha_dev;   /* NULL, or a float32 scalar in DEVICE memory multiplied into alpha when the kernel runs:
                                 1 / (sigma + eps) of a spectrally-normalised layer (W / sigma feeds a linear op, so the
                                 scale commutes with the convolution: xmcgan/libml/layers.py:209-233) -- the prepared
                                 weights are then a pure cast of W and need not wait for the power iteration */
} xmc_conv_desc;

int xmc_conv2d_nhwc(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                    const void* mask, const void* res, void* y, void* stream);

/* Layers with too few output tiles to fill the chip (the 4x4 / 8x8 layers) run split-K when the caller lends a
 * scratch buffer: xmc_conv2d_workspace_bytes(d) is the size xmc_conv2d_nhwc_ws wants for this descriptor (0: no
 * split; ws may then be NULL).  The buffer needs no initialisation; each split writes its own float32 slice and a
 * finishing kernel applies the epilogue.  xmc_conv2d_nhwc == xmc_conv2d_nhwc_ws(..., ws = NULL, ...). */
int64_t xmc_conv2d_workspace_bytes(const xmc_conv_desc* d);
/* 1 when a descriptor with w_packed bit 4 (16-tap phase weights) is inside the phase-decomposed kernels' domain, else 0 (the
 * launch would return XMC_EINVAL): lets the caller decide between the phase copies and the plain 3x3 copies of a layer */
int xmc_conv2d_phase_supported(const xmc_conv_desc* d);
int xmc_conv2d_nhwc_ws(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                       const void* mask, const void* res, void* y, void* ws, void* stream);
/* ReLU masks as BITS (kernels on fragment-packed weights, cout % 16 == 0): a mask tensor m (pixels, c) is also kept as
 * (pixels, c / 16) uint16 words, bit k of word j = (m[16 j + k] > 0).
 *   y_bits    (may be NULL): the launch also writes the bits of its OUTPUT (the stored value > 0) -- for the tensors that
 *             later serve as the `mask` of a data-gradient launch (jax.vjp of nn.relu: xmcgan/nets/common.py:63,66,129);
 *   mask_bits (may be NULL): used INSTEAD of `mask` (same shape convention): one 2-byte word per 16 couts and pixel
 *             in place of 32 bytes of bf16 that the epilogue would wait for (D 128^2 data gradient: 302 -> ~205 us).
 * A launch that takes the split-K route (xmc_conv2d_workspace_bytes(d) != 0 and ws given) cannot write y_bits (EINVAL) and
 * falls back to `mask` (which must then be given as well).  mask_bits == y_bits == NULL: xmc_conv2d_nhwc_ws. */
int xmc_conv2d_nhwc_bits(const xmc_conv_desc* d, const void* x, const void* w, const float* bias, const void* mask,
                         const void* res, void* y, void* ws, const void* mask_bits, void* y_bits, void* stream);

/* Pointwise (1x1) convolution whose reduction runs over TWO concatenated sources (round 6): y = epilogue([x | x2'] W^T), x2' pixel
 * (n, y, x) = x2[n, stride2 * y, stride2 * x, :] of an (n, h2, w2, cin2) tensor.  Replaces, in the frozen ResNet-50's down-sampling
 * bottleneck blocks, relu(bn3(conv3(h)) + proj_bn(proj_conv(x_in))) (xmcgan/utils/resnet_v1.py:74-86: the 1x1 stride-`strides`
 * projection shortcut and the block's last 1x1, eval-mode BatchNorm folded into W = [W3 | Wp], bias = b3 + bp) by ONE launch: the
 * projection's output is never written and re-read as the residual, and the sub-sampling copy in front of it is gone.
 * d as for xmc_conv2d_nhwc with ks = 1, dtype = XMC_BF16, cin = channels of x, fragment-packed w (K = cin + cin2; cin, cin2 multiples of
 * 32), COMPACT (w_packed bits 0 and 6, valid_h == valid_w = v with 0 < v < hi: only the v x v corner of every canvas is walked, the
 * margins of y are not written); relu_out, mask_after_res honoured; no ups / res_ups / pool_out / relu_in / split-K.  mask / mask_bits /
 * res (each may be NULL) and y_bits (may be NULL; cout % 16 == 0) as in xmc_conv2d_nhwc_bits.
 * stride2 = 1 or 2: x2' as above.  stride2 = -2: the ADJOINT sampling -- x2' pixel (n, y, x) = x2[n, y / 2, x / 2, :] where y and x are both
 * even, zero elsewhere: the block's data gradient mask(conv1^T(dh1) + scatter2(proj^T(g))) of the same blocks as one launch over
 * [dh1 | g'] with W = [W1^T | Wp^T] (jax.vjp of resnet_v1.py:74-86): the projection's gradient is never written, zero-scattered or re-read. */
int xmc_conv2d_pw_dual(const xmc_conv_desc* d, const void* x, const void* x2, int32_t cin2, int32_t h2, int32_t w2,
                       int32_t stride2, const void* w, const float* bias, const void* mask, const void* res, void* y,
                       const void* mask_bits, void* y_bits, void* stream);

/* ---- MX-fp8 3x3 convolution (BASELINE config #5: fp8 MFMA convolutions; replaces the conv_general_dilated of
 * xmcgan/libml/layers.py:221-233 and the flax nn.Conv of xmcgan/nets/common.py:152-159 when config.conv_fp8 is set).
 * Operands are OCP MX blocks: e4m3 elements with one e8m0 scale byte per 32 channels, multiplied by the gfx950
 * block-scaled MFMA (K = 64 per instruction, twice the bf16 rate), float32 accumulation, the bf16 kernel's epilogue.
 *
 * xmc_mx8_quantize: bf16 x [pixels][c] (c % 8 == 0) -> x8 [pixels][cp / 64][80] bytes (cp = c rounded up to 64, zero filled):
 *   per 64-channel chunk one 80-byte packet = 64 elements + the two scale bytes of its 32-channel blocks (bytes 64, 65) +
 *   pad; relu != 0 applies max(., 0) first (the `relu_in` of the convolution).
 * xmc_mx8_pack_conv_weight: bf16 fragment-packed weights (xmc_pack_conv_weight / the packed prep outputs: rows x 9 taps x
 *   k) -> w8 (ceil(rows / 32) * ceil(k / 64) * 9 * 2048 bytes) and wscale (ceil(rows / 32) * ceil(k / 64) * 3 * 256 bytes).
 * xmc_conv2d_mx8: y = epilogue(conv3x3(x8, w8)); d as for xmc_conv2d_nhwc with ks = 3, cin = the TRUE channel count,
 *   relu_in = mask_after_res = valid_* = 0 (relu_out: see xmc_conv2d_mx8_bits); ws (may be NULL) of xmc_conv2d_mx8_workspace_bytes(d) bytes
 *   enables split-K on few-tile layers.  y8 (may be NULL; needs bf16 output, cout % 64 == 0 and a launch without
 *   split-K): the epilogue also writes y as packets for the NEXT convolution, y8_relu = that convolution's relu_in --
 *   byte for byte what xmc_mx8_quantize(y, relu) would write, without the extra pass.
 * xmc_mx8_probe: one scaled MFMA on a8 [32][64] / b8 [32][64] (B transposed) bytes with scales as / bs [32][2] ->
 *   d [32][32] float32; pins the operand layout (tests). */
int xmc_mx8_quantize(const void* x, void* x8, int64_t pixels, int32_t c, int32_t relu, void* stream);
int xmc_mx8_pack_conv_weight(const void* w_packed, void* w8, void* wscale, int32_t rows, int32_t taps, int32_t k,
                             void* stream);
int64_t xmc_conv2d_mx8_workspace_bytes(const xmc_conv_desc* d);
int xmc_conv2d_mx8(const xmc_conv_desc* d, const void* x8, const void* w8, const void* wscale,
                   const float* bias, const void* mask, const void* res, void* y, void* y8, int32_t y8_relu,
                   void* ws, void* stream);
/* ... with the bf16 kernel's epilogue features (ABI 21): d->relu_out = 1 allowed (y = max(., 0); not with pool_out), mask_bits
 * (may be NULL) used INSTEAD of mask, y_bits (may be NULL) receives (y > 0) -- both as in xmc_conv2d_nhwc_bits: cout % 16 == 0 and
 * a launch without split-K (XMC_EINVAL otherwise; pass ws = NULL or check xmc_conv2d_mx8_workspace_bytes(d) == 0). */
int xmc_conv2d_mx8_bits(const xmc_conv_desc* d, const void* x8, const void* w8, const void* wscale,
                        const float* bias, const void* mask, const void* res, void* y, void* y8, int32_t y8_relu,
                        void* ws, const void* mask_bits, void* y_bits, void* stream);
int xmc_mx8_probe(const void* a8, const void* as, const void* b8, const void* bs, float* d, void* stream);

/* Weight gradient of the convolution above (jax.vjp of the same call sites):
 *   dw[cout][tap][cin] += alpha * sum_p dy'(p, cout) * a(p + tap, cin)
 * with a() as in xmc_conv2d_nhwc and dy' = dy_ups ? nearest_upsample2(dy) : dy.
 * dw is float32 in the master layout and is ACCUMULATED (atomic adds): zero it first. */
typedef struct {
    int32_t n, hi, wi, cin;   /* x is (n, hi, wi, cin) */
    int32_t cout;
    int32_t ks;
    int32_t x_ups, x_relu;
    int32_t dy_ups;           /* dy is (n, ho/2, wo/2, cout), nearest-upsampled on load */
    int32_t dtype;            /* dtype of x and dy */
    int32_t variant;          /* bits 0-3, kernel choice: 0 = generic split-K kernel only (bring-up / float32),
                                 1 = auto (LDS-DMA kernel, else register-staged patch kernel, else generic),
                                 2 = as 1 without the LDS-DMA kernel (A/B benchmarks);
                                 XMC_WGRAD_OVERWRITE: dw / db = alpha * (...) instead of += -- the FIRST write of a gradient
                                 nobody zeroed (the optimiser then need not clear what it consumed, and the reducing pass
                                 reads no old value); needs the workspace of xmc_conv2d_wgrad_ws for split launches */
    float alpha;
} xmc_wgrad_desc;
#define XMC_WGRAD_OVERWRITE 0x1000

/* db (may be NULL): the bias gradient of the same convolution, db[cout] += alpha * sum_p dy'(p, cout),
 * fused into the weight-gradient kernel. */
int xmc_conv2d_wgrad(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                     void* stream);
/* Deterministic split-K: with a workspace of xmc_conv2d_wgrad_workspace_bytes(d) bytes (no initialisation
 * needed) every pixel split writes its partial dW / db slab with plain stores and a second kernel adds the
 * slabs to dw / db in a FIXED order -- bit-reproducible gradients, no float atomics.  ws == NULL behaves
 * like xmc_conv2d_wgrad. */
int64_t xmc_conv2d_wgrad_workspace_bytes(const xmc_wgrad_desc* d);
int xmc_conv2d_wgrad_ws(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                        void* ws, int64_t ws_bytes, void* stream);

/* float32 master [cout][taps][cin] -> forward copy [cout][taps][cin] and dgrad copy
 * [cin][taps flipped][cout] in `dtype`, both multiplied by *inv_sigma when inv_sigma != NULL
 * (kernel / (sigma + eps), xmcgan/libml/layers.py:219-221).  Either output may be NULL.
 * packed bit 0 / bit 1: write the forward / dgrad copy in MFMA-fragment order (see xmc_pack_conv_weight;
 * bf16 only, cin % 32 == 0 resp. cout % 32 == 0; the buffer holds ceil(rows / 32) * 32 rows). */
int xmc_prep_conv_weight(const float* w, const float* inv_sigma, void* w_fwd, void* w_dgrad,
                         int32_t cout, int32_t taps, int32_t cin, int32_t dtype, int32_t packed, void* stream);

/* Prepared bf16 weights [cout][taps][cin] (forward or dgrad copy) -> MFMA-fragment order
 *   [ceil(cout/32)][cin/32][tap][k16 half][lane 0..63][8]   lane = (k8 half) * 32 + cout % 32
 * (ceil(cout/32) * 32 * taps * cin elements; rows >= cout are zero) consumed by xmc_conv2d_nhwc with
 * desc.w_packed = 1: every MFMA A operand is then one coalesced 1 KiB load, no LDS staging of weights. */
int xmc_pack_conv_weight(const void* w, void* out, int32_t cout, int32_t taps, int32_t cin, void* stream);

/* Phase weights of a 3x3 layer that sits next to a 2x resampling (the generator blocks' conv3x3(upsample(.)),
 * xmcgan/nets/common.py:152-159, and the discriminator blocks' avg_pool(conv3x3(.)), common.py:76-78,131).  Two of the
 * three rows / columns of such a window read the same low-resolution pixel, so the layer equals four 2x2 convolutions
 * whose taps are SUMS of the 3x3 taps (formed here in float32, x *inv_sigma, rounded to bf16 once):
 *   fwd_mode 0 (layer = conv(upsample2(x))):  w_fwd in "out" order for desc.ups launches, w_dgrad (rows = cin) in "in"
 *                                             order for the pool_out launch that computes its data gradient;
 *   fwd_mode 1 (layer = avg_pool2(conv(x))):  w_fwd in "in" order for desc.pool_out launches, w_dgrad in "out" order for
 *                                             the ups launch that computes its data gradient.
 *   fwd_mode 2 (layer = a STRIDE-2 SAME 3x3 convolution of an even-sized map, flax padding (0, 1): y[o] = sum_r w[r] x[2o + r]
 *               -- xmcgan/utils/resnet_v1.py:70): single taps instead of sums (7 of the 16 entries are zero); w_fwd for a
 *               pool_out launch with alpha = 4 (y at half the resolution), w_dgrad for the ups launch that computes the adjoint.
 * Both outputs: fragment order with 16 taps, rows * 16 * k bf16 elements (cout % 32 == 0, cin % 32 == 0); pass them
 * with desc.w_packed = 1 | 16.  Either may be NULL. */
int xmc_phase_conv_weight(const float* w, const float* inv_sigma, void* w_fwd, void* w_dgrad, int32_t cout, int32_t cin,
                          int32_t fwd_mode, void* stream);

/* ------------------------------------------------------------------------ dense / small GEMMs (K2)
 * C[b] = alpha * (*alpha_dev) * A[b] x B[b] + beta * C[b], float32, arbitrary element strides
 * (so NN / NT / TN need no copies); MFMA 32x32x2 f32 (exact fp32).  Replaces flax nn.Dense and
 * lax.dot_general (xmcgan/libml/layers.py:104-112), jnp.matmul in
 * xmcgan/libml/attention_lib.py:64-67,120,126,210,218.  alpha_dev may be NULL.
 * ws (may be NULL): xmc_gemm_ws_floats(m, n, k, batch, bf16_mfma) floats of scratch.  Products with few
 * output tiles and K >= 1024 split K over several workgroups ONLY when ws is given: each K range writes its
 * partial product to ws and a second kernel adds the partials in a fixed order (no float atomics). */
int64_t xmc_gemm_ws_floats(int32_t m, int32_t n, int32_t k, int32_t batch, int32_t bf16_mfma);
int xmc_gemm_f32(const float* a, const float* b, float* c, int32_t m, int32_t n, int32_t k,
                 int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn,
                 int64_t scb, int64_t ldc, float alpha, const float* alpha_dev, float beta,
                 int32_t batch, float* ws, void* stream);

/* Same contract, but the float32 operands are rounded to bf16 on their way into LDS and multiplied on
 * v_mfma_f32_32x32x16_bf16 (float32 accumulate): the bf16 training mode's region-word similarity GEMMs
 * (xmcgan/libml/attention_lib.py:143-166, B^2*R*T*E products).  The float32 parity mode never calls it. */
int xmc_gemm_f32_bf16mfma(const float* a, const float* b, float* c, int32_t m, int32_t n, int32_t k,
                 int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn,
                 int64_t scb, int64_t ldc, float alpha, const float* alpha_dev, float beta,
                 int32_t batch, float* ws, void* stream);

/* y[a][c] (+)= scale * sum_r f(x[a][r][c]),  f = relu or identity; x in `dtype`, y float32.
 * Bias gradients, the projection head's spatial SUM (xmcgan/nets/xmc_net.py:97-98) and
 * the tile/broadcast adjoints. */
int xmc_reduce_mid(const void* x, float* y, int64_t a, int64_t r, int64_t c, int32_t dtype,
                   int32_t relu, float scale, int32_t accumulate, void* stream);
/* Atomic-free two-stage variant (partial rows in `ws`, xmc_reduce_mid_ws_floats(a, r, c) floats, no
 * initialisation; fixed summation order -> bit-reproducible).  ws == NULL behaves like xmc_reduce_mid. */
int64_t xmc_reduce_mid_ws_floats(int64_t a, int64_t r, int64_t c);
int xmc_reduce_mid_ws(const void* x, float* y, float* ws, int64_t a, int64_t r, int64_t c, int32_t dtype,
                      int32_t relu, float scale, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------- batch norm + conditional affine (K3)
 * flax.linen.BatchNorm(use_scale=False, use_bias=False, momentum .9, eps 1e-5) as configured at
 * xmcgan/nets/xmc_net.py:192-201, fused with (Local)ConditionalBatchNorm's affine
 * x_hat * (gamma + 1) + beta (xmcgan/libml/layers.py:256-257,271-272) and the following ReLU.
 * sums: 2*C floats {sum, sum of squares}, ACCUMULATED (zero first). */
int xmc_bn_stats(const void* x, float* sums, int64_t pixels, int32_t c, int32_t dtype, void* stream);
int xmc_bn_finalize(const float* sums, float* mean, float* rstd, float* run_mean, float* run_var,
                    int64_t pixels, int32_t c, float eps, float momentum, int32_t update_running,
                    void* stream);
/* eval mode: mean/rstd from running statistics */
/* Batch statistics + finalize without same-address atomics (the path the step uses): <= 512 workgroups
 * write one row of partial [sum, sum of squares] each into `ws` (xmc_bn_stats_ws_floats(pixels, c) floats,
 * no initialisation needed), a second small kernel reduces the rows in a FIXED order -- bit-reproducible
 * statistics -- and produces mean / rstd / the running statistics exactly as xmc_bn_finalize. */
int64_t xmc_bn_stats_ws_floats(int64_t pixels, int32_t c);
int xmc_bn_batch_stats(const void* x, float* ws, float* mean, float* rstd, float* run_mean, float* run_var,
                       int64_t pixels, int32_t c, int32_t dtype, float eps, float momentum,
                       int32_t update_running, void* stream);
int xmc_bn_from_running(const float* run_mean, const float* run_var, float* mean, float* rstd,
                        int32_t c, float eps, void* stream);
/* gamma/beta: one row of c values per conditioning cell (n * hc * hc cells, hc | h; hc == 1:
 * per-sample conditional BN), rows `cstride` ELEMENTS apart (cstride >= c): gamma and beta are normally
 * the two halves of ONE (cells, 2c) conv / dense output (gamma = p, beta = p + c, cstride = 2c).
 * gb_dtype = XMC_F32, or XMC_BF16 (round 5: what LocalConditionalBatchNorm's nn.Conv(dtype=bfloat16) produces in the
 * reference's bf16 mode, xmcgan/libml/layers.py:261-273; needs dtype == XMC_BF16, c % 8 == 0, cstride % 8 == 0 and
 * 16-byte aligned gamma / beta -- and, in the backward passes, dgamma / dbeta of the same type). */
int xmc_cbn_act_fwd(const void* x, const float* mean, const float* rstd, const void* gamma,
                    const void* beta, void* y, int32_t n, int32_t h, int32_t w, int32_t c,
                    int32_t hc, int32_t cstride, int32_t relu, int32_t dtype, int32_t gb_dtype, void* stream);
/* The same with an MX-fp8 twin of y (bf16, c % 64 == 0): y8 [pixels][c / 64][80] as xmc_mx8_quantize(y, 0) would write
 * it, for the 3x3 convolution that consumes y when config.conv_fp8 is set. */
int xmc_cbn_act_fwd_mx8(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                        void* y, void* y8, int32_t n, int32_t h, int32_t w, int32_t c, int32_t hc, int32_t cstride,
                        int32_t relu, void* stream);
/* pass 1: dgamma/dbeta per conditioning cell (exclusive writes, same row stride as gamma/beta) */
int xmc_cbn_act_bwd_cells(const void* dy, const void* x, const float* mean, const float* rstd,
                          const void* gamma, const void* beta, void* dgamma, void* dbeta,
                          int32_t n, int32_t h, int32_t w, int32_t c, int32_t hc, int32_t cstride,
                          int32_t relu, int32_t dtype, int32_t gb_dtype, void* stream);
/* s[0:C] = sum_cells (gamma+1)*dbeta ; s[C:2C] = sum_cells (gamma+1)*dgamma.  Two-stage through `ws`
 * (xmc_cbn_bwd_sums_ws_floats(cells, c) floats; neither ws nor s needs initialising): workgroups write partial
 * rows, a fixed-order row reduction writes s -- atomic-free, bit-reproducible. */
int64_t xmc_cbn_bwd_sums_ws_floats(int64_t cells, int32_t c);
int xmc_cbn_bwd_sums(const void* gamma, const void* dgamma, const void* dbeta, float* s, float* ws,
                     int64_t cells, int32_t c, int32_t cstride, int32_t gb_dtype, void* stream);
/* pass 2: dx = rstd * (g*(gamma+1) - s1/P - x_hat * s2/P) */
int xmc_cbn_act_bwd_dx(const void* dy, const void* x, const float* mean, const float* rstd,
                       const void* gamma, const void* beta, const float* s, void* dx,
                       int32_t n, int32_t h, int32_t w, int32_t c, int32_t hc, int32_t cstride,
                       int32_t relu, int32_t dtype, int32_t gb_dtype, void* stream);

/* ------------------------------------------------------------------------------ resampling / pointwise
 * y = scale * sum_{2x2} x (+ res): dsample (xmcgan/nets/common.py:23-55) with scale .25 and the
 * adjoint of nearest upsample with scale 1. */
int xmc_pool2(const void* x, const void* res, void* y, int32_t n, int32_t h, int32_t w, int32_t c,
              float scale, int32_t dtype, void* stream);
/* xmc_pool2 that also writes xr = max(x, 0) at full resolution (xr NULL: xmc_pool2).  A down-sampling DiscBlock
 * (xmcgan/nets/common.py:67-78) pools its input x for the shortcut and feeds relu(x) to its first convolution; the
 * pooling pass holds every element of x, so it emits relu(x) once and neither the convolution nor its weight gradient
 * applies the ReLU again. */
int xmc_pool2_relu(const void* x, const void* res, void* y, void* xr, int32_t n, int32_t h, int32_t w, int32_t c,
                   float scale, int32_t dtype, void* stream);
/* y (n,h,w,32)[tap*c + j] = x (n,h,w,c)[pixel + sign * offset(tap)][j], zero outside the image and for the
 * padding channels (ks*ks*c <= 32): the im2col of an RGB-like tensor (sign = +1), or the shifted copies of a
 * 3-channel output gradient (sign = -1).  Lets the 3-channel first / last convolutions
 * (xmcgan/nets/common.py:127, xmcgan/nets/xmc_net.py:245) and their weight gradients run as 1x1 convolutions
 * on the MFMA kernels. */
int xmc_expand_taps(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks,
                    int32_t sign, int32_t dtype, void* stream);
/* adjoint of xmc_reduce_mid(relu=1): dx[a][r][c] = x[a][r][c] > 0 ? dpool[a][c] : 0
 * (backward of activation_fn + jnp.sum(x, axis=(1,2)), xmcgan/nets/xmc_net.py:97-98). */
int xmc_bcast_relu_bwd(const float* dpool, const void* x, void* dx, int64_t a, int64_t r, int64_t c,
                       int32_t dtype, void* stream);
/* (tanh(x)+1)/2 (xmcgan/nets/xmc_net.py:246-247) */
int xmc_tanh_out_fwd(const void* x, void* y, int64_t n, int32_t dtype, void* stream);
/* dx = dy * 0.5 * (1 - (2y-1)^2) */
int xmc_tanh_out_bwd(const void* dy, const void* y, void* dx, int64_t n, int32_t dtype, void* stream);
int xmc_cast(const void* x, int32_t dtype_in, void* y, int32_t dtype_out, int64_t n, void* stream);
/* out = a + b (same dtype) */
int xmc_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream);

/* ---------------------------------------------------------------------------------- attention (K7)
 * attention_for_g, xmcgan/libml/attention_lib.py:194-219 with the mask of
 * xmcgan/nets/xmc_net.py:225-228: region (b, r, e) in `dtype`; words_n (b, t, e) float32 already
 * l2-normalised; ctx (b, r, e) in `dtype`; attn (b, r, t) float32; rinv (b, r) float32. */
int xmc_attn_g_fwd(const void* region, const float* words_n, const float* max_len, void* ctx,
                   float* attn, float* rinv, int32_t b, int32_t r, int32_t t, int32_t e,
                   float gamma, int32_t dtype, void* stream);
int xmc_attn_g_bwd(const void* dctx, const void* region, const float* words_n, const float* attn,
                   const float* rinv, void* dregion, int32_t b, int32_t r, int32_t t, int32_t e,
                   float gamma, int32_t dtype, void* stream);
/* attention_for_g on the matrix cores (bf16 tensors; r % 128 == 0, e % 64 == 0, t <= 32 -- xmc_attn_g_mfma_supported): both
 * products as MFMA 32x32x16 tiles with the words as the A operand, softmax in registers, float32 accumulation; same
 * arguments and results as xmc_attn_g_fwd / xmc_attn_g_bwd with dtype = XMC_BF16 (the float32 parity mode keeps those: the
 * attention indices are held bit-exact against the float32 oracle there). */
int xmc_attn_g_mfma_supported(int32_t b, int32_t r, int32_t t, int32_t e);
int xmc_attn_g_fwd_mfma(const void* region, const float* words_n, const float* max_len, void* ctx, float* attn,
                        float* rinv, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream);
int xmc_attn_g_bwd_mfma(const void* dctx, const void* region, const float* words_n, const float* attn, const float* rinv,
                        void* dregion, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream);
/* The same two launches with ctx / dctx as a COLUMN SLICE of a wider tensor (row pitch ld_* elements, % 8 == 0, >= e): the
 * generator concatenates the context with the global condition along the channel axis (xmcgan/nets/xmc_net.py:231-235) --
 * the forward writes its context straight into that tensor, the backward reads its cotangent out of that tensor's gradient,
 * and neither a concatenation nor a contiguous copy of the slice is made. */
int xmc_attn_g_fwd_mfma_ld(const void* region, const float* words_n, const float* max_len, void* ctx, int32_t ld_ctx, float* attn,
                           float* rinv, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream);
int xmc_attn_g_bwd_mfma_ld(const void* dctx, int32_t ld_dctx, const void* region, const float* words_n, const float* attn,
                           const float* rinv, void* dregion, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream);


/* l2_normalize along the last axis, xmcgan/libml/attention_lib.py:30-33:
 * y = x * rsqrt(max(sum x^2, 1e-12)); x in dtype_in, y float32, inv (rows) float32. */
int xmc_l2norm_rows_fwd(const void* x, float* y, float* inv, int64_t rows, int32_t cols,
                        int32_t dtype_in, void* stream);
/* dx (dtype_out) = inv * (dy - y * <y, dy>)  (or inv * dy when the clamp was active) */
int xmc_l2norm_rows_bwd(const float* dy, const float* y, const float* inv, void* dx, int64_t rows,
                        int32_t cols, int32_t dtype_out, void* stream);

/* ------------------------------------------------------------------------- word-level loss (K8)
 * word_loss / attention, xmcgan/libml/attention_lib.py:105-191, restructured:
 *   S[(j,r)][(i,t)] = R^_j[r] . W^_i[t]  (GEMM),  G_j = R^_j R^_j^T  (GEMM),
 *   alpha = softmax_r(gamma1 * S + mask),  nn = sum_r alpha S,  H = G_j alpha (GEMM),
 *   q = sum_r alpha H,  cos = nn / sqrt(q)   (== cosine_similarity(word, context)).
 * Matrices are float32 (b*r) x (b*t), row-major.  max_len (b) float32. */
int xmc_wl_softmax(const float* s, const float* max_len, float* alpha, float* nn, int32_t b,
                   int32_t r, int32_t t, float gamma1, void* stream);
int xmc_wl_qdot(const float* alpha, const float* h, float* q, int32_t b, int32_t r, int32_t t,
                void* stream);
/* sim_t[i][j] = gamma3/gamma2 * logsumexp_t(gamma2 * nn/sqrt(q) + mask); pi = softmax_t(...) */
int xmc_wl_rows(const float* nn, const float* q, const float* max_len, float* sim_t, float* pi,
                int32_t b, int32_t t, float gamma2, float gamma3, void* stream);
/* backward of the column stage: given dsim_t (b x b), writes dS (in place over h) and
 * alpha_scaled = alpha * dq (for dG_j = alpha_scaled alpha^T). */
int xmc_wl_bwd_cols(const float* s, const float* alpha, float* h_ds, const float* nn, const float* q,
                    const float* pi, const float* dsim_t, float* alpha_scaled, int32_t b, int32_t r,
                    int32_t t, float gamma1, float gamma3, void* stream);

/* word_loss fused on the matrix cores (round 4; bf16 mode, r == 256, e % 128 == 0 -- xmc_wl_fused_supported): the same
 * quantities as the entries above (xmcgan/libml/attention_lib.py:105-127 `attention`, :130-191 `word_loss`) without the
 * float32 (b*r) x (b*t) score / probability / H tensors.  ldp = xmc_wl_fused_ldp(b, t) = b*t rounded up to 64: the padded
 * column count of every [.., ldp] tensor below (padding columns hold zeros).
 *   xmc_wl_prep_regions  x (b, r, e) bf16 -> rn = l2_normalize(x) (:30-33) bf16 (b, r, e), rnT (b, e, r), rinv (b*r) float32
 *   xmc_wl_prep_words    words_n (ld = b*t, e) float32, already normalised -> w (ldp, e) bf16 (rows >= ld zero), wT (e, ldp)
 *   xmc_wl_tn_gemm       out[z][x][y] = alpha * (sum_k x0[z][x][k] y0[z][y][k] + sum_k x1[z][x][k] y1[z][y][k]); bf16 operands
 *                        with k contiguous (row pitches ld*, batch strides s* in elements, 0 = shared); k0, k1 % 64 == 0
 *                        (k1 = 0: one segment); rows_x, rows_y % 128 == 0; out bf16 or float32 (out_f32), row pitch ldo
 *   xmc_wl_cols_fwd      per (image j, column (i,t)): S = R^_j W^^T, alpha = softmax over the r regions of gamma1 * S (+ mask),
 *                        nn = sum alpha S, q = alpha^T G_j alpha with g = R^_j R^_j^T (b, r, r) bf16 -> nn, q (b, b*t) float32,
 *                        the inputs of xmc_wl_rows
 *   xmc_wl_cols_bwd      recomputes S / alpha / H and writes dS, alpha * dq and alpha as bf16 (b, r, ldp) given dsim_t (b, b)
 *                        and pi (b, b*t) (same formulas as xmc_wl_bwd_cols)
 *   xmc_l2norm_rows_bwd_bf16y   xmc_l2norm_rows_bwd with the normalised rows y in bf16 */
int xmc_wl_fused_supported(int32_t b, int32_t r, int32_t t, int32_t e);
int xmc_wl_fused_ldp(int32_t b, int32_t t);
int xmc_wl_prep_regions(const void* x, void* rn, void* rnT, float* rinv, int32_t b, int32_t r, int32_t e, void* stream);
int xmc_wl_prep_words(const float* words_n, void* w, void* wT, int32_t ld, int32_t ldp, int32_t e, void* stream);
int xmc_wl_tn_gemm(const void* x0, int64_t sx0, int32_t ldx0, const void* y0, int64_t sy0, int32_t ldy0, int32_t k0,
                   const void* x1, int64_t sx1, int32_t ldx1, const void* y1, int64_t sy1, int32_t ldy1, int32_t k1,
                   void* out, int64_t so, int32_t ldo, int32_t out_f32, float alpha, int32_t rows_x, int32_t rows_y,
                   int32_t batch, void* stream);
int xmc_wl_cols_fwd(const void* rn, const void* w, const void* g, const float* max_len, float* nn, float* q, int32_t b,
                    int32_t t, int32_t e, int32_t ldp, float gamma1, void* stream);
int xmc_wl_cols_bwd(const void* rn, const void* w, const void* g, const float* max_len, const float* dsim_t,
                    const float* pi, void* ds, void* as, void* al, int32_t b, int32_t t, int32_t e, int32_t ldp,
                    float gamma1, float gamma3, void* stream);
int xmc_l2norm_rows_bwd_bf16y(const float* dy, const void* y, const float* inv, void* dx, int64_t rows, int32_t cols,
                              int32_t dtype_out, void* stream);

/* --------------------------------------------------------------- contrastive / GAN scalar losses (K9, K11)
 * Symmetric cross-entropy with identity labels over a b x b logit matrix L (row direction)
 * and its transpose: contrastive_loss / word_loss tails, xmcgan/libml/attention_lib.py:61-74,
 * :175-182, xmcgan/libml/losses.py:47-51.  *loss += weight * (mean_i CE(L[i,:], i) + mean_i
 * CE(L[:,i], i)); dlogits (may be NULL) = weight * d loss / dL; stats (may be NULL) receives
 * get_statistics (attention_lib.py:36-43) averaged over both directions: {accuracy, entropy}. */
int xmc_xent_sym(const float* logits, int32_t b, float weight, float* loss, float* dlogits,
                 float* stats, void* stream);
/* out[4] = {d_loss, g_loss, c_loss_d, c_loss_g} (xmcgan/xmc_gan.py:58-71,146-154): loss_vec[5] = {fake word, real word, fake sentence,
 * real sentence, image contrastive}, hinge[2] = {hinge_d, hinge_g}. */
int xmc_loss_assemble(const float* loss_vec, const float* hinge, float* out, void* stream);
/* contrastive_loss (xmcgan/libml/attention_lib.py:46-79) without the normalised copies and the GEMM launches (round 5):
 * xmc_cl_logits: logits[i][j] = <a_i, b_j> / (|a_i| |b_j|) * inv_temperature for a, b (n, d) float32, with l2_normalize's clamp
 * (attention_lib.py:30-33); ainv / binv (n) receive 1 / |row| for the backward pass.  Follow with xmc_xent_sym.
 * xmc_cl_bwd: the pullback onto ONE operand x (trans = 0: x = a, coefficients dlogits[i][j]; trans = 1: x = b, dlogits[j][i]),
 * through the normalisation: out[i] (+)= xinv_i (g_i - xn_i <xn_i, g_i>), g_i = inv_temperature * sum_j dl(i, j) yinv_j y_j.
 * d <= 2048. */
int xmc_cl_logits(const float* a, const float* b, float* logits, float* ainv, float* binv, int32_t n, int32_t d,
                  float inv_temperature, void* stream);
int xmc_cl_bwd(const float* dlogits, const float* x, const float* y, const float* xinv, const float* yinv, float* out,
               int32_t n, int32_t d, float inv_temperature, int32_t trans, int32_t accumulate, void* stream);
/* hinge_loss, xmcgan/libml/losses.py:30-35: logit (2b) = [real; fake].
 * *d_loss += mean(relu(1-real)+relu(1+fake)); *g_loss += -mean(fake); gradients (2b) each. */
int xmc_hinge(const float* logit, int32_t b, float* d_loss, float* g_loss, float* dlogit_d,
              float* dlogit_g, void* stream);
/* projection head, xmcgan/nets/xmc_net.py:99-104: out[n] = bias + sum_c pool[n][c] *
 * (w[c] * (*inv_sigma) + emb[n % b][c]) */
int xmc_proj_head_fwd(const float* pool, const float* w, const float* inv_sigma, const float* bias,
                      const float* emb, float* out, int32_t n2, int32_t b, int32_t c, void* stream);
/* dpool[n][c] (+)= dout[n] * (w[c]*inv_sigma + emb[n%b][c]); demb[i][c] (+)= sum_{n%b==i} dout[n]*pool[n][c] */
int xmc_proj_head_bwd(const float* dout, const float* pool, const float* w, const float* inv_sigma,
                      const float* emb, float* dpool, float* demb, int32_t n2, int32_t b, int32_t c,
                      int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------ spectral norm (K6)
 * One power-iteration step, xmcgan/libml/layers.py:92-101, :209-220, on W viewed as
 * rows x cols float32 with u0 along `u_axis` (0: u over rows -- conv master [cout][K];
 * 1: u over cols -- dense kernels (in, out)).  Writes v (other axis), u_new, and
 * scal = {sigma, 1/(sigma+eps)}.  tmp: rows + cols + 4 floats of scratch. */
int xmc_spectral_power_iter(const float* w, const float* u0, float* u_new, float* v, float* scal,
                            float* tmp, int32_t rows, int32_t cols, int32_t u_axis, float eps,
                            void* stream);
/* Gradient through sigma: g <- (g - (<g,w> * inv_s) * outer(u, v)) * inv_s  in place
 * (u indexed along u_axis).  tmp: 1 float of scratch. */
int xmc_spectral_grad_fix(float* g, const float* w, const float* u, const float* v,
                          const float* scal, float* tmp, int32_t rows, int32_t cols, int32_t u_axis,
                          void* stream);

/* Batched form over ALL spectrally-normalised weights of a network (one descriptor table in device
 * memory; 4 + 1 launches per forward, 2 per backward instead of ~9 per weight).  Offsets are in floats
 * into the parameter / gradient arena (w_off), the flat u / v buffers (u_off, v_off) and the prepared
 * weight buffers (wf_off, wd_off, in elements).  blk_a / blk_b / blk_p are exclusive prefix sums of the
 * workgroups each entry owns in the three grids (host-computed, see ops.py::SpectralBank):
 *   A: ceil(rows / 4)                        B: ceil(cols / 128)   (one workgroup per 128 columns, all rows)
 *   P (prep table): taps * ceil(cin/32) * ceil(cout/32) for conv entries, 0 otherwise
 *   P (grad-fix table, a second copy of the table): ceil(rows * cols / 65536)
 * No float atomics anywhere: every matvec output element has one writer, <g,w> is kept as one partial per
 * 64K-element chunk (`dots`: `blocks` floats) and summed in a fixed order -- bit-reproducible.  u / v slices
 * (u_off, v_off) must start 16-byte aligned. */
typedef struct {
    int64_t w_off;
    int32_t rows, cols, u_axis;
    int32_t u_off, v_off;
    int32_t blk_a, blk_b;
    int32_t taps, is_conv;
    int64_t wf_off, wd_off;
    int32_t blk_p;
    int32_t packed;           /* as xmc_prep_conv_weight: bit 0 forward, bit 1 dgrad copy in fragment order; bit 2: a layer
                                 next to a 2x resampling whose launches read only the 16-tap copies of xmc_phase_conv_weight --
                                 xmc_sn_batched_prep leaves its 3x3 copies UNWRITTEN when bit 8 of its `dtype` argument is set */
} xmc_sn_entry;

int xmc_sn_batched_power_iter(const void* table, int32_t n, const float* params, const float* u0,
                              float* u_new, float* v, float* u_raw, float* scal, int32_t blocks_a,
                              int32_t blocks_b, int32_t nu_total, int32_t nv_total, float eps,
                              void* stream);
int xmc_sn_batched_prep(const void* table, int32_t n, const float* params, const float* scal,
                        void* wf_buf, void* wd_buf, int32_t blocks_p, int32_t dtype, void* stream);
int xmc_sn_batched_grad_fix(const void* table, int32_t n, const float* params, float* grads,
                            const float* u, const float* v, const float* scal, float* dots,
                            int32_t blocks, void* stream);

/* Round 4 -- prepared weights as a pure cast of W (1 / sigma rides in xmc_conv_desc.alpha_dev): one pass over each weight's
 * float32 master writes the fragment-ordered forward / data-gradient copies, the 16-tap phase copies of the layers next to a
 * 2x resampling (xmc_phase_conv_weight's outputs, modes 0 / 1) and -- for spectrally-normalised weights -- the row tiles'
 * partial sums of the power iteration's first product v_raw = W^T u0 (xmcgan/libml/layers.py:209-214).  Table entries: bf16
 * mode only, cout % 32 == cin % 32 == 0, taps 1 or 9.  part: sum over entries of (cout / 32) * taps * cin floats. */
typedef struct {
    int64_t w_off;            /* floats into the parameter arena ([cout][taps][cin] master) */
    int64_t wf_off, wd_off;   /* bf16 elements into the plain forward / data-gradient buffers */
    int64_t pf_off, pd_off;   /* bf16 elements into the phase forward / data-gradient buffers */
    int64_t part_off;         /* floats into `part` */
    int32_t cout, cin, taps;
    int32_t blk0;             /* first workgroup of the entry in the tile grid ((cout / 32) * (cin / 32) workgroups each) */
    int32_t flags;            /* bit 0 / 1: write the plain forward / data-gradient copy; bits 2-3: 0 no phase copies, 1 the
                                 layer is conv3x3(upsample2(.)), 2 avg_pool2(conv3x3(.)); bit 4: write the W^T u0 partials */
    int32_t u_off, v_off;     /* slices of the flat u0 / v buffers (floats) */
    int32_t blk_c;            /* first workgroup of the entry in the column-sum grid (ceil(taps * cin / 256) workgroups each) */
} xmc_wprep_entry;
int xmc_wprep_batched(const void* table, int32_t n, const float* params, const float* u0, void* wf_buf, void* wd_buf,
                      void* pf_buf, void* pd_buf, float* part, int32_t blocks, void* stream);
/* The power iteration of xmc_sn_batched_power_iter with the first product of the `wtab` weights taken from `part`
 * (xmc_wprep_batched) and that of the remaining weights (`irr`: an xmc_sn_entry table with its own blk_a / blk_b prefixes;
 * n_irr may be 0) computed here; `table` lists all n weights. */
int xmc_sn_power_iter_fused(const void* table, int32_t n, int32_t blocks_a, int32_t blocks_b, const void* irr, int32_t n_irr,
                            int32_t irr_blocks_a, int32_t irr_blocks_b, const void* wtab, int32_t n_w, int32_t blocks_c,
                            const float* params, const float* u0, const float* part, float* u_new, float* v, float* u_raw,
                            float* scal, float eps, void* stream);
/* kvec[i] = <G_i, W_i> / (sigma_i + eps) (table = the dot-chunk table of xmc_sn_batched_grad_fix): first half of the gradient
 * through sigma; the second half is applied by xmc_adam_ema_dev_sn while it reads the gradient. */
int xmc_sn_batched_dot(const void* table, int32_t n, const float* params, const float* grads, const float* scal,
                       float* dots, float* kvec, int32_t blocks, void* stream);

/* ---------------------------------------------------------------------------------- optimiser (K12)
 * flax.optim.Adam.apply_gradient (xmcgan/xmc_gan.py:172-173,252) over a flat float32 arena, with
 * the 1/world gradient scale of lax.pmean (xmc_gan.py:170-171,251) and the EMA of
 * xmc_gan.py:174-177 fused.  ema may be NULL.  c1 = 1-beta1^t, c2 = 1-beta2^t. */
int xmc_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr,
                 float beta1, float beta2, float eps, float c1, float c2, float grad_scale,
                 float ema_decay, void* stream);
/* Same update with the optimiser's step counter in DEVICE memory so that a captured hipGraph replays
 * consecutive steps: step_state is 4 float32 slots, [0] = t as int32 bits (0 before the first step),
 * [1], [2] = 1/(1-beta1^t), 1/(1-beta2^t), refreshed (in double) by a one-thread kernel launched in
 * front of the update (beta1 / beta2 are doubles so that 1-beta^t matches the host formula).  Each call
 * advances t by one. */
int xmc_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr,
                     double beta1, double beta2, float eps, float* step_state, float grad_scale,
                     float ema_decay, void* stream);

/* xmc_adam_ema_dev that also (a) overwrites the gradient it consumed with zeros (zero_grads == 1; 2: writes the gradient with
 * the sigma term applied back instead, 0: leaves it) and (b) applies the
 * gradient through sigma of the spectrally-normalised tensors while reading it (map != NULL: one int16 per 64 arena elements
 * = index of the owning entry of `table`, or -1, or -2 = "leave this tensor alone" (xmc_adam_wprep_tiles updates it); kvec
 * from xmc_sn_batched_dot; u, v, scal of the forward's power iteration):
 * G <- (G - kvec_i u (x) v) / (sigma_i + eps), xmcgan/libml/layers.py:217-219.  `map` without `table`: skip marks only. */
int xmc_adam_ema_dev_sn(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, double beta1,
                        double beta2, float eps, float* step_state, float grad_scale, float ema_decay,
                        int32_t zero_grads, const void* map, const void* table, int32_t n_entries,
                        const float* kvec, const float* scal, const float* u, const float* vv, void* stream);
/* Round 5 -- the optimiser emits the prepared weights (W-bar's copies are a pure function of W, xmcgan/libml/layers.py:209-221,
 * and the optimiser holds the new W in registers): the Adam (+ EMA) update of the weights of an xmc_wprep_batched `table`,
 * element for element the arithmetic of xmc_adam_ema_dev_sn (gradient through sigma, zero_grads modes), followed by that
 * table's preparation from the UPDATED tiles -- fragment-ordered / phase copies and, for spectral entries (flags bit 4, bank
 * index in flags >> 8), W_new^T u as partial rows (`u` = this half step's new u = the next power iteration's u0).  Call
 * xmc_adam_ema_dev_sn on the same arena FIRST, with a `map` that carries -2 on these tensors (it advances `step_state` and
 * leaves them alone; a map without `table` is allowed there: skip marks only).  kvec / scal / u / vv NULL: plain arena. */
int xmc_adam_wprep_tiles(const void* table, int32_t n, int32_t blocks, float* p, float* g, float* m, float* v, float* ema,
                         float lr, double beta1, double beta2, float eps, const float* step_state, float grad_scale,
                         float ema_decay, int32_t zero_grads, const float* kvec, const float* scal, const float* u,
                         const float* vv, void* wf_buf, void* wd_buf, void* pf_buf, void* pd_buf, float* part, void* stream);


/* -------------------------------------------------------- frozen ResNet-50 feature path
 * xmcgan/xmc_gan.py:74-90, xmcgan/utils/pretrained_model_utils.py:102-127, xmcgan/utils/resnet_v1.py:60-186.
 * ResNet's feature maps (112^2 .. 7^2) live on power-of-two CANVASES (valid region top-left, margin zero) so its
 * convolutions run on xmc_conv2d_nhwc; these entry points are the rest.  dtype = XMC_F32 | XMC_BF16.
 *  - xmc_resize_bilinear: jax.image.resize(..., "bilinear") of x (n, hs, ws, c) into the hd x wd region of the
 *    canvas y (n, hc, wc, c): triangle filter on half-pixel centres, widened by max(hs / hd, 1) (jax anti-aliases by
 *    default when shrinking, e.g. 256 px -> 224), weights normalised over the in-bounds taps; backward = 1: x is dy
 *    on the canvas, y receives dx.
 *  - xmc_stem_im2col: the 7x7 stride-2 SAME stem conv (resnet_v1.py:148-154) as im2col of the (n, hc, wc, 3) image
 *    canvas (valid hv x wv) into col (n, ho, wo, kp >= 147), k = tap * 3 + ch; backward = 1: col2im (x = dcol).
 *  - xmc_maxpool3x3s2: nn.max_pool((3,3), strides (2,2), "SAME") (resnet_v1.py:156) canvas -> half-size canvas; idx
 *    (uint8, shape of y, may be NULL) receives the in-window position 0..8 of each window's first maximum;
 *    xmc_maxpool3x3s2_bwd: the adjoint (the gradient of a window goes to that element, as XLA's select-and-scatter).
 *  - xmc_zero_margin: zero a canvas outside its valid region, in place.
 *  - xmc_subsample2: small[o] = large[2 o + off] (the stride-2 view of a stride-1 convolution: off = 1 for flax's
 *    3x3, 0 for its 1x1 SAME stride-2 convs); scatter = 1: the adjoint (zero insertion into `large`).
 *  - xmc_add_relu: o = relu(a + b) (post-activation residual, resnet_v1.py:86); xmc_relu_bwd: g = (dy + dy2) * (out > 0). */
int xmc_resize_bilinear(const void* x, void* y, int32_t n, int32_t hs, int32_t ws, int32_t c, int32_t hd,
                        int32_t wd, int32_t hc, int32_t wc, int32_t backward, int32_t dtype, void* stream);
int xmc_stem_im2col(const void* x, void* col, int32_t n, int32_t hc, int32_t wc, int32_t hv, int32_t wv,
                    int32_t ho, int32_t wo, int32_t kp, int32_t backward, int32_t dtype, void* stream);
/* The stem as ONE implicit-GEMM launch (round 6, bf16): y canvas (n, ho, wo, 64), valid (hov, wov), <- conv 7x7 stride 2 SAME (2 before,
 * 3 after) of the image canvas x (n, hc, wc, 3) whose valid rows are hv and whose margin is ZERO (xmc_resize_bilinear's output), plus
 * bias (the folded init_bn; no ReLU: xmcgan/utils/resnet_v1.py:148-156).  Replaces xmc_stem_im2col + the pointwise GEMM: the 160-wide
 * columns (587 MB per 112 images) are never written.  wfrag: 64 x 176 bf16 in MFMA A-fragment order [cout / 32][k-step 0..10][lane][8] --
 * element e of lane l of fragment (cb, ks) = W[cb * 32 + (l & 31)][ky][kx][ch] with ky * 24 + kx * 3 + ch = ks * 16 + (l >> 5) * 8 + e
 * (zero where kx * 3 + ch >= 21 or ky >= 7).  wc <= 256, wov <= 128.  Only the valid corner of y is written (the caller keeps the
 * margins zero, as for the COMPACT pointwise launches). */
int xmc_stem_conv7x7s2(const void* x, const void* wfrag, const float* bias, void* y, int32_t n, int32_t hc, int32_t wc,
                       int32_t hv, int32_t ho, int32_t wo, int32_t hov, int32_t wov, void* stream);
/* ... and its data gradient onto the image as ONE launch (jax.vjp of the same convolution; replaces the pointwise GEMM 64 -> 160 into
 * im2col columns + xmc_stem_im2col(backward = 1)): dx canvas (n, hc, wc, 3), valid corner 2 hov x 2 wov written (margin untouched), <-
 * ds canvas (n, ho, wo, 64) with valid (hov, wov) (margin not read); bf16.  Per low-resolution pixel (Y, X) the 2 x 2 image pixels it
 * covers are 12 outputs r = (2 py + px) * 3 + c of a 4 x 4-tap correlation over ds: dx[2Y + py][2X + px][c] = sum_{t, u = 0..3} sum_co
 * ds[Y + 1 - t][X + 1 - u][co] W[co][2t + py][2u + px][c].  wfrag: 64 x 1 KiB fragments [channel half][tap t * 4 + u][k-step 0..1][lane][8],
 * element e of lane l = W[co][2t + py][2u + px][c] with l & 31 = (2 py + px) * 3 + c (rows >= 12 and taps beyond 6: zero),
 * co = half * 32 + k-step * 16 + (l >> 5) * 8 + e.  wov <= 128. */
int xmc_stem_conv7x7s2_dgrad(const void* ds, const void* wfrag, void* dx, int32_t n, int32_t ho, int32_t wo, int32_t hov,
                             int32_t wov, int32_t hc, int32_t wc, void* stream);
int xmc_maxpool3x3s2(const void* x, void* y, void* idx, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv,
                     int32_t wv, int32_t dtype, void* stream);
int xmc_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int32_t n, int32_t hc, int32_t wc, int32_t c,
                         int32_t hv, int32_t wv, int32_t dtype, void* stream);
int xmc_zero_margin(void* x, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv, int32_t wv,
                    int32_t dtype, void* stream);
int xmc_subsample2(void* large, void* small, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t off,
                   int32_t scatter, int32_t dtype, void* stream);
int xmc_add_relu(const void* a, const void* b, void* o, int64_t n, int32_t dtype, void* stream);
int xmc_relu_bwd(const void* dy, const void* dy2, const void* out, void* g, int64_t n, int32_t dtype,
                 void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XMCGAN_HIP_H_ */
