// attention_for_g (reference xmcgan/libml/attention_lib.py:194-219) on the matrix cores -- bf16 training mode.
//
//   R^ = l2n(region), S = gamma * R^ W^^T + mask, P = softmax_t(S), ctx = P W^      (W^ = l2-normalised words, T <= 32)
//
// The round-1..3 kernel (losses.hip: attn_g_fwd / bwd_kernel) is a VALU kernel: one wave per region, T dot products of E
// elements each through wave reductions (72 / 104 us per launch).  Here both products are MFMA 32x32x16 bf16 tiles with the
// words as the A operand (rows = word t, padded to 32) and 32 regions of one image as the B operand (columns):
//   S^T[t][r] = sum_e W^[t][e] R[r][e]      K = E: the region rows stream from global memory as B fragments (one 16-byte load per
//                                           lane and k-step), the words sit in LDS ([32][E] bf16); the regions' squared norms
//                                           are accumulated from the same registers, so R^ is never materialised
//   softmax over t: a lane of the 32x32 C layout holds 16 of its region's 32 words, its partner lane (lane ^ 32) the rest
//   ctx^T[e][r] = sum_t W^[t][e] P[t][r]    K = 32: P goes from the C layout to the B-operand layout with four
//                                           v_permlane32_swap per k-step; the transposed words sit in LDS ([E][32] bf16)
// and the result leaves through the convolution kernels' epilogue (16 consecutive channels per lane, common.h).
// Backward: dP^T = W^ dctx^T and S^T again in one K loop (two accumulators on one A fragment), ds = P (dP - <P, dP>) gamma,
// dr^ = ds^T W^ as above, and the l2-normalisation's adjoint dR = iv (dr^ - R^ <R^, dr^>) with <R^, dr^> = sum_t ds[t] S^[t]
// (no second pass over dr^): alpha = iv and res_scale = -iv^2 <R^, dr^> on `region` in the shared epilogue.
// float32 accumulation everywhere; probabilities / scores never leave registers.  Domain: bf16, R % 128 == 0, E % 32 == 0,
// T <= 32 (the float32 parity mode keeps the VALU kernel: bit-exact attention indices against the float32 oracle).
#include "common.h"

namespace {

constexpr int WPITCH_PAD = 8;      // bf16 elements of padding per LDS row of W^ [32][E + pad]: rows land 16 bytes apart in the banks
constexpr int WT_PITCH = 40;       // bf16 elements per LDS row of W^^T [E][32 + 8]

struct AttnArgs {
    const bf16_t* region; const float* words_n; const float* max_len;
    bf16_t* ctx; float* attn; float* rinv;
    const bf16_t* dctx; bf16_t* dregion;
    int B, R, T, E;
    float gamma;
    int ld_ctx, ld_dctx;          // row pitch (elements) of ctx / dctx: E, or the pitch of a wider tensor they are a column slice of
};

__device__ __forceinline__ bf16x8 ld_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

// words of image b -> LDS: w[t][e] (row pitch E + 8) and wt[e][t] (row pitch 40), bf16, rows t >= T zero
__device__ __forceinline__ void stage_words(const float* __restrict__ wb, bf16_t* __restrict__ w, bf16_t* __restrict__ wt, int T, int E) {
    const int wp = E + WPITCH_PAD;
    for (int i = threadIdx.x; i < 32 * E; i += 256) {
        const int t = i / E, e = i - t * E;
        const float v = t < T ? wb[(size_t)min(t, T - 1) * E + e] : 0.f;
        const bf16_t h = f2bf(t < T ? v : 0.f);
        w[t * wp + e] = h;
        wt[e * WT_PITCH + t] = h;
    }
}

// C layout of one 32x32 block: register q of lane (l31, lhi) holds row (q & 3) + 8 * (q >> 2) + 4 * lhi, column l31
__device__ __forceinline__ int c_row(int q, int lhi) { return (q & 3) + 8 * (q >> 2) + 4 * lhi; }

// 16 values per lane in the C layout (rows = k index) -> the two B-operand fragments of a K = 32 product (k-steps 0, 1):
// lane (l31, lhi) of k-step kk needs rows 16 kk + 8 lhi .. + 7; it owns four of them, its partner lane the other four
__device__ __forceinline__ void c_to_b_frags(const float* v, bf16x8* f0, bf16x8* f1) {
    uint32_t o[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        float lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * kk + q]), __float_as_uint(v[8 * kk + 4 + q]), false, false);
            lo[q] = __uint_as_float(r[0]);           // lower lanes: own rows 16kk + 0..3;  upper lanes: partner's rows 16kk + 8..11
            hi[q] = __uint_as_float(r[1]);           // lower lanes: partner's rows 16kk + 4..7;  upper lanes: own rows 16kk + 12..15
        }
        o[kk][0] = pack_bf2(lo[0], lo[1]); o[kk][1] = pack_bf2(lo[2], lo[3]);
        o[kk][2] = pack_bf2(hi[0], hi[1]); o[kk][3] = pack_bf2(hi[2], hi[3]);
    }
    *f0 = __builtin_bit_cast(bf16x8, make_uint4(o[0][0], o[0][1], o[0][2], o[0][3]));
    *f1 = __builtin_bit_cast(bf16x8, make_uint4(o[1][0], o[1][1], o[1][2], o[1][3]));
}

// out^T[e][r] = sum_t wt[e][t] pf[t][r] for all E / 32 row blocks, stored through the convolution epilogue
__device__ __forceinline__ void times_words_store(const bf16_t* __restrict__ wt, bf16x8 pf0, bf16x8 pf1, int E, int l31, int lhi,
                                                  size_t obase, ConvEpi ep) {
    for (int eb = 0; eb < E / 32; ++eb) {
        const bf16_t* a = wt + (eb * 32 + l31) * WT_PITCH + lhi * 8;
        const bf16x8 a0 = ld_frag(a), a1 = ld_frag(a + 16);
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, pf0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, pf1, acc, 0, 0, 0);
        conv_epilogue_block(acc, eb * 32, lhi, obase, obase, ep);
    }
}

__global__ __launch_bounds__(256) void attn_g_mfma_fwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int E = p.E, T = p.T, wp = E + WPITCH_PAD;
    bf16_t* w = reinterpret_cast<bf16_t*>(smem);
    bf16_t* wt = w + 32 * wp;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    stage_words(p.words_n + (size_t)b * T * E, w, wt, T, E);
    __syncthreads();
    const long long row = (long long)b * p.R + blockIdx.x * 128 + wave * 32 + l31;          // this lane's region
    const bf16_t* rr = p.region + row * E + lhi * 8;
    const bf16_t* wa = w + l31 * wp + lhi * 8;
    f32x16 acc;
#pragma unroll
    for (int q = 0; q < 16; ++q) acc[q] = 0.f;
    float ss = 0.f;
    for (int k0 = 0; k0 < E; k0 += 64) {             // four k-steps per trip: the region fragments are requested together
        bf16x8 bfr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) bfr[u] = ld_frag(rr + k0 + 16 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ld_frag(wa + k0 + 16 * u), bfr[u], acc, 0, 0, 0);
            const uint4 raw = __builtin_bit_cast(uint4, bfr[u]);
            const uint32_t d[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float lo = __uint_as_float(d[q] << 16), hi = __uint_as_float(d[q] & 0xffff0000u);
                ss += lo * lo + hi * hi;
            }
        }
    }
    ss += __shfl_xor(ss, 32);
    const float iv = rsqrtf(fmaxf(ss, 1e-12f));
    const float ml = p.max_len[b];
    float s[16], mx = -INFINITY;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int t = c_row(q, lhi);
        float v = acc[q] * iv * p.gamma;
        v = v + (((float)t >= ml) ? 1.0f : 0.0f) * (-1e9f);      // mask * (-1e9), as the reference adds it
        s[q] = t < T ? v : -INFINITY;
        mx = fmaxf(mx, s[q]);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32));
    float sum = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) { s[q] = expf(s[q] - mx); sum += s[q]; }      // exp(-inf) = 0 for the padded words
    sum += __shfl_xor(sum, 32);
    const float is = 1.f / sum;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        s[q] *= is;
        const int t = c_row(q, lhi);
        if (t < T) p.attn[row * T + t] = s[q];
    }
    if (lhi == 0) p.rinv[row] = iv;
    bf16x8 pf0, pf1;
    c_to_b_frags(s, &pf0, &pf1);
    ConvEpi ep;
    ep.bias = nullptr; ep.mask = nullptr; ep.res = nullptr; ep.y = p.ctx;
    ep.Cout = E; ep.out_f32 = 0; ep.alpha = 1.f; ep.res_scale = 0.f;
    times_words_store(wt, pf0, pf1, E, l31, lhi, (size_t)row * p.ld_ctx, ep);
}

__global__ __launch_bounds__(256) void attn_g_mfma_bwd_kernel(const AttnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int E = p.E, T = p.T, wp = E + WPITCH_PAD;
    bf16_t* w = reinterpret_cast<bf16_t*>(smem);
    bf16_t* wt = w + 32 * wp;
    const int b = blockIdx.y, lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, lhi = lane >> 5;
    stage_words(p.words_n + (size_t)b * T * E, w, wt, T, E);
    __syncthreads();
    const long long row = (long long)b * p.R + blockIdx.x * 128 + wave * 32 + l31;
    const bf16_t* rr = p.region + row * E + lhi * 8;
    const bf16_t* dr = p.dctx + row * p.ld_dctx + lhi * 8;
    const bf16_t* wa = w + l31 * wp + lhi * 8;
    f32x16 sacc, dacc;
#pragma unroll
    for (int q = 0; q < 16; ++q) { sacc[q] = 0.f; dacc[q] = 0.f; }
    for (int k0 = 0; k0 < E; k0 += 32) {
        bf16x8 rf[2], df[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { rf[u] = ld_frag(rr + k0 + 16 * u); df[u] = ld_frag(dr + k0 + 16 * u); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const bf16x8 a = ld_frag(wa + k0 + 16 * u);
            sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, rf[u], sacc, 0, 0, 0);
            dacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, df[u], dacc, 0, 0, 0);
        }
    }
    const float iv = p.rinv[row];
    float pr[16], pdp = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int t = c_row(q, lhi);
        const float a = p.attn[row * T + min(t, T - 1)];
        pr[q] = t < T ? a : 0.f;
        pdp += pr[q] * dacc[q];
    }
    pdp += __shfl_xor(pdp, 32);
    float ds[16], dot = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        ds[q] = pr[q] * (dacc[q] - pdp) * p.gamma;   // d loss / d (r^ . w^_t)
        dot += ds[q] * sacc[q];
    }
    dot += __shfl_xor(dot, 32);
    dot *= iv;                                       // <r^, dr^> = sum_t ds[t] (r^ . w^_t)
    const bool clamped = iv >= 999999.0f;            // sum x^2 <= 1e-12: r^ = x * 1e6, no norm term (l2norm_bwd_kernel)
    bf16x8 pf0, pf1;
    c_to_b_frags(ds, &pf0, &pf1);
    ConvEpi ep;
    ep.bias = nullptr; ep.mask = nullptr; ep.res = p.region; ep.y = p.dregion;
    ep.Cout = E; ep.out_f32 = 0; ep.alpha = iv; ep.res_scale = clamped ? 0.f : -iv * iv * dot;
    times_words_store(wt, pf0, pf1, E, l31, lhi, (size_t)row * E, ep);
}

bool attn_domain(int b, int r, int t, int e) { return b > 0 && r > 0 && (r % 128) == 0 && t > 0 && t <= 32 && e >= 64 && (e % 64) == 0; }
size_t attn_lds(int e) { return (size_t)(32 * (e + WPITCH_PAD) + e * WT_PITCH) * sizeof(bf16_t); }

}  // namespace

extern "C" int xmc_attn_g_mfma_supported(int32_t b, int32_t r, int32_t t, int32_t e) {
    return attn_domain(b, r, t, e) && attn_lds(e) <= 160 * 1024 ? 1 : 0;
}

static int attn_optin() {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(&attn_g_mfma_fwd_kernel), reinterpret_cast<const void*>(&attn_g_mfma_bwd_kernel)}, 160 * 1024)
               ? XMC_OK : XMC_EINVAL;
}

extern "C" int xmc_attn_g_fwd_mfma_ld(const void* region, const float* words_n, const float* max_len, void* ctx, int32_t ld_ctx,
                                      float* attn, float* rinv, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream) {
    XMC_REQUIRE(region && words_n && max_len && ctx && attn && rinv);
    XMC_REQUIRE(xmc_attn_g_mfma_supported(b, r, t, e) && ld_ctx >= e && (ld_ctx % 8) == 0);
    XMC_REQUIRE(((uintptr_t)region % 16) == 0 && ((uintptr_t)ctx % 16) == 0);
    if (attn_optin() != XMC_OK) return XMC_EINVAL;
    AttnArgs a{};
    a.region = static_cast<const bf16_t*>(region); a.words_n = words_n; a.max_len = max_len;
    a.ctx = static_cast<bf16_t*>(ctx); a.attn = attn; a.rinv = rinv;
    a.B = b; a.R = r; a.T = t; a.E = e; a.gamma = gamma; a.ld_ctx = ld_ctx; a.ld_dctx = e;
    hipLaunchKernelGGL(attn_g_mfma_fwd_kernel, dim3((unsigned)(r / 128), (unsigned)b), dim3(256), attn_lds(e), static_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_attn_g_fwd_mfma(const void* region, const float* words_n, const float* max_len, void* ctx, float* attn,
                                   float* rinv, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream) {
    return xmc_attn_g_fwd_mfma_ld(region, words_n, max_len, ctx, e, attn, rinv, b, r, t, e, gamma, stream);
}

extern "C" int xmc_attn_g_bwd_mfma_ld(const void* dctx, int32_t ld_dctx, const void* region, const float* words_n, const float* attn,
                                      const float* rinv, void* dregion, int32_t b, int32_t r, int32_t t, int32_t e, float gamma,
                                      void* stream) {
    XMC_REQUIRE(dctx && region && words_n && attn && rinv && dregion);
    XMC_REQUIRE(xmc_attn_g_mfma_supported(b, r, t, e) && ld_dctx >= e && (ld_dctx % 8) == 0);
    XMC_REQUIRE(((uintptr_t)region % 16) == 0 && ((uintptr_t)dctx % 16) == 0 && ((uintptr_t)dregion % 16) == 0);
    if (attn_optin() != XMC_OK) return XMC_EINVAL;
    AttnArgs a{};
    a.region = static_cast<const bf16_t*>(region); a.words_n = words_n; a.attn = const_cast<float*>(attn); a.rinv = const_cast<float*>(rinv);
    a.dctx = static_cast<const bf16_t*>(dctx); a.dregion = static_cast<bf16_t*>(dregion);
    a.B = b; a.R = r; a.T = t; a.E = e; a.gamma = gamma; a.ld_ctx = e; a.ld_dctx = ld_dctx;
    hipLaunchKernelGGL(attn_g_mfma_bwd_kernel, dim3((unsigned)(r / 128), (unsigned)b), dim3(256), attn_lds(e), static_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_attn_g_bwd_mfma(const void* dctx, const void* region, const float* words_n, const float* attn, const float* rinv,
                                   void* dregion, int32_t b, int32_t r, int32_t t, int32_t e, float gamma, void* stream) {
    return xmc_attn_g_bwd_mfma_ld(dctx, e, region, words_n, attn, rinv, dregion, b, r, t, e, gamma, stream);
}
