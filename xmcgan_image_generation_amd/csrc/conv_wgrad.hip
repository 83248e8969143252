// Weight gradient of the NHWC implicit-GEMM convolution for gfx950:
//   dW[cout][j] += alpha * sum_p dY'(p, cout) * A(p, j),    j = (tap, cin)
// GEMM view: rows = cout, cols = j, reduction over PIXELS.  Both operands are stored
// channel-contiguous (NHWC) while MFMA wants the reduction index contiguous per lane, so the
// [pixel][channel] tiles staged in LDS are read transposed:
//   bf16: ds_read_b64_tr_b16 (hardware 4x4-block transpose read, TR_READ=1) or eight strided
//         ds_read_u16 per fragment (TR_READ=0, the bring-up fallback);
//   f32 : v_mfma_f32_32x32x2_f32 takes one element per lane, so [pixel][channel] is already
//         the conflict-free layout.
// The pixel range is split across blockIdx.y (split-K); partial tiles are combined with
// float32 atomic adds into the (pre-zeroed / accumulating) master-layout gradient.
#include <cstdlib>

#include "common.h"

namespace {

template <typename T, bool TR> struct WT;
template <bool TR> struct WT<bf16_t, TR> {
    static constexpr int VE = 8;
    static constexpr int BKP = 32;     // pixels per LDS tile
    // tr-read: pitch == 64 B (mod 256 B) keeps the 4 rows x 2 lane-groups of one
    // ds_read_b64_tr_b16 on disjoint banks; u16 reads: any 16-byte-aligned pitch works.
    static constexpr int PITCH = TR ? 160 : 136;
    using VT = uint4;
};
template <bool TR> struct WT<float, TR> {
    static constexpr int VE = 4;
    static constexpr int BKP = 16;
    static constexpr int PITCH = 132;
    using VT = float4;
};

struct WgArgs {
    const void* x; const void* dy; float* dw;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Hd, Wd;
    int ks, x_ups, x_relu, dy_ups;
    int log2_wo, log2_howo;
    int M, J, tiles_i, tiles_j, pix_per_split;
    float alpha;
    float* part; long long L;        // deterministic split-K: partial slabs part[blockIdx.y][L] (nullptr: float atomics)
};

__device__ __forceinline__ uint4 relu_v(uint4 v) {
    return make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
}
__device__ __forceinline__ float4 relu_v(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void zero_v(uint4& v) { v = make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void zero_v(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }
__device__ __forceinline__ void set_e(uint4& v, int e, bf16_t x) {
    reinterpret_cast<uint32_t*>(&v)[e >> 1] |= ((uint32_t)x) << ((e & 1) * 16);
}
__device__ __forceinline__ void set_e(float4& v, int e, float x) { reinterpret_cast<float*>(&v)[e] = x; }
__device__ __forceinline__ bf16_t relu_s(bf16_t v) { return (v & 0x8000u) ? (bf16_t)0 : v; }
__device__ __forceinline__ float relu_s(float v) { return fmaxf(v, 0.f); }

template <typename T, bool VECX, bool VECY, bool TR>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(const WgArgs p) {
    using W = WT<T, TR>;
    using VT = typename W::VT;
    constexpr int VE = W::VE, BKP = W::BKP, PITCH = W::PITCH;
    constexpr int VPR = 128 / VE;          // 16-byte vectors per 128-channel row
    constexpr int RSTEP = 256 / VPR;       // pixel rows covered per pass (2 passes per tile)
    __shared__ __attribute__((aligned(16))) T lds[2 * 2 * BKP * PITCH];
    T* const Ys = lds;                     // [buf][pixel][cout]
    T* const Xs = lds + 2 * BKP * PITCH;   // [buf][pixel][j]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = blockIdx.x;
    const int ti = tile / p.tiles_j, tj = tile - ti * p.tiles_j;
    const int i0 = ti * 128, j0 = tj * 128;
    const int p_begin = blockIdx.y * p.pix_per_split;
    const int p_end = min(p.M, p_begin + p.pix_per_split);

    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ dy = static_cast<const T*>(p.dy);

    // loader geometry
    const int vcol = tid % VPR, prow = tid / VPR;
    const int ci = i0 + vcol * VE;                  // this thread's cout slot
    const int jj = j0 + vcol * VE;                  // this thread's (tap, cin) slot
    const int half = p.ks >> 1;
    int tap_v = 0, c_v = 0, ddy = 0, ddx = 0;
    if (VECX) {                                     // a vector never straddles taps (Cin % VE == 0)
        tap_v = jj / p.Cin;
        c_v = jj - tap_v * p.Cin;
        ddy = tap_v / p.ks - half;
        ddx = tap_v - (tap_v / p.ks) * p.ks - half;
    }
    const int howo_mask = (1 << p.log2_howo) - 1;

    struct St { VT y[2], x[2]; };
    auto load_tile = [&](int pbase, St& s) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            zero_v(s.y[r]);
            zero_v(s.x[r]);
            const int pix = pbase + prow + RSTEP * r;
            if (pix >= p_end) continue;
            const int n = pix >> p.log2_howo, rem = pix & howo_mask;
            const int oy = rem >> p.log2_wo, ox = rem & (p.Wo - 1);
            // ---- dY operand
            if (ci < p.Cout) {
                const size_t dp = p.dy_ups ? ((size_t)(n * p.Hd + (oy >> 1)) * p.Wd + (ox >> 1)) : (size_t)pix;
                const T* src = dy + dp * p.Cout + ci;
                if (VECY) {
                    s.y[r] = *reinterpret_cast<const VT*>(src);
                } else {
                    for (int e = 0; e < VE; ++e)
                        if (ci + e < p.Cout) set_e(s.y[r], e, src[e]);
                }
            }
            // ---- gathered input operand
            if (VECX) {
                if (jj < p.J) {
                    const int iy = oy + ddy, ix = ox + ddx;
                    if ((unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo) {
                        const int sy = p.x_ups ? (iy >> 1) : iy, sx = p.x_ups ? (ix >> 1) : ix;
                        s.x[r] = *reinterpret_cast<const VT*>(
                            x + ((size_t)((n * p.Hi + sy) * p.Wi + sx) * p.Cin + c_v));
                        if (p.x_relu) s.x[r] = relu_v(s.x[r]);
                    }
                }
            } else {
                for (int e = 0; e < VE; ++e) {
                    const int j = jj + e;
                    if (j >= p.J) break;
                    const int tap = j / p.Cin, c = j - tap * p.Cin;
                    const int iy = oy + tap / p.ks - half, ix = ox + (tap - (tap / p.ks) * p.ks) - half;
                    if ((unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo) {
                        const int sy = p.x_ups ? (iy >> 1) : iy, sx = p.x_ups ? (ix >> 1) : ix;
                        T v = x[(size_t)((n * p.Hi + sy) * p.Wi + sx) * p.Cin + c];
                        if (p.x_relu) v = relu_s(v);
                        set_e(s.x[r], e, v);
                    }
                }
            }
        }
    };
    auto store_tile = [&](int buf, const St& s) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = prow + RSTEP * r;
            *reinterpret_cast<VT*>(Ys + (buf * BKP + row) * PITCH + vcol * VE) = s.y[r];
            *reinterpret_cast<VT*>(Xs + (buf * BKP + row) * PITCH + vcol * VE) = s.x[r];
        }
    };

    // wave -> 64(cout) x 64(j) sub-tile as 2x2 MFMA 32x32 blocks
    const int wi = wave >> 1, wj = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    auto compute = [&](int buf) {
        const T* yb = Ys + buf * BKP * PITCH + wi * 64;
        const T* xb = Xs + buf * BKP * PITCH + wj * 64;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < BKP / 16; ++kk) {
                bf16x8 yf[2], xf[2];
                if constexpr (TR) {
                // 16-lane group g = lane>>4: channels (g&1)*16 + 0..15, k rows (g>>1)*8 + {0..3 | 4..7}.
                // Within a group lane q supplies the 8-byte chunk (row q>>2, cols (q&3)*4..+3) of a
                // 4x16 block and receives column q of it (rows = 4 consecutive pixels).
                const int q = lane & 15, g = lane >> 4;
                const int krow = kk * 16 + (g >> 1) * 8 + (q >> 2);
                const int ccol = (g & 1) * 16 + (q & 3) * 4;
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    typedef __attribute__((address_space(3))) short4v* lptr;
                    short4v y0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + krow * PITCH + a * 32 + ccol));
                    short4v y1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + (krow + 4) * PITCH + a * 32 + ccol));
                    short4v x0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xb + krow * PITCH + a * 32 + ccol));
                    short4v x1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xb + (krow + 4) * PITCH + a * 32 + ccol));
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    short8v ys = {y0[0], y0[1], y0[2], y0[3], y1[0], y1[1], y1[2], y1[3]};
                    short8v xs = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
                    yf[a] = __builtin_bit_cast(bf16x8, ys);
                    xf[a] = __builtin_bit_cast(bf16x8, xs);
                }
                } else {
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    typedef __attribute__((ext_vector_type(8))) short short8v;
                    short8v ys, xs;
#pragma unroll
                    for (int e = 0; e < 8; ++e) {
                        const int k = kk * 16 + lhi * 8 + e;
                        ys[e] = (short)yb[k * PITCH + a * 32 + l31];
                        xs[e] = (short)xb[k * PITCH + a * 32 + l31];
                    }
                    yf[a] = __builtin_bit_cast(bf16x8, ys);
                    xf[a] = __builtin_bit_cast(bf16x8, xs);
                }
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(yf[a], xf[b], acc[a][b], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < BKP / 2; ++kk) {
                float yf[2], xf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) {
                    yf[a] = yb[(kk * 2 + lhi) * PITCH + a * 32 + l31];
                    xf[a] = xb[(kk * 2 + lhi) * PITCH + a * 32 + l31];
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(yf[a], xf[b], acc[a][b], 0, 0, 0);
            }
        }
    };

    const int ntile = (p_end - p_begin + BKP - 1) / BKP;
    if (ntile <= 0) return;
    St st;
    load_tile(p_begin, st);
    store_tile(0, st);
    __syncthreads();
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        const bool more = t + 1 < ntile;
        if (more) load_tile(p_begin + (t + 1) * BKP, st);
        compute(buf);
        if (more) store_tile(buf ^ 1, st);
        __syncthreads();
    }

    // D[cout][j]: col = lane&31 -> j (contiguous in dW: coalesced atomics), rows -> cout
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int j = j0 + wj * 64 + b * 32 + l31;
        if (j >= p.J) continue;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wi * 64 + a * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (i < p.Cout) {
                    if (p.part) p.part[(size_t)blockIdx.y * p.L + (size_t)i * p.J + j] = acc[a][b][e];
                    else atomicAdd(p.dw + (size_t)i * p.J + j, p.alpha * acc[a][b][e]);
                }
            }
    }
}

}  // namespace

extern "C" int xmc_conv2d_wgrad_patch_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw,
                                          float* db, float* ws, long long* query, void* stream);

extern "C" int xmc_conv2d_wgrad_phase_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                                          float* ws, long long* query, void* stream);
extern "C" int xmc_conv2d_wgrad_dma_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw,
                                        float* db, float* ws, long long* query, void* stream);

extern "C" int64_t xmc_reduce_mid_ws_floats(int64_t a, int64_t r, int64_t c);
extern "C" int xmc_reduce_mid_ws(const void* x, float* y, float* ws, int64_t a, int64_t r, int64_t c, int32_t dtype,
                                 int32_t relu, float scale, int32_t accumulate, void* stream);

namespace {

// dw[e] += alpha * sum_s part[s][e] (e < n_w) and db[e - n_w] += alpha * sum_s part[s][e] (n_w <= e < L), splits
// added in a fixed order: SG split groups per float4 column, each summing its splits in sequence, the groups
// combined through LDS in order.  OVR: "=" instead of "+=" (first write of a gradient arena nobody zeroed: no read of dw).
template <int SG, bool OVR>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, int nsplit, long long L,
                                                           long long n_w, float* __restrict__ dw,
                                                           float* __restrict__ db, float alpha) {
    constexpr int COLS = 256 / SG;
    __shared__ float4 red[SG][COLS];
    const int cq = threadIdx.x % COLS, sg = threadIdx.x / COLS;
    const long long e0 = ((long long)blockIdx.x * COLS + cq) * 4;
    const long long n_tot = db ? L : n_w;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (e0 < n_tot) {
        // four slabs per trip, loaded together (a slab index past the end re-reads the last slab and adds nothing): one
        // dependent load per trip was nsplit / SG serial memory latencies; same order of the sum
        for (int s = sg; s < nsplit; s += 4 * SG) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const float4*>(part + (long long)min(s + u * SG, nsplit - 1) * L + e0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = s + u * SG < nsplit;
                t.x += ok ? v[u].x : 0.f; t.y += ok ? v[u].y : 0.f; t.z += ok ? v[u].z : 0.f; t.w += ok ? v[u].w : 0.f;
            }
        }
    }
    if (SG > 1) {
        red[sg][cq] = t;
        __syncthreads();
        if (sg == 0) {
#pragma unroll
            for (int k = 1; k < SG; ++k) { t.x += red[k][cq].x; t.y += red[k][cq].y; t.z += red[k][cq].z; t.w += red[k][cq].w; }
        }
    }
    if (sg == 0 && e0 < n_tot) {
        float4* dst = reinterpret_cast<float4*>(e0 < n_w ? dw + e0 : db + (e0 - n_w));
        if constexpr (OVR) {
            *dst = make_float4(alpha * t.x, alpha * t.y, alpha * t.z, alpha * t.w);
        } else {
            float4 d = *dst;
            d.x += alpha * t.x; d.y += alpha * t.y; d.z += alpha * t.z; d.w += alpha * t.w;
            *dst = d;
        }
    }
}

__global__ __launch_bounds__(256) void wgrad_reduce_scalar_kernel(const float* __restrict__ part, int nsplit, long long L,
                                                                  long long n_w, float* __restrict__ dw,
                                                                  float* __restrict__ db, float alpha, int overwrite) {
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (db ? L : n_w)) return;
    float t = 0.f;
    for (int s = 0; s < nsplit; ++s) t += part[(long long)s * L + e];
    float* dst = e < n_w ? dw + e : db + (e - n_w);
    *dst = overwrite ? alpha * t : *dst + alpha * t;
}

}  // namespace

extern "C" int xmc_internal_wgrad_reduce(const float* part, int nsplit, long long L, long long n_w, float* dw,
                                         float* db, float alpha, int overwrite, void* stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n_tot = db ? L : n_w;
    const bool vec = (L % 4) == 0 && (n_w % 4) == 0 && ((uintptr_t)part % 16) == 0 && ((uintptr_t)dw % 16) == 0 &&
                     (!db || ((uintptr_t)db % 16) == 0);
#define XMC_WR(SG_, GRID_)                                                                                                     \
    do {                                                                                                                       \
        if (overwrite) hipLaunchKernelGGL((wgrad_reduce_kernel<SG_, true>), dim3((unsigned)(GRID_)), dim3(256), 0, s, part, nsplit, L, n_w, dw, db, alpha); \
        else hipLaunchKernelGGL((wgrad_reduce_kernel<SG_, false>), dim3((unsigned)(GRID_)), dim3(256), 0, s, part, nsplit, L, n_w, dw, db, alpha);         \
    } while (0)
    if (!vec) {
        hipLaunchKernelGGL(wgrad_reduce_scalar_kernel, dim3((unsigned)((n_tot + 255) / 256)), dim3(256), 0, s, part, nsplit, L, n_w,
                           dw, db, alpha, overwrite);
    } else if (nsplit <= 4) {
        XMC_WR(1, (n_tot / 4 + 255) / 256);
    } else if (nsplit <= 32) {
        XMC_WR(4, (n_tot / 4 + 63) / 64);
    } else {
        XMC_WR(16, (n_tot / 4 + 15) / 16);
    }
#undef XMC_WR
    return xmc_hip_err(hipGetLastError());
}

// Shared body of xmc_conv2d_wgrad (ws == NULL: split-K by float atomics), xmc_conv2d_wgrad_ws (deterministic
// split-K through the caller's workspace) and xmc_conv2d_wgrad_workspace_bytes (query != NULL: no launch).
static int wgrad_dispatch(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db, float* ws,
                          long long* query, void* stream) {
    XMC_REQUIRE(d && (query || (x && dy && dw)));
    XMC_REQUIRE(d->ks == 1 || d->ks == 3);
    XMC_REQUIRE(d->dtype == XMC_F32 || d->dtype == XMC_BF16);
    const int variant = d->variant & 15;   // (bits 4.. : tuning bits of the LDS-DMA kernel, A/B benchmarks only)
    if (variant != 0) {                    // variant: 0 generic kernel only, 1 auto, 2 skip the LDS-DMA kernel (A/B benchmarks)
        // next to a 2x resampling (x_ups / dy_ups): 16 (phase, tap) products per low-resolution pixel instead of 36
        // (conv_wgrad_phase.hip; needs the workspace; bit 8 of variant: off)
        int rc = variant == 2 ? 1 : xmc_conv2d_wgrad_phase_try(d, x, dy, dw, db, ws, query, stream);
        if (rc != 1) return rc;
        rc = variant == 2 ? 1 : xmc_conv2d_wgrad_dma_try(d, x, dy, dw, db, ws, query, stream);   // LDS-DMA staged, 3-stage ring
        if (rc != 1) return rc;
    }
    // XMC_WGRAD_OVERWRITE on the two kernels that only know "+=" (register-staged patch kernel, generic kernel: the float32
    // parity mode and channel counts outside the MFMA kernels' domain): clear dw / db first, then accumulate
    xmc_wgrad_desc dacc;
    if (!query && (d->variant & XMC_WGRAD_OVERWRITE)) {
        hipStream_t s0 = static_cast<hipStream_t>(stream);
        const int rc0 = xmc_hip_err(hipMemsetAsync(dw, 0, sizeof(float) * (size_t)d->cout * d->ks * d->ks * d->cin, s0));
        if (rc0 != XMC_OK) return rc0;
        if (db) {
            const int rc1 = xmc_hip_err(hipMemsetAsync(db, 0, sizeof(float) * (size_t)d->cout, s0));
            if (rc1 != XMC_OK) return rc1;
        }
        dacc = *d;
        dacc.variant &= ~XMC_WGRAD_OVERWRITE;
        d = &dacc;
    }
    if (variant != 0) {
        const int rc = xmc_conv2d_wgrad_patch_try(d, x, dy, dw, db, ws, query, stream);              // register staged
        if (rc != 1) return rc;
    }
    WgArgs a;
    a.x = x; a.dy = dy; a.dw = dw;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->x_ups ? 2 * d->hi : d->hi;
    a.Wo = d->x_ups ? 2 * d->wi : d->wi;
    a.Hd = d->dy_ups ? a.Ho / 2 : a.Ho;
    a.Wd = d->dy_ups ? a.Wo / 2 : a.Wo;
    a.ks = d->ks; a.x_ups = d->x_ups; a.x_relu = d->x_relu; a.dy_ups = d->dy_ups;
    a.log2_wo = ilog2_exact(a.Wo);
    const int l2h = ilog2_exact(a.Ho);
    XMC_REQUIRE(a.log2_wo >= 0 && l2h >= 0);
    XMC_REQUIRE(!d->dy_ups || (a.Ho >= 2 && a.Wo >= 2));
    a.log2_howo = a.log2_wo + l2h;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    XMC_REQUIRE(m < (1ll << 31));
    a.M = (int)m;
    a.J = a.ks * a.ks * a.Cin;
    a.alpha = d->alpha;
    a.tiles_i = (a.Cout + 127) / 128;
    a.tiles_j = (a.J + 127) / 128;
    const int bkp = d->dtype == XMC_BF16 ? 32 : 16;
    const int ve = d->dtype == XMC_BF16 ? 8 : 4;
    // split the pixel range so the launch has >= ~1024 workgroups (256 CUs x 4)
    const int tiles = a.tiles_i * a.tiles_j;
    int nsplit = (1024 + tiles - 1) / tiles;
    const int max_split = (a.M + 8 * bkp - 1) / (8 * bkp);     // >= 8 LDS tiles per block
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    int pps = (a.M + nsplit - 1) / nsplit;
    pps = ((pps + bkp - 1) / bkp) * bkp;
    nsplit = (a.M + pps - 1) / pps;
    a.pix_per_split = pps;
    // deterministic mode: [nsplit][cout * J] weight partials, then the workspace of the bias reduction
    a.L = (long long)a.Cout * a.J;
    const long long pix_b = (long long)d->n * (d->x_ups ? 4 : 1) * d->hi * d->wi / (d->dy_ups ? 4 : 1);
    const long long w_floats = nsplit > 1 ? (long long)nsplit * a.L : 0;
    if (query) { *query = w_floats + xmc_reduce_mid_ws_floats(1, pix_b, d->cout); return XMC_OK; }
    a.part = (ws && nsplit > 1) ? ws : nullptr;
    if (db) {     // generic path: bias gradient = alpha * sum_p dy'(p) as a separate reduction
        const int rc = xmc_reduce_mid_ws(dy, db, ws ? ws + w_floats : nullptr, 1, pix_b, d->cout, d->dtype, 0,
                                         d->alpha * (d->dy_ups ? 4.f : 1.f), 1, stream);
        if (rc != XMC_OK) return rc;
    }
    const bool vecx = (a.Cin % ve) == 0 && ((uintptr_t)x % 16) == 0;
    const bool vecy = (a.Cout % ve) == 0 && ((uintptr_t)dy % 16) == 0;
    dim3 grid(tiles, nsplit), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
#define XMC_WG_LAUNCH(T, VX, VY, TR) hipLaunchKernelGGL((conv_wgrad_kernel<T, VX, VY, TR>), grid, block, 0, s, a)
    if (d->dtype == XMC_BF16 && variant == 1) {
        if (vecx && vecy) XMC_WG_LAUNCH(bf16_t, true, true, true);
        else if (vecx) XMC_WG_LAUNCH(bf16_t, true, false, true);
        else if (vecy) XMC_WG_LAUNCH(bf16_t, false, true, true);
        else XMC_WG_LAUNCH(bf16_t, false, false, true);
    } else if (d->dtype == XMC_BF16) {
        if (vecx && vecy) XMC_WG_LAUNCH(bf16_t, true, true, false);
        else if (vecx) XMC_WG_LAUNCH(bf16_t, true, false, false);
        else if (vecy) XMC_WG_LAUNCH(bf16_t, false, true, false);
        else XMC_WG_LAUNCH(bf16_t, false, false, false);
    } else {
        if (vecx && vecy) XMC_WG_LAUNCH(float, true, true, false);
        else if (vecx) XMC_WG_LAUNCH(float, true, false, false);
        else if (vecy) XMC_WG_LAUNCH(float, false, true, false);
        else XMC_WG_LAUNCH(float, false, false, false);
    }
#undef XMC_WG_LAUNCH
    if (a.part) return xmc_internal_wgrad_reduce(a.part, nsplit, a.L, a.L, dw, nullptr, a.alpha, 0, stream);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_conv2d_wgrad(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                                void* stream) {
    return wgrad_dispatch(d, x, dy, dw, db, nullptr, nullptr, stream);
}

extern "C" int64_t xmc_conv2d_wgrad_workspace_bytes(const xmc_wgrad_desc* d) {
    long long floats = 0;
    if (wgrad_dispatch(d, nullptr, nullptr, nullptr, nullptr, nullptr, &floats, nullptr) != XMC_OK) return 0;
    return (int64_t)floats * 4;
}

extern "C" int xmc_conv2d_wgrad_ws(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                                   void* ws, int64_t ws_bytes, void* stream) {
    if (!ws) return wgrad_dispatch(d, x, dy, dw, db, nullptr, nullptr, stream);
    long long need = 0;           // the kernel choice can depend on the operands' alignment: size against THESE pointers
    const int rc = wgrad_dispatch(d, x, dy, dw, db, nullptr, &need, nullptr);
    if (rc != XMC_OK) return rc;
    XMC_REQUIRE(need * 4 <= ws_bytes && ((uintptr_t)ws % 16) == 0);
    return wgrad_dispatch(d, x, dy, dw, db, static_cast<float*>(ws), nullptr, stream);
}
