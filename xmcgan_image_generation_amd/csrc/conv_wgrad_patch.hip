// Patch-based weight gradient (bf16) for gfx950.
//
//   dW[cout][tap][c] += alpha * sum_p dY'(p, cout) * patch(p + tap, c)         (+ fused bias gradient)
//
// One workgroup owns a [128 cout] x [32 cin] x [all ks*ks taps] slab of dW and walks its share of the
// pixels (split-K) in tiles of 64 output pixels (4 rows x 16 columns).  Per tile it stages
//   Ys [64 pixels][128 cout]           (the output gradient, nearest-upsampled on the fly if dy_ups)
//   Xp [patch pixels][32 cin]          (the input rows of the tile plus a one-pixel halo; upsample /
//                                       ReLU fused; zero at the image border)
// in LDS once, and every tap reads its B operand from the SAME patch at a constant row offset --
// the input is fetched once instead of ks*ks times and the inner loop has no gather arithmetic.
// Both MFMA operands need the reduction index (pixels) contiguous per lane while NHWC keeps
// channels contiguous: fragments are read with ds_read_b64_tr_b16 (hardware 4x16 transpose read).
// Wave w of the 4 owns cout block w (32 channels) x 9 taps: 9 accumulators, 36 MFMAs per wave per
// tile between barriers.  Partial slabs are combined with float32 atomic adds.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int WPT = 64;                 // output pixels per tile (4 MFMA k-steps)
constexpr int YP = 160;                 // Ys pitch (bf16): 320 B == 64 B (mod 256 B): tr-read conflict-free
constexpr int XP = 32;                  // Xp pitch (bf16): 64-byte rows, 4 consecutive rows = all 64 banks
constexpr int WPP_MAX = 200;            // patch pixels: 3 x 66 for a 64-pixel row segment

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct WPArgs {
    const void* x; const void* dy; float* dw; float* db;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Hd, Wd;
    int x_ups, x_relu, dy_ups;
    int log2_tx, log2_ty;                           // pixel tiles per image row / column
    int M, tiles_i, cchunks, tiles_per_split, ntiles, nsplit;
    int Wt, Rt, imgs, PW, PP;
    unsigned x_bytes, dy_bytes;
    float alpha;
    float* part; long long L;        // deterministic split-K: partial slabs part[split][L] (nullptr: float atomics)
};

__device__ __forceinline__ uint4 relu4w(uint4 v) {
    return make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
}

template <int KS, int OCC>
__global__ __launch_bounds__(256, OCC) void conv_wgrad_patch_kernel(const WPArgs p) {
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];      // 2*64*YP + 2*WPP_MAX*XP bf16 = 65 KiB
    bf16_t* const Ys = lds;                           // [2][64][YP]
    bf16_t* const Xs = lds + 2 * WPT * YP;            // [2][PP][XP]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // (placing all slabs of one pixel split on the same XCD, so that one L2 fetches those pixels once, was measured
    //  SLOWER: 0.50 -> 0.56 ms on the 96-channel 128^2 layer, 0.40 -> 1.1 ms on the 1536-channel layers)
    const int slabs = p.tiles_i * p.cchunks;
    const int slab = blockIdx.x % slabs, split = blockIdx.x / slabs;
    const int ti = slab / p.cchunks, cc = slab - ti * p.cchunks;
    const int i0 = ti * 128, c0 = cc * 32;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.ntiles, t_begin + p.tiles_per_split);
    if (t_begin >= t_end) return;
    const int PR1 = p.Rt + 2 * HALO;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, p.dy_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- dY tile: 64 pixels x 128 cout = 1024 16-byte vectors, 4 per thread: (pixel = tid/16 + 16 r, slot)
    const int yslot = tid & 15, ypix = tid >> 4;
    unsigned yvoff[4];
    {
        const int co = i0 + yslot * 8;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int t = ypix + 16 * r;                      // tile-local pixel
            const int c = t & (p.Wt - 1), rowi = t / p.Wt;
            const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
            // local offset inside dY relative to the tile origin (n0, y0, x0); origin parity is even
            // whenever the tile spans rows, so (y0 + rj) >> 1 == (y0 >> 1) + (rj >> 1)
            const int ly = p.dy_ups ? (rj >> 1) : rj, lx = p.dy_ups ? (c >> 1) : c;
            yvoff[r] = co < p.Cout ? (unsigned)(((im * p.Hd + ly) * p.Wd + lx) * p.Cout + co) * 2u : OOB;
        }
    }
    // ---- patch: PP pixels x 4 slots
    const int nvec = p.PP * 4;
    int prr[4], ppc[4], pim[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = tid + 256 * i;
        const int pp = v >> 2;
        const int pr = pp / p.PW;
        ppc[i] = pp - pr * p.PW;
        pim[i] = pr / PR1;
        prr[i] = pr - pim[i] * PR1;
    }
    const int pkv = tid & 3;

    u32x4 yreg[4], xreg[4];
    auto load_tile = [&](int t) {
        // tile t = Rt rows x Wt columns of one image (or `imgs` whole small images)
        const int x0 = (t & ((1 << p.log2_tx) - 1)) * p.Wt;
        const int y0 = ((t >> p.log2_tx) & ((1 << p.log2_ty) - 1)) * p.Rt;
        const int n0 = (t >> (p.log2_tx + p.log2_ty)) * p.imgs;
        const int ybase = p.dy_ups ? (((n0 * p.Hd + (y0 >> 1)) * p.Wd + (x0 >> 1)) * p.Cout) * 2
                                   : (((n0 * p.Ho + y0) * p.Wo + x0) * p.Cout) * 2;
#pragma unroll
        for (int r = 0; r < 4; ++r) yreg[r] = __builtin_amdgcn_raw_buffer_load_b128(yr, yvoff[r], ybase, 0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (tid + 256 * i < nvec) {
                const int y = y0 + prr[i] - HALO, xx = x0 + ppc[i] - HALO;
                unsigned off = OOB;
                if ((unsigned)y < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo) {
                    const int sy = p.x_ups ? (y >> 1) : y, sx = p.x_ups ? (xx >> 1) : xx;
                    off = (unsigned)((((n0 + pim[i]) * p.Hi + sy) * p.Wi + sx) * p.Cin + c0 + pkv * 8) * 2u;
                }
                xreg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, 0, 0);
            }
        }
    };
    float bsum[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bsum[e] = 0.f;
    const bool do_bias = p.db != nullptr && cc == 0;
    auto store_tile = [&](int buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            *reinterpret_cast<u32x4*>(Ys + (buf * WPT + ypix + 16 * r) * YP + yslot * 8) = yreg[r];
            if (do_bias) {
                const unsigned w4[4] = {yreg[r].x, yreg[r].y, yreg[r].z, yreg[r].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    bsum[2 * e] += __uint_as_float(w4[e] << 16);
                    bsum[2 * e + 1] += __uint_as_float(w4[e] & 0xffff0000u);
                }
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int v = tid + 256 * i;
            if (v < nvec) {
                uint4 q = make_uint4(xreg[i].x, xreg[i].y, xreg[i].z, xreg[i].w);
                if (p.x_relu) q = relu4w(q);
                *reinterpret_cast<uint4*>(Xs + (buf * WPP_MAX + (v >> 2)) * XP + pkv * 8) = q;
            }
        }
    };

    // ---- fragment geometry (transpose reads).  16-lane group g = lane >> 4: channel half g & 1,
    //      pixel half g >> 1; lane q of the group supplies row (q >> 2) and 4-channel chunk (q & 3).
    const int q = lane & 15, g = lane >> 4;
    const int kro = (g >> 1) * 8 + (q >> 2);            // pixel row offset inside a 16-pixel k-step
    const int cco = (g & 1) * 16 + (q & 3) * 4;         // channel offset of this lane's 8-byte chunk
    int xrow[8];                                        // patch row (tap 0,0) of the lane's 8 source pixels
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int t = (s >> 1) * 16 + kro + (s & 1) * 4;
        const int c = t & (p.Wt - 1), rowi = t / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        xrow[s] = ((im * PR1 + rj) * p.PW + c) * XP + cco;
    }
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    typedef __attribute__((address_space(3))) short4v* lptr;
    typedef __attribute__((ext_vector_type(8))) short short8v;
    // Fragment reads run ONE group (3 taps) ahead of the MFMAs that consume them, in double-buffered registers,
    // with sched_barrier(0) pinning that order: left alone, the scheduler puts each ds_read_b64_tr right before
    // its MFMA behind an s_waitcnt lgkmcnt(0), and every group pays the LDS latency.
    constexpr int GRP = TAPS == 9 ? 3 : 1, NG = TAPS / GRP, UNITS = 4 * NG;
    auto compute = [&](int buf) {
        const bf16_t* yb = Ys + buf * WPT * YP + wave * 32 + cco;
        const bf16_t* xb = Xs + buf * WPP_MAX * XP;
        auto rd_a = [&](int kk) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + (kk * 16 + kro) * YP));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + (kk * 16 + kro + 4) * YP));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int t) {
            const int toff = ((t / KS) * p.PW + (t % KS)) * XP;
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xb + xrow[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(xb + xrow[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2], bfr[2][GRP];
        af[0] = rd_a(0);
#pragma unroll
        for (int t = 0; t < GRP; ++t) bfr[0][t] = rd_b(0, t);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int kk = u / NG, g = u % NG;
            if (u + 1 < UNITS) {
                const int kk1 = (u + 1) / NG, g1 = (u + 1) % NG;
                if (g1 == 0) af[kk1 & 1] = rd_a(kk1);
#pragma unroll
                for (int t = 0; t < GRP; ++t) bfr[(u + 1) & 1][t] = rd_b(kk1, g1 * GRP + t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < GRP; ++t)
                acc[g * GRP + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1], bfr[u & 1][t], acc[g * GRP + t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    load_tile(t_begin);
    store_tile(0);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        const bool more = t + 1 < t_end;
        if (more) load_tile(t + 1);
        compute(buf);
        if (more) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- D[i = cout][j = cin]: col = lane & 31 -> cin (contiguous in dW), rows -> cout
    const int l31 = lane & 31, lhi = lane >> 5;
    const int J = TAPS * p.Cin;
    float* const pr = p.part ? p.part + (size_t)split * p.L : nullptr;     // this split's slab (plain stores)
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int i = i0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
            if (i < p.Cout) {
                const size_t o = (size_t)i * J + t * p.Cin + c0 + l31;
                if (pr) pr[o] = acc[t][e];
                else atomicAdd(p.dw + o, p.alpha * acc[t][e]);
            }
        }
    if (do_bias) {       // workgroup-level reduction in LDS (the main loop ended on a barrier), then ONE atomic
                         // per output channel per workgroup (per-thread atomics to 96 addresses cost 0.9 ms)
        float* red = reinterpret_cast<float*>(lds);                  // [16 pixel rows][128 cout]
#pragma unroll
        for (int e = 0; e < 8; ++e) red[ypix * 128 + yslot * 8 + e] = bsum[e];
        __syncthreads();
        if (tid < 128) {
            float sum = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += red[r * 128 + tid];
            if (i0 + tid < p.Cout) {
                if (pr) pr[(size_t)p.Cout * J + i0 + tid] = sum;
                else atomicAdd(p.db + i0 + tid, p.alpha * sum);
            }
        }
    }
}

}  // namespace

// Returns XMC_OK when launched, 1 when the shape is not eligible, or a negative error.  `db` (may be
// NULL) receives alpha * sum_p dy'(p, cout) -- the bias gradient of the same convolution.
constexpr int WGP_LDS_BYTES = (2 * WPT * YP + 2 * WPP_MAX * XP) * 2;
extern "C" int xmc_internal_optin_wgrad_patch(void) {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(conv_wgrad_patch_kernel<3, 2>),
                          reinterpret_cast<const void*>(conv_wgrad_patch_kernel<1, 1>)}, WGP_LDS_BYTES) ? XMC_OK : XMC_EINVAL;
}

extern "C" int xmc_conv2d_wgrad_patch_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw,
                                          float* db, float* ws, long long* query, void* stream) {
    if (d->dtype != XMC_BF16 || (d->cin % 32) != 0 || (d->cout % 8) != 0) return 1;
    WPArgs a;
    a.x = x; a.dy = dy; a.dw = dw; a.db = db;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->x_ups ? 2 * d->hi : d->hi;
    a.Wo = d->x_ups ? 2 * d->wi : d->wi;
    a.Hd = d->dy_ups ? a.Ho / 2 : a.Ho;
    a.Wd = d->dy_ups ? a.Wo / 2 : a.Wo;
    a.x_ups = d->x_ups; a.x_relu = d->x_relu; a.dy_ups = d->dy_ups;
    const int l2w = ilog2_exact(a.Wo), l2h = ilog2_exact(a.Ho);
    if (l2w < 0 || l2h < 0) return 1;
    if (d->dy_ups && (a.Ho < 2 || a.Wo < 2)) return 1;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    if (m % WPT != 0 || m >= (1ll << 31)) return 1;
    a.M = (int)m;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2, yb = (long long)a.N * a.Hd * a.Wd * a.Cout * 2;
    if (xb >= 0xfffffff0ll || yb >= 0xfffffff0ll) return 1;
    if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16)) return 1;
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    const int halo = d->ks / 2;
    // 4 rows x 16 columns where the image allows: 108 patch pixels per 64 outputs (a 1 x 64 row segment needs 198)
    constexpr int wt_max = 16;
    a.Wt = a.Wo < wt_max ? a.Wo : wt_max;
    const int rows = WPT / a.Wt;
    a.Rt = rows < a.Ho ? rows : a.Ho;
    a.imgs = WPT / (a.Wt * a.Rt);
    if (a.N % a.imgs != 0) return 1;
    a.log2_tx = l2w - ilog2_exact(a.Wt);
    a.log2_ty = l2h - ilog2_exact(a.Rt);
    // with dy_ups a multi-row tile must start on an even row and a row segment on an even column:
    // true for every power-of-two geometry with Wt >= 2
    if (d->dy_ups && a.Wt < 2) return 1;
    a.PW = a.Wt + 2 * halo;
    a.PP = a.imgs * (a.Rt + 2 * halo) * a.PW;
    if (a.PP > WPP_MAX || a.PP * 4 > 1024) return 1;
    a.tiles_i = (a.Cout + 127) / 128;
    a.cchunks = a.Cin / 32;
    a.ntiles = a.M / WPT;
    const int slabs = a.tiles_i * a.cchunks;
    constexpr int target_wg = 1024;
    int nsplit = (target_wg + slabs - 1) / slabs;             // >= ~target workgroups
    const int max_split = (a.ntiles + 3) / 4;                 // >= 4 tiles (256 pixels) per workgroup
    if (nsplit > max_split) nsplit = max_split;
    if (nsplit < 1) nsplit = 1;
    a.tiles_per_split = (a.ntiles + nsplit - 1) / nsplit;
    nsplit = (a.ntiles + a.tiles_per_split - 1) / a.tiles_per_split;
    a.nsplit = nsplit;
    a.alpha = d->alpha;
    a.L = (long long)a.Cout * d->ks * d->ks * a.Cin + a.Cout;
    if (query) { *query = nsplit > 1 ? (long long)nsplit * a.L : 0; return XMC_OK; }
    a.part = (ws && nsplit > 1) ? ws : nullptr;
    dim3 grid(slabs * nsplit), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    constexpr int lds_bytes = (2 * WPT * YP + 2 * WPP_MAX * XP) * 2;
    static_assert(lds_bytes == WGP_LDS_BYTES, "opt-in size");
    if (xmc_internal_optin_wgrad_patch() != XMC_OK) return 1;      // > 64 KiB of LDS needs the opt-in attribute
    if (d->ks == 3) hipLaunchKernelGGL((conv_wgrad_patch_kernel<3, 2>), grid, block, lds_bytes, s, a);   // 2 waves/SIMD: +19 %
    else if (d->ks == 1) hipLaunchKernelGGL((conv_wgrad_patch_kernel<1, 1>), grid, block, lds_bytes, s, a);
    else return 1;
    if (a.part) return xmc_internal_wgrad_reduce(a.part, nsplit, a.L, a.L - a.Cout, dw, db, a.alpha, 0, stream);
    return xmc_hip_err(hipGetLastError());
}
