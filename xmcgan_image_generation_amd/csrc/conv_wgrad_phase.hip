// Weight gradient (bf16) of a 3x3 convolution that sits next to a 2x resampling, phase-decomposed (round 3).
//
// conv_stream.hip's conv_phase_kernel runs conv3x3(nearest_upsample2(x)) and avg_pool2x2(conv3x3(x)) as four 2x2 convolutions
// on the low-resolution ("V") grid: 16 instead of 36 multiply-adds per V pixel and channel pair.  The weight gradient has
// the same structure -- per V pixel only 16 distinct (dY pixel, x pixel) pairs occur:
//   FORM 0  (layer = conv(upsample2(x)); wgrad_desc.x_ups):   x on the V grid, dY at twice the resolution
//       dE[(a,b)][(tu,tv)] = sum_{n,i,j} dY[n][2i+a][2j+b] (x) x[n][i+a-1+tu][j+b-1+tv]
//   FORM 1  (layer = avg_pool2(conv(x)); wgrad_desc.dy_ups):  dY on the V grid, x at twice the resolution
//       dE[(a',b')][(tu,tv)] = sum_{n,i,j} dY[n][i][j] (x) x[n][2(i+tu)-a'][2(j+tv)-b']
// and every 3x3 tap gradient is the sum of the four dE entries whose tap sums contain it (the transpose of
// xmc_phase_conv_weight's map): dW[dy][dx] = sum_{(a,tu) : dy in S(a,tu)} sum_{(b,tv) : dx in S(b,tv)} dE[(a,b)][(tu,tv)].
// 2.25x fewer MFMAs than the 3x3 formulation over the high-resolution pixels, exact in real arithmetic.
//
// Skeleton of conv_wgrad_dma.hip (both operands global -> LDS by `buffer_load_dwordx4 ... lds`, fragments by
// ds_read_b64_tr_b16 with immediate tap offsets, split-K over pixel tiles with plain-store partial slabs), re-cut for 16
// "taps": one workgroup = 64 cout x 32 cin x 16 (phase, tap) pairs = 32 accumulator blocks, wave w = phase w (4 taps x 2
// cout blocks = 8 blocks, 8 MFMAs per 16-pixel k-step from 2 A + 4 B fragments); tile = 64 V pixels; 2-stage LDS ring
//   FORM 0: Ys [4 phases][64 px][64 cout] (wave w gathers and reads ITS phase of dY: rows 128 B, the 16-byte slot s of row r
//           is stored at slot s ^ (((r >> 1) & 1) << 2)), Xp [patch (Rt + 2) x (Wt + 2)][32 cin]            40 KB / stage
//   FORM 1: Ys [64 px][64 cout] shared by the four waves, Xp = the (2 Rt + 2) x (2 Wt + 2) high-resolution window with its
//           columns de-interleaved by parity (the 16 pixels of a k-step are then consecutive 64-byte rows for every tap)
//                                                                                                          <= 36 KB / stage
// The 16 entries are folded into the 9 taps of dW inside the workgroup (through LDS, after the pixel loop); the result goes
// to conv_wgrad_dma's partial-slab layout (fixed-order reduction by xmc_internal_wgrad_reduce) or, for a single split,
// straight into dW / db: bit-reproducible, no atomics.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr int DPT = 64;                  // V pixels per tile

struct WPArgs {
    const void* x; const void* dy; float* part; float* dw; float* db;   // part == nullptr (single split): dw / db are updated in place
    int N, Hv, Wv, Cin, Cout;            // V grid; FORM 0: x (N,Hv,Wv,Cin), dy (N,2Hv,2Wv,Cout); FORM 1: x (N,2Hv,2Wv,Cin), dy (N,Hv,Wv,Cout)
    int x_relu, do_bias;
    int overwrite;                       // single split: dw / db = alpha * sum (no read of the old value)
    int log2_tx, log2_ty;
    int tiles_i, cchunks, tiles_per_split, ntiles, nsplit;
    int Wt, Rt, imgs, PR1, PP, magic_pw, magic_pr1;
    unsigned x_bytes, dy_bytes;
    long long L;                         // floats per split slab: Cout * 9 * Cin + Cout (conv_wgrad_dma's layout)
    float alpha;
};

// NY = dY DMA instructions per wave per tile (8 / 2), XI = x-patch DMA instructions per wave per tile (16 patch pixels each),
// PWC = patch row width in pixels (compile time: tap offsets are immediates)
template <int FORM, int XI, int PWC>
__global__ __launch_bounds__(256, 2) void conv_wgrad_phase_kernel(const WPArgs p) {
    constexpr int NY = FORM == 0 ? 8 : 2;
    constexpr int YS_BYTES = FORM == 0 ? 4 * DPT * 128 : DPT * 128;
    constexpr int STAGE_BYTES = YS_BYTES + XI * 4 * 1024;
    constexpr int PER_TILE = NY + XI;
    constexpr int HALF = PWC / 2;
    constexpr unsigned OOB = 0xfffffff0u;
    static_assert(PER_TILE <= 16, "one DMA piece per MFMA unit");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slabs = p.tiles_i * p.cchunks;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int slab = wid % slabs, split = wid / slabs;
    const int ti = slab / p.cchunks, cc = slab - ti * p.cchunks;
    const int i0 = ti * 64, c0 = cc * 32;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.ntiles, t_begin + p.tiles_per_split);
    float* const pr = p.part ? p.part + (size_t)split * p.L : nullptr;
    const int Hx = FORM == 0 ? p.Hv : 2 * p.Hv, Wx = FORM == 0 ? p.Wv : 2 * p.Wv;      // x resolution
    const int Hd = FORM == 0 ? 2 * p.Hv : p.Hv, Wd = FORM == 0 ? 2 * p.Wv : p.Wv;      // dY resolution

    const v4i32 xr = make_srd(p.x, p.x_bytes), yr = make_srd(p.dy, p.dy_bytes);
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);

    // ---- dY DMA: 8 rows (V pixels) x 8 slots of 16 bytes per instruction
    unsigned yvoff[NY];
#pragma unroll
    for (int k = 0; k < NY; ++k) {
        const int I = wave * NY + k;                             // FORM 0: phase = I >> 3 = wave, rows (I & 7) * 8 ..
        const int r = (FORM == 0 ? (I & 7) : I) * 8 + (lane >> 3);
        const int slot = (lane & 7) ^ (((r >> 1) & 1) << 2);     // source slot of LDS position (lane & 7)
        const int co = i0 + slot * 8;
        const int c = r & (p.Wt - 1), rowi = r / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        int pix;
        if (FORM == 0) pix = (im * Hd + 2 * rj + (wave >> 1)) * Wd + 2 * c + (wave & 1);
        else pix = (im * Hd + rj) * Wd + c;
        yvoff[k] = co < p.Cout ? (unsigned)(pix * p.Cout + co) * 2u : OOB;
    }
    // ---- x DMA: instruction k covers LDS patch positions (wave * XI + k) * 16 .. + 15; lane -> (position, 16-byte slot)
    int prr[XI], ppc[XI], pim[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int pp = (wave * XI + k) * 16 + (lane >> 2);
        int pr_, pc_;
        if (FORM == 0) {
            pr_ = (pp * p.magic_pw) >> 16; pc_ = pp - pr_ * PWC;
        } else {                                                 // [row][column parity][PWC / 2]
            const int q = (pp * p.magic_pw) >> 16, ch = pp - q * HALF;
            pr_ = q >> 1; pc_ = 2 * ch + (q & 1);
        }
        ppc[k] = pp < p.PP ? pc_ : -1000000;
        pim[k] = (pr_ * p.magic_pr1) >> 16;
        prr[k] = pr_ - pim[k] * p.PR1;
    }
    const int pkv = lane & 3;

    struct TileOrg { int x0, y0, n0, ybase; unsigned sb; };
    auto tile_org = [&](int t, int stage) {
        TileOrg o;
        o.x0 = (t & ((1 << p.log2_tx) - 1)) * p.Wt;
        o.y0 = ((t >> p.log2_tx) & ((1 << p.log2_ty) - 1)) * p.Rt;
        o.n0 = (t >> (p.log2_tx + p.log2_ty)) * p.imgs;
        o.ybase = FORM == 0 ? (((o.n0 * Hd + 2 * o.y0) * Wd + 2 * o.x0) * p.Cout) * 2 : (((o.n0 * Hd + o.y0) * Wd + o.x0) * p.Cout) * 2;
        o.sb = lds0 + stage * STAGE_BYTES;
        return o;
    };
    auto issue_piece = [&](const TileOrg& o, int k, bool live) {
        if (k < NY) {
            dma16(yr, live ? yvoff[k] : OOB, o.ybase, o.sb + (wave * NY + k) * 1024);
        } else {
            const int kx = k - NY;
            const int y = (FORM == 0 ? o.y0 : 2 * o.y0) + prr[kx] - 1, xx = (FORM == 0 ? o.x0 : 2 * o.x0) + ppc[kx] - 1;
            unsigned off = OOB;
            if (live && (unsigned)y < (unsigned)Hx && (unsigned)xx < (unsigned)Wx)
                off = (unsigned)((((o.n0 + pim[kx]) * Hx + y) * Wx + xx) * p.Cin + c0 + pkv * 8) * 2u;
            dma16(xr, off, 0, o.sb + YS_BYTES + (wave * XI + kx) * 1024);
        }
    };
    auto relu_own = [&](int stage) {                   // the 16 bytes each lane's x DMA wrote
        unsigned char* xb = lds + stage * STAGE_BYTES + YS_BYTES;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            uint4* q = reinterpret_cast<uint4*>(xb + (wave * XI + k) * 1024 + lane * 16);
            const uint4 v = *q;
            *q = make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
        }
    };

    // ---- fragment geometry (transpose reads): 16-lane group g = lane >> 4: channel half g & 1, pixel half g >> 1; lane q of
    //      the group supplies row (q >> 2) and 4-channel chunk (q & 3)
    const int q = lane & 15, g = lane >> 4;
    const int kro = (g >> 1) * 8 + (q >> 2);
    const int cco = (g & 1) * 16 + (q & 3) * 4;
    int ya2[2];                                         // A (dY) byte offsets of the two cout blocks inside this wave's Ys region
#pragma unroll
    for (int b = 0; b < 2; ++b)
        ya2[b] = (FORM == 0 ? wave * (DPT * 128) : 0) + kro * 128 + (((b * 4 + (cco >> 3)) ^ (((q >> 3) & 1) << 2)) * 16) + (cco & 7) * 2;
    int xrow[8];                                        // byte offset of (V pixel, window origin) in Xp
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int t = (s >> 1) * 16 + kro + (s & 1) * 4;
        const int c = t & (p.Wt - 1), rowi = t / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        xrow[s] = FORM == 0 ? (((im * p.PR1 + rj) * PWC + c) * 32 + cco) * 2 : (((im * p.PR1 + 2 * rj) * PWC + c) * 32 + cco) * 2;
    }

    f32x16 acc[8];                                      // acc[b * 4 + tap]
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum4[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // independent chains (conv_wgrad_dma.hip)
    const bool do_bias = p.do_bias && cc == 0;

    typedef __attribute__((address_space(3))) short4v* lptr;
    typedef __attribute__((ext_vector_type(8))) short short8v;
    // units u = (k-step kk, tap): 2 MFMAs each (both cout blocks); the B fragment of unit u + 2 is read before the MFMAs of u
    auto compute = [&](int stage, auto ph_tag, auto&& dma) {
        constexpr int PH = decltype(ph_tag)::value;
        constexpr int PA = PH >> 1, PB = PH & 1;
        const unsigned char* yb = lds + stage * STAGE_BYTES;
        int xs[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xs[s] = xrow[s] + stage * STAGE_BYTES + YS_BYTES;
        auto rd_a = [&](int kk, int b) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + kk * 16 * 128));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + (kk * 16 + 4) * 128));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int tap) {
            const int tu = tap >> 1, tv = tap & 1;
            int toff;                                                       // compile-time after unrolling
            if (FORM == 0) toff = ((PA + tu) * PWC + (PB + tv)) * 64;
            else {
                const int drow = 2 * tu - PA + 1, dcol = 2 * tv - PB + 1;
                toff = ((drow * 2 + (dcol & 1)) * HALF + (dcol >> 1)) * 64;
            }
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xs[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xs[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2][2], bfr[3];
        af[0][0] = rd_a(0, 0); af[0][1] = rd_a(0, 1);
        bfr[0] = rd_b(0, 0);
        bfr[1] = rd_b(0, 1);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int kk = u >> 2, tap = u & 3;
            if (u + 2 < 16) {
                const int kk2 = (u + 2) >> 2, tap2 = (u + 2) & 3;
                if (tap2 == 0) { af[kk2 & 1][0] = rd_a(kk2, 0); af[kk2 & 1][1] = rd_a(kk2, 1); }
                bfr[(u + 2) % 3] = rd_b(kk2, tap2);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][0], bfr[u % 3], acc[tap], 0, 0, 0);
            acc[4 + tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][1], bfr[u % 3], acc[4 + tap], 0, 0, 0);
            // bias partials: FORM 0 every wave sums its phase; FORM 1 (shared dY tile) wave w sums k-step w
            if (tap == 0 && do_bias && (FORM == 0 || kk == PH)) {
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const uint4 w4 = __builtin_bit_cast(uint4, af[kk & 1][b]);
                    const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) bsum4[b][e] = bf2_sum_acc(ws[e], bsum4[b][e]);
                }
            }
            if (u < PER_TILE) dma(u);
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- 2-stage ring: the DMA of tile t + 1 is issued (in pieces) under the MFMAs of tile t
    auto ring = [&](auto ph_tag) {
        if (t_begin < t_end) {
            const TileOrg o0 = tile_org(t_begin, 0);
#pragma unroll
            for (int k = 0; k < PER_TILE; ++k) issue_piece(o0, k, true);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (p.x_relu) relu_own(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int t = t_begin; t < t_end; ++t) {
            const bool more = t + 1 < t_end;
            const TileOrg org = tile_org(t + 1, stage ^ 1);     // stage ^ 1 was last read in iteration t - 1 (barrier since)
            compute(stage, ph_tag, [&](int k) { issue_piece(org, k, more); });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p.x_relu && more) relu_own(stage ^ 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            stage ^= 1;
        }
    };
    if (wave == 0) ring(std::integral_constant<int, 0>{});
    else if (wave == 1) ring(std::integral_constant<int, 1>{});
    else if (wave == 2) ring(std::integral_constant<int, 2>{});
    else ring(std::integral_constant<int, 3>{});

    // ---- fold the 16 (phase, tap) entries into the 9 taps inside the workgroup, through LDS (the ring is idle now), one
    //      cout block at a time: every wave publishes its 4 entries of the block [wave][tap][reg][lane], then wave w sums
    //      the four entries of taps w, w + 4, w + 8.  The result goes to this split's slab (conv_wgrad_dma's layout:
    //      xmc_internal_wgrad_reduce adds the splits in a fixed order) or, when the launch has a single split, straight into
    //      dW / db (every element has exactly one owner: no atomics).
    const int l31 = lane & 31, lhi = lane >> 5;
    const int J = 9 * p.Cin;
    float* const fl = reinterpret_cast<float*>(lds);
    // (phase bit, window position) of the two entries per axis that contain tap row / column d (xmc_phase_conv_weight's sets)
    auto pair_of = [&](int d, int which, int& a, int& tu) {
        a = which;
        tu = d == 0 ? 0 : (d == 1 ? (which == 0 ? 1 : 0) : 1);
        if (FORM == 1) a = 1 - a;
    };
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 4; ++tap)
#pragma unroll
            for (int e = 0; e < 16; ++e) fl[((wave * 4 + tap) * 16 + e) * 64 + lane] = acc[b * 4 + tap][e];
        __syncthreads();
        for (int t9 = wave; t9 < 9; t9 += 4) {
            const int dy = t9 / 3, dx = t9 - dy * 3;
            int src[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                int a, tu, bb, tv;
                pair_of(dy, k >> 1, a, tu);
                pair_of(dx, k & 1, bb, tv);
                src[k] = (((a * 2 + bb) * 4 + tu * 2 + tv) * 16) * 64 + lane;
            }
            float* const dst = (pr ? pr : p.dw) + (size_t)t9 * p.Cin + c0 + l31;
            float old[16];
            if (!pr && p.overwrite) {
#pragma unroll
                for (int e = 0; e < 16; ++e) old[e] = 0.f;
            } else if (!pr) {                            // single split: all 16 reads of dW in flight before the first store
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int i = i0 + b * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                    old[e] = i < p.Cout ? __builtin_nontemporal_load(dst + (size_t)i * J) : 0.f;
                }
            }
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float sum = (fl[src[0] + e * 64] + fl[src[1] + e * 64]) + (fl[src[2] + e * 64] + fl[src[3] + e * 64]);
                const int i = i0 + b * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (i < p.Cout) dst[(size_t)i * J] = pr ? sum : old[e] + p.alpha * sum;
            }
        }
    }
    if (p.do_bias && cc == 0) {                         // db = sum over the high-resolution dY pixels (FORM 1: each low-res pixel counts 4x)
        __syncthreads();
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const float bs = (bsum4[b][0] + bsum4[b][1]) + (bsum4[b][2] + bsum4[b][3]);
            const float tot = bs + __shfl_xor(bs, 32);
            if (lhi == 0) fl[(wave * 2 + b) * 32 + l31] = tot;
        }
        __syncthreads();
        if (wave == 0) {
            const int b = lhi, i = i0 + b * 32 + l31;
            const float tot = ((fl[(0 * 2 + b) * 32 + l31] + fl[(1 * 2 + b) * 32 + l31]) + (fl[(2 * 2 + b) * 32 + l31] + fl[(3 * 2 + b) * 32 + l31])) *
                              (FORM == 1 ? 4.f : 1.f);
            if (i < p.Cout) {
                if (pr) pr[(size_t)p.Cout * J + i] = tot;
                else p.db[i] = (p.overwrite ? 0.f : p.db[i]) + p.alpha * tot;
            }
        }
    }
}

}  // namespace

#define XMC_WP_VARIANTS(X) X(0, 2, 18) X(0, 2, 10) X(0, 3, 6) X(1, 6, 34) X(1, 6, 18) X(1, 7, 10)
extern "C" int xmc_internal_optin_wgrad_phase(void) {
    static XmcLdsOptIn opt_in;
#define XMC_WP_PTR(F_, XI_, PW_) reinterpret_cast<const void*>(conv_wgrad_phase_kernel<F_, XI_, PW_>),
    return opt_in.ensure({XMC_WP_VARIANTS(XMC_WP_PTR)}, 160 * 1024) ? XMC_OK : XMC_EINVAL;
#undef XMC_WP_PTR
}

// Returns 1 when the launch is outside this kernel's domain (the caller falls back to the 3x3 formulation).
// `query` != NULL: no launch, *query = workspace floats.  The kernel always works through the workspace.
extern "C" int xmc_conv2d_wgrad_phase_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw, float* db,
                                          float* ws, long long* query, void* stream) {
    if (d->dtype != XMC_BF16 || d->ks != 3 || (d->cin % 32) != 0 || (d->cout % 32) != 0) return 1;
    if ((d->x_ups != 0) == (d->dy_ups != 0)) return 1;
    if ((d->variant >> 8) & 1) return 1;                  // A/B switch (XMC_PHASE_CONV=0)
    if (d->x_ups && d->x_relu) return 1;
    const int form = d->x_ups ? 0 : 1;
    WPArgs a{};
    a.x = x; a.dy = dy; a.part = ws; a.dw = dw; a.db = db; a.alpha = d->alpha;
    a.N = d->n; a.Cin = d->cin; a.Cout = d->cout;
    a.Hv = form == 0 ? d->hi : d->hi / 2; a.Wv = form == 0 ? d->wi : d->wi / 2;
    if (form == 1 && ((d->hi & 1) || (d->wi & 1))) return 1;
    a.x_relu = d->x_relu; a.do_bias = db != nullptr || query != nullptr;
    const int l2w = ilog2_exact(a.Wv), l2h = ilog2_exact(a.Hv);
    if (l2w < 2 || l2h < 2) return 1;                     // V grid >= 4 x 4
    const long long m = (long long)a.N * a.Hv * a.Wv;
    if (m % DPT != 0 || m >= (1ll << 29)) return 1;
    const long long xb = (long long)a.N * d->hi * d->wi * a.Cin * 2;
    const long long yb = (long long)a.N * (form == 0 ? 4 : 1) * a.Hv * a.Wv * a.Cout * 2;
    if (xb >= 0x7ffffff0ll || yb >= 0x7ffffff0ll) return 1;
    if (!query && (((uintptr_t)x % 16) || ((uintptr_t)dy % 16) || (ws && ((uintptr_t)ws % 16)))) return 1;
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    a.Wt = a.Wv < 16 ? a.Wv : 16;
    const int rows = DPT / a.Wt;
    a.Rt = rows < a.Hv ? rows : a.Hv;
    a.imgs = DPT / (a.Wt * a.Rt);
    if (a.N % a.imgs != 0) return 1;
    a.log2_tx = l2w - ilog2_exact(a.Wt);
    a.log2_ty = l2h - ilog2_exact(a.Rt);
    const int pw = form == 0 ? a.Wt + 2 : 2 * a.Wt + 2;
    a.PR1 = form == 0 ? a.Rt + 2 : 2 * a.Rt + 2;
    a.PP = a.imgs * a.PR1 * pw;
    const int xi = (a.PP + 63) / 64;
    a.magic_pw = 65536 / (form == 0 ? pw : pw / 2) + 1; a.magic_pr1 = 65536 / a.PR1 + 1;
    a.tiles_i = (a.Cout + 63) / 64;
    a.cchunks = a.Cin / 32;
    a.ntiles = (int)(m / DPT);
    const int slabs = a.tiles_i * a.cchunks;
    const int max_split = (a.ntiles + 3) / 4;
    static const int targets[8] = {768, 768, 512, 1536, 2048, 3072, 4096, 1024};      // (bits 5-7 of variant: A/B sweep of tools/)
    const int t_env = xmc_internal_tuning(XMC_TUNE_WGRAD_TARGET_PHASE);
    int ns = ((((d->variant >> 5) & 7) == 0 && t_env ? t_env : targets[(d->variant >> 5) & 7]) + slabs - 1) / slabs;
    if (ns > max_split) ns = max_split;
    if (ns < 1) ns = 1;
    const int tps = (a.ntiles + ns - 1) / ns;
    const int nsplit = (a.ntiles + tps - 1) / tps;
    a.tiles_per_split = tps; a.nsplit = nsplit;
    a.L = (long long)a.Cout * 9 * a.Cin + a.Cout;
    bool known = false;
#define XMC_WP_KNOWN(F_, XI_, PW_) if (form == F_ && xi == XI_ && pw == PW_) known = true;
    XMC_WP_VARIANTS(XMC_WP_KNOWN)
#undef XMC_WP_KNOWN
    if (!known) return 1;
    if (query) { *query = nsplit > 1 ? (long long)nsplit * a.L : 0; return XMC_OK; }
    if (nsplit > 1 && !ws) return 1;                      // several splits need the workspace (no atomics here)
    a.part = nsplit > 1 ? ws : nullptr;
    a.do_bias = db != nullptr;
    const int overwrite = (d->variant & XMC_WGRAD_OVERWRITE) ? 1 : 0;
    a.overwrite = overwrite && nsplit == 1;
    dim3 grid(slabs * nsplit), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (xmc_internal_optin_wgrad_phase() != XMC_OK) return 1;
#define XMC_WP_LAUNCH(F_, XI_, PW_)                                                                                        \
    if (form == F_ && xi == XI_ && pw == PW_)                                                                              \
        hipLaunchKernelGGL((conv_wgrad_phase_kernel<F_, XI_, PW_>), grid, block,                                           \
                           2 * (size_t)((F_ == 0 ? 4 * DPT * 128 : DPT * 128) + XI_ * 4 * 1024), s, a);
    XMC_WP_VARIANTS(XMC_WP_LAUNCH)
#undef XMC_WP_LAUNCH
    if (a.part) return xmc_internal_wgrad_reduce(a.part, nsplit, a.L, a.L - a.Cout, dw, db, d->alpha, overwrite, stream);
    return xmc_hip_err(hipGetLastError());
}
