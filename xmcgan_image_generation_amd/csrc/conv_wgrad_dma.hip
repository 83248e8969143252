// Weight gradient (bf16) with LDS-DMA staging for gfx950.
//
//   dW[cout][tap][c] += alpha * sum_p dY'(p, cout) * patch(p + tap, c)         (+ fused bias gradient)
//
// Same decomposition as conv_wgrad_patch.hip (one workgroup = 128 cout x 32 cin x all taps, split-K over pixel
// tiles of 64, fragments by ds_read_b64_tr_b16), but that kernel stages each tile through 32 VGPRs with a
// prefetch distance of ONE tile, and with 254 VGPRs per wave there is no room for a second register set: with the
// MFMAs removed its load -> ds_write -> barrier pipeline alone takes 88 % of the kernel's time on the
// 96-channel 128^2 layers.  Here both operands travel global -> LDS by `buffer_load_dwordx4 ... lds` (no staging
// registers at all) into a 3-stage LDS ring, TWO tiles ahead of the MFMAs, retired by counted s_waitcnt vmcnt
// and raw s_barrier (a __syncthreads would drain the DMA queue).
//
// The DMA writes lane-linearly (M0 base + lane * 16 B), so the LDS image cannot be padded:
//   Ys [64 pixels][128 cout]  256-byte rows; the 16-byte slot s of pixel row r is stored at slot s ^ ((r & 3) << 2)
//      (the involution is applied to the per-lane SOURCE address and to the per-lane constant of the transpose
//      read): the 4 rows x 64 B of one ds_read_b64_tr_b16 pass land on 4 disjoint bank quarters;
//   Xp [patch pixels][32 cin] 64-byte rows, linear (4 consecutive rows = all 64 banks); out-of-image pixels use
//      an out-of-range buffer offset -> the DMA writes zeros.
// ReLU on the input (x_relu) cannot ride the DMA: each lane rewrites the 16 bytes IT fetched (ds_read, relu,
// ds_write) after its own vmcnt wait and before the barrier that publishes the stage.  The bias gradient is
// summed from the A fragments already in registers.
#include <cstdlib>
#include <type_traits>

#include "common.h"

#ifndef WG_ABL
#define WG_ABL 0    // timing ablations (results wrong): 1 no B fragment reads, 2 no A fragment reads, 4 no DMA after the first tiles, 8 no per-tile barrier, 16 no partial-slab stores, 32 DMA issued but never waited for, 64 no dY DMA, 128 no x DMA (both without the counted wait)
#endif

namespace {

constexpr int DPT = 64;                  // output pixels per tile
constexpr int YS_BYTES = DPT * 256;      // 16 KB

struct WDArgs {
    const void* x; const void* dy; float* dw; float* db;
    int N, Hi, Wi, Cin, Ho, Wo, Cout, Hd, Wd;
    int x_ups, x_relu, dy_ups;
    int log2_tx, log2_ty;
    int tiles_i, cchunks, tiles_per_split, ntiles, nsplit;
    int Wt, Rt, imgs, PW, PR1, PP, magic_pw, magic_pr1;
    int stage_bytes, xcd;
    unsigned x_bytes, dy_bytes;
    float alpha;
    float* part; long long L;        // deterministic split-K: partial slabs part[split][L] (nullptr: float atomics)
    int overwrite;                   // single-split launch (every element of dw / db has ONE owner): plain store of alpha * acc
};

// XI = x-patch DMA instructions per wave per tile (16 patch pixels each): 2 (<= 128 patch pixels) or 3 (<= 192).
// PWC = patch row width in pixels as a COMPILE-TIME constant (tile width + 2 * halo: 18 / 10 / 6 for 3x3, 16 / 8 / 4
// for 1x1): the LDS byte offset of every filter tap is then an immediate of its ds_read_b64_tr_b16.  With PW a
// kernel argument each of the 18 transpose reads of a k-step paid a v_add (PMC on the 96-channel 128^2 layer: 3.4
// VALU per MFMA, 41 % of wave cycles issue-stalled).
// CB (1x1 only) = 32-channel cin blocks per workgroup: a 1x1 tile stages 16 KB of dY for FOUR MFMAs per wave with one cin
// block -- the kernel then runs at the speed of its DMA (the 4224 x 1024 x 14336 gradient of the batched local-cBN
// projections: 396 us = 313 TF/s).  With CB blocks the x tile is [CB][64 px][32 cin] and block j plays the part of "tap" j
// (LDS offset j * 4 KB): CB MFMAs per k-step per wave from the same A fragment.  XI == CB (one DMA instruction per wave
// per block).
// C96 (3x3 only; round 5) = 96-cout workgroup tiles for layers whose cout count is 96 / 192 / 288 ...: every channel count of the
// network is a multiple of 96, and in a 128-cout tile those layers multiply a quarter of their MFMAs by a zero cout block (the
// three worst rows of the per-layer table: D 64^2 96 -> 192, G 64^2 192 -> 192, G 128^2 96 -> 96).  The 27 (cout block, tap)
// products of a k-step go to the four waves as 7 / 7 / 7 / 6: wave r < 3 = block r x taps 0..6 (1 A + 7 B fragments), wave 3 =
// taps 7, 8 x the three blocks (3 A + 2 B).  Staging, LDS image (128-cout rows, the fourth block never fetched) and slab layout
// are the 128-cout kernel's.
// NST = stages of the LDS ring.  3 everywhere except the 4 x 4 maps (XI = 3, PWC = 6: four images per tile, 144 patch pixels): their
// 28 KB stages x 3 leave ONE workgroup per CU (84 KB of 160), and the 576 slabs of a 1536 -> 1536 layer then take three rounds of
// 256 -- 484 TF/s, the slowest row of the per-layer table.  Two stages (56 KB: two workgroups per CU, the second covers the one-tile
// prefetch distance) as conv_wgrad_phase.hip's ring.
template <int KS, int XI, int PWC, int CB = 1, bool C96 = false, int NST = 3>
__global__ __launch_bounds__(256, 2) void conv_wgrad_dma_kernel(const WDArgs p) {
    static_assert(NST == 2 || NST == 3, "ring depth");
    static_assert(CB == 1 || (KS == 1 && XI == CB), "cin blocks: pointwise only, one x DMA instruction per block");
    static_assert(!C96 || (KS == 3 && CB == 1), "96-cout tiles: 3x3 only");
    constexpr int TAPS = KS == 1 ? CB : KS * KS, HALO = KS / 2;
    constexpr int STAGE_BYTES = YS_BYTES + XI * 4 * 1024;
    constexpr int PER_TILE = 4 + XI;                  // DMA instructions per wave per tile
    constexpr unsigned OOB = 0xfffffff0u;
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slabs = p.tiles_i * p.cchunks;
    // XCD-aware order: the slabs of one pixel split read the SAME dY tiles (every 32-channel slab of a cout tile) and the
    // same x patches (every cout tile of a channel slab).  The dispatcher puts block b on XCD b % 8, so in launch order the
    // sharers of a tile sit on 8 different L2s and each fetches it from HBM (PMC round 2: 3.6x the algorithmic bytes);
    // with a contiguous range of (split, slab) per XCD they run side by side on one XCD and re-read from its L2.
    const int wid = p.xcd ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const int slab = wid % slabs, split = wid / slabs;
    const int ti = slab / p.cchunks, cc = slab - ti * p.cchunks;
    const int i0 = ti * (C96 ? 96 : 128), c0 = cc * 32 * CB;
    const int t_begin = split * p.tiles_per_split;
    const int t_end = min(p.ntiles, t_begin + p.tiles_per_split);
    if (t_begin >= t_end) return;

    const v4i32 xr = make_srd(p.x, p.x_bytes), yr = make_srd(p.dy, p.dy_bytes);
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);

    // ---- dY DMA: instruction k of this wave covers tile pixel rows (wave * 4 + k) * 4 .. + 3; lane -> (row, slot)
    unsigned yvoff[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int t = (wave * 4 + k) * 4 + (lane >> 4);          // tile-local pixel
        const int slot = (lane & 15) ^ ((t & 3) << 2);           // source slot of LDS position (lane & 15)
        const int co = i0 + slot * 8;
        const int c = t & (p.Wt - 1), rowi = t / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        const int ly = p.dy_ups ? (rj >> 1) : rj, lx = p.dy_ups ? (c >> 1) : c;
        yvoff[k] = (co < p.Cout && (!C96 || slot < 12)) ? (unsigned)(((im * p.Hd + ly) * p.Wd + lx) * p.Cout + co) * 2u : OOB;
    }
    // ---- x DMA: instruction k covers patch pixels (wave * XI + k) * 16 .. + 15; lane -> (pixel, 16-byte slot)
    int prr[XI], ppc[XI], pim[XI];
#pragma unroll
    for (int k = 0; k < XI; ++k) {
        const int pp = CB > 1 ? wave * 16 + (lane >> 2) : (wave * XI + k) * 16 + (lane >> 2);   // CB > 1: instruction k = cin block k
        const int pr = (pp * p.magic_pw) >> 16;
        ppc[k] = pp < p.PP ? pp - pr * p.PW : -1000000;          // dead pixels: always out of the image
        pim[k] = (pr * p.magic_pr1) >> 16;
        prr[k] = pr - pim[k] * p.PR1;
    }
    const int pkv = lane & 3;

    auto issue_tile = [&](int t, int stage) {
        const int x0 = (t & ((1 << p.log2_tx) - 1)) * p.Wt;
        const int y0 = ((t >> p.log2_tx) & ((1 << p.log2_ty) - 1)) * p.Rt;
        const int n0 = (t >> (p.log2_tx + p.log2_ty)) * p.imgs;
        const int ybase = p.dy_ups ? (((n0 * p.Hd + (y0 >> 1)) * p.Wd + (x0 >> 1)) * p.Cout) * 2
                                   : (((n0 * p.Ho + y0) * p.Wo + x0) * p.Cout) * 2;
        const unsigned sb = lds0 + stage * STAGE_BYTES;
#pragma unroll
        for (int k = 0; k < 4; ++k) dma16(yr, yvoff[k], ybase, sb + (wave * 4 + k) * 1024);
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            const int y = y0 + prr[k] - HALO, xx = x0 + ppc[k] - HALO;
            unsigned off = OOB;
            if ((unsigned)y < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo) {
                const int sy = p.x_ups ? (y >> 1) : y, sx = p.x_ups ? (xx >> 1) : xx;
                off = (unsigned)((((n0 + pim[k]) * p.Hi + sy) * p.Wi + sx) * p.Cin + c0 + (CB > 1 ? k * 32 : 0) + pkv * 8) * 2u;
            }
            dma16(xr, off, 0, sb + YS_BYTES + (CB > 1 ? k * 4096 + wave * 1024 : (wave * XI + k) * 1024));
        }
    };
    // The same tile, one DMA instruction at a time (piece k of PER_TILE), for the steady state: the pieces are spread over
    // the MFMA units of the tile in front instead of being issued as a clump at its top (ablation: the clump costs 19 % of
    // the kernel although nothing ever waits for its data -- back-pressure of 6-7 VMEM issues in a row, with the MFMAs of
    // the wave queued behind them).  `live` false: out-of-range offsets (the DMA writes zeros to a stage nobody reads), so
    // the loop body has no branch and the vmcnt bookkeeping is the same for every tile.
    struct TileOrg { int x0, y0, n0, ybase; unsigned sb; };
    auto tile_org = [&](int t, int stage) {
        TileOrg o;
        o.x0 = (t & ((1 << p.log2_tx) - 1)) * p.Wt;
        o.y0 = ((t >> p.log2_tx) & ((1 << p.log2_ty) - 1)) * p.Rt;
        o.n0 = (t >> (p.log2_tx + p.log2_ty)) * p.imgs;
        o.ybase = p.dy_ups ? (((o.n0 * p.Hd + (o.y0 >> 1)) * p.Wd + (o.x0 >> 1)) * p.Cout) * 2
                           : (((o.n0 * p.Ho + o.y0) * p.Wo + o.x0) * p.Cout) * 2;
        o.sb = lds0 + stage * STAGE_BYTES;
        return o;
    };
    auto issue_piece = [&](const TileOrg& o, int k, bool live) {
        if (k < 4) {
            if constexpr (!(WG_ABL & 64)) dma16(yr, live ? yvoff[k] : OOB, o.ybase, o.sb + (wave * 4 + k) * 1024);
        } else if constexpr (!(WG_ABL & 128)) {
            const int kx = k - 4;
            const int y = o.y0 + prr[kx] - HALO, xx = o.x0 + ppc[kx] - HALO;
            unsigned off = OOB;
            if (live && (unsigned)y < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo) {
                const int sy = p.x_ups ? (y >> 1) : y, sx = p.x_ups ? (xx >> 1) : xx;
                off = (unsigned)((((o.n0 + pim[kx]) * p.Hi + sy) * p.Wi + sx) * p.Cin + c0 + (CB > 1 ? kx * 32 : 0) + pkv * 8) * 2u;
            }
            dma16(xr, off, 0, o.sb + YS_BYTES + (CB > 1 ? kx * 4096 + wave * 1024 : (wave * XI + kx) * 1024));
        }
    };
    auto relu_own = [&](int stage) {                   // the 16 bytes each lane's x DMA wrote
        unsigned char* xb = lds + stage * STAGE_BYTES + YS_BYTES;
#pragma unroll
        for (int k = 0; k < XI; ++k) {
            uint4* q = reinterpret_cast<uint4*>(xb + (CB > 1 ? k * 4096 + wave * 1024 : (wave * XI + k) * 1024) + lane * 16);
            const uint4 v = *q;
            *q = make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
        }
    };

    // ---- fragment geometry (transpose reads).  16-lane group g = lane >> 4: channel half g & 1,
    //      pixel half g >> 1; lane q of the group supplies row (q >> 2) and 4-channel chunk (q & 3).
    const int q = lane & 15, g = lane >> 4;
    const int kro = (g >> 1) * 8 + (q >> 2);            // pixel row offset inside a 16-pixel k-step
    const int cco = (g & 1) * 16 + (q & 3) * 4;         // channel offset of this lane's 8-byte chunk
    // A (dY): row kk * 16 + kro (+4); unswizzled slot = wave * 4 + cco / 8; row & 3 == (q >> 2)
    const int ya = kro * 256 + (((wave * 4 + (cco >> 3)) ^ ((q >> 2) << 2)) * 16) + (cco & 7) * 2;
    int xrow[8];                                        // byte offset of (source pixel, tap (0,0)) in Xp
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int t = (s >> 1) * 16 + kro + (s & 1) * 4;
        const int c = t & (p.Wt - 1), rowi = t / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        xrow[s] = (((im * p.PR1 + rj) * PWC + c) * 32 + cco) * 2;
    }
    // ---- work split inside the workgroup (128 cout x 32 cin x TAPS).
    // 1x1: wave w = cout block w.  3x3: wave = (cout half h2 = w >> 1: blocks 2 h2, 2 h2 + 1) x (tap group tg = w & 1): taps
    // 4 tg .. 4 tg + 3 for BOTH blocks plus tap 8 for block tg only -- 9 MFMAs per 16-pixel k-step in EVERY wave (36 per
    // tile), fed by 2 A + 5 B fragments = 1.56 transpose reads per MFMA instead of the 2.2 of "one cout block x all 9
    // taps".  (Round 2 split the taps 5 / 4: 40 vs 32 MFMAs per tile, and the per-tile barrier made every wave wait for
    // the 40s.  The kernel is bound by per-wave issue, not by the matrix pipe: PMC 36 % issuing / 38 % issue-stalled.)
    constexpr bool SPLIT_TAPS = KS == 3;
    constexpr int NACC = C96 ? 7 : (SPLIT_TAPS ? 9 : TAPS);
    const int h2 = wave >> 1, tg = wave & 1;
    int ya2[C96 ? 3 : 2];                               // A (dY) byte offsets of the wave's two cout blocks (3x3 mapping; C96: of blocks 0..2)
#pragma unroll
    for (int b = 0; b < (C96 ? 3 : 2); ++b)
        ya2[b] = kro * 256 + (((((C96 ? 0 : h2 * 2) + b) * 4 + (cco >> 3)) ^ ((q >> 2) << 2)) * 16) + (cco & 7) * 2;
    f32x16 acc[NACC];                                   // 3x3: acc[b * 4 + local tap], acc[8] = tap 8 of block tg
#pragma unroll
    for (int t = 0; t < NACC; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;
    float bsum4[4] = {0.f, 0.f, 0.f, 0.f};              // four independent chains: a single one serialises 4 dependent dots per k-step
    const bool do_bias = p.db != nullptr && cc == 0 && (!C96 || wave < 3);   // 3x3: wave tg sums cout block tg of its half (C96: wave r block r)

    typedef __attribute__((address_space(3))) short4v* lptr;
    typedef __attribute__((ext_vector_type(8))) short short8v;
    constexpr int GRP = TAPS == 9 ? 3 : 1, NG = TAPS / GRP, UNITS = 4 * NG;
    auto compute = [&](int stage, auto&& dma) {
        const unsigned char* yb = lds + stage * STAGE_BYTES + ya;
        const unsigned char* xb = lds + stage * STAGE_BYTES + YS_BYTES;
        int xr[8];                                      // this stage's x rows: one add per tile, taps are immediates
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = xrow[s] + stage * STAGE_BYTES + YS_BYTES;
        (void)xb;
        auto rd_a = [&](int kk) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + kk * 16 * 256));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + (kk * 16 + 4) * 256));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int t) {
            const int toff = CB > 1 ? t * 4096 : ((t / KS) * PWC + (t % KS)) * 64;     // compile-time after unrolling
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2], bfr[2][GRP];
        af[0] = rd_a(0);
#pragma unroll
        for (int t = 0; t < GRP; ++t) bfr[0][t] = rd_b(0, t);
#pragma unroll
        for (int u = 0; u < UNITS; ++u) {
            const int kk = u / NG, gi = u % NG;
            if (u + 1 < UNITS) {
                const int kk1 = (u + 1) / NG, g1 = (u + 1) % NG;
                if (g1 == 0) af[kk1 & 1] = rd_a(kk1);
#pragma unroll
                for (int t = 0; t < GRP; ++t) bfr[(u + 1) & 1][t] = rd_b(kk1, g1 * GRP + t);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int t = 0; t < GRP; ++t)
                acc[gi * GRP + t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1], bfr[u & 1][t], acc[gi * GRP + t], 0, 0, 0);
            if (gi == 0 && do_bias) {                   // this lane's 8 pixels of output channel (lane & 31)
                const uint4 w4 = __builtin_bit_cast(uint4, af[kk & 1]);
                const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum4[e] = bf2_sum_acc(ws[e], bsum4[e]);
            }
            if (UNITS >= 2 * PER_TILE ? (u % 2 == 1 && u / 2 < PER_TILE) : u < PER_TILE) dma(UNITS >= 2 * PER_TILE ? u / 2 : u);
            if (UNITS < PER_TILE && u == UNITS - 1)
                for (int k = UNITS; k < PER_TILE; ++k) dma(k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // 3x3 mapping: units u = (k-step kk, slot sl): slots 0..3 = taps 4 TG + sl on both cout blocks (2 MFMAs), slot 4 = tap 8
    // on block TG (1 MFMA); the B fragment of unit u + 2 is read before the MFMAs of unit u
    auto compute3 = [&](int stage, auto tg_tag, auto&& dma) {
        constexpr int TG = decltype(tg_tag)::value;
        constexpr int NS = 5, UN = 4 * NS;
        const unsigned char* yb = lds + stage * STAGE_BYTES;
        int xr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = xrow[s] + stage * STAGE_BYTES + YS_BYTES;
        auto rd_a = [&](int kk, int b) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + kk * 16 * 256));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + (kk * 16 + 4) * 256));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int sl) {
            const int t = sl == 4 ? 8 : TG * 4 + sl;
            const int toff = ((t / KS) * PWC + (t % KS)) * 64;              // compile-time after unrolling
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2][2], bfr[3];
        af[0][0] = rd_a(0, 0); af[0][1] = rd_a(0, 1);
        bfr[0] = rd_b(0, 0);
        bfr[1] = rd_b(0, 1);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = u / NS, sl = u % NS;
            if (u + 2 < UN) {
                const int kk2 = (u + 2) / NS, sl2 = (u + 2) % NS;
                if (sl2 == 0) {
                    if constexpr (!(WG_ABL & 2)) { af[kk2 & 1][0] = rd_a(kk2, 0); af[kk2 & 1][1] = rd_a(kk2, 1); }
                    else { af[kk2 & 1][0] = af[0][0]; af[kk2 & 1][1] = af[0][1]; }
                }
                if constexpr (!(WG_ABL & 1)) bfr[(u + 2) % 3] = rd_b(kk2, sl2);
                else bfr[(u + 2) % 3] = bfr[0];
            }
            __builtin_amdgcn_sched_barrier(0);
            if (sl < 4) {
                acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][0], bfr[u % 3], acc[sl], 0, 0, 0);
                acc[4 + sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][1], bfr[u % 3], acc[4 + sl], 0, 0, 0);
            } else {
                acc[8] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][TG], bfr[u % 3], acc[8], 0, 0, 0);
            }
            if (sl == 0 && do_bias) {                   // this lane's 8 pixels of output channel (lane & 31) of block TG
                const uint4 w4 = __builtin_bit_cast(uint4, af[kk & 1][TG]);
                const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum4[e] = bf2_sum_acc(ws[e], bsum4[e]);
            }
            if (u % 2 == 1 && u / 2 < PER_TILE) dma(u / 2);     // 20 units, <= 7 pieces: one after every other unit
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // 96-cout mapping, waves 0..2 (ROLE = cout block): units u = (k-step kk, tap sl in 0..6), one MFMA each; the B fragment of unit
    // u + 2 and the A fragment of the next k-step are read ahead of the MFMA of unit u
    auto compute96 = [&](int stage, auto role_tag, auto&& dma) {
        constexpr int ROLE = decltype(role_tag)::value;
        constexpr int NS = 7, UN = 4 * NS;
        const unsigned char* yb = lds + stage * STAGE_BYTES;
        int xr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = xrow[s] + stage * STAGE_BYTES + YS_BYTES;
        auto rd_a = [&](int kk) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[ROLE] + kk * 16 * 256));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[ROLE] + (kk * 16 + 4) * 256));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int t) {
            const int toff = ((t / KS) * PWC + (t % KS)) * 64;                 // compile-time after unrolling
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2], bfr[3];
        af[0] = rd_a(0);
        bfr[0] = rd_b(0, 0);
        bfr[1] = rd_b(0, 1);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = u / NS, sl = u % NS;
            if (u + 2 < UN) {
                const int kk2 = (u + 2) / NS, sl2 = (u + 2) % NS;
                if (sl2 == 0) af[kk2 & 1] = rd_a(kk2);
                bfr[(u + 2) % 3] = rd_b(kk2, sl2);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1], bfr[u % 3], acc[sl], 0, 0, 0);
            if (sl == 0 && do_bias) {                   // this lane's 8 pixels of output channel (lane & 31) of block ROLE
                const uint4 w4 = __builtin_bit_cast(uint4, af[kk & 1]);
                const unsigned ws[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) bsum4[e] = bf2_sum_acc(ws[e], bsum4[e]);
            }
            if (u % 2 == 1 && u / 2 < PER_TILE) dma(u / 2);     // 28 units, <= 7 pieces
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // 96-cout mapping, wave 3: taps 7 and 8 on the three cout blocks -- units u = (kk, sl): tap 7 + sl / 3, block sl % 3; the five
    // fragments of the next k-step are read at the top of this one
    auto compute96w3 = [&](int stage, auto&& dma) {
        constexpr int NS = 6, UN = 4 * NS;
        const unsigned char* yb = lds + stage * STAGE_BYTES;
        int xr[8];
#pragma unroll
        for (int s = 0; s < 8; ++s) xr[s] = xrow[s] + stage * STAGE_BYTES + YS_BYTES;
        auto rd_a = [&](int kk, int b) {
            const short4v a0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + kk * 16 * 256));
            const short4v a1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(yb + ya2[b] + (kk * 16 + 4) * 256));
            const short8v av = {a0[0], a0[1], a0[2], a0[3], a1[0], a1[1], a1[2], a1[3]};
            return __builtin_bit_cast(bf16x8, av);
        };
        auto rd_b = [&](int kk, int t) {
            const int toff = ((t / KS) * PWC + (t % KS)) * 64;
            const short4v b0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk] + toff));
            const short4v b1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(lds + xr[2 * kk + 1] + toff));
            const short8v bv = {b0[0], b0[1], b0[2], b0[3], b1[0], b1[1], b1[2], b1[3]};
            return __builtin_bit_cast(bf16x8, bv);
        };
        bf16x8 af[2][3], bfr[2][2];
#pragma unroll
        for (int b = 0; b < 3; ++b) af[0][b] = rd_a(0, b);
        bfr[0][0] = rd_b(0, 7); bfr[0][1] = rd_b(0, 8);
#pragma unroll
        for (int u = 0; u < UN; ++u) {
            const int kk = u / NS, sl = u % NS;
            if (sl == 0 && kk + 1 < 4) {
#pragma unroll
                for (int b = 0; b < 3; ++b) af[(kk + 1) & 1][b] = rd_a(kk + 1, b);
                bfr[(kk + 1) & 1][0] = rd_b(kk + 1, 7); bfr[(kk + 1) & 1][1] = rd_b(kk + 1, 8);
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[sl] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[kk & 1][sl % 3], bfr[kk & 1][sl / 3], acc[sl], 0, 0, 0);
            if (u % 2 == 1 && u / 2 < PER_TILE) dma(u / 2);     // 24 units, <= 7 pieces
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    // ---- 3-stage ring, two tiles in flight.  The 3x3 mapping runs one of two instantiations of the whole loop (5 or 4
    //      tap group per wave: the tap offsets stay immediates); every wave meets the same barriers in either.
    auto ring = [&](auto&& compute_fn) {
        if constexpr (NST == 2) {
            issue_tile(t_begin, 0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p.x_relu) relu_own(0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            int stage = 0;
            for (int t = t_begin; t < t_end; ++t) {
                const bool more = t + 1 < t_end;
                const TileOrg org = tile_org(t + 1, stage ^ 1);     // stage ^ 1 was last read in iteration t - 1 (barrier since)
                compute_fn(stage, [&](int k) { issue_piece(org, k, more); });
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // tile t + 1 (or its dummy) landed
                if (p.x_relu && more) relu_own(stage ^ 1);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                stage ^= 1;
            }
            return;
        }
        issue_tile(t_begin, 0);
        if (t_begin + 1 < t_end) {
            issue_tile(t_begin + 1, 1);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (p.x_relu) relu_own(0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        int stage = 0;
        for (int t = t_begin; t < t_end; ++t) {
            const int s1 = stage == 2 ? 0 : stage + 1, s2 = s1 == 2 ? 0 : s1 + 1;
            const bool more2 = t + 2 < t_end;
            const TileOrg org = tile_org(t + 2, s2);        // stage s2 was last read in iteration t - 1 (barrier since)
            compute_fn(stage, [&](int k) { if constexpr (!(WG_ABL & 4)) issue_piece(org, k, more2); });
            if constexpr (!(WG_ABL & 4) && !(WG_ABL & 32) && !(WG_ABL & 192))
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER_TILE) : "memory");   // tile t + 1 landed, t + 2 (or its dummy) in flight
            if (p.x_relu && t + 1 < t_end) relu_own(s1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if constexpr (!(WG_ABL & 8)) __builtin_amdgcn_s_barrier();
            stage = s1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    if constexpr (C96) {
        if (wave == 0) ring([&](int stage, auto&& dma) { compute96(stage, std::integral_constant<int, 0>{}, dma); });
        else if (wave == 1) ring([&](int stage, auto&& dma) { compute96(stage, std::integral_constant<int, 1>{}, dma); });
        else if (wave == 2) ring([&](int stage, auto&& dma) { compute96(stage, std::integral_constant<int, 2>{}, dma); });
        else ring([&](int stage, auto&& dma) { compute96w3(stage, dma); });
    } else if constexpr (SPLIT_TAPS) {
        if (tg == 0) ring([&](int stage, auto&& dma) { compute3(stage, std::integral_constant<int, 0>{}, dma); });
        else ring([&](int stage, auto&& dma) { compute3(stage, std::integral_constant<int, 1>{}, dma); });
    } else {
        ring(compute);
    }

    // ---- D[i = cout][j = cin]: col = lane & 31 -> cin (contiguous in dW), rows -> cout
    const int l31 = lane & 31, lhi = lane >> 5;
    const int J = (KS == 1 ? 1 : TAPS) * p.Cin;
    float* const pr = p.part ? p.part + (size_t)split * p.L : nullptr;     // this split's slab (plain stores)
    if constexpr (C96) {
#pragma unroll
        for (int ai = 0; ai < 7; ++ai) {
            if (wave == 3 && ai == 6) continue;                            // wave 3 holds six accumulators
            const int b = wave < 3 ? wave : ai % 3;                        // cout block of this accumulator
            const int t = wave < 3 ? ai : 7 + ai / 3;                      // filter tap
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + b * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (i < p.Cout) {
                    const size_t o = (size_t)i * J + t * p.Cin + c0 + l31;
                    if (pr) pr[o] = acc[ai][e];
                    else if (p.overwrite) p.dw[o] = p.alpha * acc[ai][e];
                    else atomicAdd(p.dw + o, p.alpha * acc[ai][e]);
                }
            }
        }
        if (do_bias) {
            const float bsum = (bsum4[0] + bsum4[1]) + (bsum4[2] + bsum4[3]);
            const float tot = bsum + __shfl_xor(bsum, 32);
            const int i = i0 + wave * 32 + l31;
            if (lhi == 0 && i < p.Cout) {
                if (pr) pr[(size_t)p.Cout * J + i] = tot;
                else if (p.overwrite) p.db[i] = p.alpha * tot;
                else atomicAdd(p.db + i, p.alpha * tot);
            }
        }
    } else if constexpr (SPLIT_TAPS) {
#pragma unroll
        for (int ai = 0; ai < 9; ++ai) {
            const int b = ai == 8 ? tg : ai >> 2;                          // cout block of this accumulator
            const int t = ai == 8 ? 8 : tg * 4 + (ai & 3);                 // filter tap
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + (h2 * 2 + b) * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (i < p.Cout) {
                    const size_t o = (size_t)i * J + t * p.Cin + c0 + l31;
                    if constexpr ((WG_ABL & 16) != 0) { if (acc[ai][e] == -12345.678f) pr[o] = 1.f; }
                    else if (pr) pr[o] = acc[ai][e];
                    else if (p.overwrite) p.dw[o] = p.alpha * acc[ai][e];
                    else atomicAdd(p.dw + o, p.alpha * acc[ai][e]);
                }
            }
        }
        if (do_bias) {
            const float bsum = (bsum4[0] + bsum4[1]) + (bsum4[2] + bsum4[3]);
            const float tot = bsum + __shfl_xor(bsum, 32);
            const int i = i0 + (h2 * 2 + tg) * 32 + l31;
            if (lhi == 0 && i < p.Cout) {
                if (pr) pr[(size_t)p.Cout * J + i] = tot;
                else if (p.overwrite) p.db[i] = p.alpha * tot;
                else atomicAdd(p.db + i, p.alpha * tot);
            }
        }
    } else {
#pragma unroll
        for (int t = 0; t < TAPS; ++t)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int i = i0 + wave * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (i < p.Cout) {
                    const size_t o = (size_t)i * J + (CB > 1 ? t * 32 : t * p.Cin) + c0 + l31;
                    if (pr) pr[o] = acc[t][e];
                    else if (p.overwrite) p.dw[o] = p.alpha * acc[t][e];
                    else atomicAdd(p.dw + o, p.alpha * acc[t][e]);
                }
            }
        if (do_bias) {
            const float bsum = (bsum4[0] + bsum4[1]) + (bsum4[2] + bsum4[3]);
            const float tot = bsum + __shfl_xor(bsum, 32);
            const int i = i0 + wave * 32 + l31;
            if (lhi == 0 && i < p.Cout) {
                if (pr) pr[(size_t)p.Cout * J + i] = tot;
                else if (p.overwrite) p.db[i] = p.alpha * tot;
                else atomicAdd(p.db + i, p.alpha * tot);
            }
        }
    }
}

}  // namespace

#define XMC_WD_VARIANTS(X) X(3, 2, 18, 1, false) X(3, 2, 10, 1, false) X(3, 3, 6, 1, false) X(3, 2, 6, 1, false) X(3, 3, 18, 1, false) X(3, 3, 10, 1, false) \
                           X(1, 2, 16, 1, false) X(1, 2, 8, 1, false) X(1, 2, 4, 1, false) X(1, 3, 16, 1, false) X(1, 3, 8, 1, false) X(1, 3, 4, 1, false) \
                           X(1, 2, 16, 2, false) X(1, 2, 8, 2, false) X(1, 2, 4, 2, false) X(1, 4, 16, 4, false) X(1, 4, 8, 4, false) X(1, 4, 4, 4, false) \
                           X(3, 2, 18, 1, true)
#define XMC_WD_NST2(X) X(3, 3, 6, 1, false)
extern "C" int xmc_internal_optin_wgrad_dma(void) {
    static XmcLdsOptIn opt_in;
#define XMC_WD_PTR(KS_, XI_, PW_, CB_, C96_) reinterpret_cast<const void*>(conv_wgrad_dma_kernel<KS_, XI_, PW_, CB_, C96_>),
#define XMC_WD_PTR2(KS_, XI_, PW_, CB_, C96_) reinterpret_cast<const void*>(conv_wgrad_dma_kernel<KS_, XI_, PW_, CB_, C96_, 2>),
    return opt_in.ensure({XMC_WD_VARIANTS(XMC_WD_PTR) XMC_WD_NST2(XMC_WD_PTR2)}, 160 * 1024) ? XMC_OK : XMC_EINVAL;
#undef XMC_WD_PTR2
#undef XMC_WD_PTR
}

// `query` != NULL: no launch, *query = workspace floats the deterministic mode needs for this shape (0: single split).
// `ws` != NULL (and more than one split): partial slabs + fixed-order reduction instead of float atomics.
extern "C" int xmc_conv2d_wgrad_dma_try(const xmc_wgrad_desc* d, const void* x, const void* dy, float* dw,
                                        float* db, float* ws, long long* query, void* stream) {
    if (d->dtype != XMC_BF16 || (d->cin % 32) != 0 || (d->cout % 8) != 0) return 1;
    WDArgs a;
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
    if (m % DPT != 0 || m >= (1ll << 31)) return 1;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2, yb = (long long)a.N * a.Hd * a.Wd * a.Cout * 2;
    if (xb >= 0x7ffffff0ll || yb >= 0x7ffffff0ll) return 1;
    if (((uintptr_t)x % 16) || ((uintptr_t)dy % 16)) return 1;
    a.x_bytes = (unsigned)xb; a.dy_bytes = (unsigned)yb;
    const int halo = d->ks / 2;
    a.Wt = a.Wo < 16 ? a.Wo : 16;
    const int rows = DPT / a.Wt;
    a.Rt = rows < a.Ho ? rows : a.Ho;
    a.imgs = DPT / (a.Wt * a.Rt);
    if (a.N % a.imgs != 0) return 1;
    if (d->dy_ups && a.Wt < 2) return 1;
    a.log2_tx = l2w - ilog2_exact(a.Wt);
    a.log2_ty = l2h - ilog2_exact(a.Rt);
    a.PW = a.Wt + 2 * halo; a.PR1 = a.Rt + 2 * halo;
    a.PP = a.imgs * a.PR1 * a.PW;
    if (a.PP > 192) return 1;
    a.magic_pw = 65536 / a.PW + 1; a.magic_pr1 = 65536 / a.PR1 + 1;
    // 1x1: cin blocks per workgroup (bits 9-10 of variant force 1 / 2 / 4 for A/B runs): 2 where the channel count allows
    int cb = 1;
    if (d->ks == 1 && a.PP <= 64) {
        // measured (tools/bench_wgrad_1x1.py): 2 blocks -10..-25 % from 2k pixels up, 4 blocks only on the widest launch
        // (4224 couts: 344 -> 230 us); below 2k pixels the single block's 2x workgroups win
        if (m >= 2048 && (a.Cin % 128) == 0 && a.Cout >= 2048) cb = 4;
        else if (m >= 2048 && (a.Cin % 64) == 0) cb = 2;
        const int force = (d->variant >> 9) & 3;
        if (force == 1) cb = 1;
        else if (force == 2 && (a.Cin % 64) == 0) cb = 2;
        else if (force == 3 && (a.Cin % 128) == 0) cb = 4;
    }
    const int xi = cb > 1 ? cb : (a.PP <= 128 ? 2 : 3);
    a.stage_bytes = YS_BYTES + xi * 4 * 1024;
    // 96-cout tiles where they waste fewer zero couts than 128-cout ones (Cout = 96, 192, 288: every channel count of the
    // network is a multiple of 96); the one instantiation covers the 16-pixel-wide tiles (maps >= 16 x 16), where those
    // layers live.  (bit 11 of variant: off -- A/B)
    const bool c96 = d->ks == 3 && xi == 2 && a.PW == 18 && cb == 1 && !((d->variant >> 11) & 1) &&
                     ((a.Cout + 95) / 96) * 96 < ((a.Cout + 127) / 128) * 128;
    a.tiles_i = c96 ? (a.Cout + 95) / 96 : (a.Cout + 127) / 128;
    a.cchunks = a.Cin / (32 * cb);
    a.ntiles = (int)(m / DPT);
    const int slabs = a.tiles_i * a.cchunks;
    // Pixel split count and launch order.  A sweep over the 21 C1 layer shapes (profiles/r03_wgrad_split_sweep.txt; variants
    // timed in interleaved rounds -- back-to-back timing favours whichever variant runs first after an idle gap by up to
    // 10 %: DVFS) puts every (order, target) pair within +-1.5 % of each other IN TOTAL: the kernel is bound by per-wave issue,
    // not by its re-reads.  Per layer the table does split: ~1536 workgroups are 5-11 % faster on the >= 64^2 maps (few
    // slabs, long pixel loops) and 8-15 % slower on the <= 32^2 ones, hence the rule below.  The XCD-aware order is kept for
    // its HBM traffic (the sharers of a dY tile / x patch hit one L2).
    // (tune = variant >> 4 of tools/bench_conv.py overrides: bit 0 launch order, bits 1-3 workgroup target)
    const int tune = (d->variant >> 4) & 15;
    static const int targets[8] = {0, 768, 512, 1536, 2048, 3072, 4096, 1024};
    const int max_split = (a.ntiles + 3) / 4;
    auto split_for = [&](int target_wg) {
        int ns = (target_wg + slabs - 1) / slabs;
        if (slabs >= 448) ns = 1;            // >= 1.75 workgroups per CU already: skip the split-K partials + reduction (-0.4 ms/step; 256: +0.6)
        if (ns > max_split) ns = max_split;
        if (ns < 1) ns = 1;
        const int tps = (a.ntiles + ns - 1) / ns;
        return (a.ntiles + tps - 1) / tps;
    };
    // Round 4: the targets as A/B'd INSIDE the step (profiles/r04_ksplit_target_ab.txt), where the partial slabs of a split and
    // their reduction compete with the neighbouring launches for HBM: 384 workgroups on the >= 64^2 maps and pointwise layers,
    // 512 below (the isolated per-layer sweep above had 1536 / 1024 level or ahead; in the step they cost 0.4 ms).
    const int t_hi = xmc_internal_tuning(XMC_TUNE_WGRAD_TARGET_HI), t_lo = xmc_internal_tuning(XMC_TUNE_WGRAD_TARGET_LO);
    int nsplit = split_for((a.Ho >= 64 || d->ks == 1) ? t_hi : t_lo);
    a.xcd = 1;
    if (tune) { nsplit = split_for(targets[(tune >> 1) & 7] ? targets[(tune >> 1) & 7] : 1024); a.xcd = (tune & 1) ? 0 : 1; }
    a.tiles_per_split = (a.ntiles + nsplit - 1) / nsplit;
    a.nsplit = nsplit;
    a.alpha = d->alpha;
    a.L = (long long)a.Cout * d->ks * d->ks * a.Cin + a.Cout;
    if (query) { *query = nsplit > 1 ? (long long)nsplit * a.L : 0; return XMC_OK; }
    a.part = (ws && nsplit > 1) ? ws : nullptr;
    const int overwrite = (d->variant & XMC_WGRAD_OVERWRITE) ? 1 : 0;
    if (overwrite && nsplit > 1 && !a.part) return 1;         // several splits without a workspace add with atomics: not a first write
    a.overwrite = overwrite && nsplit == 1;
    dim3 grid(slabs * nsplit), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    // 4 x 4 maps: two-stage ring, two workgroups per CU (see the kernel's NST; bit 13 of variant: three stages -- A/B)
    const bool nst2 = d->ks == 3 && xi == 3 && a.PW == 6 && cb == 1 && !c96 && !((d->variant >> 13) & 1);
    const size_t lds_bytes = (nst2 ? 2 : 3) * (size_t)a.stage_bytes;
    if (xmc_internal_optin_wgrad_dma() != XMC_OK) return 1;
    bool launched = false;
#define XMC_WD_LAUNCH(KS_, XI_, PW_, CB_, C96_)                                                              \
    if (!launched && d->ks == KS_ && xi == XI_ && a.PW == PW_ && cb == CB_ && c96 == C96_) {                 \
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<KS_, XI_, PW_, CB_, C96_>), grid, block, lds_bytes, s, a); \
        launched = true;                                                                                     \
    }
    if (nst2) {
        hipLaunchKernelGGL((conv_wgrad_dma_kernel<3, 3, 6, 1, false, 2>), grid, block, lds_bytes, s, a);
        launched = true;
    }
    XMC_WD_VARIANTS(XMC_WD_LAUNCH)
#undef XMC_WD_LAUNCH
    if (launched) {}
    else return 1;
    if (a.part) return xmc_internal_wgrad_reduce(a.part, nsplit, a.L, a.L - a.Cout, dw, db, a.alpha, overwrite, stream);
    return xmc_hip_err(hipGetLastError());
}
