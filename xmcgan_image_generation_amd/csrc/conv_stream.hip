// Weight-streaming im2col convolution for gfx950 (bf16, 3x3 / 1x1, forward and data gradient).
//
// conv_patch.hip publishes one 128 x 32 weight tile per filter tap through LDS: one workgroup barrier per
// 8 MFMAs per wave, and rocprofv3 PMC shows ~56 % of its wave cycles parked in s_waitcnt / s_barrier.
// This kernel removes the weights from LDS altogether.  The prepared weights are stored in MFMA-FRAGMENT
// order (xmc_pack_conv_weight / the `packed` mode of the prep kernels):
//
//     [cout/32][cin/32][tap][k16 half][lane 0..63][8 bf16]      lane = (k8 half) * 32 + (cout % 32)
//
// so the A operand of every MFMA is ONE fully coalesced 1 KiB buffer_load_b128 per wave whose per-lane
// address never changes (voffset = block base + lane * 16, soffset = a scalar that advances 1 KiB per
// k-step): the weight stream of a wave is a linear walk, prefetched three k-steps ahead into a register
// ring.  LDS holds only the input patch (the tile's pixels plus the one-pixel halo, 32 channels), double
// buffered, so there is ONE barrier per 32-channel chunk = per 144 MFMAs per wave, and the next chunk's
// patch is loaded / stored in three small register groups under the current chunk's MFMAs.
//
// Tile 256 pixels x 128 output channels, 4 wave64 as 2 (pixels) x 2 (cout); each wave owns 128 pixels x
// 64 channels = 2 x 4 MFMA 32x32x16 blocks (128 accumulator registers -> AGPRs, 2 waves per SIMD):
// 0.5 LDS fragment reads per MFMA (conv_patch: 1.0).  The pixel tile is at most 64 wide (4 rows x 64
// columns on the 64^2 .. 256^2 layers), which keeps the halo overhead at 396 / 256 patch pixels and two
// workgroups' double-buffered patches (2 x 63 KB) inside the 160 KB LDS.
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace {

constexpr int SBM = 256, SPITCH_B = 80;              // tile pixels; patch row pitch in bytes (32 bf16 + 16 B pad)
constexpr int NV_MAX = 9, NGRP = 3;                  // patch 16-byte vectors per thread (<= 576 patch pixels)

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct SArgs {
    const void* x; const void* w; const float* bias; const void* mask; const void* res; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int ups, relu_in, res_ups, out_f32, pool_out;
    int relu_out, mask_after, valid_h, valid_w;
    int nchunks, tiles_m, tiles_n;
    int log2_wt, log2_rt, log2_imgs, log2_tx, log2_ty;       // tile geometry (all powers of two)
    int PW, PR1, PP, pbuf_bytes;
    int magic_pw, magic_pr1;                                 // q = (x * magic) >> 16 == x / d for x < 1024
    unsigned x_bytes, w_bytes;
    float alpha, res_scale;
    const float* alpha_dev;          // optional device scalar multiplied into alpha (1 / sigma of a spectral layer)
    int ksplit, chunks_per_split;    // split-K over 32-channel chunks: split s writes its partial tile to ws[s][M][Cout] (float32)
    float* ws;
    const unsigned short* mask_bits; unsigned short* y_bits;      // ReLU masks as bits (ConvEpi), nullptr: off
    // pointwise kernel, DUAL launches (xmc_conv2d_pw_dual): the reduction runs over [x | x2] -- the first nch1 stages read x, the rest
    // the Cin2 channels of x2, an (N, H2, W2, Cin2) tensor sampled at (s2 * y, s2 * x) of the tile pixel (n, y, x) (compact mode only)
    const void* x2; unsigned x2_bytes; int Cin2, nch1, H2, W2, s2;
    int vh; unsigned magic_vv, magic_vh;     // pointwise kernel, compact mode: only the vh x vh valid corner of every (Ho x Wo) canvas is
                                             // processed -- tile pixels index that region; the margins of y are neither read nor written
};

__device__ __forceinline__ unsigned pk_max_i16(unsigned a, unsigned b) {
    unsigned r;
    asm("v_pk_max_i16 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ u32x4 relu4v(u32x4 v) {
    return u32x4{relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w)};
}

// WCB x WPB = 32-cout x 32-pixel accumulator blocks per wave, WN = waves along cout (4 / WN along pixels):
//   <3, 2, 4, 2>  256 pixels x 128 couts, wave = 64 couts x 128 pixels -- the general shape;
//   <3, 3, 2, 1>  256 pixels x  96 couts, wave = 96 couts x  64 pixels -- for Cout = 96 / 192 (every channel count of the
//                 network is a multiple of 96): in a 128-wide tile those layers leave one wave pair with HALF the MFMAs
//                 of the other (2 + 1 blocks), and the kernel is bound by per-wave issue, not by the matrix pipe
//                 (PMC: 22 % issuing / 42 % issue-stalled): here every wave carries 6 MFMAs per k-step.
//   <3, 3, 1, 1>  128 pixels x 96 couts, wave = 96 couts x 32 pixels (round 4): 48 accumulator registers, ~130 VGPRs and 42 KiB of
//                 LDS -> THREE workgroups per CU.  For the 96 / 192-channel 64^2 .. 128^2 layers, which are HBM-bound by their
//                 roofline (a tile moves 125 KB for 4.3 us of MFMAs) but ran at 1.8 TB/s: with two 4-wave workgroups per CU
//                 and the next chunk's patch requested 5 k-steps ahead there are ~17 KB in flight per CU (Little: 2 TB/s at
//                 2 us), and a tile's exposed prologue + epilogue is as long as its 3-chunk K loop.
// CS_ABL (compile-time ablation of conv_stream_kernel, tools/cs_abl.sh; 0 in the product): bit 0 no weight refills, bit 1 no
// patch staging of the next chunk, bit 2 no LDS fragment reads, bit 3 no barrier, bit 4 no epilogue (nothing stored), bit 5 the epilogue without its store instructions
#ifndef CS_ABL
#define CS_ABL 0
#endif
//   <3, 1, 8, 3, 3>  256 pixels x 96 couts on THREE waves, wave = ONE cout block x all EIGHT pixel blocks (round 4): a weight fragment
//                 feeds 8 MFMAs instead of 2.  The CS_ABL ablations and a byte count say the 96 / 192-cout 64^2 .. 128^2 launches
//                 are bound by the CU's vector-memory pipeline, not by the matrix pipe: per 256-pixel tile <3, 3, 2, 1> pulls
//                 4 waves x 162 KB of weight fragments + 76 KB of patch through it and pushes 49 KB of output (337 us of issue
//                 time on the D 128^2 96 -> 96 layer against 131 us of MFMAs); here the weight stream is 3 x 54 KB.
template <int KS, int WCB, int WPB, int WN, int NW = 4>
__global__ __launch_bounds__(NW * 64, ((WCB == 3 && WPB == 1) || (NW == 3 && WPB == 4)) ? 3 : 2) void conv_stream_kernel(const SArgs p) {
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int NT = NW * 64;                      // threads per workgroup
    constexpr int TILE_N = WN * WCB * 32;            // couts per tile
    constexpr int TILE_PX = (NW / WN) * WPB * 32;
    static_assert(TILE_PX == SBM || TILE_PX == 128, "tile is 256 (or 128) pixels");
    static_assert(NW == 4 || (NW == 3 && WN == 3 && WCB == 1 && (WPB == 8 || WPB == 4)), "3 waves: one cout block each, all pixels");
    constexpr int STEPS = TAPS * 2;                  // k16 steps per 32-channel chunk
    constexpr int D = KS == 3 ? 3 : 2;               // weight register ring depth (divides STEPS)
    constexpr int GSTEP = STEPS / NGRP;              // patch group g: loaded at step g * GSTEP, stored GSTEP - 1 later
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int wid = xcd_remap(blockIdx.x, total_tiles * p.ksplit);
    const int split = wid / total_tiles, tile = wid - split * total_tiles;
    const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int Wt = 1 << p.log2_wt, Rt = 1 << p.log2_rt;
    const int tx = tm & ((1 << p.log2_tx) - 1), rest = tm >> p.log2_tx;
    const int ty = rest & ((1 << p.log2_ty) - 1);
    const int img0 = (rest >> p.log2_ty) << p.log2_imgs;
    const int y0 = ty << p.log2_rt, x0 = tx << p.log2_wt;
    // COMPACT 3x3 launches (w_packed bit 6 + valid_h, round 6): a tile that lies entirely in the canvas margin -- rows 56 .. 63 of the
    // frozen ResNet-50's 64^2 canvases: two of sixteen row tiles -- is not computed; nobody reads those pixels (the pointwise
    // consumers are compact themselves)
    if (p.vh && (y0 >= p.vh || x0 >= p.vh)) return;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;            // beyond any buffer: the load returns zeros

    // ---- per-thread patch vectors: voffset into x (bytes), fixed for all chunks
    unsigned pvoff[NV_MAX];
    const int nvec = p.PP * 4;
#pragma unroll
    for (int i = 0; i < NV_MAX; ++i) {
        const int v = tid + NT * i;
        pvoff[i] = OOB;
        if (v < nvec) {
            const int pp = v >> 2, kv = v & 3;
            const int pr = (pp * p.magic_pw) >> 16, pc = pp - pr * p.PW;        // prologue is exposed per tile: no integer divides
            const int im = (pr * p.magic_pr1) >> 16, rr = pr - im * p.PR1;
            const int y = y0 + rr - HALO, xx = x0 + pc - HALO;
            if ((unsigned)y < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo && img0 + im < p.N) {
                const int sy = p.ups ? (y >> 1) : y, sx = p.ups ? (xx >> 1) : xx;
                pvoff[i] = (unsigned)((((img0 + im) * p.Hi + sy) * p.Wi + sx) * p.Cin + kv * 8) * 2u;
            }
        }
    }
    u32x4 preg[NV_MAX / NGRP];
    auto load_group = [&](int g, int chunk) {        // unconditional: vectors past the patch have an OOB voffset
        const int so = chunk * 64;                   // 32 channels * 2 bytes
#pragma unroll
        for (int k = 0; k < NV_MAX / NGRP; ++k)
            preg[k] = __builtin_amdgcn_raw_buffer_load_b128(xr, pvoff[g * (NV_MAX / NGRP) + k], so, 0);
    };
    auto store_group = [&](int g, int bufoff) {
#pragma unroll
        for (int k = 0; k < NV_MAX / NGRP; ++k) {
            const int v = tid + NT * (g * (NV_MAX / NGRP) + k);
            if (v < nvec) {
                u32x4 q = preg[k];
                if (p.relu_in) q = relu4v(q);
                *reinterpret_cast<u32x4*>(lds + bufoff + (v >> 2) * SPITCH_B + (v & 3) * 16) = q;
            }
        }
    };

    // ---- MFMA geometry: wave -> 64 cout x 128 pixels (2 x 4 blocks)
    // cout half of this wave; flipped on every other workgroup so that, when a ragged cout tile leaves one half
    // lighter, the two workgroups sharing a CU put their heavy waves on different SIMDs
    const int wp = WN == 2 ? wave >> 1 : (WN == 3 ? 0 : wave), wc = WN == 2 ? (wave ^ (blockIdx.x >> 3)) & 1 : (WN == 3 ? wave : 0);
    const int l31 = lane & 31, lhi = lane >> 5;
    int pbase[WPB];                                  // LDS byte offset of (lane's pixel, tap (0,0), k8 half) in buffer 0
    int opix[WPB];                                   // output pixel index (or -1)
    // tile pixel of accumulator block j: row-major over the tile, except that a 2-block wave with a pooled epilogue on
    // 64-wide tile rows takes a 2 x 32 patch (its blocks = rows r, r + 1) so that the 2x2 windows stay inside the wave
    auto tile_pixel = [&](int j) {
        if (WPB == 2 && p.pool_out && p.log2_wt == 6) return (((wp >> 1) * 2 + j) << 6) + (wp & 1) * 32 + l31;
        return wp * (WPB * 32) + j * 32 + l31;
    };
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const int t = tile_pixel(j);
        const int c = t & (Wt - 1), rowi = t >> p.log2_wt;
        const int im = rowi >> p.log2_rt, rj = rowi & (Rt - 1);
        pbase[j] = ((im * p.PR1 + rj) * p.PW + c) * SPITCH_B + lhi * 16;
        opix[j] = (img0 + im < p.N) ? ((img0 + im) * p.Ho + y0 + rj) * p.Wo + x0 + c : -1;
    }
    // ---- weight stream: block cb = tn * (WN * WCB) + wc * WCB + i, linear over (chunk, tap, k16 half)
    const int ncb = (p.Cout + 31) >> 5;
    unsigned wvoff[WCB];
#pragma unroll
    for (int i = 0; i < WCB; ++i) {
        const int cb = tn * (WN * WCB) + wc * WCB + i;
        wvoff[i] = cb < ncb ? (unsigned)(cb * p.nchunks + c_begin) * (unsigned)(STEPS * 1024) + lane * 16 : OOB;
    }
    u32x4 wreg[D][WCB];
    auto load_w = [&](int slot, int unit) {
#pragma unroll
        for (int i = 0; i < WCB; ++i) wreg[slot][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], unit * 1024, 0);
    };

    f32x16 acc[WCB][WPB];
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int j = 0; j < WPB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16x8 xf[WPB == 8 ? 1 : 2][WPB];              // (8 pixel blocks: ONE set, each fragment re-read right behind the MFMA that consumed it)
    auto read_x = [&](int set, int bufoff, int s) {
        const int tap = s >> 1, kk = s & 1;
        const int off = bufoff + ((tap / KS) * p.PW + (tap % KS)) * SPITCH_B + kk * 32;
#pragma unroll
        for (int j = 0; j < WPB; ++j) xf[set][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off);
    };

    // ---- prologue: whole patch of chunk 0 -> buffer 0 (all loads in flight at once); weight units 0 .. D-1
    {
        u32x4 p0[NV_MAX];
#pragma unroll
        for (int i = 0; i < NV_MAX; ++i) p0[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, pvoff[i], c_begin * 64, 0);
#pragma unroll
        for (int u = 0; u < D; ++u) load_w(u, u);
#pragma unroll
        for (int i = 0; i < NV_MAX; ++i) {
            const int v = tid + NT * i;
            if (v < nvec) {
                u32x4 q = p0[i];
                if (p.relu_in) q = relu4v(q);
                *reinterpret_cast<u32x4*>(lds + (v >> 2) * SPITCH_B + (v & 3) * 16) = q;
            }
        }
    }
    __syncthreads();
    read_x(0, 0, 0);

    // The K loop, specialised on the number of 32-cout blocks this wave really has (2, or 1 / 0 in a ragged cout
    // tile: Cout = 96 or 192 leave a quarter of a 128-wide tile empty): a wave skips the MFMAs and weight loads of
    // blocks past Cout but still stages the patch and meets the barriers.
    auto k_loop = [&](auto nv_tag) {
        constexpr int NVB = decltype(nv_tag)::value;
        int unit = 0;
        for (int chunk = c_begin; chunk < c_end; ++chunk) {
            const bool next_chunk = chunk + 1 < c_end;
            const int cur = (CS_ABL & 2) ? 0 : ((chunk - c_begin) & 1) * p.pbuf_bytes, nxt = p.pbuf_bytes - cur;
#pragma unroll
            for (int s = 0; s < STEPS; ++s, ++unit) {
                // sched_barrier(0): keep the issue order written here -- the scheduler otherwise sinks the prefetches
                // (weights 3 steps ahead, fragments 1 step ahead) down to their uses and exposes their latency
                if (!(CS_ABL & 2) && next_chunk && (s % GSTEP) == 0) load_group(s / GSTEP, chunk + 1);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (WPB == 8) {
                    // one cout block x 8 pixel blocks: 8 MFMAs on ONE weight fragment; fragment j of step s + 1 is requested
                    // right behind MFMA j of step s (into the same registers), the weight refill behind the last MFMA
                    static_assert(NVB == 1 || NVB == 0, "one cout block per wave");
                    if constexpr (NVB == 1) {
                        const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
                        const bool rd = s + 1 < STEPS && !(CS_ABL & 4);
                        const int tap1 = (s + 1) >> 1, kk1 = (s + 1) & 1;
                        const int off1 = cur + ((tap1 / KS) * p.PW + (tap1 % KS)) * SPITCH_B + kk1 * 32;
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[0][j], acc[0][j], 0, 0, 0);
                            if (rd) xf[0][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                            if (j == 7 && !(CS_ABL & 1)) wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], (unit + D) * 1024, 0);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else if constexpr (WCB == 3) {
                    // 96-cout wave: per pixel block three MFMAs (one per cout block); the fragment read of step s + 1
                    // rides behind the first of them, each weight register is refilled behind its last reader
                    static_assert(NVB == 3 || NVB == 0, "the 96-wide tile is launched for Cout % 96 == 0 only");
                    if constexpr (NVB == 3) {
                        const bool rd = s + 1 < STEPS && !(CS_ABL & 4);
                        const int tap1 = (s + 1) >> 1, kk1 = (s + 1) & 1;
                        const int off1 = cur + ((tap1 / KS) * p.PW + (tap1 % KS)) * SPITCH_B + kk1 * 32;
#pragma unroll
                        for (int j = 0; j < WPB; ++j) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[s % D][i]), xf[s & 1][j], acc[i][j], 0, 0, 0);
                                if (i == 0 && rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                                if (j == WPB - 1 && !(CS_ABL & 1)) wreg[s % D][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], (unit + D) * 1024, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else if constexpr (NVB == 2) {
                    // One memory instruction per MFMA gap (an in-order wave hides <= 5 single-issue instructions beside a
                    // 32-cycle MFMA, MI355X guide): the four fragment reads of step s + 1 ride in the gaps after MFMAs
                    // 1..4, the two weight refills after the last MFMA that reads each register.  Issued as one clump
                    // between the MFMA groups they left the matrix pipe idle for the length of the clump every step.
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
                    const bf16x8 w1 = __builtin_bit_cast(bf16x8, wreg[s % D][1]);
                    const bool rd = s + 1 < STEPS && !(CS_ABL & 4);
                    const int tap1 = (s + 1) >> 1, kk1 = (s + 1) & 1;
                    const int off1 = cur + ((tap1 / KS) * p.PW + (tap1 % KS)) * SPITCH_B + kk1 * 32;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[s & 1][j], acc[0][j], 0, 0, 0);
                        if (rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[s & 1][j], acc[1][j], 0, 0, 0);
                        if (j == 3 && !(CS_ABL & 1)) wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], (unit + D) * 1024, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!(CS_ABL & 1)) wreg[s % D][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[1], (unit + D) * 1024, 0);
                } else if constexpr (NVB == 1) {
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
                    const bool rd = s + 1 < STEPS;
                    const int tap1 = (s + 1) >> 1, kk1 = (s + 1) & 1;
                    const int off1 = cur + ((tap1 / KS) * p.PW + (tap1 % KS)) * SPITCH_B + kk1 * 32;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[s & 1][j], acc[0][j], 0, 0, 0);
                        if (rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], (unit + D) * 1024, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(CS_ABL & 2) && next_chunk && (s % GSTEP) == GSTEP - 1) store_group(s / GSTEP, nxt);
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(CS_ABL & 8)) __syncthreads();    // next patch published; everyone is done reading the current one
            if (NVB > 0 && next_chunk && !(CS_ABL & 4)) read_x(0, nxt, 0);
        }
    };
    const int left = ncb - (tn * (WN * WCB) + wc * WCB);
    if constexpr (WCB == 3) {
        if (left >= 3) k_loop(std::integral_constant<int, 3>{});
        else k_loop(std::integral_constant<int, 0>{});
    } else if constexpr (WCB == 1) {                 // 64-cout tile / 3-wave 96-cout tile: one block per wave
        if (left >= 1) k_loop(std::integral_constant<int, 1>{});
        else k_loop(std::integral_constant<int, 0>{});
    } else {
        if (left >= 2) k_loop(std::integral_constant<int, 2>{});
        else if (left == 1) k_loop(std::integral_constant<int, 1>{});
        else k_loop(std::integral_constant<int, 0>{});
    }

    if constexpr (CS_ABL & 16) {                     // no epilogue: keep the accumulators alive, store nothing
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < WCB; ++i)
#pragma unroll
            for (int j = 0; j < WPB; ++j)
#pragma unroll
                for (int q = 0; q < 16; ++q) t += acc[i][j][q];
        if (t == 123456.789f) static_cast<bf16_t*>(p.y)[0] = 0;
        return;
    }
    // ---- epilogue (common.h: lanes trade runs so each holds 16 consecutive couts of its pixel)
    if (p.ksplit > 1) {                              // raw float32 partial tile -> ws[split]; conv_splitk_finish_kernel does the rest
        ConvEpi e;
        e.bias = nullptr; e.mask = nullptr; e.res = nullptr; e.y = p.ws + (size_t)split * ((size_t)p.N * p.Ho * p.Wo * p.Cout);
        e.Cout = p.Cout; e.out_f32 = 1; e.alpha = 1.f; e.res_scale = 0.f;
#pragma unroll
        for (int j = 0; j < WPB; ++j) {
            const bool live = opix[j] >= 0;
            ConvEpi ej = e;
            if (!live) ej.Cout = 0;
            const size_t obase = (size_t)(live ? opix[j] : 0) * p.Cout;
#pragma unroll
            for (int i = 0; i < WCB; ++i) conv_epilogue_block(acc[i][j], tn * TILE_N + wc * (WCB * 32) + i * 32, lhi, obase, obase, ej);
        }
        return;
    }
    ConvEpi e;
    e.bias = p.bias; e.mask = static_cast<const bf16_t*>(p.mask); e.res = static_cast<const bf16_t*>(p.res); e.y = p.y;
    e.Cout = p.Cout; e.out_f32 = p.out_f32; e.alpha = conv_alpha(p.alpha, p.alpha_dev); e.res_scale = p.res_scale;
    e.relu_out = p.relu_out; e.mask_after = p.mask_after;
    e.mask_bits = p.mask_bits; e.y_bits = p.y_bits;
    const int n0 = tn * TILE_N;
    if (p.pool_out) {
        // y = avg_pool2x2(conv) (+ res at the pooled resolution): the wave's pixels are whole 2x2 windows -- the
        // vertical partner of a pixel is the same lane of another accumulator block (tile rows are 64 or 32 pixels
        // wide), the horizontal partner is the neighbouring lane.  The full-resolution tensor is never written.
        e.alpha = 0.25f * e.alpha;
        const int jstep = (WPB == 4 && p.log2_wt == 6) ? 2 : 1;    // blocks (j, j + jstep) hold rows (r, r + 1)
#pragma unroll
        for (int q = 0; q < WPB / 2; ++q) {
            const int ja = jstep == 2 ? q : 2 * q;
            const int t = tile_pixel(ja);            // tile pixel of the window's top-left corner (even lanes)
            const int col = t & (Wt - 1), rowi = t >> p.log2_wt;
            const int im = rowi >> p.log2_rt, rj = rowi & (Rt - 1);
            const bool live = (l31 & 1) == 0 && img0 + im < p.N;
            const size_t obase = live ? ((size_t)((img0 + im) * (p.Ho >> 1) + ((y0 + rj) >> 1)) * (p.Wo >> 1) + ((x0 + col) >> 1)) * p.Cout : 0;
            ConvEpi ej = e;
            if (!live) ej.Cout = 0;
#pragma unroll
            for (int i = 0; i < WCB; ++i) {
                f32x16 sacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = 0.f;
                    if constexpr (WPB == 4) v = (jstep == 2 ? acc[i][q][r] + acc[i][q + 2][r] : acc[i][2 * q][r] + acc[i][2 * q + 1][r]);
                    else if constexpr (WPB == 2) v = acc[i][0][r] + acc[i][1][r];
                    sacc[r] = v + __shfl_xor(v, 1);
                }
                conv_epilogue_block(sacc, n0 + wc * (WCB * 32) + i * 32, lhi, obase, obase, ej);
            }
        }
        return;
    }
    int mword[WPB][WCB];                             // ReLU-mask words of all blocks, requested together (one exposed latency)
#pragma unroll
    for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int i = 0; i < WCB; ++i)
            mword[j][i] = conv_epilogue_mask_word(e, n0 + wc * (WCB * 32) + i * 32, lhi, (size_t)(opix[j] >= 0 ? opix[j] : 0) * p.Cout, opix[j] >= 0);
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const int pix = opix[j];
        const bool live = pix >= 0;
        const size_t obase = (size_t)(live ? pix : 0) * p.Cout;
        size_t rbase = obase;
        if (e.res && p.res_ups && live) {
            // Ho, Wo are powers of two (checked by the launcher): shifts, not the ~40-instruction integer divisions
            const int l2w = __builtin_ctz(p.Wo), l2hw = l2w + __builtin_ctz(p.Ho);
            const int n = pix >> l2hw, rem = pix & ((1 << l2hw) - 1);
            const int y2 = (rem >> l2w) >> 1, x2 = (rem & (p.Wo - 1)) >> 1;
            rbase = ((size_t)(n * (p.Ho >> 1) + y2) * (p.Wo >> 1) + x2) * p.Cout;
        }
        ConvEpi ej = e;
        if (!live) ej.Cout = 0;                      // the lane still takes part in the swaps, but stores nothing
        if (p.valid_h) {
            const int rem = pix & (p.Ho * p.Wo - 1);
            ej.zero = (rem >> __builtin_ctz(p.Wo)) >= p.valid_h || (rem & (p.Wo - 1)) >= p.valid_w;
        }
#pragma unroll
        for (int i = 0; i < WCB; ++i) {
            ej.pre_bits = mword[j][i];
            conv_epilogue_block(acc[i][j], n0 + wc * (WCB * 32) + i * 32, lhi, obase, rbase, ej);
        }
    }
}

// ---- phase-decomposed 3x3 convolutions around a 2x resampling (round 3) -------------------------------------------------
// conv3x3(nearest_upsample2(x)) and avg_pool2x2(conv3x3(x)) -- every generator block's first convolution, every
// down-sampling discriminator block's second one, and (as each other's adjoints) their data gradients -- multiply each
// input pixel by sums of filter taps: two of the three rows (columns) of the window read the SAME low-resolution row.
//   MODE 0 "out":  y[2i+a][2j+b] = sum_{tu,tv in {0,1}} E_ab[tu][tv] x[i+a-1+tu][j+b-1+tv]      (x at the LOW resolution)
//                  E_a[tu] = sum of the taps dy in {-1},{0,1} (a = 0) / {-1,0},{1} (a = 1)
//   MODE 1 "in":   P[i][j] = 1/4 sum_{a',b'} sum_{tu,tv} F_a'b'[tu][tv] x[2(i+tu)-a'][2(j+tv)-b']  (x at the HIGH resolution)
//                  F_a'[tu] = sum of the taps dy in {-1,0},{1} (a' = 0) / {-1},{0,1} (a' = 1)
// i.e. four 2x2 convolutions -- 16 instead of 36 multiply-adds per low-resolution pixel and channel pair, 2.25x fewer
// MFMAs, exact in real arithmetic (the tap sums are formed in float32 by xmc_phase_conv_weight and rounded to bf16 once).
// Same skeleton as conv_stream_kernel (weights streamed in fragment order, packed with 16 "taps" = phase * 4 + tu * 2 + tv;
// double-buffered 32-channel patch in LDS, one barrier per stage), with a stage = one (chunk, phase) pair = 4 taps = 8
// k-steps.  MODE 0: the output phase is a grid dimension (the four workgroups of a tile are neighbours on one XCD and
// share the patch in L2), outputs are stored with stride 2.  MODE 1: the input phase is part of the K loop (all four
// accumulate into the same tile), the patch of phase (a', b') gathers every other pixel.  Patch = (Wt + 1) x (Rt + 1).
// PH_ABL (compile-time ablation of the phase kernels' k loop, tools/phase_abl.sh; 0 in the product): bit 0 no weight refills,
// bit 1 no patch staging of the next stage, bit 2 no LDS fragment reads, bit 3 no barrier; conv_phase_kernel only: bit 4 the
// staging loads without their LDS stores, bit 5 the stores without the loads
#ifndef PH_ABL
#define PH_ABL 0
#endif

constexpr int NVP = 8, NGP = 4;                      // patch vectors per thread (<= 512 patch pixels), loaded / stored in 4 groups of 2

template <int MODE, int WCB, int WPB, int WN>
__global__ __launch_bounds__(256, 2) void conv_phase_kernel(const SArgs p) {
    constexpr int TILE_N = WN * WCB * 32;
    static_assert((4 / WN) * WPB * 32 == SBM, "tile is 256 pixels");
    constexpr int STEPS = 8, D = 2;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int wid = xcd_remap(blockIdx.x, total_tiles * p.ksplit * (MODE == 0 ? 4 : 1));
    const int oph = MODE == 0 ? (wid & 3) : 0;                   // output phase (a, b) = (oph >> 1, oph & 1)
    const int w2 = MODE == 0 ? (wid >> 2) : wid;
    const int split = w2 / total_tiles, tile = w2 - split * total_tiles;
    const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int Hv = MODE == 0 ? p.Hi : p.Ho, Wv = MODE == 0 ? p.Wi : p.Wo;      // the grid the 2x2 convolutions run on
    const int Wt = 1 << p.log2_wt, Rt = 1 << p.log2_rt;
    const int tx = tm & ((1 << p.log2_tx) - 1), rest = tm >> p.log2_tx;
    const int ty = rest & ((1 << p.log2_ty) - 1);
    const int img0 = (rest >> p.log2_ty) << p.log2_imgs;
    const int y0 = ty << p.log2_rt, x0 = tx << p.log2_wt;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- per-thread patch vectors.  MODE 0: the byte offset into x of (patch pixel, 8-channel group), or OOB.
    //      MODE 1: that offset for input phase (0, 0) in bits 4..31, validity per input phase in bits 0..3
    unsigned pvoff[NVP];
    const int nvec = p.PP * 4;
#pragma unroll
    for (int i = 0; i < NVP; ++i) {
        const int v = tid + 256 * i;
        pvoff[i] = MODE == 0 ? OOB : 0u;                 // MODE 0: the final offset (the phase is fixed per workgroup)
        if (v < nvec) {
            const int pp = v >> 2, kv = v & 3;
            const int pr = (pp * p.magic_pw) >> 16, pc = pp - pr * p.PW;
            const int im = (pr * p.magic_pr1) >> 16, rr = pr - im * p.PR1;
            if (img0 + im < p.N) {
                if constexpr (MODE == 0) {
                    const int y = y0 + rr + (oph >> 1) - 1, xx = x0 + pc + (oph & 1) - 1;
                    if ((unsigned)y < (unsigned)p.Hi && (unsigned)xx < (unsigned)p.Wi)
                        pvoff[i] = (unsigned)((((img0 + im) * p.Hi + y) * p.Wi + xx) * p.Cin + kv * 8) * 2u;
                    else pvoff[i] = OOB;
                } else {
                    const int yv = y0 + rr, xv = x0 + pc;        // 0 .. Hv / Wv inclusive
                    const unsigned ry0 = yv < Hv, ry1 = yv >= 1, cx0 = xv < Wv, cx1 = xv >= 1;
                    const unsigned m = (ry0 & cx0) | ((ry0 & cx1) << 1) | ((ry1 & cx0) << 2) | ((ry1 & cx1) << 3);
                    pvoff[i] = ((unsigned)((((img0 + im) * p.Hi + 2 * yv) * p.Wi + 2 * xv) * p.Cin + kv * 8) * 2u) | m;
                }
            }
        }
    }
    u32x4 preg[NVP];
    // stage st -> (32-channel chunk, input phase): MODE 0 one chunk per stage; MODE 1 the four phases of a chunk in turn
    // MODE 1 walks PHASE-major (all chunks of input phase 0, then phase 1, ...): consecutive stages then read the
    // neighbouring 64-byte pieces of the SAME pixels, so the other half of every 128-byte line is used one stage later
    // instead of four (chunk-major order: PMC fetch 2.2x the tensor bytes on the 96-channel 128^2 layer -- the lines
    // were gone from L2 by then)
    auto load_vec = [&](int i, int ph, int chunk) {
        unsigned off = pvoff[i];
        if constexpr (MODE == 1) {
            const unsigned delta = 0u - (unsigned)(((ph >> 1) * p.Wi + (ph & 1)) * p.Cin * 2);
            off = (pvoff[i] & (1u << ph)) ? (pvoff[i] & ~15u) + delta : OOB;
        }
        preg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, off, chunk * 64, 0);
    };
    const int st_base = (tid >> 2) * SPITCH_B + (tid & 3) * 16;      // vector i of this thread: + i * 64 * SPITCH_B (an immediate)
    auto store_vec = [&](int i, int bufoff) {
        if (tid + 256 * i < nvec) {
            u32x4 q = preg[i];
            if (p.relu_in) q = relu4v(q);
            *reinterpret_cast<u32x4*>(lds + (bufoff + st_base) + i * (64 * SPITCH_B)) = q;
        }
    };

    // ---- MFMA geometry (as conv_stream_kernel)
    const int wp = WN == 2 ? wave >> 1 : wave, wc = WN == 2 ? (wave ^ (blockIdx.x >> 3)) & 1 : 0;
    const int l31 = lane & 31, lhi = lane >> 5;
    int pbase[WPB], opix[WPB];
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const int t = wp * (WPB * 32) + j * 32 + l31;
        const int c = t & (Wt - 1), rowi = t >> p.log2_wt;
        const int im = rowi >> p.log2_rt, rj = rowi & (Rt - 1);
        pbase[j] = ((im * p.PR1 + rj) * p.PW + c) * SPITCH_B + lhi * 16;
        if (img0 + im >= p.N) opix[j] = -1;
        else if (MODE == 0) opix[j] = ((img0 + im) * p.Ho + 2 * (y0 + rj) + (oph >> 1)) * p.Wo + 2 * (x0 + c) + (oph & 1);
        else opix[j] = ((img0 + im) * p.Ho + y0 + rj) * p.Wo + x0 + c;
    }
    // ---- weight stream: [cout block][chunk][16 taps][k16 half] fragments of 1 KiB; this workgroup walks
    //      MODE 0: (chunk, its phase's 4 taps), MODE 1: phase-major (all chunks of phase 0, then phase 1, ...)
    const int ncb = (p.Cout + 31) >> 5;
    unsigned wvoff[WCB];
#pragma unroll
    for (int i = 0; i < WCB; ++i) {
        const int cb = tn * (WN * WCB) + wc * WCB + i;
        wvoff[i] = cb < ncb ? (unsigned)(cb * p.nchunks) * (32u * 1024u) + lane * 16 : OOB;
    }
    const int wbase0 = (c_begin * 32 + oph * 8) * 1024;                      // stage (phase ph, chunk c): (c * 32 + ph * 8) KiB
    u32x4 wreg[D][WCB];

    f32x16 acc[WCB][WPB];
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int j = 0; j < WPB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16x8 xf[2][WPB];
    auto frag_off = [&](int s) { return (((s >> 2) & 1) * p.PW + ((s >> 1) & 1)) * SPITCH_B + (s & 1) * 32; };
    auto read_x = [&](int set, int bufoff, int s) {
        const int off = bufoff + frag_off(s);
#pragma unroll
        for (int j = 0; j < WPB; ++j) xf[set][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off);
    };

    const int nst = (c_end - c_begin) * (MODE == 0 ? 1 : 4);
    // ---- prologue: stage 0's patch -> buffer 0; weight units 0 .. D-1
    {
#pragma unroll
        for (int i = 0; i < NVP; ++i) load_vec(i, 0, c_begin);
#pragma unroll
        for (int u = 0; u < D; ++u)
#pragma unroll
            for (int i = 0; i < WCB; ++i) wreg[u][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], wbase0 + u * 1024, 0);
#pragma unroll
        for (int i = 0; i < NVP; ++i) store_vec(i, 0);
    }
    __syncthreads();
    read_x(0, 0, 0);

    auto k_loop = [&](auto nv_tag) {
        constexpr int NVB = decltype(nv_tag)::value;
        int wbase = wbase0, ph = 0, chunk = c_begin;
        for (int st = 0; st < nst; ++st) {
            const bool next = st + 1 < nst;
            int nph = ph, nchunk = chunk + 1;            // the stage after this one
            if (MODE == 1 && nchunk == c_end) { nchunk = c_begin; nph = ph + 1; }
            const int wbase_n = (nchunk * 32 + (MODE == 0 ? oph : nph) * 8) * 1024;
            const int cur = (PH_ABL & 2) ? 0 : (st & 1) * p.pbuf_bytes, nxt = p.pbuf_bytes - cur;
#pragma unroll
            for (int s = 0; s < STEPS; ++s) {
                // the next stage's patch: group g is loaded at step g and stored at step 4 + g (4 steps = 32 MFMAs of cover)
                if (!(PH_ABL & (2 | 32)) && next && s < NGP) { load_vec(2 * s, nph, nchunk); load_vec(2 * s + 1, nph, nchunk); }
                __builtin_amdgcn_sched_barrier(0);
                const int wnext = s + D < STEPS ? wbase + (s + D) * 1024 : wbase_n + (s + D - STEPS) * 1024;
                const bool rd = s + 1 < STEPS && !(PH_ABL & 4);
                const int off1 = cur + frag_off(s + 1);
                if constexpr (WCB == 3) {
                    static_assert(NVB == 3 || NVB == 0, "the 96-wide tile is launched for Cout % 96 == 0 only");
                    if constexpr (NVB == 3) {
#pragma unroll
                        for (int j = 0; j < WPB; ++j) {
#pragma unroll
                            for (int i = 0; i < 3; ++i) {
                                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wreg[s % D][i]), xf[s & 1][j], acc[i][j], 0, 0, 0);
                                if (i == 0 && rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                                if (j == WPB - 1) wreg[s % D][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], wnext, 0);
                                __builtin_amdgcn_sched_barrier(0);
                            }
                        }
                    }
                } else if constexpr (NVB == 2) {
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
                    const bf16x8 w1 = __builtin_bit_cast(bf16x8, wreg[s % D][1]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[s & 1][j], acc[0][j], 0, 0, 0);
                        if (rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        __builtin_amdgcn_sched_barrier(0);
                        acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[s & 1][j], acc[1][j], 0, 0, 0);
                        if (j == 3 && !(PH_ABL & 1)) wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], wnext, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!(PH_ABL & 1)) wreg[s % D][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[1], wnext, 0);
                } else if constexpr (NVB == 1) {
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[s & 1][j], acc[0][j], 0, 0, 0);
                        if (rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], wnext, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(PH_ABL & (2 | 16)) && next && s >= STEPS - NGP) { store_vec(2 * (s - (STEPS - NGP)), nxt); store_vec(2 * (s - (STEPS - NGP)) + 1, nxt); }
                if ((PH_ABL & 16) && next && s >= STEPS - NGP) {       // keep the loads alive (and waited for) without the LDS stores
                    asm volatile("" ::"v"(preg[2 * (s - (STEPS - NGP))]), "v"(preg[2 * (s - (STEPS - NGP)) + 1]));
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            if constexpr (!(PH_ABL & 8)) __syncthreads();
            if (NVB > 0 && next && !(PH_ABL & 4)) read_x(0, (PH_ABL & 2) ? 0 : nxt, 0);
            wbase = wbase_n; ph = nph; chunk = nchunk;
        }
    };
    const int left = ncb - (tn * (WN * WCB) + wc * WCB);
    if constexpr (WCB == 3) {
        if (left >= 3) k_loop(std::integral_constant<int, 3>{});
        else k_loop(std::integral_constant<int, 0>{});
    } else {
        if (left >= 2) k_loop(std::integral_constant<int, 2>{});
        else if (left == 1) k_loop(std::integral_constant<int, 1>{});
        else k_loop(std::integral_constant<int, 0>{});
    }

    // ---- epilogue
    ConvEpi e;
    if (p.ksplit > 1) {
        e.bias = nullptr; e.mask = nullptr; e.res = nullptr; e.y = p.ws + (size_t)split * ((size_t)p.N * p.Ho * p.Wo * p.Cout);
        e.Cout = p.Cout; e.out_f32 = 1; e.alpha = 1.f; e.res_scale = 0.f;
    } else {
        e.bias = p.bias; e.mask = static_cast<const bf16_t*>(p.mask); e.res = static_cast<const bf16_t*>(p.res); e.y = p.y;
        e.Cout = p.Cout; e.out_f32 = p.out_f32; e.alpha = conv_alpha(p.alpha, p.alpha_dev); e.res_scale = p.res_scale;
        e.relu_out = p.relu_out;
        e.mask_bits = p.mask_bits; e.y_bits = p.y_bits;
    }
    const int n0 = tn * TILE_N;
    int mword[WPB][WCB];
#pragma unroll
    for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int i = 0; i < WCB; ++i)
            mword[j][i] = conv_epilogue_mask_word(e, n0 + wc * (WCB * 32) + i * 32, lhi, (size_t)(opix[j] >= 0 ? opix[j] : 0) * p.Cout, opix[j] >= 0);
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const bool live = opix[j] >= 0;
        const size_t obase = (size_t)(live ? opix[j] : 0) * p.Cout;
        ConvEpi ej = e;
        if (!live) ej.Cout = 0;
#pragma unroll
        for (int i = 0; i < WCB; ++i) {
            ej.pre_bits = mword[j][i];
            conv_epilogue_block(acc[i][j], n0 + wc * (WCB * 32) + i * 32, lhi, obase, obase, ej);
        }
    }
}

// ---- "out" form with the four output phases as the four WAVES of a workgroup (round 3) -------------------------------------
// conv_phase_kernel<0, ...> makes the output phase a grid dimension: four workgroups stage the same patch, and each writes
// every other pixel of every other row.  Here a workgroup = 64 low-resolution pixels x TILE_N couts x ALL four phases:
// wave w = phase (w >> 1, w & 1) computes the tile's 64 pixels (2 pixel blocks) x WCB cout blocks from ONE staged patch
// ((Wt + 2) x (Rt + 2): 37 % of the per-MFMA staging of the phase-per-workgroup form) and its own 4 taps of the weight
// stream; together the four waves write whole 2 x (2 Wt) blocks of the output (full lines while they are hot in L2).
// Per k-step and wave: WCB x 2 MFMAs, 2 fragment reads, WCB weight loads.
constexpr int NV4 = 3;                               // patch vectors per thread: <= 192 patch pixels

// WPB = 32-pixel blocks per wave.  <4, 2> / <3, 2>: 64 low-resolution pixels x 128 / 96 couts.  Round 4, <2, 4>: 128 pixels x 64
// couts -- every 1 KiB weight fragment a wave streams then feeds FOUR MFMAs instead of two: with two pixel blocks the four
// waves of a workgroup pull 4 x WCB KiB of weights per k-step through the CU's 64 B/clk L1 path for 8 x WCB MFMAs, i.e. at
// the matrix pipe's peak rate the weight stream alone needs ALL of that path (0.5 KiB per MFMA; conv_stream_kernel: 0.25).
// D = depth of the weight register ring (k-steps a fragment is requested ahead of its MFMAs; divides the 8 steps of a stage).
// Round 4 ablation (tools/phase_abl.sh, PH_ABL=1): with the weight refills compiled out the "out"-form layers run 15-30 %
// faster -- each wave streams its OWN four taps, two k-steps of cover (16 MFMAs) do not hide an L2 round trip under two waves
// per SIMD; D = 4 where the registers allow it (<2, 4>: 238, <3, 2>: 214 VGPRs; <4, 2> is at 240 with D = 2).
#ifndef PH4_DEPTH
#define PH4_DEPTH 4
#endif
template <int WCB, int WPB = 2, int D = 2>
__global__ __launch_bounds__(256, 2) void conv_phase4_kernel(const SArgs p) {
    constexpr int TILE_N = WCB * 32;
    constexpr int STEPS = 8;
    static_assert(STEPS % D == 0, "ring slots are indexed by the step within a stage");
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pa = wave >> 1, pb = wave & 1;                     // this wave's output phase
    const int total_tiles = p.tiles_m * p.tiles_n;
    const int wid = xcd_remap(blockIdx.x, total_tiles * p.ksplit);
    const int split = wid / total_tiles, tile = wid - split * total_tiles;
    const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int Wt = 1 << p.log2_wt, Rt = 1 << p.log2_rt;
    const int tx = tm & ((1 << p.log2_tx) - 1), rest = tm >> p.log2_tx;
    const int ty = rest & ((1 << p.log2_ty) - 1);
    const int img0 = (rest >> p.log2_ty) << p.log2_imgs;
    const int y0 = ty << p.log2_rt, x0 = tx << p.log2_wt;

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;

    // ---- patch vectors (halo 1 on every side: the union of the four phases' 2x2 windows)
    unsigned pvoff[NV4];
    const int nvec = p.PP * 4;
#pragma unroll
    for (int i = 0; i < NV4; ++i) {
        const int v = tid + 256 * i;
        pvoff[i] = OOB;
        if (v < nvec) {
            const int pp = v >> 2, kv = v & 3;
            const int pr = (pp * p.magic_pw) >> 16, pc = pp - pr * p.PW;
            const int im = (pr * p.magic_pr1) >> 16, rr = pr - im * p.PR1;
            const int y = y0 + rr - 1, xx = x0 + pc - 1;
            if (img0 + im < p.N && (unsigned)y < (unsigned)p.Hi && (unsigned)xx < (unsigned)p.Wi)
                pvoff[i] = (unsigned)((((img0 + im) * p.Hi + y) * p.Wi + xx) * p.Cin + kv * 8) * 2u;
        }
    }
    u32x4 preg[NV4];
    const int st_base = (tid >> 2) * SPITCH_B + (tid & 3) * 16;
    auto load_vec = [&](int i, int chunk) { preg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, pvoff[i], chunk * 64, 0); };
    auto store_vec = [&](int i, int bufoff) {
        if (tid + 256 * i < nvec) {
            u32x4 q = preg[i];
            if (p.relu_in) q = relu4v(q);
            *reinterpret_cast<u32x4*>(lds + (bufoff + st_base) + i * (64 * SPITCH_B)) = q;
        }
    };

    // ---- MFMA geometry: every wave covers the tile's 64 pixels (2 blocks of 32) for its own phase
    const int l31 = lane & 31, lhi = lane >> 5;
    int pbase[WPB], opix[WPB];
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const int t = j * 32 + l31;
        const int c = t & (Wt - 1), rowi = t >> p.log2_wt;
        const int im = rowi >> p.log2_rt, rj = rowi & (Rt - 1);
        // window origin of phase (pa, pb) at patch position (rj + pa, c + pb)
        pbase[j] = ((im * p.PR1 + rj + pa) * p.PW + c + pb) * SPITCH_B + lhi * 16;
        opix[j] = img0 + im < p.N ? ((img0 + im) * p.Ho + 2 * (y0 + rj) + pa) * p.Wo + 2 * (x0 + c) + pb : -1;
    }
    const int ncb = (p.Cout + 31) >> 5;
    unsigned wvoff[WCB];
#pragma unroll
    for (int i = 0; i < WCB; ++i) {
        const int cb = tn * WCB + i;
        wvoff[i] = cb < ncb ? (unsigned)(cb * p.nchunks) * (32u * 1024u) + lane * 16 : OOB;
    }
    int wbase = (c_begin * 32 + wave * 8) * 1024;                // stage = chunk: + 32 KiB
    u32x4 wreg[D][WCB];

    f32x16 acc[WCB][WPB];
#pragma unroll
    for (int i = 0; i < WCB; ++i)
#pragma unroll
        for (int j = 0; j < WPB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    bf16x8 xf[2][WPB];
    auto frag_off = [&](int s) { return (((s >> 2) & 1) * p.PW + ((s >> 1) & 1)) * SPITCH_B + (s & 1) * 32; };
    auto read_x = [&](int set, int bufoff, int s) {
        const int off = bufoff + frag_off(s);
#pragma unroll
        for (int j = 0; j < WPB; ++j) xf[set][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off);
    };

    const int nst = c_end - c_begin;
    {
#pragma unroll
        for (int i = 0; i < NV4; ++i) load_vec(i, c_begin);
#pragma unroll
        for (int u = 0; u < D; ++u)
#pragma unroll
            for (int i = 0; i < WCB; ++i) wreg[u][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], wbase + u * 1024, 0);
#pragma unroll
        for (int i = 0; i < NV4; ++i) store_vec(i, 0);
    }
    __syncthreads();
    read_x(0, 0, 0);

    const int left = ncb - tn * WCB;                             // cout blocks this tile really has (ragged last tile)
    auto k_loop = [&](auto full_tag) {
    constexpr bool FULL = decltype(full_tag)::value;
    for (int st = 0; st < nst; ++st, wbase += 32 * 1024) {
        const bool next = st + 1 < nst;
        const int cur = (PH_ABL & 2) ? 0 : (st & 1) * p.pbuf_bytes, nxt = p.pbuf_bytes - cur;
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            // next chunk's patch: vector i loaded at step i, stored at step 5 + i
            if (!(PH_ABL & 2) && next && s < NV4) load_vec(s, c_begin + st + 1);
            __builtin_amdgcn_sched_barrier(0);
            const int wnext = wbase + (s + D < STEPS ? (s + D) * 1024 : 32 * 1024 + (s + D - STEPS) * 1024);
            const bool rd = s + 1 < STEPS && !(PH_ABL & 4);
            const int off1 = cur + frag_off(s + 1);
            if constexpr (WPB == 4) {
                // 2 cout blocks x 4 pixel blocks: conv_stream_kernel's order -- one memory instruction per MFMA gap
                static_assert(WCB == 2, "the 128-pixel tile carries 64 couts");
                if (FULL || left >= 1) {
                    const bf16x8 w0 = __builtin_bit_cast(bf16x8, wreg[s % D][0]);
                    const bf16x8 w1 = __builtin_bit_cast(bf16x8, wreg[s % D][1]);
                    const bool two = FULL || left >= 2;
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        acc[0][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w0, xf[s & 1][j], acc[0][j], 0, 0, 0);
                        if (rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        __builtin_amdgcn_sched_barrier(0);
                        if (two) acc[1][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1, xf[s & 1][j], acc[1][j], 0, 0, 0);
                        if (j == 3 && !(PH_ABL & 1)) wreg[s % D][0] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[0], wnext, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (!(PH_ABL & 1)) wreg[s % D][1] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[1], wnext, 0);
                }
            } else {
#pragma unroll
            for (int i = 0; i < WCB; ++i) {
                if (FULL || i < left) {
                    const bf16x8 wv = __builtin_bit_cast(bf16x8, wreg[s % D][i]);
#pragma unroll
                    for (int j = 0; j < WPB; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wv, xf[s & 1][j], acc[i][j], 0, 0, 0);
                        if (i == 0 && rd) xf[(s + 1) & 1][j] = *reinterpret_cast<const bf16x8*>(lds + pbase[j] + off1);
                        if (j == WPB - 1 && !(PH_ABL & 1)) wreg[s % D][i] = __builtin_amdgcn_raw_buffer_load_b128(wr, wvoff[i], wnext, 0);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            }

            __builtin_amdgcn_sched_barrier(0);
            if (!(PH_ABL & 2) && next && s >= STEPS - NV4) store_vec(s - (STEPS - NV4), nxt);
            __builtin_amdgcn_sched_barrier(0);
        }
        if constexpr (!(PH_ABL & 8)) __syncthreads();
        if (next && !(PH_ABL & 4)) read_x(0, (PH_ABL & 2) ? 0 : nxt, 0);
    }
    };
    if (left >= WCB) k_loop(std::true_type{});
    else k_loop(std::false_type{});

    // ---- epilogue
    ConvEpi e;
    if (p.ksplit > 1) {
        e.bias = nullptr; e.mask = nullptr; e.res = nullptr; e.y = p.ws + (size_t)split * ((size_t)p.N * p.Ho * p.Wo * p.Cout);
        e.Cout = p.Cout; e.out_f32 = 1; e.alpha = 1.f; e.res_scale = 0.f;
    } else {
        e.bias = p.bias; e.mask = static_cast<const bf16_t*>(p.mask); e.res = nullptr; e.y = p.y;
        e.Cout = p.Cout; e.out_f32 = p.out_f32; e.alpha = conv_alpha(p.alpha, p.alpha_dev); e.res_scale = 0.f;
        e.relu_out = p.relu_out;
        e.mask_bits = p.mask_bits; e.y_bits = p.y_bits;
    }
    const int n0 = tn * TILE_N;
    int mword[WPB][WCB];                             // the D data gradients' ReLU-mask words: all requested before the first block
#pragma unroll
    for (int j = 0; j < WPB; ++j)
#pragma unroll
        for (int i = 0; i < WCB; ++i)
            mword[j][i] = conv_epilogue_mask_word(e, n0 + i * 32, lhi, (size_t)(opix[j] >= 0 ? opix[j] : 0) * p.Cout, opix[j] >= 0);
#pragma unroll
    for (int j = 0; j < WPB; ++j) {
        const bool live = opix[j] >= 0;
        const size_t obase = (size_t)(live ? opix[j] : 0) * p.Cout;
        ConvEpi ej = e;
        if (!live) ej.Cout = 0;
#pragma unroll
        for (int i = 0; i < WCB; ++i) {
            ej.pre_bits = mword[j][i];
            conv_epilogue_block(acc[i][j], n0 + i * 32, lhi, obase, obase, ej);
        }
    }
}

// ---- pointwise (1x1) convolution on fragment-packed weights: Y[M][Cout] = epi(X[M][Cin] W^T), M = N * Ho * Wo ----------
// A 1x1 layer has 2 MFMA k-steps per 32-channel chunk instead of the 3x3's 18, so the patch machinery above (stage,
// barrier, 18 steps) would spend its time in barriers, and most of these layers (the frozen ResNet-50's bottleneck
// projections, the generator's shortcuts) move more bytes than they multiply: 28 FLOP/B at 64 -> 256 channels.  Hence:
//   * the tile is 256 CONSECUTIVE pixels x 128 couts (no spatial structure), 4 waves as 2 (pixel halves) x 2 (cout halves),
//     each wave 128 px x 64 couts like the 3x3 kernel (same accumulator layout, same epilogue);
//   * BOTH operands travel global -> LDS by `buffer_load_dwordx4 ... lds` (no staging registers, no ds_write) into a
//     ring of NS stages of KC channels, NS - 1 stages ahead of the MFMAs, retired by counted s_waitcnt vmcnt: the
//     kernel has no compiler-visible vector memory load, so nothing drains the queue;
//       X [256 px][KC]: the 16-byte slot s of tile pixel t is stored at slot s ^ swz(t) so that the 16 lanes of a
//         ds_read_b128 pass touch 16 distinct 16-byte bank groups (the DMA writes lane-linearly: the involution is
//         applied to the per-lane SOURCE address);
//       W: fragment-packed already -- one DMA instruction per 1 KiB fragment, read back lane-linearly;
//   * ONE barrier per stage;
//   * layers with very few tiles split K over workgroups through the same workspace + finishing kernel as the 3x3 path.
//   * DUAL (the frozen ResNet-50's down-sampling blocks, round 6): the reduction is the concatenation [x | x2] of two tensors --
//     relu(bn3(conv3(h)) + proj_bn(proj_conv(x_in))) is ONE product [h | x_in(2y, 2x)] [W3 | Wp]^T: the projection's output is never
//     written and re-read as the residual, its launch and the sub-sampling copy in front of it are gone.
template <int KC, int NS, int TM = 256, bool DUAL = false>
__global__ __launch_bounds__(256, TM == 128 ? 3 : 2) void conv_pw_kernel(const SArgs p) {
    static_assert(TM == 256 || TM == 128, "pixel tile");
    constexpr int JB = TM / 64;                      // 32-pixel blocks per wave (a wave owns TM / 2 pixels x 64 couts)
    constexpr int SLOTS = KC / 8, ROWB = KC * 2, XBYTES = TM * ROWB;
    constexpr int KSTEPS = KC / 16;
    constexpr int WBYTES = 4 * KSTEPS * 1024;        // 4 row blocks x KSTEPS fragments of 1 KiB
    constexpr int STAGE = XBYTES + WBYTES;
    constexpr int XDMA = XBYTES / 4096, WDMA = WBYTES / 4096, PER = XDMA + WDMA;   // DMA instructions per wave per stage
    constexpr int PPI = 64 / SLOTS;                  // pixels per X DMA instruction: 8 / 16
    constexpr int SWSH = KC == 64 ? 1 : 2;           // swz(t) = (t >> SWSH) & (SLOTS - 1)
    constexpr int D = NS - 1;                        // prefetch distance in stages
    constexpr unsigned OOB = 0xfffffff0u;
    static_assert(D >= 1 && D <= 3 && PER * (D - 1) <= 63, "vmcnt range");
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total_tiles = p.tiles_m * p.tiles_n;
    // PERSISTENT workgroups (ksplit == 1): workgroup b walks tiles b, b + G, b + 2G, ... and the DMA ring runs straight
    // across tile boundaries, so the next tile's operands are in flight under this tile's epilogue -- a 64-channel layer
    // is only two stages long, and its residual loads + stores would otherwise run with nothing else in the air.
    // Split-K launches keep one (tile, split) per workgroup.
    const int G = gridDim.x;
    const int wid = xcd_remap(blockIdx.x, G);
    const int split = p.ksplit > 1 ? wid / total_tiles : 0;
    const int tile0 = p.ksplit > 1 ? wid - split * total_tiles : wid;
    const int tstep = p.ksplit > 1 ? total_tiles : G;                       // split-K: exactly one tile
    const int my_tiles = tile0 < total_tiles ? (total_tiles - 1 - tile0) / tstep + 1 : 0;
    const int c_begin = split * p.chunks_per_split;
    const int c_end = min(p.nchunks, c_begin + p.chunks_per_split);
    const int nch = c_end - c_begin;
    const int total = my_tiles * nch;                                      // stages this workgroup walks
    const int M = p.vh ? p.N * p.vh * p.vh : p.N * p.Ho * p.Wo;            // pixels the tiles walk
    const int hw = p.Ho * p.Wo;
    if (total <= 0) return;
    // compact mode: tile pixel m (image n, row y, column x of the valid corner) -> canvas pixel; vh is not a power of two:
    // two divisions by vh as 32-bit magic multiplies (exact for m < 2^32 / vh)
    auto canvas_pix = [&](int m) {
        if (!p.vh) return m;
        const int yg = (int)__umulhi((unsigned)m, p.magic_vh), x = m - yg * p.vh;       // row index over all images
        const int n = (int)__umulhi((unsigned)yg, p.magic_vh), y = yg - n * p.vh;
        return (n * p.Ho + y) * p.Wo + x;
    };

    const v4i32 xr = make_srd(p.x, p.x_bytes), wr = make_srd(p.w, p.w_bytes);
    const v4i32 x2r = DUAL ? make_srd(p.x2, p.x2_bytes) : xr;
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);
    const int kch32 = DUAL ? (p.Cin + p.Cin2) >> 5 : p.Cin >> 5;
    const unsigned wlane = lane * 16;

    // ---- DMA addressing of the tile the ISSUE pointer is in.  X: instruction k of this wave covers tile pixels
    // (wave * XDMA + k) * PPI .. + PPI - 1;  W: instruction k = LDS fragment wave * WDMA + k = row block * KSTEPS + k-step
    unsigned xvoff[XDMA];
    unsigned xvoff2[DUAL ? XDMA : 1];
    int wsoff[WDMA];
    auto setup_issue_tile = [&](int tile) {
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;       // the cout tiles of one pixel tile are neighbours
#pragma unroll
        for (int k = 0; k < XDMA; ++k) {
            const int t = (wave * XDMA + k) * PPI + lane / SLOTS;
            const int slot = (lane % SLOTS) ^ ((t >> SWSH) & (SLOTS - 1));
            const int pix = tm * TM + t;
            xvoff[k] = OOB;
            if constexpr (DUAL) {
                xvoff2[k] = OOB;
                if (pix < M) {                       // compact mode (launcher): pix -> (image, row, column) of the valid corner
                    const int yg = (int)__umulhi((unsigned)pix, p.magic_vh), xx = pix - yg * p.vh;
                    const int n = (int)__umulhi((unsigned)yg, p.magic_vh), yy = yg - n * p.vh;
                    if (p.s2 > 0) xvoff2[k] = (unsigned)(((n * p.H2 + yy * p.s2) * p.W2 + xx * p.s2) * p.Cin2 + slot * 8) * 2u;
                    else if (!((yy | xx) & 1))       // s2 == -2: the adjoint of the stride-2 sampling -- x2 lives at the EVEN pixels, zeros elsewhere
                        xvoff2[k] = (unsigned)(((n * p.H2 + (yy >> 1)) * p.W2 + (xx >> 1)) * p.Cin2 + slot * 8) * 2u;
                }
            }
            if (pix < M) {
                int src = canvas_pix(pix);
                if (p.ups) {
                    const int l2w = __builtin_ctz(p.Wo), l2hw = l2w + __builtin_ctz(p.Ho);      // powers of two (launcher)
                    const int n = pix >> l2hw, rem = pix & (hw - 1);
                    const int y = rem >> l2w, x = rem & (p.Wo - 1);
                    src = (n * p.Hi + (y >> 1)) * p.Wi + (x >> 1);
                }
                xvoff[k] = (unsigned)(src * p.Cin + slot * 8) * 2u;
            }
        }
#pragma unroll
        for (int k = 0; k < WDMA; ++k) {
            const int f = wave * WDMA + k, rbl = f / KSTEPS, ks = f - rbl * KSTEPS;
            wsoff[k] = ((tn * 4 + rbl) * kch32 + (ks >> 1)) * 2048 + (ks & 1) * 1024;
        }
    };
    auto issue = [&](int chunk, int slot) {
        const unsigned sb = lds0 + slot * STAGE;
#pragma unroll
        for (int k = 0; k < XDMA; ++k) {
            if constexpr (DUAL) {
                if (chunk >= p.nch1) dma16(x2r, xvoff2[k], (chunk - p.nch1) * ROWB, sb + (wave * XDMA + k) * 1024);
                else dma16(xr, xvoff[k], chunk * ROWB, sb + (wave * XDMA + k) * 1024);
            } else dma16(xr, xvoff[k], chunk * ROWB, sb + (wave * XDMA + k) * 1024);
        }
#pragma unroll
        for (int k = 0; k < WDMA; ++k) dma16(wr, wlane, wsoff[k] + chunk * (KC / 32) * 2048, sb + XBYTES + (wave * WDMA + k) * 1024);
    };

    // ---- fragments
    const int wp = wave >> 1, wc = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    int xrow[JB], xsw[JB];
#pragma unroll
    for (int j = 0; j < JB; ++j) {
        const int t = wp * (TM / 2) + j * 32 + l31;
        xrow[j] = t * ROWB;
        xsw[j] = ((t >> SWSH) & (SLOTS - 1)) * 16;
    }
    const int wfrag = XBYTES + wc * 2 * KSTEPS * 1024 + lane * 16;

    f32x16 acc[2][JB];
    auto zero_acc = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < JB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    };
    zero_acc();

    // -DPW_ABL=bits: compile-time ablations for tools/pw_abl.sh (results are wrong): 1 no MFMAs, 2 no LDS fragment reads,
    // 4 no DMA after the prologue, 8 no barrier
#ifndef PW_ABL
#define PW_ABL 0
#endif
    // ReLU on the input WITHOUT a branch between the fragment reads (a branch put an s_waitcnt lgkmcnt(0) behind every
    // ds_read: eight exposed LDS latencies per stage): max as packed int16 against 0 (negative bf16 = negative int16,
    // same bits as relu_bf2) or against INT16_MIN (identity)
    unsigned relu_floor = p.relu_in ? 0u : 0x80008000u;
    asm volatile("" : "+v"(relu_floor));          // one VGPR, set once: no scalar reload (lgkmcnt) between the fragment reads
    // the fragments of TWO k-steps are read before the first MFMA: with one workgroup per CU (the few-tile layers) a wave is
    // alone on its SIMD and nothing else hides the LDS latency between a k-step's reads and its MFMAs
    auto compute = [&](int slot) {
        const unsigned char* sb = lds + slot * STAGE;
#pragma unroll
        for (int s0 = 0; s0 < KSTEPS; s0 += 2) {
            bf16x8 xf[2][JB], wf[2][2];
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                const int s = s0 + ss;
                if constexpr (PW_ABL & 2) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) wf[ss][i] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, (unsigned)s, 1u, 2u});
#pragma unroll
                    for (int j = 0; j < JB; ++j) xf[ss][j] = __builtin_bit_cast(bf16x8, u32x4{(unsigned)lane, (unsigned)j, 3u, 4u});
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) wf[ss][i] = *reinterpret_cast<const bf16x8*>(sb + wfrag + (i * KSTEPS + s) * 1024);
#pragma unroll
                    for (int j = 0; j < JB; ++j) {
                        u32x4 q = *reinterpret_cast<const u32x4*>(sb + xrow[j] + (((s * 2 + lhi) * 16) ^ xsw[j]));
                        q = u32x4{pk_max_i16(q.x, relu_floor), pk_max_i16(q.y, relu_floor), pk_max_i16(q.z, relu_floor), pk_max_i16(q.w, relu_floor)};
                        xf[ss][j] = __builtin_bit_cast(bf16x8, q);
                    }
                }
            }
#pragma unroll
            for (int ss = 0; ss < 2; ++ss) {
                if constexpr (PW_ABL & 1) {
#pragma unroll
                    for (int i = 0; i < 2; ++i) asm volatile("" ::"v"(wf[ss][i]));
#pragma unroll
                    for (int j = 0; j < JB; ++j) asm volatile("" ::"v"(xf[ss][j]));
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < JB; ++j)
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[ss][i], xf[ss][j], acc[i][j], 0, 0, 0);
                }
            }
        }
    };

    // ---- epilogue of one tile (common.h), as the 3x3 kernel's.  Its operands (five pointers, scales, flags: ~20 SGPRs) are
    // re-read from the kernel-argument segment once per tile through a pointer the compiler cannot see through: kept live
    // across the stage loop they pushed the kernel over the 102-SGPR budget (40 spilled SGPRs = v_readlane / v_writelane
    // in a loop that has no other stray issue slots).
    auto epilogue = [&](int tile) {
        const SArgs* kp = (const SArgs*)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(kp));
        const SArgs& q = *kp;
        const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n, m0 = tm * TM;
        ConvEpi e;
        const int res_ups = q.res_ups, valid_h = q.valid_h, valid_w = q.valid_w;
        if (p.ksplit > 1) {
            e.bias = nullptr; e.mask = nullptr; e.res = nullptr; e.y = q.ws + (size_t)split * ((size_t)M * p.Cout);
            e.Cout = p.Cout; e.out_f32 = 1; e.alpha = 1.f; e.res_scale = 0.f;
        } else {
            e.bias = q.bias; e.mask = static_cast<const bf16_t*>(q.mask); e.res = static_cast<const bf16_t*>(q.res); e.y = q.y;
            e.Cout = p.Cout; e.out_f32 = q.out_f32; e.alpha = conv_alpha(q.alpha, q.alpha_dev); e.res_scale = q.res_scale;
            e.relu_out = q.relu_out; e.mask_after = q.mask_after;
            e.mask_bits = q.mask_bits; e.y_bits = q.y_bits;
        }
#pragma unroll
        for (int j = 0; j < JB; ++j) {
            const int mpix = m0 + wp * (TM / 2) + j * 32 + l31;
            const bool live = mpix < M;
            const int pix = live ? canvas_pix(mpix) : 0;
            const size_t obase = (size_t)pix * p.Cout;
            size_t rbase = obase;
            ConvEpi ej = e;
            if (!live) ej.Cout = 0;
            if (p.ksplit == 1 && live && (valid_h || (e.res && res_ups))) {
                const int l2w = __builtin_ctz(p.Wo), l2hw = l2w + __builtin_ctz(p.Ho);          // powers of two (launcher)
                const int n = pix >> l2hw, rem = pix & (hw - 1);
                const int y = rem >> l2w, x = rem & (p.Wo - 1);
                if (e.res && res_ups) rbase = ((size_t)(n * (p.Ho >> 1) + (y >> 1)) * (p.Wo >> 1) + (x >> 1)) * p.Cout;
                if (valid_h) ej.zero = y >= valid_h || x >= valid_w;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) conv_epilogue_block<false, true>(acc[i][j], tn * 128 + wc * 64 + i * 32, lhi, obase, rbase, ej);
        }
    };

    // ---- main loop over the stages g = (tile index, chunk) of this workgroup: stage g lives in ring slot g % NS and
    //      is issued D stages before it is consumed
    int itile = tile0, ichunk = c_begin, ig = 0;                           // issue pointer
    setup_issue_tile(itile);
    auto issue_next = [&](int slot) {
        issue(ichunk, slot);
        ++ig;
        if (++ichunk == c_end) {
            ichunk = c_begin;
            itile += tstep;
            if (ig < total) setup_issue_tile(itile);
        }
    };
#pragma unroll
    for (int s = 0; s < D; ++s)
        if (s < total) issue_next(s);
    int slot = 0, islot = D % NS, ctile = tile0, cchunk = 0;
    const bool full_cout = (p.Cout & 127) == 0 && !p.out_f32 && p.ksplit == 1;
    bool after_epi = false;
    for (int g = 0; g < total; ++g) {
        const int younger = min(D - 1, total - 1 - g);                     // stages issued after g: their DMAs may stay in flight
        // ... and so may the 16 output stores of a full bf16 tile's epilogue when that was the last thing this wave issued
        // (stores count in vmcnt on gfx9 and retire in order: a LOWER bound on their number is safe)
        if (after_epi) {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER + 4 * JB) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER + 4 * JB) : "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * JB) : "memory");
        } else {
            if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PER) : "memory");
            else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PER) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        after_epi = false;
        if constexpr (!(PW_ABL & 8)) __builtin_amdgcn_s_barrier();         // stage g landed in every wave; everyone left stage g - 1
        if constexpr (PW_ABL & 4) { if (ig < total) { ++ig; if (++ichunk == c_end) { ichunk = c_begin; itile += tstep; } } }
        else if (ig < total) issue_next(islot);                            // ... whose slot this is
        compute(slot);
        slot = slot + 1 == NS ? 0 : slot + 1;
        islot = islot + 1 == NS ? 0 : islot + 1;
        if (++cchunk == nch) {
            epilogue(ctile);
            after_epi = full_cout && ((ctile / p.tiles_n) * TM + TM <= M);     // exactly 2 * JB blocks x 2 16-byte stores per wave
            zero_acc();
            cchunk = 0;
            ctile += tstep;
        }
    }
}

// y = epilogue(sum_s ws[s]) for the split-K launches: alpha, bias, ReLU-backward mask, residual, output dtype.
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const SArgs p, long long nvec) {
    const long long v = (long long)blockIdx.x * 256 + threadIdx.x;
    if (v >= nvec) return;
    const int cv = p.Cout >> 2;
    const long long pix = v / cv;
    const int c = (int)(v - pix * cv) * 4;
    const size_t off = (size_t)pix * p.Cout + c, slice = (size_t)p.N * p.Ho * p.Wo * p.Cout;
    // Every load of this thread is issued before the first use (partials four at a time, optional operands from a valid dummy
    // address + select): the split loop was ksplit dependent round trips, each optional operand one more.
    float4 a = *reinterpret_cast<const float4*>(p.ws + off);
    size_t rb = off;
    if (p.res && p.res_ups) {
        const int hw = p.Ho * p.Wo;
        const int n = (int)(pix / hw), rem = (int)(pix - (long long)n * hw);
        const int y2 = (rem / p.Wo) >> 1, x2 = (rem % p.Wo) >> 1;
        rb = ((size_t)(n * (p.Ho >> 1) + y2) * (p.Wo >> 1) + x2) * p.Cout + c;
    }
    const float4 bv = *reinterpret_cast<const float4*>(p.bias ? p.bias + c : p.ws + off);
    const uint2 mraw = *reinterpret_cast<const uint2*>(p.mask ? static_cast<const bf16_t*>(p.mask) + off : reinterpret_cast<const bf16_t*>(p.ws + off));
    const uint2 rraw = *reinterpret_cast<const uint2*>(p.res ? static_cast<const bf16_t*>(p.res) + rb : reinterpret_cast<const bf16_t*>(p.ws + off));
    for (int s = 1; s < p.ksplit; s += 4) {
        float4 b4[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) b4[u] = *reinterpret_cast<const float4*>(p.ws + (size_t)min(s + u, p.ksplit - 1) * slice + off);
#pragma unroll
        for (int u = 0; u < 4; ++u) {                // same order of the sum; splits past the last add nothing
            const bool ok = s + u < p.ksplit;
            a.x += ok ? b4[u].x : 0.f; a.y += ok ? b4[u].y : 0.f; a.z += ok ? b4[u].z : 0.f; a.w += ok ? b4[u].w : 0.f;
        }
    }
    const float alpha = conv_alpha(p.alpha, p.alpha_dev);
    float r[4] = {a.x * alpha, a.y * alpha, a.z * alpha, a.w * alpha};
    if (p.bias) { r[0] += bv.x; r[1] += bv.y; r[2] += bv.z; r[3] += bv.w; }
    const float mf[4] = {bf2f((bf16_t)(mraw.x & 0xffffu)), bf2f((bf16_t)(mraw.x >> 16)), bf2f((bf16_t)(mraw.y & 0xffffu)), bf2f((bf16_t)(mraw.y >> 16))};
    const float rf[4] = {bf2f((bf16_t)(rraw.x & 0xffffu)), bf2f((bf16_t)(rraw.x >> 16)), bf2f((bf16_t)(rraw.y & 0xffffu)), bf2f((bf16_t)(rraw.y >> 16))};
    if (p.mask && !p.mask_after) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(mf[e] > 0.f)) r[e] = 0.f;
    }
    if (p.res) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] += p.res_scale * rf[e];
    }
    if (p.mask && p.mask_after) {
#pragma unroll
        for (int e = 0; e < 4; ++e) if (!(mf[e] > 0.f)) r[e] = 0.f;
    }
    if (p.relu_out) {
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = fmaxf(r[e], 0.f);
    }
    if (p.valid_h) {
        const int rem = (int)(pix & (long long)(p.Ho * p.Wo - 1));
        if (rem / p.Wo >= p.valid_h || (rem & (p.Wo - 1)) >= p.valid_w) r[0] = r[1] = r[2] = r[3] = 0.f;
    }
    if (p.out_f32) *reinterpret_cast<float4*>(static_cast<float*>(p.y) + off) = make_float4(r[0], r[1], r[2], r[3]);
    else *reinterpret_cast<uint2*>(static_cast<bf16_t*>(p.y) + off) = make_uint2(pack_bf2(r[0], r[1]), pack_bf2(r[2], r[3]));
}

// plain [cout][taps][cin] -> fragment order (see the header comment); rows >= cout are zero
__global__ void pack_weight_kernel(const bf16_t* __restrict__ w, bf16_t* __restrict__ out, int cout, int taps, int cin,
                                   long long nvec) {
    const long long o = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (o >= nvec) return;
    const int lane = (int)(o & 63);
    long long r = o >> 6;
    const int kk = (int)(r & 1); r >>= 1;
    const int tap = (int)(r % taps); r /= taps;
    const int nchunks = cin >> 5;
    const int chunk = (int)(r % nchunks);
    const int cb = (int)(r / nchunks);
    const int n = cb * 32 + (lane & 31), c0 = chunk * 32 + kk * 16 + (lane >> 5) * 8;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (n < cout) v = *reinterpret_cast<const uint4*>(w + ((size_t)n * taps + tap) * cin + c0);
    *reinterpret_cast<uint4*>(out + o * 8) = v;
}

// float32 master [cout][9][cin] (x inv_sigma) -> the two phase-summed, fragment-packed bf16 copies conv_phase_kernel
// streams (16 "taps" = phase * 4 + tu * 2 + tv).  fwd_mode 0: the layer is conv3x3(upsample2(.)) -- forward copy in "out"
// order, data-gradient copy (rows = cin) in "in" order; fwd_mode 1: the layer is avg_pool2(conv3x3(.)) -- forward "in",
// data gradient "out".  The adjoint of either form is the other one with the 2x2 window reversed.
// One workgroup = a 32 x 32 (cout x cin) tile of all 9 taps.
__global__ __launch_bounds__(256) void phase_weight_kernel(const float* __restrict__ w, const float* __restrict__ inv_sigma,
                                                           bf16_t* __restrict__ wf, bf16_t* __restrict__ wd, int cout, int cin,
                                                           int fwd_mode) {
    __shared__ float t9[9][32][33];
    const float is = inv_sigma ? *inv_sigma : 1.f;
    const int n0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
    // cout and cin are multiples of 32 (checked by the launcher): NO bounds test around the loads -- a per-element condition
    // became 36 x (branch, global_load_dword, s_waitcnt vmcnt(0), ds_write): 36 serial memory latencies per workgroup,
    // 229 us for the 1536 x 1536 layer.  Nine 16-byte loads per thread (8 threads = the 128 bytes of one (row, tap)), all in
    // flight before the first LDS write.
    {
        const int row = threadIdx.x >> 3, c4 = (threadIdx.x & 7) * 4;
        float4 v[9];
        if ((reinterpret_cast<uintptr_t>(w) & 15) == 0) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap)
                v[tap] = *reinterpret_cast<const float4*>(w + ((size_t)(n0 + row) * 9 + tap) * cin + c0 + c4);
        } else {                                         // a tensor at an odd offset of the parameter arena: dword loads
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float* q = w + ((size_t)(n0 + row) * 9 + tap) * cin + c0 + c4;
                v[tap] = make_float4(q[0], q[1], q[2], q[3]);
            }
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            t9[tap][row][c4] = v[tap].x * is; t9[tap][row][c4 + 1] = v[tap].y * is;
            t9[tap][row][c4 + 2] = v[tap].z * is; t9[tap][row][c4 + 3] = v[tap].w * is;
        }
    }
    __syncthreads();
    // tap rows (columns) summed by ("out" phase a, window position tu): lo .. hi of dy + 1.
    // fwd_mode 2 (stride-2 SAME convolution of an even-sized map, flax padding (0, 1): y[o] = sum_r w[r] x[2o + r], in the
    // "in" form x[2(o + tu) - a']): single taps -- (a', tu) = (0,0) -> w[0], (1,1) -> w[1], (0,1) -> w[2], (1,0) -> none;
    // written below in terms of the complementary phase a = 1 - a' like the "in" sets of fwd_mode 1
    const bool s2 = fwd_mode == 2;
    auto lo_of = [s2](int a, int tu) { return s2 ? (a == 1 ? (tu == 0 ? 0 : 2) : 1) : (a == 0 ? (tu == 0 ? 0 : 1) : (tu == 0 ? 0 : 2)); };
    auto hi_of = [s2](int a, int tu) { return s2 ? (a == 1 ? (tu == 0 ? 0 : 2) : (tu == 0 ? 0 : 1)) : (a == 0 ? (tu == 0 ? 0 : 2) : (tu == 0 ? 1 : 2)); };
    // every thread writes 16-byte runs (8 consecutive k of one row) of the fragment order: item = (row r, k8 group), two
    // of the 16 entries per pass
    const int item = threadIdx.x & 127, r = item & 31, k8 = item >> 5;
    for (int t = threadIdx.x >> 7; t < 16; t += 2) {
        const int ph = t >> 2, tu = (t >> 1) & 1, tv = t & 1;
        // "out" sets for fwd_mode 0; "in" sets = "out" sets of the complementary phase for fwd_mode 1
        const int a = fwd_mode == 0 ? (ph >> 1) : 1 - (ph >> 1), b = fwd_mode == 0 ? (ph & 1) : 1 - (ph & 1);
        if (wf) {                                        // forward copy: rows = cout, k = cin
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int dy = lo_of(a, tu); dy <= hi_of(a, tu); ++dy)
                for (int dx = lo_of(b, tv); dx <= hi_of(b, tv); ++dx)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t9[dy * 3 + dx][r][k8 * 8 + e];
            *reinterpret_cast<uint4*>(wf + packed_w_index(n0 + r, t, c0 + k8 * 8, 16, cin >> 5)) =
                make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        }
        if (wd) {                                        // data-gradient copy: rows = cin, k = cout; the OTHER form with its
            const int ru = 1 - tu, rv = 1 - tv;          // window reversed = this form's (phase, 1 - tu, 1 - tv) entry transposed
            float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            for (int dy = lo_of(a, ru); dy <= hi_of(a, ru); ++dy)
                for (int dx = lo_of(b, rv); dx <= hi_of(b, rv); ++dx)
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += t9[dy * 3 + dx][k8 * 8 + e][r];
            *reinterpret_cast<uint4*>(wd + packed_w_index(c0 + r, t, n0 + k8 * 8, 16, cout >> 5)) =
                make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
        }
    }
}

}  // namespace

// > 64 KiB of dynamic LDS is an opt-in per kernel per device (also called by xmc_create for its device)
extern "C" int xmc_internal_optin_conv_stream(void) {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(&conv_stream_kernel<3, 2, 4, 2>), reinterpret_cast<const void*>(&conv_stream_kernel<3, 3, 2, 1>),
                          reinterpret_cast<const void*>(&conv_stream_kernel<3, 1, 4, 2>), reinterpret_cast<const void*>(&conv_stream_kernel<3, 1, 2, 1>),
                          reinterpret_cast<const void*>(&conv_pw_kernel<64, 3>),
                          reinterpret_cast<const void*>(&conv_phase_kernel<0, 2, 4, 2>), reinterpret_cast<const void*>(&conv_phase_kernel<1, 2, 4, 2>),
                          reinterpret_cast<const void*>(&conv_phase_kernel<0, 3, 2, 1>), reinterpret_cast<const void*>(&conv_phase_kernel<1, 3, 2, 1>),
                          reinterpret_cast<const void*>(&conv_pw_kernel<32, 3>), reinterpret_cast<const void*>(&conv_pw_kernel<32, 4>),
                          reinterpret_cast<const void*>(&conv_pw_kernel<32, 3, 128>),
                          reinterpret_cast<const void*>(&conv_pw_kernel<32, 3, 128, true>), reinterpret_cast<const void*>(&conv_pw_kernel<32, 3, 256, true>)}, 160 * 1024) ? XMC_OK : XMC_EINVAL;
}

extern "C" int xmc_pack_conv_weight(const void* w, void* out, int32_t cout, int32_t taps, int32_t cin, void* stream) {
    XMC_REQUIRE(w && out && cout > 0 && taps > 0 && cin > 0 && (cin % 32) == 0);
    const long long nvec = (long long)((cout + 31) / 32) * (cin / 32) * taps * 2 * 64;
    hipLaunchKernelGGL(pack_weight_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(w), static_cast<bf16_t*>(out), cout, taps, cin, nvec);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_phase_conv_weight(const float* w, const float* inv_sigma, void* w_fwd, void* w_dgrad, int32_t cout,
                                     int32_t cin, int32_t fwd_mode, void* stream) {
    XMC_REQUIRE(w && (w_fwd || w_dgrad) && cout > 0 && cin > 0 && (cout % 32) == 0 && (cin % 32) == 0);
    XMC_REQUIRE(fwd_mode >= 0 && fwd_mode <= 2);
    hipLaunchKernelGGL(phase_weight_kernel, dim3((unsigned)(cin / 32), (unsigned)(cout / 32)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       w, inv_sigma, static_cast<bf16_t*>(w_fwd), static_cast<bf16_t*>(w_dgrad), cout, cin, fwd_mode);
    XMC_LAUNCH_RET();
}

// Geometry of the phase-decomposed launch (w_packed bit 4: `w` holds the 16-tap phase weights): the 2x2 convolutions run
// on the LOW-resolution grid -- the input grid of an `ups` launch, the pooled output grid of a `pool_out` launch.
struct PhaseGeom { int mode, hv, wv, wt, rt, imgs, pp; long long tiles_m; bool tile96; int tiles_n, ksplit; bool waves4, px128; };
static bool phase_geom(const xmc_conv_desc* d, PhaseGeom* g) {
    if (!((d->w_packed >> 4) & 1) || d->dtype != XMC_BF16 || d->ks != 3 || (d->cin % 32) != 0 || (d->cout % 4) != 0) return false;
    if ((d->ups != 0) == (d->pool_out != 0)) return false;
    if (d->res_ups || d->mask_after_res || d->valid_h) return false;
    g->mode = d->ups ? 0 : 1;
    g->hv = d->ups ? d->hi : d->hi / 2; g->wv = d->ups ? d->wi : d->wi / 2;
    if (g->hv < 2 || g->wv < 2 || ilog2_exact(g->hv) < 0 || ilog2_exact(g->wv) < 0) return false;
    if (!d->ups && ((d->hi & 1) || (d->wi & 1))) return false;
    // "out" form: the four phases as the four waves of a 64-pixel tile (conv_phase4_kernel) unless bit 5 of w_packed asks
    // for the phase-per-workgroup form (A/B runs)
    g->waves4 = g->mode == 0 && !((d->w_packed >> 5) & 1);
    // 128-pixel x 64-cout tiles of the phases-as-waves form (conv_phase4_kernel<2, 4>: half the weight traffic per MFMA) where
    // the patch of a 16 x 8 pixel tile fits the three staging vectors per thread; w_packed bit 7: off (A/B)
    g->px128 = g->waves4 && !((d->w_packed >> 7) & 1) && (d->cout % 64) == 0 && g->wv >= 16 && g->hv >= 8;
    const int tile_px = g->px128 ? 128 : g->waves4 ? 64 : SBM;
    g->wt = g->waves4 ? (g->wv < 16 ? g->wv : 16) : (g->wv < 64 ? g->wv : 64);
    g->rt = tile_px / g->wt; if (g->rt > g->hv) g->rt = g->hv;
    g->imgs = tile_px / (g->wt * g->rt);
    g->pp = g->waves4 ? g->imgs * (g->rt + 2) * (g->wt + 2) : g->imgs * (g->rt + 1) * (g->wt + 1);
    if (g->pp * 4 > (g->waves4 ? NV4 : NVP) * 256) return false;
    g->tiles_m = (long long)((d->n + g->imgs - 1) / g->imgs) * (g->wv / g->wt) * (g->hv / g->rt);
    g->tile96 = !g->px128 && (d->cout % 96) == 0 && (d->cout % 128) != 0 && d->cout <= 192;
    g->tiles_n = g->px128 ? d->cout / 64 : g->tile96 ? d->cout / 96 : (d->cout + 127) / 128;
    const long long wgs = g->tiles_m * g->tiles_n * (g->mode == 0 && !g->waves4 ? 4 : 1);
    const int nchunks = d->cin / 32;
    int ks = 1;
    const int target = xmc_internal_tuning(XMC_TUNE_KSPLIT_TARGET_PHASE);
    if (wgs < 384 && nchunks >= 16) {
        ks = (int)((target + wgs / 2) / wgs);
        if (ks > nchunks / 4) ks = nchunks / 4;
        if (ks < 2) ks = 1;
    }
    g->ksplit = ks;
    return true;
}

static int conv2d_phase(const xmc_conv_desc* d, const PhaseGeom& g, const void* x, const void* w, const float* bias,
                        const void* mask, const void* res, void* y, void* ws, const void* mask_bits, void* y_bits, void* stream) {
    SArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.mask = mask; a.res = res; a.y = y;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = g.mode == 0 ? 2 * d->hi : d->hi / 2; a.Wo = g.mode == 0 ? 2 * d->wi : d->wi / 2;
    a.relu_in = d->relu_in; a.out_f32 = d->out_f32; a.relu_out = d->relu_out;
    if (g.mode == 0 && res) return XMC_EINVAL;
    if (g.mode == 1 && (mask || mask_bits)) return XMC_EINVAL;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2;
    const int ncb = (a.Cout + 31) / 32;
    const long long wb = (long long)ncb * 32 * 16 * a.Cin * 2;
    if (m >= (1ll << 31) || xb >= 0xfffffff0ll || wb >= 0xfffffff0ll) return XMC_EINVAL;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16)) return XMC_EINVAL;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.nchunks = a.Cin / 32;
    a.alpha = g.mode == 1 ? 0.25f * d->alpha : d->alpha; a.res_scale = d->res_scale; a.alpha_dev = d->alpha_dev;
    a.log2_wt = ilog2_exact(g.wt); a.log2_rt = ilog2_exact(g.rt); a.log2_imgs = ilog2_exact(g.imgs);
    a.log2_tx = ilog2_exact(g.wv) - a.log2_wt; a.log2_ty = ilog2_exact(g.hv) - a.log2_rt;
    a.PW = g.wt + (g.waves4 ? 2 : 1); a.PR1 = g.rt + (g.waves4 ? 2 : 1); a.PP = g.pp;
    a.pbuf_bytes = ((a.PP + 7) & ~7) * SPITCH_B;
    a.magic_pw = 65536 / a.PW + 1; a.magic_pr1 = 65536 / a.PR1 + 1;
    a.tiles_m = (int)g.tiles_m; a.tiles_n = g.tiles_n;
    a.ksplit = ws ? g.ksplit : 1;
    a.chunks_per_split = (a.nchunks + a.ksplit - 1) / a.ksplit;
    a.ksplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.ws = static_cast<float*>(ws);
    if (a.ksplit > 1 && (y_bits || (mask_bits && !mask))) return XMC_EINVAL;     // the finishing kernel knows bf16 masks only
    if ((mask_bits || y_bits) && (a.Cout % 16) != 0) return XMC_EINVAL;
    a.mask_bits = a.ksplit > 1 ? nullptr : static_cast<const unsigned short*>(mask_bits);
    a.y_bits = static_cast<unsigned short*>(y_bits);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (xmc_internal_optin_conv_stream() != XMC_OK) return XMC_EINVAL;
    dim3 grid((unsigned)(a.tiles_m * a.tiles_n * a.ksplit * (g.mode == 0 && !g.waves4 ? 4 : 1)));
    const size_t lds_bytes = 2 * (size_t)a.pbuf_bytes;
    if (g.px128) hipLaunchKernelGGL((conv_phase4_kernel<2, 4, PH4_DEPTH>), grid, dim3(256), lds_bytes, s, a);
    else if (g.waves4) {
        if (g.tile96) hipLaunchKernelGGL((conv_phase4_kernel<3, 2, PH4_DEPTH>), grid, dim3(256), lds_bytes, s, a);
        else hipLaunchKernelGGL((conv_phase4_kernel<4>), grid, dim3(256), lds_bytes, s, a);
    } else if (g.mode == 0) {
        if (g.tile96) hipLaunchKernelGGL((conv_phase_kernel<0, 3, 2, 1>), grid, dim3(256), lds_bytes, s, a);
        else hipLaunchKernelGGL((conv_phase_kernel<0, 2, 4, 2>), grid, dim3(256), lds_bytes, s, a);
    } else {
        if (g.tile96) hipLaunchKernelGGL((conv_phase_kernel<1, 3, 2, 1>), grid, dim3(256), lds_bytes, s, a);
        else hipLaunchKernelGGL((conv_phase_kernel<1, 2, 4, 2>), grid, dim3(256), lds_bytes, s, a);
    }
    if (a.ksplit > 1) {
        const long long nvec = m * (a.Cout / 4);
        hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, a, nvec);
    }
    return xmc_hip_err(hipGetLastError());
}

// Launches the weight-streaming kernel on fragment-packed weights.  Returns XMC_OK, or XMC_EINVAL when the
// shape is outside its domain (packed weights have no other consumer).
// Split-K factor of the weight-streaming kernel: layers with too few 256 x 128 tiles to occupy the chip (the 4x4 and
// 8x8 layers: 84-336 workgroups walking 24-48 chunks each) split the 32-channel chunks over several workgroups.
static int stream_ksplit(const xmc_conv_desc* d) {
    if (d->dtype != XMC_BF16 || (d->cin % 32) != 0 || (d->ks != 3 && d->ks != 1) || !d->w_packed || (d->cout % 4) != 0 || d->pool_out)
        return 1;
    const int ho = d->ups ? 2 * d->hi : d->hi, wo = d->ups ? 2 * d->wi : d->wi;
    if (d->ks == 1) {                                 // pointwise kernel: 256-pixel x 128-cout tiles, KC-channel stages
        const int kc = (d->cin % 64) == 0 ? 64 : 32;
        const long long tiles = (((long long)d->n * ho * wo + 255) / 256) * ((d->cout + 127) / 128);
        const int nchunks = d->cin / kc;
        // the partial sums cost 8 bytes of workspace traffic per output element and split: worth it only for very few tiles
        const int target_pw = xmc_internal_tuning(XMC_TUNE_KSPLIT_TARGET_PW);
        // (round 4, full-step A/B at batch 56: no pointwise launch of the step gains from its split -- 0.15 ms per step without
        //  them, profiles/r04_ksplit_target_ab.txt; the split stays for launches with fewer than 48 tiles: batch-2-sized work)
        if (tiles >= 48 || nchunks < 8) return 1;
        int ks = (int)((target_pw + tiles / 2) / tiles);
        if (ks > nchunks / 4) ks = nchunks / 4;
        return ks < 2 ? 1 : ks;
    }
    const int wt = wo < 64 ? wo : 64;
    int rt = SBM / wt; if (rt > ho) rt = ho;
    const int imgs = SBM / (wt * rt);
    const long long tiles_m = (long long)((d->n + imgs - 1) / imgs) * (wo / wt) * (ho / rt);
    const long long tiles = tiles_m * ((d->cout + 127) / 128);
    const int nchunks = d->cin / 32;
    // (>= 16 chunks: at 8 chunks -- the ResNet-50's 256-channel 16^2 layers -- two splits of 4 chunks plus the float32
    //  round trip cost 63 us against 39 us unsplit)
    if (tiles >= 384 || nchunks < 16) return 1;
    // target = workgroups the split aims at.  One per CU: a full-step A/B over 1 / 128 / 192 / 256 / 320 / 384 / 640 on one box
    // (profiles/r04_ksplit_target_ab.txt) has 256 ahead of 640 (round 3's value: 2.5 per CU) by 0.5 ms per step -- each split
    // beyond the first full wave of workgroups adds a float32 partial slab to write and re-read and hides nothing.
    const int target = xmc_internal_tuning(XMC_TUNE_KSPLIT_TARGET);
    int ks = (int)((target + tiles / 2) / tiles);
    if (ks > nchunks / 4) ks = nchunks / 4;
    return ks < 2 ? 1 : ks;
}

// 1 when this descriptor (w_packed bit 4 set) is inside the phase kernels' domain, else 0: the host-side mirror of phase_geom
extern "C" int xmc_conv2d_phase_supported(const xmc_conv_desc* d) {
    PhaseGeom g;
    return d && phase_geom(d, &g) ? 1 : 0;
}

extern "C" int64_t xmc_conv2d_workspace_bytes(const xmc_conv_desc* d) {
    if (!d) return 0;
    PhaseGeom g;
    if (phase_geom(d, &g))
        return g.ksplit <= 1 ? 0 : (int64_t)g.ksplit * ((long long)d->n * (g.mode == 0 ? 4 : 1) * d->hi * d->wi / (g.mode == 0 ? 1 : 4)) * d->cout * 4;
    if ((d->w_packed >> 4) & 1) return 0;
    const int ks = stream_ksplit(d);
    if (ks <= 1) return 0;
    const long long m = (long long)d->n * (d->ups ? 4 : 1) * d->hi * d->wi;
    return (int64_t)ks * m * d->cout * 4;
}

extern "C" int xmc_conv2d_stream(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                 const void* mask, const void* res, void* y, void* ws, const void* mask_bits, void* y_bits,
                                 void* stream) {
    if (d->dtype != XMC_BF16 || (d->cin % 32) != 0 || (d->ks != 3 && d->ks != 1)) return XMC_EINVAL;
    if ((d->w_packed >> 4) & 1) {                    // 16-tap phase weights: no other kernel can read them
        PhaseGeom g;
        if (!phase_geom(d, &g)) return XMC_EINVAL;
        return conv2d_phase(d, g, x, w, bias, mask, res, y, ws, mask_bits, y_bits, stream);
    }
    SArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.mask = mask; a.res = res; a.y = y;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->ups ? 2 * d->hi : d->hi;
    a.Wo = d->ups ? 2 * d->wi : d->wi;
    a.ups = d->ups; a.relu_in = d->relu_in; a.res_ups = d->res_ups; a.out_f32 = d->out_f32; a.pool_out = d->pool_out;
    a.relu_out = d->relu_out; a.mask_after = d->mask_after_res; a.valid_h = d->valid_h; a.valid_w = d->valid_w;
    XMC_REQUIRE(!(d->pool_out && (d->relu_out || d->mask_after_res || d->valid_h)));
    if (d->pool_out && (a.Wo < 32 || mask || d->res_ups)) return XMC_EINVAL;   // pooled epilogue: 2x2 windows inside a wave
    const int l2w = ilog2_exact(a.Wo), l2h = ilog2_exact(a.Ho);
    if (l2w < 0 || l2h < 0) return XMC_EINVAL;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2;
    const int ncb = (a.Cout + 31) / 32;
    const long long wb = (long long)ncb * 32 * d->ks * d->ks * a.Cin * 2;
    if (m >= (1ll << 31) || xb >= 0xfffffff0ll || wb >= 0xfffffff0ll) return XMC_EINVAL;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16)) return XMC_EINVAL;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.nchunks = a.Cin / 32;
    a.alpha = d->alpha; a.res_scale = d->res_scale; a.alpha_dev = d->alpha_dev;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->ks == 1) {
        if (d->pool_out) return XMC_EINVAL;
        const int kc = (a.Cin % 64) == 0 ? 64 : 32;
        a.nchunks = a.Cin / kc;
        a.tiles_n = (a.Cout + 127) / 128;
        a.ksplit = ws ? stream_ksplit(d) : 1;
        long long m_walk = m;                        // pixels the tiles walk (compact mode: the valid corners only)
        // bit 6 of w_packed: compact -- walk only the valid_h x valid_h corner of every canvas; the margins of y are left
        // untouched (the caller keeps them zero).  Not with split-K (its finishing kernel walks the whole canvas and zeroes
        // the margins itself) and not with an upsampling gather.
        a.vh = 0; a.magic_vv = a.magic_vh = 0;
        if (((d->w_packed >> 6) & 1) && d->valid_h > 0 && d->valid_h == d->valid_w && d->valid_h < a.Ho && !d->ups && !d->res_ups && a.ksplit == 1) {
            a.vh = d->valid_h;
            a.magic_vv = 0;
            a.magic_vh = (unsigned)(0x100000000ull / (unsigned)a.vh) + 1u;
            const long long mv = (long long)a.N * a.vh * a.vh;
            if (mv * a.vh >= 0x100000000ll) return XMC_EINVAL;
            m_walk = mv;
            a.valid_h = a.valid_w = 0;               // every pixel the kernel touches is valid
        }
        // pixel tile: 128 (48 KiB of LDS, 144 VGPRs: three workgroups per CU) up to 200k pixels, 256 (two per CU, twice the
        // FLOP per weight byte) above.  With few 256-pixel tiles a workgroup is alone on its CU -- ONE wave per SIMD -- and every
        // LDS / barrier / DMA latency of its stage loop is exposed (tools/pw_abl.sh: the MFMAs of a 1024 -> 256 layer on 22k
        // pixels are 6.8 us of a 35 us launch); measured on every 1x1 shape of the frozen ResNet-50 (tools/bench_resnet.py
        // --detail --pw-variant 4 / 8): 128 wins by 3-30 % up to 175k pixels and on the <= 64-cout layers at 351k, 256 by
        // 2-8 % on the others.  w_packed bits 14-15: 1 forces 256, 2 forces 128.
        const int tm_force = (d->w_packed >> 14) & 3;
        // Round 4: a LONG reduction with enough 256-pixel tiles to give every CU one takes the 256-pixel tile whatever the pixel
        // count -- the generator's fused conditional-BatchNorm projection (14,336 pixels, 1024 -> 4,224) and its data gradient
        // (4,224 -> 1024) re-read their weights half as often: 197 -> 182 us and 221 -> 152 us (tools/bench_pw.py --pw-variant 0 / 4);
        // the 384 -> 768 and shorter layers lose 20 % that way and keep 128.
        const bool long_k = d->cin >= 1024 && a.ksplit == 1 && ((m_walk + 255) / 256) * (long long)a.tiles_n >= 256;
        const int TMv = tm_force == 1 ? 256 : tm_force == 2 ? 128
                        : ((m_walk <= 200000 && !long_k) || (a.Cout <= 64 && m_walk <= 500000)) ? 128 : 256;
        a.tiles_m = (int)((m_walk + TMv - 1) / TMv);
        a.chunks_per_split = (a.nchunks + a.ksplit - 1) / a.ksplit;
        a.ksplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
        a.ws = static_cast<float*>(ws);
        if (a.ksplit > 1 && (y_bits || (mask_bits && !mask))) return XMC_EINVAL;
        if ((mask_bits || y_bits) && (a.Cout % 16) != 0) return XMC_EINVAL;
        a.mask_bits = a.ksplit > 1 ? nullptr : static_cast<const unsigned short*>(mask_bits);
        a.y_bits = static_cast<unsigned short*>(y_bits);
        if (xmc_internal_optin_conv_stream() != XMC_OK) return XMC_EINVAL;
        // persistent workgroups walk the tiles (two per CU: 72 KiB of LDS each); split-K launches stay one per (tile, split)
        long long nwg = (long long)a.tiles_m * a.tiles_n * a.ksplit;
        const int per_cu = TMv == 128 ? 3 : 2;
        if (a.ksplit == 1 && nwg > per_cu * xmc_cu_count()) nwg = per_cu * xmc_cu_count();
        dim3 grid((unsigned)nwg);
        // bits 12-13 of w_packed: A/B hook of tools/bench_resnet.py (1 <32,3>, 2 <64,3>, 3 <32,4>); 0 = the shipped choice.
        // Measured: 32-channel stages x 3 (two workgroups per CU) wins on every ResNet-50 shape.  (No process-wide state.)
        const int variant = ((d->w_packed >> 12) & 3) ? ((d->w_packed >> 12) & 3) : 1;
        if (variant == 2 && kc == 64 && TMv == 256) hipLaunchKernelGGL((conv_pw_kernel<64, 3>), grid, dim3(256), 3 * (256 * 128 + 16384), s, a);
        else {
            if (kc == 64) { a.nchunks *= 2; a.chunks_per_split *= 2; }       // 32-channel stages
            if (TMv == 128) hipLaunchKernelGGL((conv_pw_kernel<32, 3, 128>), grid, dim3(256), 3 * (128 * 64 + 8192), s, a);
            else if (variant == 3) hipLaunchKernelGGL((conv_pw_kernel<32, 4>), grid, dim3(256), 4 * (256 * 64 + 8192), s, a);
            else hipLaunchKernelGGL((conv_pw_kernel<32, 3>), grid, dim3(256), 3 * (256 * 64 + 8192), s, a);
        }
        if (a.ksplit > 1) {
            const long long nvec = m * (a.Cout / 4);
            hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, a, nvec);
        }
        return xmc_hip_err(hipGetLastError());
    }
    const int halo = d->ks / 2;
    a.vh = 0;
    if (d->ks == 3 && ((d->w_packed >> 6) & 1) && d->valid_h > 0 && d->valid_h == d->valid_w && !d->ups && !d->pool_out) {
        a.vh = d->valid_h;                           // compact: margin tiles are skipped, margin pixels of the other tiles are whatever
        a.valid_h = a.valid_w = 0;                   // the convolution gives there (not zeroed) -- the caller reads the valid corner only
    }
    // 96-cout tiles (waves 4 x 1, 3 x 2 blocks each) where a 128-wide tile would leave a quarter of its MFMA slots and half
    // of one wave pair's work empty: Cout = 96, 192 (the pooled epilogue needs the 128-pixel waves of the general shape)
    const bool tile96 = d->ks == 3 && (a.Cout % 96) == 0 && (((a.Cout % 128) != 0 && a.Cout <= 192) || ((d->w_packed >> 9) & 1)) && !((d->w_packed >> 8) & 1);
    // ... on 128-pixel tiles (three workgroups per CU) for the unsplit many-tile launches; w_packed bit 11: off (A/B)
    // (round 4, measured and removed: 128-pixel x 96-cout tiles on four waves -- three workgroups per CU, 25-35 % slower -- and
    //  on three waves of 1 cout block x 4 pixel blocks -- half the weight bytes per MFMA, 8-10 % slower: DESIGN 11)
    const int TPX = SBM;
    const int wt = a.Wo < 64 ? a.Wo : 64;
    int rt = TPX / wt; if (rt > a.Ho) rt = a.Ho;
    const int imgs = TPX / (wt * rt);
    a.log2_wt = ilog2_exact(wt); a.log2_rt = ilog2_exact(rt); a.log2_imgs = ilog2_exact(imgs);
    a.log2_tx = l2w - a.log2_wt; a.log2_ty = l2h - a.log2_rt;
    a.PW = wt + 2 * halo; a.PR1 = rt + 2 * halo;
    a.PP = imgs * a.PR1 * a.PW;
    if (a.PP * 4 > NV_MAX * 256) return XMC_EINVAL;
    a.pbuf_bytes = ((a.PP + 7) & ~7) * SPITCH_B;
    a.magic_pw = 65536 / a.PW + 1; a.magic_pr1 = 65536 / a.PR1 + 1;
    a.tiles_m = ((a.N + imgs - 1) / imgs) << (a.log2_tx + a.log2_ty);
    a.tiles_n = tile96 ? a.Cout / 96 : (a.Cout + 127) / 128;
    const size_t lds_bytes = 2 * (size_t)a.pbuf_bytes;
    a.ksplit = ws ? stream_ksplit(d) : 1;
    if (a.ksplit > 1) a.vh = 0;                      // (the finishing kernel walks every pixel: no uninitialised partial slabs)
    // 64-cout tiles (waves 2 x 2 as in the general shape, ONE cout block per wave) for unsplit launches with at most one
    // 128-wide tile per CU: a lone workgroup has one wave per SIMD and every latency of its chunk loop is exposed (the frozen
    // ResNet-50's 256-channel 16^2 layers: 224 workgroups, 43 us for 15 us of MFMAs); twice the workgroups at half the
    // accumulators each put two on a CU.  w_packed bit 10: off (A/B).
    const int tile64_pct = xmc_internal_tuning(XMC_TUNE_TILE64_PCT);
    const bool tile64 = d->ks == 3 && !tile96 && a.ksplit == 1 && (a.Cout % 64) == 0 && !((d->w_packed >> 10) & 1) &&
                        (long long)a.tiles_m * a.tiles_n * 100 <= (long long)xmc_cu_count() * tile64_pct;
    if (tile64) a.tiles_n = a.Cout / 64;
    // 32-cout tiles (four waves, ALL on pixels: 2 pixel blocks x 1 cout block each) for the <= 32-channel outputs -- the generator's
    // to-RGB convolution and the discriminator's image gradient (96 -> 3 at 128^2, three launches per step): in the 128-wide tile
    // two of the four waves have no cout block at all and only stage.  w_packed bit 11: off (A/B).
    const bool tile32 = d->ks == 3 && !tile96 && !tile64 && a.ksplit == 1 && a.Cout <= 32 && !((d->w_packed >> 11) & 1);
    if (tile32) a.tiles_n = 1;
    a.chunks_per_split = (a.nchunks + a.ksplit - 1) / a.ksplit;
    a.ksplit = (a.nchunks + a.chunks_per_split - 1) / a.chunks_per_split;
    a.ws = static_cast<float*>(ws);
    if (a.ksplit > 1 && (y_bits || (mask_bits && !mask))) return XMC_EINVAL;
    if ((mask_bits || y_bits) && (a.Cout % 16) != 0) return XMC_EINVAL;
    a.mask_bits = a.ksplit > 1 ? nullptr : static_cast<const unsigned short*>(mask_bits);
    a.y_bits = static_cast<unsigned short*>(y_bits);
    dim3 grid(a.tiles_m * a.tiles_n * a.ksplit);
    if (xmc_internal_optin_conv_stream() != XMC_OK) return XMC_EINVAL;
    if (d->ks == 3 && tile96) hipLaunchKernelGGL((conv_stream_kernel<3, 3, 2, 1>), grid, dim3(256), lds_bytes, s, a);
    else if (d->ks == 3 && tile64) hipLaunchKernelGGL((conv_stream_kernel<3, 1, 4, 2>), grid, dim3(256), lds_bytes, s, a);
    else if (d->ks == 3 && tile32) hipLaunchKernelGGL((conv_stream_kernel<3, 1, 2, 1>), grid, dim3(256), lds_bytes, s, a);
    else if (d->ks == 3) hipLaunchKernelGGL((conv_stream_kernel<3, 2, 4, 2>), grid, dim3(256), lds_bytes, s, a);
    if (a.ksplit > 1) {
        const long long nvec = m * (a.Cout / 4);
        hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((nvec + 255) / 256)), dim3(256), 0, s, a, nvec);
    }
    return xmc_hip_err(hipGetLastError());
}


// y = epilogue([x | x2'] W^T): the pointwise kernel over TWO sources (DUAL instantiations of conv_pw_kernel).  d describes the launch
// as for xmc_conv2d_nhwc (ks = 1, bf16, fragment-packed w with K = d->cin + cin2, COMPACT: w_packed bit 6 and valid_h == valid_w = v,
// 0 < v < ho); x2 is (n, h2, w2, cin2) and tile pixel (n, y, x) reads x2[n, stride2 * y, stride2 * x, :].  No split-K, no mask.
extern "C" int xmc_conv2d_pw_dual(const xmc_conv_desc* d, const void* x, const void* x2, int32_t cin2, int32_t h2, int32_t w2,
                                  int32_t stride2, const void* w, const float* bias, const void* mask, const void* res, void* y,
                                  const void* mask_bits, void* y_bits, void* stream) {
    XMC_REQUIRE(d && x && x2 && w && y);
    XMC_REQUIRE(d->dtype == XMC_BF16 && d->ks == 1 && (d->w_packed & 1) && ((d->w_packed >> 6) & 1));
    XMC_REQUIRE((d->cin % 32) == 0 && cin2 > 0 && (cin2 % 32) == 0 && (d->cout % 4) == 0 && (stride2 == 1 || stride2 == 2 || stride2 == -2));
    XMC_REQUIRE(!d->ups && !d->res_ups && !d->pool_out && !d->out_f32 && !d->relu_in);
    XMC_REQUIRE(d->valid_h > 0 && d->valid_h == d->valid_w && d->valid_h < d->hi && d->hi == d->wi);
    if (stride2 > 0) XMC_REQUIRE(h2 >= stride2 * (d->valid_h - 1) + 1 && w2 >= stride2 * (d->valid_w - 1) + 1);
    else XMC_REQUIRE(h2 >= (d->valid_h + 1) / 2 && w2 >= (d->valid_w + 1) / 2);
    XMC_REQUIRE((!y_bits && !mask_bits) || (d->cout % 16) == 0);
    SArgs a{};
    a.x = x; a.w = w; a.bias = bias; a.res = res; a.y = y;
    a.mask = mask; a.mask_bits = static_cast<const unsigned short*>(mask_bits); a.mask_after = d->mask_after_res;
    a.N = d->n; a.Hi = a.Ho = d->hi; a.Wi = a.Wo = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.relu_out = d->relu_out;
    a.x2 = x2; a.Cin2 = cin2; a.H2 = h2; a.W2 = w2; a.s2 = stride2;
    if (ilog2_exact(a.Wo) < 0 || ilog2_exact(a.Ho) < 0) return XMC_EINVAL;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2, x2b = (long long)a.N * h2 * w2 * cin2 * 2;
    const int ncb = (a.Cout + 31) / 32;
    const long long wb = (long long)ncb * 32 * (a.Cin + cin2) * 2;
    if (xb >= 0xfffffff0ll || x2b >= 0xfffffff0ll || wb >= 0xfffffff0ll) return XMC_EINVAL;
    if (((uintptr_t)x % 16) || ((uintptr_t)x2 % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16)) return XMC_EINVAL;
    a.x_bytes = (unsigned)xb; a.x2_bytes = (unsigned)x2b; a.w_bytes = (unsigned)wb;
    a.nch1 = a.Cin / 32;
    a.nchunks = (a.Cin + cin2) / 32;                 // 32-channel stages
    a.alpha = d->alpha; a.res_scale = d->res_scale; a.alpha_dev = d->alpha_dev;
    a.tiles_n = (a.Cout + 127) / 128;
    a.ksplit = 1; a.chunks_per_split = a.nchunks;
    a.vh = d->valid_h;
    a.magic_vh = (unsigned)(0x100000000ull / (unsigned)a.vh) + 1u;
    const long long mv = (long long)a.N * a.vh * a.vh;
    if (mv * a.vh >= 0x100000000ll) return XMC_EINVAL;
    const int tm_force = (d->w_packed >> 14) & 3;
    const int TMv = tm_force == 1 ? 256 : tm_force == 2 ? 128 : (mv <= 200000 || (a.Cout <= 64 && mv <= 500000)) ? 128 : 256;
    a.tiles_m = (int)((mv + TMv - 1) / TMv);
    a.y_bits = static_cast<unsigned short*>(y_bits);
    if (xmc_internal_optin_conv_stream() != XMC_OK) return XMC_EINVAL;
    long long nwg = (long long)a.tiles_m * a.tiles_n;
    const int per_cu = TMv == 128 ? 3 : 2;
    if (nwg > per_cu * xmc_cu_count()) nwg = per_cu * xmc_cu_count();
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (TMv == 128) hipLaunchKernelGGL((conv_pw_kernel<32, 3, 128, true>), dim3((unsigned)nwg), dim3(256), 3 * (128 * 64 + 8192), s, a);
    else hipLaunchKernelGGL((conv_pw_kernel<32, 3, 256, true>), dim3((unsigned)nwg), dim3(256), 3 * (256 * 64 + 8192), s, a);
    return xmc_hip_err(hipGetLastError());
}
