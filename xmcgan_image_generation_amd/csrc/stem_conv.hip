// The frozen ResNet-50's stem -- conv 7x7, stride 2, SAME, 3 -> 64 channels on 224^2 images, eval-mode BatchNorm folded in
// (xmcgan/utils/resnet_v1.py:148-156) -- as ONE implicit-GEMM kernel (round 6).
//
// Rounds 2-5 ran it as im2col + a pointwise GEMM: 587 MB of bf16 columns written and re-read per 112 images (0.33 ms of launches)
// for a layer that reads 22 MB of image and writes 180 MB.  Here the columns never leave the CU:
//   * a workgroup (4 waves) owns R = 4 output rows of one image and stages the 2R + 5 image rows they read in LDS (coalesced
//     16-byte loads; 8 leading zero elements = the left SAME padding, zero rows above / below the image, the canvas margin on the
//     right is zero already);
//   * K is laid out as k' = ky * 24 + (kx * 3 + ch): the 21 values of one filter row are CONTIGUOUS in the staged image row,
//     starting at element 6 ox + 2 (+ 8 leading zeros) of row 2 oy + ky - 2, so the B fragment of output pixel ox for an 8-wide
//     k' slice is four aligned 4-byte LDS reads (the three pad slots per filter row read the neighbouring pixel and meet zero
//     weights); K' = 168 -> 11 k-steps of 16 (176);
//   * the weights (64 x 176 bf16 = 22 fragments of 1 KiB in MFMA A-operand order, packed by the host: include/xmcgan_hip.h) stay
//     in registers for the whole workgroup;
//   * each wave walks its output row in four 32-pixel blocks x two 32-cout blocks; epilogue = the library's conv_epilogue_block
//     (bias, 16-byte stores).  Only the valid 112 x 112 corner of the 128^2 output canvas is written (COMPACT: the caller keeps the
//     margins zero).
#include "common.h"

namespace {

constexpr int ST_R = 4;                       // output rows per workgroup (one per wave)
constexpr int ST_NROWS = 2 * ST_R + 5;        // staged image rows
constexpr int ST_ROWE = 800;                  // elements per staged row: 8 zeros + 256 * 3 + 24 zeros
constexpr int ST_KSTEPS = 11;

typedef unsigned int st_u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stem_conv_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wfrag,
                                                        const float* __restrict__ bias, bf16_t* __restrict__ y, int N, int Hc, int Wc,
                                                        int Hv, int Ho, int Wo, int Hov, int Wov, int tiles) {
    __shared__ __attribute__((aligned(16))) bf16_t rows[ST_NROWS * ST_ROWE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);         // neighbouring row tiles of an image share 9 of 13 staged rows: same XCD, same L2
    const int n = wid / tiles, oy0 = (wid - n * tiles) * ST_R;
    // ---- stage the image rows 2 oy0 - 2 ... 2 oy0 + 2 R + 2
    const int vec_row = ST_ROWE / 8, data_vecs = Wc * 3 / 8;                    // uint4 per staged row; data vectors 1 .. data_vecs
    for (int v = tid; v < ST_NROWS * vec_row; v += 256) {
        const int r = v / vec_row, c = v - r * vec_row;
        const int yy = 2 * oy0 - 2 + r;
        uint4 q = make_uint4(0, 0, 0, 0);
        if (c >= 1 && c <= data_vecs && (unsigned)yy < (unsigned)Hv)
            q = *reinterpret_cast<const uint4*>(x + ((size_t)(n * Hc + yy) * Wc) * 3 + (size_t)(c - 1) * 8);
        *reinterpret_cast<uint4*>(rows + r * ST_ROWE + c * 8) = q;
    }
    // ---- this wave's weights: 2 cout blocks x 11 k-steps, one 16-byte fragment piece per lane each
    bf16x8 wf[2][ST_KSTEPS];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int ks = 0; ks < ST_KSTEPS; ++ks)
            wf[cb][ks] = *reinterpret_cast<const bf16x8*>(wfrag + ((size_t)(cb * ST_KSTEPS + ks) * 64 + lane) * 8);
    __syncthreads();
    const int oy = oy0 + wave;
    if (oy >= Hov) return;
    const int l31 = lane & 31, lhi = lane >> 5;
    // byte offset of this lane's 8-wide k' slice of k-step ks inside the staged rows, before the pixel term: slice sl = 2 ks + lhi
    // -> filter row ky = sl / 3 (the pad slice 21 re-reads row 6: zero weights), in-row offset j0 = (sl % 3) * 8
    int koff[ST_KSTEPS];
#pragma unroll
    for (int ks = 0; ks < ST_KSTEPS; ++ks) {
        const int sl = 2 * ks + lhi;
        const int ky = sl / 3 > 6 ? 6 : sl / 3, j0 = (sl - (sl / 3) * 3) * 8;
        koff[ks] = ((2 * wave + ky) * ST_ROWE + 8 + j0 - 6) * 2;                  // element 8 + (2 ox - 2) * 3 + j0 = 6 ox + 2 + j0
    }
    ConvEpi e;
    e.bias = bias; e.mask = nullptr; e.res = nullptr; e.y = y; e.Cout = 64; e.out_f32 = 0; e.alpha = 1.f; e.res_scale = 0.f;
    const unsigned char* lds = reinterpret_cast<const unsigned char*>(rows);
#pragma unroll 1
    for (int pb = 0; pb < 4; ++pb) {
        const int ox = pb * 32 + l31;
        const int pbase = ox * 12;                                               // 6 elements = 12 bytes per output pixel
        f32x16 acc[2];
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[cb][k] = 0.f;
#pragma unroll
        for (int ks = 0; ks < ST_KSTEPS; ++ks) {
            const unsigned char* p = lds + pbase + koff[ks];
            st_u32x4 q;
            q.x = *reinterpret_cast<const unsigned*>(p); q.y = *reinterpret_cast<const unsigned*>(p + 4);
            q.z = *reinterpret_cast<const unsigned*>(p + 8); q.w = *reinterpret_cast<const unsigned*>(p + 12);
            const bf16x8 xf = __builtin_bit_cast(bf16x8, q);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[0][ks], xf, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[1][ks], xf, acc[1], 0, 0, 0);
        }
        const bool live = ox < Wov;
        const size_t obase = ((size_t)(n * Ho + oy) * Wo + (live ? ox : 0)) * 64;
        ConvEpi ej = e;
        if (!live) ej.Cout = 0;
        conv_epilogue_block(acc[0], 0, lhi, obase, obase, ej);
        conv_epilogue_block(acc[1], 32, lhi, obase, obase, ej);
    }
}

}  // namespace

// y canvas (n, ho, wo, 64) valid (hov, wov) <- stem(x canvas (n, hc, wc, 3) valid rows hv; zero margin), bf16.  wfrag: the folded
// 7x7x3 -> 64 weights in fragment order [cout / 32][k-step 0..10][lane 0..63][8]: element e of lane l of fragment (cb, ks) =
// W[cb * 32 + (l & 31)][k'] with k' = ks * 16 + (l >> 5) * 8 + e = ky * 24 + kx * 3 + ch (zero for kx * 3 + ch >= 21 or ky >= 7).
// Only the valid corner of y is written.
extern "C" int xmc_stem_conv7x7s2(const void* x, const void* wfrag, const float* bias, void* y, int32_t n, int32_t hc, int32_t wc,
                                  int32_t hv, int32_t ho, int32_t wo, int32_t hov, int32_t wov, void* stream) {
    XMC_REQUIRE(x && wfrag && bias && y && n > 0 && wc * 3 + 32 <= ST_ROWE && (wc * 3) % 8 == 0 && hv > 0 && hv <= hc);
    XMC_REQUIRE(hov > 0 && hov <= ho && wov > 0 && wov <= wo && wov <= 128 && 2 * (wov - 1) + 5 < wc + 3);
    XMC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)wfrag % 16) == 0 && ((uintptr_t)y % 16) == 0);
    const int tiles = (hov + ST_R - 1) / ST_R;
    XMC_REQUIRE((long long)n * tiles < (1ll << 31) && (long long)n * ho * wo * 64 < (1ll << 40));
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)(n * tiles)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(x), static_cast<const bf16_t*>(wfrag), bias, static_cast<bf16_t*>(y), n, hc, wc, hv, ho, wo,
                       hov, wov, tiles);
    XMC_LAUNCH_RET();
}

// ---- the stem's data gradient (the generated half only) as ONE launch (round 6) ---------------------------------------------
// jax.vjp of the 7x7 stride-2 convolution onto the image: dx[y][x][c] = sum_{ky, kx, co} ds[(y + 2 - ky) / 2][(x + 2 - kx) / 2][co]
// W[co][ky][kx][c] over the taps with even y + 2 - ky, x + 2 - kx.  Rounds 2-5: a pointwise GEMM 64 -> 160 into im2col columns (294 MB
// written per 56 images) + a col2im gather (0.25 ms together).  Here, per LOW-resolution pixel (Y, X) of the 112^2 map, the 2 x 2
// image pixels (2Y + py, 2X + px) it covers are 12 outputs r = (2 py + px) * 3 + c of a 4 x 4-tap correlation over ds:
//     dx[2Y + py][2X + px][c] = sum_{t, u = 0..3} sum_co ds[Y + 1 - t][X + 1 - u][co] W[co][2t + py][2u + px][c]     (ky, kx <= 6)
// -- an implicit GEMM with M = 12 (of a 32-row MFMA block), N = pixels, K = 16 taps x 64 channels, the same LDS-patch form as the
// library's phase-decomposed kernels.  A workgroup (4 waves) owns 4 low-resolution rows of one image; the ds patch (7 rows x 116
// columns) is staged 32 channels at a time (80-byte pixel pitch: conflict-free 16-byte fragment reads), the chunk's 32 weight
// fragments (host-packed, include/xmcgan_hip.h) sit in registers.  The 17 MB of image gradient are the only bytes written.
constexpr int SD_R = 4, SD_PR = SD_R + 3, SD_PC = 132, SD_PITCH = 80;

__global__ __launch_bounds__(256, 2) void stem_dgrad_kernel(const bf16_t* __restrict__ ds, const bf16_t* __restrict__ wfrag,
                                                           bf16_t* __restrict__ dx, int N, int Ho, int Wo, int Hov, int Wov, int Hc, int Wc,
                                                           int tiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char patch[];        // SD_PR * SD_PC * SD_PITCH bytes (72 KiB: opt-in)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wid = xcd_remap(blockIdx.x, gridDim.x);         // neighbouring row tiles share 3 of their 7 patch rows: same XCD, same L2
    const int n = wid / tiles, Y0 = (wid - n * tiles) * SD_R;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[4];
#pragma unroll
    for (int pb = 0; pb < 4; ++pb)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[pb][k] = 0.f;
    const int Y = Y0 + wave;
#pragma unroll 1
    for (int ch = 0; ch < 2; ++ch) {
        // ---- stage ds[Y0 - 2 .. Y0 + 4][-2 .. 129][32 ch .. 32 ch + 31]: four 16-byte vectors per patch pixel, zeros outside the map
        for (int v = tid; v < SD_PR * SD_PC * 4; v += 256) {
            const int pp = v >> 2, kv = v & 3;
            const int pr = pp / SD_PC, pc = pp - pr * SD_PC;
            const int yy = Y0 - 2 + pr, xx = pc - 2;
            uint4 q = make_uint4(0, 0, 0, 0);
            if ((unsigned)yy < (unsigned)Hov && (unsigned)xx < (unsigned)Wov)
                q = *reinterpret_cast<const uint4*>(ds + ((size_t)(n * Ho + yy) * Wo + xx) * 64 + ch * 32 + kv * 8);
            *reinterpret_cast<uint4*>(patch + pp * SD_PITCH + kv * 16) = q;
        }
        // ---- this chunk's weights: 16 taps x 2 k-steps, one 16-byte fragment piece per lane each
        bf16x8 wf[32];
#pragma unroll
        for (int f = 0; f < 32; ++f) wf[f] = *reinterpret_cast<const bf16x8*>(wfrag + ((size_t)(ch * 32 + f) * 64 + lane) * 8);
        __syncthreads();
        if (Y < Hov) {
#pragma unroll
            for (int tap = 0; tap < 16; ++tap) {
                const int t = tap >> 2, u = tap & 3;
                const int base = ((wave + 3 - t) * SD_PC + (3 - u)) * SD_PITCH + lhi * 16;
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int pb = 0; pb < 4; ++pb) {
                        const bf16x8 xf = *reinterpret_cast<const bf16x8*>(patch + base + (pb * 32 + l31) * SD_PITCH + s * 32);
                        acc[pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[tap * 2 + s], xf, acc[pb], 0, 0, 0);
                    }
            }
        }
        __syncthreads();
    }
    if (Y >= Hov) return;
    // ---- rows r = (2 py + px) * 3 + c of the accumulator block: lane half 0 holds r = 0..3 (registers 0..3) and 8..11 (4..7),
    //      half 1 holds r = 4..7 (registers 0..3); image pixel (2Y + py, 2X + px), channel c
#pragma unroll
    for (int pb = 0; pb < 4; ++pb) {
        const int X = pb * 32 + l31;
        if (X >= Wov) continue;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            if (lhi == 1 && k >= 4) continue;
            const int r = lhi == 0 ? (k < 4 ? k : 4 + k) : 4 + k;
            const int q = r / 3, c = r - q * 3;
            dx[((size_t)(n * Hc + 2 * Y + (q >> 1)) * Wc + 2 * X + (q & 1)) * 3 + c] = f2bf(acc[pb][k]);
        }
    }
}

// dx canvas (n, hc, wc, 3): the valid 2 hov x 2 wov corner <- adjoint of xmc_stem_conv7x7s2 applied to ds canvas (n, ho, wo, 64) with
// valid (hov, wov) (bf16; the margin of ds is not read, the margin of dx is not written).  wfrag: 64 fragments of 1 KiB in MFMA
// A-operand order [channel half 0..1][tap t * 4 + u][k-step 0..1][lane][8]: element e of lane l = W[co][2t + py][2u + px][c] with row
// r = l & 31 = (2 py + px) * 3 + c (rows >= 12, and taps with 2t + py > 6 or 2u + px > 6: zero), co = half * 32 + k-step * 16 + (l >> 5) * 8 + e.
extern "C" int xmc_stem_conv7x7s2_dgrad(const void* ds, const void* wfrag, void* dx, int32_t n, int32_t ho, int32_t wo, int32_t hov,
                                        int32_t wov, int32_t hc, int32_t wc, void* stream) {
    XMC_REQUIRE(ds && wfrag && dx && n > 0 && hov > 0 && hov <= ho && wov > 0 && wov <= wo && wov <= 128 && hc >= 2 * hov && wc >= 2 * wov);
    XMC_REQUIRE(((uintptr_t)ds % 16) == 0 && ((uintptr_t)wfrag % 16) == 0);
    const int tiles = (hov + SD_R - 1) / SD_R;
    XMC_REQUIRE((long long)n * tiles < (1ll << 31));
    static XmcLdsOptIn opt_in;
    if (!opt_in.ensure({reinterpret_cast<const void*>(&stem_dgrad_kernel)}, SD_PR * SD_PC * SD_PITCH)) return XMC_EINVAL;
    hipLaunchKernelGGL(stem_dgrad_kernel, dim3((unsigned)(n * tiles)), dim3(256), SD_PR * SD_PC * SD_PITCH, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(ds), static_cast<const bf16_t*>(wfrag), static_cast<bf16_t*>(dx), n, ho, wo, hov, wov, hc, wc,
                       tiles);
    XMC_LAUNCH_RET();
}
