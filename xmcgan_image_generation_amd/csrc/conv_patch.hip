// LDS-staged im2col convolution for gfx950 (bf16, forward and data gradient).
//
// The v1 kernel (conv_igemm.hip) re-gathers the shifted input for every filter tap and spends
// ~10 VALU instructions of address arithmetic per MFMA (rocprofv3 PMC: SQ_INSTS_VALU /
// MFMA ~ 10, MFMA busy 20 %).  This kernel stages, per 32-channel chunk, the input PATCH of the
// 128-pixel output tile -- its rows plus a one-pixel halo, zero-filled at the image border,
// nearest-upsample and ReLU applied on the way in -- ONCE in LDS and feeds all ks*ks taps from it:
// the MFMA B-operand fragment of tap (dy, dx) is the same ds_read_b128 at a constant LDS offset
// (dy * patch_pitch + dx).  Only the 128 x 32 weight tile changes per tap.  All global loads are
// buffer loads whose per-lane voffset is computed once per workgroup and whose per-iteration
// offset is a scalar (soffset), so the inner loop carries no per-lane address arithmetic;
// out-of-range lanes use an out-of-bounds voffset (hardware returns 0).
//
//   D[cout][pixel] += W[cout][tap][c] * patch[pixel + tap][c]
//
// Tile 128 pixels x 128 output channels, 4 wave64 as 2x2, each wave 2x2 MFMA 32x32x16 blocks.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int PBN = 128, PBK = 32, PPITCH = 40;               // PPITCH in bf16: 80-byte rows

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

struct PArgs {
    const void* x; const void* w; const float* bias; const void* mask; const void* res; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int ups, relu_in, res_ups, out_f32;
    int relu_out, mask_after, valid_h, valid_w;
    int log2_wo, log2_howo;
    int M, nchunks, tiles_m, tiles_n;
    int Wt, Rt, imgs, PW, PP, pp_alloc; // tile geometry (output domain), patch size, LDS rows reserved for it
    unsigned x_bytes, w_bytes;
    float alpha, res_scale;
    const float* alpha_dev;
};

__device__ __forceinline__ uint4 relu4(uint4 v) {
    return make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
}

// BM = pixels per tile: 128 (4 waves) or 256 (8 waves: the weight tile is amortised over twice the pixels -> 167
// instead of 97 FLOP per byte moved L2 -> LDS).  Cout tile 128 = 4 blocks of 32; waves 2x2 MFMA blocks each.
// (96-wide cout tiles, 2x4 blocks per wave and a software-pipelined variant with a 3-deep LDS weight ring were
//  measured slower and removed: profiles/r01_conv_kernel_iterations.md.)
template <int KS, bool USE_RING, int BM>
__global__ __launch_bounds__(2 * BM, BM == 128 ? 3 : 2)
void conv_patch_kernel(const PArgs p) {     // min workgroups per CU: keeps the epilogue's temporaries from costing a wave of occupancy
    constexpr int TAPS = KS * KS, HALO = KS / 2;
    constexpr int CI = 2, PJ = 2;                                 // MFMA blocks per wave: cout x pixel
    constexpr int BN = 128;
    constexpr int T = 2 * BM;                                     // threads
    constexpr int PP_MAX = BM == 128 ? 400 : 520;                 // 3 x 130 / 4 x 130 patch pixels
    constexpr int NVEC_MAX = (PP_MAX * 4 + T - 1) / T;            // patch 16-byte vectors per thread
    constexpr int NWR = (BN * 4 + T - 1) / T;                     // weight vectors per thread (BN rows x 4 slots)
    // dynamic LDS sized to the ACTUAL patch: layers with Wo <= 32 need < 40 KB -> 4 workgroups per CU
    extern __shared__ __attribute__((aligned(16))) bf16_t lds[];
    bf16_t* const Ps = lds;                          // patch  [PP][PPITCH]
    bf16_t* const Ws = lds + p.pp_alloc * PPITCH;    // weights [2][128][PPITCH]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    // tile origin in the output domain
    const int img0 = m0 >> p.log2_howo, rem0 = m0 & ((1 << p.log2_howo) - 1);
    const int y0 = rem0 >> p.log2_wo, x0 = rem0 & (p.Wo - 1);
    const int PR1 = p.Rt + 2 * HALO;                 // patch rows per image segment

    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wr = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;            // beyond any buffer: the load returns zeros

    // ---- per-thread patch vectors: voffset into x (bytes) and LDS destination, fixed for all chunks
    unsigned pvoff[NVEC_MAX];
    const int nvec = p.PP * 4;
#pragma unroll
    for (int i = 0; i < NVEC_MAX; ++i) {
        const int v = tid + T * i;
        pvoff[i] = OOB;
        if (v < nvec) {
            const int pp = v >> 2, kv = v & 3;
            const int pr = pp / p.PW, pc = pp - pr * p.PW;
            const int im = pr / PR1, rr = pr - im * PR1;
            const int y = y0 + rr - HALO, xx = x0 + pc - HALO;
            if ((unsigned)y < (unsigned)p.Ho && (unsigned)xx < (unsigned)p.Wo) {
                const int sy = p.ups ? (y >> 1) : y, sx = p.ups ? (xx >> 1) : xx;
                pvoff[i] = (unsigned)((((img0 + im) * p.Hi + sy) * p.Wi + sx) * p.Cin + kv * 8) * 2u;
            }
        }
    }
    // ---- per-thread weight vectors (2 rows x one 16-byte slot)
    const int lrow = tid >> 2, kv = tid & 3;
    unsigned wvoff[NWR];
#pragma unroll
    for (int r = 0; r < NWR; ++r) {
        const int n = n0 + lrow + (T / 4) * r;
        wvoff[r] = (lrow + (T / 4) * r < BN && n < p.Cout) ? (unsigned)((n * TAPS) * p.Cin + kv * 8) * 2u : OOB;
    }

    constexpr int RING = (TAPS == 9 && USE_RING) ? 3 : 1;          // weight-tile register ring: loads stay in flight RING-1 taps
    u32x4 preg[NVEC_MAX], wreg[RING][NWR];
    auto load_patch = [&](int chunk) {
        const int so = chunk * PBK * 2;
#pragma unroll
        for (int i = 0; i < NVEC_MAX; ++i)
            if (tid + T * i < nvec) preg[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, pvoff[i], so, 0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int i = 0; i < NVEC_MAX; ++i) {
            const int v = tid + T * i;
            if (v < nvec) {
                uint4 q = make_uint4(preg[i].x, preg[i].y, preg[i].z, preg[i].w);
                if (p.relu_in) q = relu4(q);
                *reinterpret_cast<uint4*>(Ps + (v >> 2) * PPITCH + (v & 3) * 8) = q;
            }
        }
    };
    auto load_w = [&](int slot, int unit) {           // unit = chunk * TAPS + tap (may run past the end: OOB -> 0)
        const int chunk = unit / TAPS, tap = unit - chunk * TAPS;
        const int so = (tap * p.Cin + chunk * PBK) * 2;
        const bool live = chunk < p.nchunks;
#pragma unroll
        for (int r = 0; r < NWR; ++r)
            wreg[slot][r] = __builtin_amdgcn_raw_buffer_load_b128(wr, live ? wvoff[r] : OOB, so, 0);
    };
    auto store_w = [&](int buf, int slot) {
#pragma unroll
        for (int r = 0; r < NWR; ++r)
            if (lrow + (T / 4) * r < BN)
                *reinterpret_cast<u32x4*>(Ws + (buf * BN + lrow + (T / 4) * r) * PPITCH + kv * 8) = wreg[slot][r];
    };

    // ---- MFMA geometry: wave -> 64 (cout) x 64 (pixel); lane -> pixel within each 32-pixel block
    const int wp = wave >> 1;                                    // pixel group (PJ*32 pixels)
    const int wc = wave & 1;                                     // cout group (CI*32 channels)
    const int l31 = lane & 31, lhi = lane >> 5;
    int pbase[PJ];                                   // patch pixel index of (lane's pixel, tap 0,0)
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int t = wp * (PJ * 32) + j * 32 + l31; // tile pixel
        const int c = t & (p.Wt - 1), rowi = t / p.Wt;
        const int im = rowi / p.Rt, rj = rowi - im * p.Rt;
        pbase[j] = ((im * PR1 + rj) * p.PW + c) * PPITCH + lhi * 8;     // tap (0,0) = top-left of the halo
    }
    const bf16_t* wbase = Ws + (wc * (CI * 32) + l31) * PPITCH + lhi * 8;

    f32x16 acc[CI][PJ];
#pragma unroll
    for (int i = 0; i < CI; ++i)
#pragma unroll
        for (int j = 0; j < PJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf, int tapoff) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 wf[CI], xf[PJ];
#pragma unroll
            for (int i = 0; i < CI; ++i)
                wf[i] = *reinterpret_cast<const bf16x8*>(wbase + (buf * BN + i * 32) * PPITCH + kk * 16);
#pragma unroll
            for (int j = 0; j < PJ; ++j)
                xf[j] = *reinterpret_cast<const bf16x8*>(Ps + pbase[j] + tapoff + kk * 16);
#pragma unroll
            for (int i = 0; i < CI; ++i)
#pragma unroll
                for (int j = 0; j < PJ; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- main loop over (chunk, tap): weights double-buffered, patch single-buffered (register
    //      staged one chunk ahead: its loads fly under the ks*ks taps of the current chunk)
    load_patch(0);
#pragma unroll
    for (int u = 0; u < RING; ++u) load_w(u, u);
    store_patch();
    store_w(0, 0);
    __syncthreads();
    int it = 0;
    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const bool next_chunk = chunk + 1 < p.nchunks;
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap, ++it) {
            const bool last_tap = tap == TAPS - 1;
            const bool more = !last_tap || next_chunk;
            // slot of unit u is u % RING; TAPS % RING == 0, so slot == tap % RING is a compile-time index
            if (tap == 0 && next_chunk) load_patch(chunk + 1);
            const int tapoff = ((tap / KS) * p.PW + (tap % KS)) * PPITCH;
            if (RING == 1 && more) load_w(0, it + 1);               // no ring: classic load-next / compute / store
            compute(it & 1, tapoff);
            if (more) store_w((it + 1) & 1, (tap + 1) % RING);      // unit it+1, loaded RING-1 iterations ago
            if (RING > 1) load_w(tap % RING, it + RING);            // refill the slot unit `it` vacated
            if (last_tap && next_chunk) {
                __syncthreads();                     // every wave has finished reading the old patch
                store_patch();
            }
            __syncthreads();
        }
    }

    // ---- epilogue (common.h: the lane halves trade runs so each lane stores 16 consecutive couts of its pixel)
    ConvEpi e;
    e.bias = p.bias; e.mask = static_cast<const bf16_t*>(p.mask); e.res = static_cast<const bf16_t*>(p.res); e.y = p.y;
    e.Cout = p.Cout; e.out_f32 = p.out_f32; e.alpha = conv_alpha(p.alpha, p.alpha_dev); e.res_scale = p.res_scale;
    e.relu_out = p.relu_out; e.mask_after = p.mask_after;
#pragma unroll
    for (int j = 0; j < PJ; ++j) {
        const int pix = m0 + wp * (PJ * 32) + j * 32 + l31;
        const bool live = pix < p.M;
        const size_t obase = (size_t)(live ? pix : 0) * p.Cout;
        size_t rbase = obase;
        if (e.res && p.res_ups && live) {
            const int n = pix >> p.log2_howo, rem = pix & ((1 << p.log2_howo) - 1);
            const int y2 = (rem >> p.log2_wo) >> 1, x2 = (rem & (p.Wo - 1)) >> 1;
            rbase = ((size_t)(n * (p.Ho >> 1) + y2) * (p.Wo >> 1) + x2) * p.Cout;
        }
        ConvEpi ej = e;
        if (!live) ej.Cout = 0;                      // the lane still takes part in the swaps, but stores nothing
        if (p.valid_h) {
            const int rem = pix & ((1 << p.log2_howo) - 1);
            ej.zero = (rem >> p.log2_wo) >= p.valid_h || (rem & (p.Wo - 1)) >= p.valid_w;
        }
#pragma unroll
        for (int i = 0; i < CI; ++i) conv_epilogue_block(acc[i][j], n0 + wc * (CI * 32) + i * 32, lhi, obase, rbase, ej);
    }
}

}  // namespace

// Returns XMC_OK when the patch kernel was launched, 1 when the shape is not eligible (the caller
// then uses the generic conv_igemm kernel), or a negative error.
extern "C" int xmc_conv2d_patch_try(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                    const void* mask, const void* res, void* y, void* stream) {
    if (d->dtype != XMC_BF16 || (d->cin % PBK) != 0) return 1;
    PArgs a;
    a.x = x; a.w = w; a.bias = bias; a.mask = mask; a.res = res; a.y = y;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->ups ? 2 * d->hi : d->hi;
    a.Wo = d->ups ? 2 * d->wi : d->wi;
    a.ups = d->ups; a.relu_in = d->relu_in; a.res_ups = d->res_ups; a.out_f32 = d->out_f32;
    a.relu_out = d->relu_out; a.mask_after = d->mask_after_res; a.valid_h = d->valid_h; a.valid_w = d->valid_w;
    a.log2_wo = ilog2_exact(a.Wo);
    const int l2h = ilog2_exact(a.Ho);
    if (a.log2_wo < 0 || l2h < 0) return 1;
    a.log2_howo = a.log2_wo + l2h;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    if (m % 128 != 0 || m >= (1ll << 31)) return 1;
    a.M = (int)m;
    const long long xb = (long long)a.N * a.Hi * a.Wi * a.Cin * 2, wb = (long long)a.Cout * d->ks * d->ks * a.Cin * 2;
    if (xb >= 0xfffffff0ll || wb >= 0xfffffff0ll) return 1;
    if (((uintptr_t)x % 16) || ((uintptr_t)w % 16) || ((uintptr_t)y % 16)) return 1;
    a.x_bytes = (unsigned)xb; a.w_bytes = (unsigned)wb;
    a.nchunks = a.Cin / PBK;
    const int bn = PBN;
    a.tiles_n = (a.Cout + bn - 1) / bn;
    a.alpha = d->alpha; a.res_scale = d->res_scale; a.alpha_dev = d->alpha_dev;
    const int halo = d->ks / 2;
    auto geometry = [&](int bm) {
        a.Wt = a.Wo < bm ? a.Wo : bm;
        const int rows = bm / a.Wt;
        a.Rt = rows < a.Ho ? rows : a.Ho;
        a.imgs = bm / (a.Wt * a.Rt);
        a.PW = a.Wt + 2 * halo;
        a.PP = a.imgs * (a.Rt + 2 * halo) * a.PW;
        a.tiles_m = a.M / bm;
        return a.PP;
    };
    // 256-pixel tiles halve the weight traffic per FLOP; use them when they divide the problem and still
    // leave >= 2 workgroups per CU
    bool big = (m % 256) == 0 && (m / 256) * a.tiles_n >= 512;
    if (big && geometry(256) > 520) big = false;
    if (!big && geometry(128) > 400) return 1;
    hipStream_t s = static_cast<hipStream_t>(stream);
    // few workgroups (4x4 / 8x8 layers): latency-bound, keep weight loads in flight for 2 taps (ring);
    // many workgroups: the ring's extra registers cost more than they hide (profiles/r01_conv_kernel_iterations.md)
    const bool ring = a.tiles_m * a.tiles_n <= 512;
    a.pp_alloc = (a.PP + 7) & ~7;                    // keeps the weight tiles 256-byte aligned (8 rows x 80 B = 640 B)
    const size_t lds_bytes = (size_t)(a.pp_alloc * PPITCH + 2 * bn * PPITCH) * 2;
    dim3 grid(a.tiles_m * a.tiles_n);
    if (big) {
        if (d->ks == 3) hipLaunchKernelGGL((conv_patch_kernel<3, false, 256>), grid, dim3(512), lds_bytes, s, a);
        else hipLaunchKernelGGL((conv_patch_kernel<1, false, 256>), grid, dim3(512), lds_bytes, s, a);
    } else {
        if (d->ks == 3 && ring) hipLaunchKernelGGL((conv_patch_kernel<3, true, 128>), grid, dim3(256), lds_bytes, s, a);
        else if (d->ks == 3) hipLaunchKernelGGL((conv_patch_kernel<3, false, 128>), grid, dim3(256), lds_bytes, s, a);
        else hipLaunchKernelGGL((conv_patch_kernel<1, false, 128>), grid, dim3(256), lds_bytes, s, a);
    }
    return xmc_hip_err(hipGetLastError());
}
