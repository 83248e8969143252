// word_loss (reference xmcgan/libml/attention_lib.py:105-191) on the matrix cores -- bf16 training mode, R = 256 regions.
//
// Rounds 1-3 ran the restructured word_loss (S = R^ W^^T, G_j = R^_j R^_j^T, alpha = softmax_r(gamma1 S), nn = sum alpha S,
// q = alpha^T G_j alpha, cos = nn / sqrt(q)) as five GEMM launches on FLOAT32 tensors plus four elementwise / column kernels, with the
// (B, B, R, T) score, probability and H = G alpha tensors materialised in float32 (3 x 55 MB at B = 56): the GEMMs ran at
// 130-355 TF/s because a 128 x 128 tile of float32 operands is 32 FLOP per operand byte.  Here:
//
//   wl_prep_regions   x (bf16) -> R^ (bf16, [B][R][E]), R^^T ([B][E][R]) and 1 / |x| -- one pass, replaces l2norm_fwd's float32 copy
//   wl_prep_words     W^ (float32) -> bf16 [LDP][E] (rows >= B*T zero) and W^^T [E][LDP]       (LDP = B*T rounded up to 64)
//   wl_tn_gemm        Out[b][x][y] = alpha * sum_k X[b][x][k] Y[b][y][k] over one or two K segments, bf16 operands with k
//                     contiguous, 128 x 128 tiles, float32 accumulate: G_j, dG_j and d R^ = [dS | 2 dG] [W^^T | R^_j^T]^T
//   wl_cols<BWD>      workgroup = (image j, 64 word columns (i, t)): S tile (256 x 64, K = E) on MFMA with the regions as the A
//                     operand, so that a lane of the 32 x 32 C layout owns ONE word column and 16 of a block's 32 regions: the
//                     softmax over the 256 regions is an in-lane reduction + one shuffle + a 4-wave LDS exchange; alpha goes
//                     to LDS as B-operand fragments and H = G_j alpha (K = 256) runs on MFMA straight away; forward keeps only
//                     nn and q per column (2 x B x B*T floats) -- S, alpha and H never reach HBM.  The backward launch recomputes
//                     them (28 GFLOP at B = 56, bit-identical to the forward) and emits dS, alpha dq and alpha as bf16
//                     [B][R][LDP] for the two products above.
//
// Domain: bf16, R == 256, E % 64 == 0 (xmc_wl_fused_supported); anything else keeps the GEMM + column-kernel path (losses.hip).
#include "common.h"

namespace {

constexpr int KP32 = 40;        // LDS row pitch (bf16 elements) of a 32-k stage: 80 bytes = 5 x 16 -- conflict-free ds_read_b128
constexpr int KP64 = 72;        // of a 64-k stage: 144 bytes = 9 x 16
constexpr int ALP = 264;        // alpha^T row pitch: 256 regions + 8 (528 bytes = 33 x 16)
constexpr int WLR = 256;        // regions per image

__device__ __forceinline__ bf16x8 lds_frag(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }
__device__ __forceinline__ f32x16 mfma16(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
// C layout of one 32x32 block: register q of lane (l31, lhi) holds row (q & 3) + 8 * (q >> 2) + 4 * lhi, column l31
__device__ __forceinline__ int c_row(int q, int lhi) { return (q & 3) + 8 * (q >> 2) + 4 * lhi; }

__device__ __forceinline__ float sumsq8(uint4 v) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    float s = 0.f;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const float lo = __uint_as_float(d[q] << 16), hi = __uint_as_float(d[q] & 0xffff0000u);
        s += lo * lo + hi * hi;
    }
    return s;
}
__device__ __forceinline__ uint4 scale8(uint4 v, float s) {
    const uint32_t d[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) o[q] = pack_bf2(__uint_as_float(d[q] << 16) * s, __uint_as_float(d[q] & 0xffff0000u) * s);
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// LDS tile [32][tp] (bf16) -> dst[e][0..31] for all e < E (row pitch ldt, 64 contiguous bytes per e)
__device__ __forceinline__ void store_transposed32(const bf16_t* __restrict__ tile, int tp, bf16_t* __restrict__ dst, size_t ldt, int E) {
    for (int e = threadIdx.x; e < E; e += 256) {
        uint32_t w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = (uint32_t)tile[(2 * r) * tp + e] | ((uint32_t)tile[(2 * r + 1) * tp + e] << 16);
        uint4* o = reinterpret_cast<uint4*>(dst + (size_t)e * ldt);
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = make_uint4(w[4 * k], w[4 * k + 1], w[4 * k + 2], w[4 * k + 3]);
    }
}

// ---- x (bf16 [B*R][E]) -> rn = l2_normalize(x) (attention_lib.py:30-33), rnT[j][e][r], rinv; workgroup = 32 rows of one image
__global__ __launch_bounds__(256) void wl_prep_regions_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ rn,
                                                              bf16_t* __restrict__ rnT, float* __restrict__ rinv, int R, int E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);              // [32][E + 8]
    const int tp = E + 8;
    const int j = blockIdx.y, r0 = blockIdx.x * 32;
    const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const size_t grow = (size_t)j * R + r0 + row;
    const bf16_t* xr = x + grow * E;
    const int nv = E >> 6;                                       // 16-byte vectors per thread
    float ss = 0.f;
    for (int i = 0; i < nv; ++i) {
        const int k = (sub + 8 * i) * 8;
        const uint4 v = *reinterpret_cast<const uint4*>(xr + k);
        *reinterpret_cast<uint4*>(tile + row * tp + k) = v;
        ss += sumsq8(v);
    }
    ss += __shfl_xor(ss, 1); ss += __shfl_xor(ss, 2); ss += __shfl_xor(ss, 4);
    const float iv = rsqrtf(fmaxf(ss, 1e-12f));
    if (sub == 0) rinv[grow] = iv;
    for (int i = 0; i < nv; ++i) {                               // each thread re-reads what it wrote itself
        const int k = (sub + 8 * i) * 8;
        const uint4 v = scale8(*reinterpret_cast<const uint4*>(tile + row * tp + k), iv);
        *reinterpret_cast<uint4*>(tile + row * tp + k) = v;
        *reinterpret_cast<uint4*>(rn + grow * E + k) = v;
    }
    __syncthreads();
    store_transposed32(tile, tp, rnT + (size_t)j * E * R + r0, (size_t)R, E);
}

// ---- words_n (float32 [ld][E], already normalised) -> w (bf16 [LDP][E], rows >= ld zero) and wT (bf16 [E][LDP])
__global__ __launch_bounds__(256) void wl_prep_words_kernel(const float* __restrict__ wn, bf16_t* __restrict__ w, bf16_t* __restrict__ wT,
                                                            int ld, int LDP, int E) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* tile = reinterpret_cast<bf16_t*>(smem);              // [32][E + 8]
    const int tp = E + 8;
    const int r0 = blockIdx.x * 32;
    const int row = threadIdx.x >> 3, sub = threadIdx.x & 7;
    const int grow = r0 + row;
    const bool valid = grow < ld;
    const float* src = wn + (size_t)min(grow, ld - 1) * E;       // a padding row re-reads the last word and is zeroed
    const int nv = E >> 5;                                       // float4 vectors per thread
    for (int i = 0; i < nv; ++i) {
        const int k = (sub + 8 * i) * 4;
        const float4 v = *reinterpret_cast<const float4*>(src + k);
        const uint2 o = make_uint2(valid ? pack_bf2(v.x, v.y) : 0u, valid ? pack_bf2(v.z, v.w) : 0u);
        *reinterpret_cast<uint2*>(tile + row * tp + k) = o;
        *reinterpret_cast<uint2*>(w + (size_t)grow * E + k) = o;
    }
    __syncthreads();
    store_transposed32(tile, tp, wT + r0, (size_t)LDP, E);
}

// ---- Out[b][x][y] = alpha * sum over segments of X_s[b][x][:] . Y_s[b][y][:]  (bf16 operands, k contiguous) -----------------
struct TnArgs {
    const bf16_t *x0, *y0, *x1, *y1;
    long long sx0, sy0, sx1, sy1;      // batch strides in elements (0: shared by the batch)
    int ldx0, ldy0, ldx1, ldy1;
    int k0, k1;                        // multiples of 64 (k1 may be 0)
    void* out; long long so; int ldo; int out_f32; float alpha;
    int nx, ny, batch;                 // 128-row tiles along x / y
};

// 128 x 128 tile, 4 waves 2 x 2, 64-k stages double-buffered in LDS (one barrier per 16 MFMAs per wave).  The MFMA's A operand
// is Y (rows m = y), its B operand X (columns n = x): a lane of the C layout owns output row x and 16 consecutive y after the
// convolution epilogue's lane swaps -- 32- / 64-byte stores along the contiguous axis of Out.
__global__ __launch_bounds__(256, 2) void wl_tn_gemm_kernel(const TnArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* bufX = reinterpret_cast<bf16_t*>(smem);              // [2][128 * 72]
    bf16_t* bufY = bufX + 2 * 128 * KP64;
    const int nblk = gridDim.x;
    const int bid = xcd_remap(blockIdx.x, nblk);
    const int per = p.nx * p.ny;
    const int b = bid / per, rem = bid - b * per;
    const int xt = rem / p.ny, yt = rem - xt * p.ny;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int wx = wave >> 1, wy = wave & 1;
    const int srow = tid >> 3, skv = (tid & 7) * 8;              // 8 lanes read the 128 bytes of one row's k-stage
    const bf16_t* gx0 = p.x0 + b * p.sx0 + (size_t)(xt * 128 + srow) * p.ldx0 + skv;
    const bf16_t* gy0 = p.y0 + b * p.sy0 + (size_t)(yt * 128 + srow) * p.ldy0 + skv;
    const bf16_t* gx1 = p.k1 ? p.x1 + b * p.sx1 + (size_t)(xt * 128 + srow) * p.ldx1 + skv : gx0;
    const bf16_t* gy1 = p.k1 ? p.y1 + b * p.sy1 + (size_t)(yt * 128 + srow) * p.ldy1 + skv : gy0;
    const int nst0 = p.k0 >> 6, nst = nst0 + (p.k1 >> 6);
    epi_u32x4 rx[4], ry[4];
    auto gload = [&](int s) __attribute__((always_inline)) {
        const bool s1 = s >= nst0;                               // wave-uniform
        const bf16_t* gx = s1 ? gx1 : gx0;
        const bf16_t* gy = s1 ? gy1 : gy0;
        const size_t ldx = (size_t)(s1 ? p.ldx1 : p.ldx0) * 32, ldy = (size_t)(s1 ? p.ldy1 : p.ldy0) * 32;
        const int k = (s1 ? s - nst0 : s) * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            rx[i] = *reinterpret_cast<const epi_u32x4*>(gx + i * ldx + k);
            ry[i] = *reinterpret_cast<const epi_u32x4*>(gy + i * ldy + k);
        }
    };
    auto sstore = [&](int buf) __attribute__((always_inline)) {
        bf16_t* dx = bufX + buf * 128 * KP64 + srow * KP64 + skv;
        bf16_t* dy = bufY + buf * 128 * KP64 + srow * KP64 + skv;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            *reinterpret_cast<epi_u32x4*>(dx + i * 32 * KP64) = rx[i];
            *reinterpret_cast<epi_u32x4*>(dy + i * 32 * KP64) = ry[i];
        }
    };
    f32x16 acc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[u][v][q] = 0.f;
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const bf16_t* bx = bufX + buf * 128 * KP64 + (64 * wx + l31) * KP64 + 8 * lhi;
        const bf16_t* ay = bufY + buf * 128 * KP64 + (64 * wy + l31) * KP64 + 8 * lhi;
        bf16x8 fb[4][2], fa[4][2];                               // all sixteen fragments of the stage are requested first
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            fb[kk][0] = lds_frag(bx + 16 * kk); fb[kk][1] = lds_frag(bx + 32 * KP64 + 16 * kk);
            fa[kk][0] = lds_frag(ay + 16 * kk); fa[kk][1] = lds_frag(ay + 32 * KP64 + 16 * kk);
        }
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            acc[0][0] = mfma16(fa[kk][0], fb[kk][0], acc[0][0]); acc[0][1] = mfma16(fa[kk][1], fb[kk][0], acc[0][1]);
            acc[1][0] = mfma16(fa[kk][0], fb[kk][1], acc[1][0]); acc[1][1] = mfma16(fa[kk][1], fb[kk][1], acc[1][1]);
        }
    };
    gload(0);
    sstore(0);
    __syncthreads();
    for (int s = 0; s + 1 < nst; ++s) {
        gload(s + 1);
        asm volatile("" ::: "memory");                           // the loads stay ABOVE the MFMAs: the scheduler otherwise sinks each next to its ds_write
        compute(s & 1);
        sstore((s + 1) & 1);
        __syncthreads();
    }
    compute((nst - 1) & 1);
    ConvEpi ep;
    ep.bias = nullptr; ep.mask = nullptr; ep.res = nullptr; ep.y = p.out;
    ep.Cout = p.ny * 128; ep.out_f32 = p.out_f32; ep.alpha = p.alpha; ep.res_scale = 0.f;
#pragma unroll
    for (int u = 0; u < 2; ++u) {                                // acc[u][v]: x block u (columns of C = lanes), y block v (rows of C)
        const size_t obase = (size_t)b * p.so + (size_t)(xt * 128 + 64 * wx + 32 * u + l31) * p.ldo;
#pragma unroll
        for (int v = 0; v < 2; ++v) conv_epilogue_block(acc[u][v], yt * 128 + 64 * wy + 32 * v, lhi, obase, obase, ep);
    }
}

// ---- the column stage ------------------------------------------------------------------------------------------------------
struct WlArgs {
    const bf16_t* rn;            // [B][256][E]
    const bf16_t* w;             // [LDP][E], rows >= ld zero
    const bf16_t* g;             // [B][256][256]
    const float* max_len;        // [B]
    float* nn; float* q;         // [B][ld]           forward outputs
    const float* dsim; const float* pi;      // [B][B] (caption i, image j), [B][ld]   backward inputs
    bf16_t *ds, *as, *al;        // [B][256][LDP]     backward outputs
    int B, T, E, ld, LDP;
    float g1, g3;
};

// 16 values of one 32x32 C block (rows = k) -> the two B-operand fragments of its 32 k (attn_mfma.hip::c_to_b_frags)
__device__ __forceinline__ void c_to_frags(const f32x16& v, uint4* f0, uint4* f1) {
    uint32_t o[2][4];
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
        float lo[4], hi[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[8 * kk + q]), __float_as_uint(v[8 * kk + 4 + q]), false, false);
            lo[q] = __uint_as_float(r[0]);
            hi[q] = __uint_as_float(r[1]);
        }
        o[kk][0] = pack_bf2(lo[0], lo[1]); o[kk][1] = pack_bf2(lo[2], lo[3]);
        o[kk][2] = pack_bf2(hi[0], hi[1]); o[kk][3] = pack_bf2(hi[2], hi[3]);
    }
    *f0 = make_uint4(o[0][0], o[0][1], o[0][2], o[0][3]);
    *f1 = make_uint4(o[1][0], o[1][1], o[1][2], o[1][3]);
}

constexpr int WL_LDS_A = 2 * 256 * KP32 * 2;                     // bytes: two 32-k stages of 256 rows
constexpr int WL_LDS_T = 64 * ALP * 2;                           // alpha^T (aliases the two word stages of phase 1: 2 * 64 * 40 * 2 bytes)
constexpr int WL_LDS_RED = 3 * 4 * 64 * 4;
constexpr int WL_LDS = WL_LDS_A + WL_LDS_T + WL_LDS_RED;
// transposing output staging of the backward launch: [256][64 + 8] bf16, aliases the A stages
constexpr int OTP = 72;
static_assert(256 * OTP * 2 <= WL_LDS_A, "output tile must fit the A stages");

// workgroup = (image j, word columns c0 .. c0 + 63); wave wr owns regions 64 wr .. 64 wr + 63 of all 64 columns
template <bool BWD>
__global__ __launch_bounds__(256, 2) void wl_cols_kernel(const WlArgs p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    bf16_t* bufA = reinterpret_cast<bf16_t*>(smem);                                  // [2][256 * 40]
    bf16_t* bufB = reinterpret_cast<bf16_t*>(smem + WL_LDS_A);                       // [2][64 * 40]   phase 1
    bf16_t* alT = bufB;                                                              // [64][264]      phase 3
    float* red = reinterpret_cast<float*>(smem + WL_LDS_A + WL_LDS_T);               // [3][4][64]
    const int ncb = p.LDP >> 6;
    const int bid = xcd_remap(blockIdx.x, gridDim.x);
    const int j = bid / ncb, c0 = (bid - j * ncb) * 64;
    const int tid = threadIdx.x, lane = tid & 63, wr = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    const int E = p.E, ld = p.ld;
    const int srow = tid >> 2, skv = (tid & 3) * 8;              // 4 lanes read the 64 bytes of one row's k-stage
    epi_u32x4 ra[4], rb;

    // ---------------- phase 1: S = R^_j W^^T (raw scores, K = E)
    f32x16 sacc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 16; ++q) sacc[u][v][q] = 0.f;
    {
        const bf16_t* ga = p.rn + ((size_t)j * WLR + srow) * E + skv;
        const bf16_t* gb = p.w + ((size_t)c0 + srow) * E + skv;
        const size_t a64 = (size_t)64 * E;
        auto gload = [&](int k) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const epi_u32x4*>(ga + i * a64 + k);
            rb = *reinterpret_cast<const epi_u32x4*>(gb + k);
        };
        auto sstore = [&](int buf) __attribute__((always_inline)) {
            bf16_t* a = bufA + buf * 256 * KP32 + srow * KP32 + skv;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<epi_u32x4*>(a + i * 64 * KP32) = ra[i];
            *reinterpret_cast<epi_u32x4*>(bufB + buf * 64 * KP32 + srow * KP32 + skv) = rb;
        };
        auto compute = [&](int buf) __attribute__((always_inline)) {
            const bf16_t* a = bufA + buf * 256 * KP32 + (64 * wr + l31) * KP32 + 8 * lhi;
            const bf16_t* b = bufB + buf * 64 * KP32 + l31 * KP32 + 8 * lhi;
            bf16x8 fa[2][2], fb[2][2];                           // the eight fragments of the stage are requested first
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                fa[kk][0] = lds_frag(a + 16 * kk); fa[kk][1] = lds_frag(a + 32 * KP32 + 16 * kk);
                fb[kk][0] = lds_frag(b + 16 * kk); fb[kk][1] = lds_frag(b + 32 * KP32 + 16 * kk);
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                sacc[0][0] = mfma16(fa[kk][0], fb[kk][0], sacc[0][0]); sacc[1][0] = mfma16(fa[kk][1], fb[kk][0], sacc[1][0]);
                sacc[0][1] = mfma16(fa[kk][0], fb[kk][1], sacc[0][1]); sacc[1][1] = mfma16(fa[kk][1], fb[kk][1], sacc[1][1]);
            }
        };
        const int nst = E >> 5;
        gload(0);
        sstore(0);
        __syncthreads();
        for (int s = 0; s + 1 < nst; ++s) {
            gload((s + 1) * 32);
            asm volatile("" ::: "memory");                           // the loads stay ABOVE the MFMAs: the scheduler otherwise sinks each next to its ds_write
            compute(s & 1);
            sstore((s + 1) & 1);
            __syncthreads();
        }
        compute((nst - 1) & 1);
    }
    // the first stage of G_j is requested now and lands under the softmax
    const bf16_t* gg = p.g + ((size_t)j * WLR + srow) * WLR + skv;
#pragma unroll
    for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const epi_u32x4*>(gg + (size_t)i * 64 * WLR);
    asm volatile("" ::: "memory");

    // ---------------- phase 2: alpha = softmax over the 256 regions of gamma1 * S (+ mask), nn = sum alpha S   (attention_lib.py:119-124)
    bool live[2], masked[2];
    int col[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        col[cb] = c0 + 32 * cb + l31;
        live[cb] = col[cb] < ld;
        const int i = live[cb] ? col[cb] / p.T : 0, t = col[cb] - i * p.T;
        masked[cb] = !live[cb] || ((float)t >= p.max_len[i]);    // mask * (-1e9) absorbs every score: uniform probabilities
    }
    float mx[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float m = -INFINITY;
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
            for (int q = 0; q < 16; ++q) m = fmaxf(m, sacc[rbk][cb][q]);
        m = fmaxf(m, __shfl_xor(m, 32));
        if (lhi == 0) red[wr * 64 + 32 * cb + l31] = m;
    }
    __syncthreads();                                             // also: every wave is done with the phase-1 stages
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int cl = 32 * cb + l31;
        mx[cb] = fmaxf(fmaxf(red[cl], red[64 + cl]), fmaxf(red[128 + cl], red[192 + cl]));
    }
    f32x16 al[2][2];
    float nnv[2], inv[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float se = 0.f, ses = 0.f;
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float s = sacc[rbk][cb][q];
                const float e = masked[cb] ? 1.f : expf(p.g1 * (s - mx[cb]));
                al[rbk][cb][q] = e;
                se += e;
                ses += e * s;
            }
        se += __shfl_xor(se, 32);
        ses += __shfl_xor(ses, 32);
        if (lhi == 0) { red[256 + wr * 64 + 32 * cb + l31] = se; red[512 + wr * 64 + 32 * cb + l31] = ses; }
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int cl = 32 * cb + l31;
        const float se = red[256 + cl] + red[320 + cl] + red[384 + cl] + red[448 + cl];
        const float ses = red[512 + cl] + red[576 + cl] + red[640 + cl] + red[704 + cl];
        inv[cb] = 1.f / se;
        nnv[cb] = ses * inv[cb];
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
            for (int q = 0; q < 16; ++q) al[rbk][cb][q] *= inv[cb];
    }
    // alpha -> LDS as the B operand of H = G alpha: alT[column][region], 8 consecutive regions per 16-byte vector
#pragma unroll
    for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk) {
            uint4 f0, f1;
            c_to_frags(al[rbk][cb], &f0, &f1);
            bf16_t* d = alT + (32 * cb + l31) * ALP + 64 * wr + 32 * rbk + 8 * lhi;
            *reinterpret_cast<uint4*>(d) = f0;
            *reinterpret_cast<uint4*>(d + 16) = f1;
        }

    // ---------------- phase 3: H = G_j alpha (K = 256 regions), q = sum alpha H
    f32x16 hacc[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v)
#pragma unroll
            for (int q = 0; q < 16; ++q) hacc[u][v][q] = 0.f;
    {
        auto gload = [&](int k) __attribute__((always_inline)) {
#pragma unroll
            for (int i = 0; i < 4; ++i) ra[i] = *reinterpret_cast<const epi_u32x4*>(gg + (size_t)i * 64 * WLR + k);
        };
        auto sstore = [&](int buf) __attribute__((always_inline)) {
            bf16_t* a = bufA + buf * 256 * KP32 + srow * KP32 + skv;
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<epi_u32x4*>(a + i * 64 * KP32) = ra[i];
        };
        auto compute = [&](int buf, int k) __attribute__((always_inline)) {
            const bf16_t* a = bufA + buf * 256 * KP32 + (64 * wr + l31) * KP32 + 8 * lhi;
            const bf16_t* b = alT + l31 * ALP + k + 8 * lhi;
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const bf16x8 a0 = lds_frag(a + 16 * kk), a1 = lds_frag(a + 32 * KP32 + 16 * kk);
                const bf16x8 b0 = lds_frag(b + 16 * kk), b1 = lds_frag(b + 32 * ALP + 16 * kk);
                hacc[0][0] = mfma16(a0, b0, hacc[0][0]); hacc[1][0] = mfma16(a1, b0, hacc[1][0]);
                hacc[0][1] = mfma16(a0, b1, hacc[0][1]); hacc[1][1] = mfma16(a1, b1, hacc[1][1]);
            }
        };
        constexpr int nst = WLR / 32;
        sstore(0);                                               // stage 0 was requested before the softmax
        __syncthreads();                                         // ... and alT is complete
        for (int s = 0; s + 1 < nst; ++s) {
            gload((s + 1) * 32);
            asm volatile("" ::: "memory");                           // the loads stay ABOVE the MFMAs: the scheduler otherwise sinks each next to its ds_write
            compute(s & 1, s * 32);
            sstore((s + 1) & 1);
            __syncthreads();
        }
        compute((nst - 1) & 1, (nst - 1) * 32);
    }
    float qv[2];
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        float s = 0.f;
#pragma unroll
        for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += al[rbk][cb][q] * hacc[rbk][cb][q];
        s += __shfl_xor(s, 32);
        if (lhi == 0) red[wr * 64 + 32 * cb + l31] = s;
    }
    __syncthreads();
#pragma unroll
    for (int cb = 0; cb < 2; ++cb) {
        const int cl = 32 * cb + l31;
        qv[cb] = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
    }
    if constexpr (!BWD) {
        if (wr == 0 && lhi == 0) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
                if (live[cb]) { p.nn[(size_t)j * ld + col[cb]] = nnv[cb]; p.q[(size_t)j * ld + col[cb]] = qv[cb]; }
        }
        return;
    } else {
        // ---------------- backward of the column stage (losses.hip::wl_bwd_cols_kernel): d cos -> dS, alpha dq
        //   cos = nn / sqrt(q);  dn = dcos / sqrt(q);  dq = -dcos nn q^-3/2 / 2;  d alpha = dn S + 2 dq H;
        //   dS = alpha (dn + gamma1 (d alpha - sum alpha d alpha)),  sum alpha d alpha = dn nn + 2 dq q
        bf16_t* otile = bufA;                                    // [256][72]: the A stages are idle now (barrier above)
        const int orow = tid >> 3, okv = (tid & 7) * 8;          // 8 lanes store the 128 bytes of one output row
#pragma unroll
        for (int which = 0; which < 3; ++which) {
#pragma unroll
            for (int cb = 0; cb < 2; ++cb) {
                const int i = live[cb] ? col[cb] / p.T : 0;
                const size_t sc = (size_t)j * ld + (live[cb] ? col[cb] : 0);
                const float dcos = live[cb] ? p.g3 * p.dsim[(size_t)i * p.B + j] * p.pi[sc] : 0.f;
                const float rq = rsqrtf(qv[cb]);
                const float dn = dcos * rq;
                const float dq = -0.5f * dcos * nnv[cb] * rq * rq * rq;
                const float cst = dn * nnv[cb] + 2.f * dq * qv[cb];
#pragma unroll
                for (int rbk = 0; rbk < 2; ++rbk)
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const float a = live[cb] ? al[rbk][cb][q] : 0.f;
                        float v;
                        if (which == 0) {
                            const float dal = dn * sacc[rbk][cb][q] + 2.f * dq * hacc[rbk][cb][q];
                            v = a * (dn + p.g1 * (dal - cst));
                        } else if (which == 1) {
                            v = a * dq;
                        } else {
                            v = a;
                        }
                        otile[(64 * wr + 32 * rbk + c_row(q, lhi)) * OTP + 32 * cb + l31] = f2bf(v);
                    }
            }
            __syncthreads();
            bf16_t* dst = (which == 0 ? p.ds : which == 1 ? p.as : p.al) + ((size_t)j * WLR + orow) * p.LDP + c0 + okv;
#pragma unroll
            for (int i = 0; i < 8; ++i)
                *reinterpret_cast<uint4*>(dst + (size_t)i * 32 * p.LDP) = *reinterpret_cast<const uint4*>(otile + (orow + 32 * i) * OTP + okv);
            __syncthreads();
        }
    }
}

// dx = inv * (dy - y <y, dy>) with y in bf16 (losses.hip::l2norm_bwd_kernel reads a float32 y)
template <typename T>
__global__ __launch_bounds__(256) void wl_l2norm_bwd_kernel(const float* __restrict__ dy, const bf16_t* __restrict__ y,
                                                            const float* __restrict__ inv, T* __restrict__ dx, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float iv = inv[row];
    const bool clamped = iv >= 999999.0f;
    const float* d = dy + row * cols;
    const bf16_t* yr = y + row * cols;
    float dot = 0.f;
    for (int c = lane * 4; c < cols; c += 256) {                 // cols % 4 == 0
        const float4 dv = *reinterpret_cast<const float4*>(d + c);
        const uint2 yv = *reinterpret_cast<const uint2*>(yr + c);
        dot += dv.x * __uint_as_float(yv.x << 16) + dv.y * __uint_as_float(yv.x & 0xffff0000u) +
               dv.z * __uint_as_float(yv.y << 16) + dv.w * __uint_as_float(yv.y & 0xffff0000u);
    }
    dot = clamped ? 0.f : wave_sum(dot);
    for (int c = lane * 4; c < cols; c += 256) {
        const float4 dv = *reinterpret_cast<const float4*>(d + c);
        const uint2 yv = *reinterpret_cast<const uint2*>(yr + c);
        const float o0 = iv * (dv.x - __uint_as_float(yv.x << 16) * dot), o1 = iv * (dv.y - __uint_as_float(yv.x & 0xffff0000u) * dot);
        const float o2 = iv * (dv.z - __uint_as_float(yv.y << 16) * dot), o3 = iv * (dv.w - __uint_as_float(yv.y & 0xffff0000u) * dot);
        T* o = dx + row * cols + c;
        if constexpr (sizeof(T) == 2) *reinterpret_cast<uint2*>(o) = make_uint2(pack_bf2(o0, o1), pack_bf2(o2, o3));
        else *reinterpret_cast<float4*>(o) = make_float4(o0, o1, o2, o3);
    }
}

bool wl_domain(int b, int r, int t, int e) { return b > 0 && t > 0 && r == WLR && e >= 128 && (e % 128) == 0 && e <= 2048; }
int wl_ldp(int b, int t) { return (b * t + 63) / 64 * 64; }

int wl_optin() {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(&wl_cols_kernel<false>), reinterpret_cast<const void*>(&wl_cols_kernel<true>),
                          reinterpret_cast<const void*>(&wl_tn_gemm_kernel), reinterpret_cast<const void*>(&wl_prep_regions_kernel),
                          reinterpret_cast<const void*>(&wl_prep_words_kernel)}, 160 * 1024)
               ? XMC_OK : XMC_EINVAL;
}

}  // namespace

extern "C" int xmc_wl_fused_supported(int32_t b, int32_t r, int32_t t, int32_t e) { return wl_domain(b, r, t, e) ? 1 : 0; }
extern "C" int xmc_wl_fused_ldp(int32_t b, int32_t t) { return wl_ldp(b, t); }

extern "C" int xmc_wl_prep_regions(const void* x, void* rn, void* rnT, float* rinv, int32_t b, int32_t r, int32_t e, void* stream) {
    XMC_REQUIRE(x && rn && rnT && rinv && b > 0 && r > 0 && (r % 32) == 0 && e >= 64 && (e % 64) == 0 && e <= 2048);
    XMC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)rn % 16) == 0 && ((uintptr_t)rnT % 16) == 0);
    if (wl_optin() != XMC_OK) return XMC_EINVAL;
    hipLaunchKernelGGL(wl_prep_regions_kernel, dim3((unsigned)(r / 32), (unsigned)b), dim3(256), (size_t)32 * (e + 8) * 2,
                       static_cast<hipStream_t>(stream), static_cast<const bf16_t*>(x), static_cast<bf16_t*>(rn),
                       static_cast<bf16_t*>(rnT), rinv, r, e);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_prep_words(const float* words_n, void* w, void* wT, int32_t ld, int32_t ldp, int32_t e, void* stream) {
    XMC_REQUIRE(words_n && w && wT && ld > 0 && ldp >= ld && (ldp % 64) == 0 && e >= 64 && (e % 64) == 0 && e <= 2048);
    XMC_REQUIRE(((uintptr_t)words_n % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)wT % 16) == 0);
    if (wl_optin() != XMC_OK) return XMC_EINVAL;
    hipLaunchKernelGGL(wl_prep_words_kernel, dim3((unsigned)(ldp / 32)), dim3(256), (size_t)32 * (e + 8) * 2,
                       static_cast<hipStream_t>(stream), words_n, static_cast<bf16_t*>(w), static_cast<bf16_t*>(wT), ld, ldp, e);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_tn_gemm(const void* x0, int64_t sx0, int32_t ldx0, const void* y0, int64_t sy0, int32_t ldy0, int32_t k0,
                              const void* x1, int64_t sx1, int32_t ldx1, const void* y1, int64_t sy1, int32_t ldy1, int32_t k1,
                              void* out, int64_t so, int32_t ldo, int32_t out_f32, float alpha, int32_t rows_x, int32_t rows_y,
                              int32_t batch, void* stream) {
    XMC_REQUIRE(x0 && y0 && out && k0 > 0 && (k0 % 64) == 0 && k1 >= 0 && (k1 % 64) == 0 && batch > 0);
    XMC_REQUIRE(rows_x > 0 && (rows_x % 128) == 0 && rows_y > 0 && (rows_y % 128) == 0 && ldo >= rows_y && (ldo % 8) == 0);
    XMC_REQUIRE(ldx0 >= k0 && ldy0 >= k0 && (ldx0 % 8) == 0 && (ldy0 % 8) == 0 && (sx0 % 8) == 0 && (sy0 % 8) == 0 && (so % 8) == 0);
    XMC_REQUIRE(((uintptr_t)x0 % 16) == 0 && ((uintptr_t)y0 % 16) == 0 && ((uintptr_t)out % 16) == 0);
    if (k1) {
        XMC_REQUIRE(x1 && y1 && ldx1 >= k1 && ldy1 >= k1 && (ldx1 % 8) == 0 && (ldy1 % 8) == 0 && (sx1 % 8) == 0 && (sy1 % 8) == 0);
        XMC_REQUIRE(((uintptr_t)x1 % 16) == 0 && ((uintptr_t)y1 % 16) == 0);
    }
    if (wl_optin() != XMC_OK) return XMC_EINVAL;
    TnArgs a{};
    a.x0 = static_cast<const bf16_t*>(x0); a.y0 = static_cast<const bf16_t*>(y0);
    a.x1 = static_cast<const bf16_t*>(x1); a.y1 = static_cast<const bf16_t*>(y1);
    a.sx0 = sx0; a.sy0 = sy0; a.sx1 = sx1; a.sy1 = sy1;
    a.ldx0 = ldx0; a.ldy0 = ldy0; a.ldx1 = ldx1; a.ldy1 = ldy1; a.k0 = k0; a.k1 = k1;
    a.out = out; a.so = so; a.ldo = ldo; a.out_f32 = out_f32; a.alpha = alpha;
    a.nx = rows_x / 128; a.ny = rows_y / 128; a.batch = batch;
    const long long nblk = (long long)a.nx * a.ny * batch;
    XMC_REQUIRE(nblk < (1ll << 31));
    hipLaunchKernelGGL(wl_tn_gemm_kernel, dim3((unsigned)nblk), dim3(256), (size_t)4 * 128 * KP64 * 2, static_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_RET();
}

static int wl_cols_args(WlArgs& a, const void* rn, const void* w, const void* g, const float* max_len, int32_t b, int32_t t, int32_t e,
                        int32_t ldp, float gamma1) {
    XMC_REQUIRE(rn && w && g && max_len && wl_domain(b, WLR, t, e) && ldp == wl_ldp(b, t));
    XMC_REQUIRE(((uintptr_t)rn % 16) == 0 && ((uintptr_t)w % 16) == 0 && ((uintptr_t)g % 16) == 0);
    a.rn = static_cast<const bf16_t*>(rn); a.w = static_cast<const bf16_t*>(w); a.g = static_cast<const bf16_t*>(g);
    a.max_len = max_len; a.B = b; a.T = t; a.E = e; a.ld = b * t; a.LDP = ldp; a.g1 = gamma1;
    return wl_optin();
}

extern "C" int xmc_wl_cols_fwd(const void* rn, const void* w, const void* g, const float* max_len, float* nn, float* q, int32_t b,
                               int32_t t, int32_t e, int32_t ldp, float gamma1, void* stream) {
    WlArgs a{};
    XMC_REQUIRE(nn && q);
    const int rc = wl_cols_args(a, rn, w, g, max_len, b, t, e, ldp, gamma1);
    if (rc != XMC_OK) return rc;
    a.nn = nn; a.q = q;
    hipLaunchKernelGGL(wl_cols_kernel<false>, dim3((unsigned)(b * (ldp / 64))), dim3(256), (size_t)WL_LDS, static_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_cols_bwd(const void* rn, const void* w, const void* g, const float* max_len, const float* dsim_t,
                               const float* pi, void* ds, void* as, void* al, int32_t b, int32_t t, int32_t e, int32_t ldp,
                               float gamma1, float gamma3, void* stream) {
    WlArgs a{};
    XMC_REQUIRE(dsim_t && pi && ds && as && al);
    XMC_REQUIRE(((uintptr_t)ds % 16) == 0 && ((uintptr_t)as % 16) == 0 && ((uintptr_t)al % 16) == 0);
    const int rc = wl_cols_args(a, rn, w, g, max_len, b, t, e, ldp, gamma1);
    if (rc != XMC_OK) return rc;
    a.dsim = dsim_t; a.pi = pi; a.g3 = gamma3;
    a.ds = static_cast<bf16_t*>(ds); a.as = static_cast<bf16_t*>(as); a.al = static_cast<bf16_t*>(al);
    hipLaunchKernelGGL(wl_cols_kernel<true>, dim3((unsigned)(b * (ldp / 64))), dim3(256), (size_t)WL_LDS, static_cast<hipStream_t>(stream), a);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_l2norm_rows_bwd_bf16y(const float* dy, const void* y, const float* inv, void* dx, int64_t rows, int32_t cols,
                                         int32_t dtype_out, void* stream) {
    XMC_REQUIRE(dy && y && inv && dx && rows > 0 && cols > 0 && (cols % 4) == 0);
    XMC_REQUIRE(((uintptr_t)dy % 16) == 0 && ((uintptr_t)y % 8) == 0);
    const dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype_out == XMC_BF16)
        hipLaunchKernelGGL(wl_l2norm_bwd_kernel<bf16_t>, grid, block, 0, static_cast<hipStream_t>(stream), dy, static_cast<const bf16_t*>(y),
                           inv, static_cast<bf16_t*>(dx), (long long)rows, cols);
    else if (dtype_out == XMC_F32)
        hipLaunchKernelGGL(wl_l2norm_bwd_kernel<float>, grid, block, 0, static_cast<hipStream_t>(stream), dy, static_cast<const bf16_t*>(y),
                           inv, static_cast<float*>(dx), (long long)rows, cols);
    else
        return XMC_EINVAL;
    XMC_LAUNCH_RET();
}
