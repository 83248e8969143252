// Batched ("multi-tensor") spectral normalisation for ALL spectrally-normalised weights of the
// discriminator in a handful of launches (the per-weight path in spectral.hip costs ~6 launches per
// weight per forward: ~360 launches and 4.3 ms per training step at 21 weights).
//
// A device table describes every weight W_i (rows x cols float32 inside one parameter arena):
//   u_axis 0: conv master [cout][K]  -- u over rows, v over cols
//   u_axis 1: dense kernel (in, out) -- u over cols, v over rows
// One power-iteration step (reference xmcgan/libml/layers.py:92-101, :209-220) for all i:
//   P1  v_raw_i = W_i^T u0_i  (axis 0)  |  W_i u0_i (axis 1)
//   P2  v_i = v_raw_i * rsqrt(|v_raw_i|^2 + eps)
//   P3  u_raw_i = W_i v_i     (axis 0)  |  W_i^T v_i (axis 1)
//   P4  u_i = u_raw_i * rsqrt(|u_raw_i|^2 + eps); sigma_i = u_raw_i . u_i; scal_i = {sigma, 1/(sigma+eps)}
//   P5  prepared conv weights (activation dtype, forward + dgrad layouts, scaled by 1/(sigma+eps))
// and for the backward pass (gradient through sigma):
//   B1  dot_i = <G_i, W_i>        B2  G_i <- (G_i - dot_i * inv_i * outer(u_i, v_i)) * inv_i
// Work is split into fixed-size chunks; a workgroup finds its (weight, chunk) with a linear scan of
// the <= 64-entry prefix table.
#include "common.h"

namespace {

struct SnEntry {                 // one spectrally-normalised weight (all offsets in floats)
    long long w_off;             // into the parameter arena (and the gradient arena)
    int rows, cols, u_axis;
    int u_off, v_off;            // into the flat u / v buffers (u0, u_new, u_raw share u_off)
    int blk_a, blk_b;            // first workgroup of this weight in the "row-chunk" / "col-chunk" grids
    int taps, is_conv;           // conv: rows = cout, cols = taps * cin
    long long wf_off, wd_off;    // into the prepared forward / dgrad weight buffers (elements)
    int blk_p;                   // first workgroup in the prep grid
    int packed;                  // bit 0 / 1: forward / dgrad copy in MFMA-fragment order (bf16 only)
};

constexpr int ROWS_PER_WG = 4;       // "rows" pass: 4 rows per workgroup, walked by all 256 threads together (4 KB
                                     // contiguous per row per step, x read once for the four rows, four independent
                                     // loads of W in flight per thread); the host table's blk_a counts ceil(rows / 4)
constexpr int COLS_PER_WG = 32;      // "cols" pass: columns per workgroup (all rows); the host table's blk_b counts ceil(cols / 32)

// Entry that owns workgroup `bid` of grid `which` (0: rows, 1: cols, 2: prep / grad-fix): the last entry whose prefix is
// <= bid.  One parallel probe (lane l reads entry l's prefix, n <= 64) instead of a walk through the table: the walk was
// a chain of up to n dependent global loads in front of every workgroup -- the 4-row workgroups of the "rows" pass
// spent longer finding their entry than streaming their 220 KB.
__device__ __forceinline__ int find_entry(const SnEntry* __restrict__ tab, int n, int bid, int which) {
    const int lane = threadIdx.x & 63;
    int start = 0x7fffffff;
    if (lane < n) start = which == 0 ? tab[lane].blk_a : (which == 1 ? tab[lane].blk_b : tab[lane].blk_p);
    const unsigned long long owns = __ballot(start <= bid);
    return __popcll(owns) - 1;
}

// y[r] = sum_c W[r][c] x[c] for rows 4 chunk .. 4 chunk + 3; the four waves' partial sums are added in a fixed order
__device__ __forceinline__ void rows_pass(const float* __restrict__ W, const float* __restrict__ x,
                                          float* __restrict__ y, int rows, int cols, int chunk) {
    __shared__ float wsum[4][ROWS_PER_WG];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r0 = chunk * ROWS_PER_WG;
    if (r0 >= rows) return;
    float s[ROWS_PER_WG];
    const float* wr[ROWS_PER_WG];
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG; ++k) {
        s[k] = 0.f;
        wr[k] = W + (long long)min(r0 + k, rows - 1) * cols;      // rows past the end re-read the last row (result unused)
    }
    if ((cols & 3) == 0) {               // 16-byte loads (every arena tensor starts 256-byte aligned)
        const float4* x4 = reinterpret_cast<const float4*>(x);
        const int nq = cols >> 2;
        for (int c = tid; c < nq; c += 256) {
            const float4 b = x4[c];
            float4 a[ROWS_PER_WG];
#pragma unroll
            for (int k = 0; k < ROWS_PER_WG; ++k) a[k] = reinterpret_cast<const float4*>(wr[k])[c];
#pragma unroll
            for (int k = 0; k < ROWS_PER_WG; ++k) s[k] += a[k].x * b.x + a[k].y * b.y + a[k].z * b.z + a[k].w * b.w;
        }
    } else {
        for (int c = tid; c < cols; c += 256) {
            const float b = x[c];
#pragma unroll
            for (int k = 0; k < ROWS_PER_WG; ++k) s[k] += wr[k][c] * b;
        }
    }
#pragma unroll
    for (int k = 0; k < ROWS_PER_WG; ++k) {
        const float t = wave_sum(s[k]);
        if (lane == 0) wsum[wave][k] = t;
    }
    __syncthreads();
    if (tid < ROWS_PER_WG && r0 + tid < rows) y[r0 + tid] = (wsum[0][tid] + wsum[1][tid]) + (wsum[2][tid] + wsum[3][tid]);
}

// y[c] = sum_r x[r] W[r][c] over ALL rows for a block of 32 columns: 8 column QUADS (16-byte loads, one 128-byte line per
// row) x 32 row groups per workgroup, four independent loads in flight per thread, the row groups combined through LDS in a
// fixed order -- no cross-workgroup reduction, no atomics, no zero-initialised output: the power iteration is
// bit-reproducible.  (Round 3: 128 columns x 8 row groups with two loads in flight gave the largest weight 108 workgroups
// and the pass 2 TB/s.)
__device__ __forceinline__ void cols_pass(const float* __restrict__ W, const float* __restrict__ x,
                                          float* __restrict__ y, int rows, int cols, int chunk) {
    __shared__ float4 part[32][8];
    if ((cols & 3) == 0) {
        const int q = threadIdx.x & 7, rg = threadIdx.x >> 3;
        const int c = chunk * COLS_PER_WG + q * 4;
        float4 s[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) s[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c < cols) {
            int r = rg;
            for (; r + 96 < rows; r += 128) {
                float xa[4];
                float4 wa[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    xa[k] = x[r + 32 * k];
                    wa[k] = *reinterpret_cast<const float4*>(W + (long long)(r + 32 * k) * cols + c);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) { s[k].x += xa[k] * wa[k].x; s[k].y += xa[k] * wa[k].y; s[k].z += xa[k] * wa[k].z; s[k].w += xa[k] * wa[k].w; }
            }
            for (; r < rows; r += 32) {
                const float xa = x[r];
                const float4 wa = *reinterpret_cast<const float4*>(W + (long long)r * cols + c);
                s[0].x += xa * wa.x; s[0].y += xa * wa.y; s[0].z += xa * wa.z; s[0].w += xa * wa.w;
            }
            s[0].x = (s[0].x + s[1].x) + (s[2].x + s[3].x); s[0].y = (s[0].y + s[1].y) + (s[2].y + s[3].y);
            s[0].z = (s[0].z + s[1].z) + (s[2].z + s[3].z); s[0].w = (s[0].w + s[1].w) + (s[2].w + s[3].w);
        }
        part[rg][q] = s[0];
        __syncthreads();
        if (rg == 0 && c < cols) {
            float4 t = part[0][q];
#pragma unroll
            for (int k = 1; k < 32; ++k) { t.x += part[k][q].x; t.y += part[k][q].y; t.z += part[k][q].z; t.w += part[k][q].w; }
            *reinterpret_cast<float4*>(y + c) = t;
        }
        return;
    }
    // column counts that are not a multiple of 4 (the RGB layers: 27 and 3 columns; the 1536 -> 1 output dense: ONE column):
    // the workgroup's columns (<= 32, rounded up to a power of two CPW) x 256 / CPW row groups, the groups added through LDS
    // in group order.  (One thread per column walking all rows made the 1536 -> 1 dense a chain of 1,536 dependent-issue loads
    // in ONE thread: 130 of the 160 us of the power iteration's second product -- the 352 MB "rows" pass beside it was done
    // long before; found in round 4 when no change to the rows pass moved the launch.)
    __shared__ float ps[256];
    const int ncol = min(COLS_PER_WG, cols - chunk * COLS_PER_WG);
    int cpw = 1;
    while (cpw < ncol) cpw <<= 1;
    const int nrg = 256 / cpw;
    const int cl = threadIdx.x & (cpw - 1), rg = threadIdx.x / cpw;
    const int c = chunk * COLS_PER_WG + cl;
    float s = 0.f;
    if (cl < ncol) {
        int r = rg;
        for (; r + 3 * nrg < rows; r += 4 * nrg) {
            float xa[4], wa[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { xa[k] = x[r + k * nrg]; wa[k] = W[(long long)(r + k * nrg) * cols + c]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) s += xa[k] * wa[k];
        }
        for (; r < rows; r += nrg) s += x[r] * W[(long long)r * cols + c];
    }
    ps[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && cl < ncol) {
        float t = ps[cl];
        for (int k = 1; k < nrg; ++k) t += ps[k * cpw + cl];
        y[c] = t;
    }
}

// phase 1 (mode 0): v_raw from u0; phase 3 (mode 1): u_raw from v.  grid = blocks_a + blocks_b
__global__ __launch_bounds__(256) void sn_matvec_kernel(const SnEntry* __restrict__ tab, int n,
                                                        const float* __restrict__ params,
                                                        const float* __restrict__ uvec,     // u0 (mode 0) | unused
                                                        const float* __restrict__ vvec,     // v (mode 1)
                                                        float* __restrict__ out_v,          // v_raw (mode 0)
                                                        float* __restrict__ out_u,          // u_raw (mode 1)
                                                        int mode, int blocks_a) {
    // the "rows" grid (A) serves: mode 0 & axis 1 (v_raw = W u0), mode 1 & axis 0 (u_raw = W v)
    // the "cols" grid (B) serves: mode 0 & axis 0 (v_raw = W^T u0), mode 1 & axis 1 (u_raw = W^T v)
    const bool gridA = (int)blockIdx.x < blocks_a;
    const int bid = gridA ? blockIdx.x : blockIdx.x - blocks_a;
    const int i = find_entry(tab, n, bid, gridA ? 0 : 1);
    const SnEntry e = tab[i];
    const float* W = params + e.w_off;
    const int chunk = bid - (gridA ? e.blk_a : e.blk_b);
    if (gridA) {
        if (mode == 0 && e.u_axis == 1) rows_pass(W, uvec + e.u_off, out_v + e.v_off, e.rows, e.cols, chunk);
        else if (mode == 1 && e.u_axis == 0) rows_pass(W, vvec + e.v_off, out_u + e.u_off, e.rows, e.cols, chunk);
    } else {
        if (mode == 0 && e.u_axis == 0) cols_pass(W, uvec + e.u_off, out_v + e.v_off, e.rows, e.cols, chunk);
        else if (mode == 1 && e.u_axis == 1) cols_pass(W, vvec + e.v_off, out_u + e.u_off, e.rows, e.cols, chunk);
    }
}

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    __syncthreads();
    return t;
}

// phase 2: one workgroup per weight: v <- v_raw * rsqrt(|v_raw|^2 + eps)   (in place)
__global__ __launch_bounds__(1024) void sn_norm_v_kernel(const SnEntry* __restrict__ tab, float* __restrict__ v,
                                                         float eps) {
    __shared__ float red[16];
    const SnEntry e = tab[blockIdx.x];
    const int nv = e.u_axis == 0 ? e.cols : e.rows;
    float* y = v + e.v_off;
    float ss = 0.f;
    for (int i = threadIdx.x; i < nv; i += blockDim.x) ss += y[i] * y[i];
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss + eps);
    for (int i = threadIdx.x; i < nv; i += blockDim.x) y[i] *= inv;
}

// phase 4: one workgroup per weight: u_new, sigma, 1/(sigma + eps)
__global__ __launch_bounds__(1024) void sn_finalize_kernel(const SnEntry* __restrict__ tab,
                                                           const float* __restrict__ u_raw, float* __restrict__ u_new,
                                                           float* __restrict__ scal, float eps) {
    __shared__ float red[16];
    const SnEntry e = tab[blockIdx.x];
    const int nu = e.u_axis == 0 ? e.rows : e.cols;
    const float* ur = u_raw + e.u_off;
    float* un = u_new + e.u_off;
    float ss = 0.f;
    for (int i = threadIdx.x; i < nu; i += blockDim.x) ss += ur[i] * ur[i];
    ss = block_sum(ss, red);
    const float inv = rsqrtf(ss + eps);
    float sg = 0.f;
    for (int i = threadIdx.x; i < nu; i += blockDim.x) {
        const float u = ur[i] * inv;
        un[i] = u;
        sg += ur[i] * u;
    }
    sg = block_sum(sg, red);
    if (threadIdx.x == 0) {
        scal[2 * blockIdx.x] = sg;
        scal[2 * blockIdx.x + 1] = 1.f / (sg + eps);
    }
}

// phase 5: prepared conv weights for every conv entry; 32x32 LDS transpose tiles (see spectral.hip)
template <typename T>
__global__ __launch_bounds__(256) void sn_prep_kernel(const SnEntry* __restrict__ tab, int n,
                                                      const float* __restrict__ params,
                                                      const float* __restrict__ scal, T* __restrict__ wf_buf,
                                                      T* __restrict__ wd_buf, int skip_phase) {
    __shared__ float tile[32][33];
    const int i = find_entry(tab, n, blockIdx.x, 2);
    const SnEntry e = tab[i];
    if (!e.is_conv) return;
    if (skip_phase && (e.packed & 4)) return;        // a phase site: only its 16-tap copies are read (xmc_phase_conv_weight)
    const int cout = e.rows, taps = e.taps, cin = e.cols / e.taps;
    const int tc = (cin + 31) / 32, tn = (cout + 31) / 32;
    int b = blockIdx.x - e.blk_p;
    const int tap = b / (tc * tn);
    b -= tap * tc * tn;
    const int n0 = (b / tc) * 32, c0 = (b % tc) * 32;
    const float is = scal[2 * i + 1];
    const float* w = params + e.w_off;
    T* wf = wf_buf + e.wf_off;
    T* wd = wd_buf ? wd_buf + e.wd_off : nullptr;
    prep_weight_tile<T>(tile, w, is, wf, wd, cout, taps, cin, tap, n0, c0, e.packed);
}

// backward B1: dots[chunk] = <G_i, W_i> over this workgroup's fixed 64K-element chunk (sn_fix_kernel adds an
// entry's chunk partials in a fixed order: no atomics, bit-reproducible)
constexpr int DOT_CHUNK = 65536;
__global__ __launch_bounds__(256) void sn_dot_kernel(const SnEntry* __restrict__ tab, int n,
                                                     const float* __restrict__ params,
                                                     const float* __restrict__ grads, float* __restrict__ dots) {
    __shared__ float red[4];
    const int i = find_entry(tab, n, blockIdx.x, 2);         // reuses blk_p as the chunk prefix (dot table)
    const SnEntry e = tab[i];
    const long long total = (long long)e.rows * e.cols;
    const long long lo = (long long)(blockIdx.x - e.blk_p) * DOT_CHUNK, hi = min(total, lo + DOT_CHUNK);
    const float* w = params + e.w_off;
    const float* g = grads + e.w_off;
    float s = 0.f;
    if ((total & 3) == 0) {              // chunk bounds are multiples of 4: 16-byte loads
        const float4* g4 = reinterpret_cast<const float4*>(g);
        const float4* w4 = reinterpret_cast<const float4*>(w);
        for (long long k = (lo >> 2) + threadIdx.x; k < (hi >> 2); k += 256) {
            const float4 a = g4[k], b = w4[k];
            s += a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
        }
    } else {
        for (long long k = lo + threadIdx.x; k < hi; k += 256) s += g[k] * w[k];
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) dots[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);    // one partial per 64K-element chunk
}

// backward B2: G <- (G - dot * inv * outer(u, v)) * inv
__global__ __launch_bounds__(256) void sn_fix_kernel(const SnEntry* __restrict__ tab, int n,
                                                     float* __restrict__ grads, const float* __restrict__ u,
                                                     const float* __restrict__ v, const float* __restrict__ scal,
                                                     const float* __restrict__ dots) {
    const int i = find_entry(tab, n, blockIdx.x, 2);
    const SnEntry e = tab[i];
    const long long total = (long long)e.rows * e.cols;
    const long long lo = (long long)(blockIdx.x - e.blk_p) * DOT_CHUNK, hi = min(total, lo + DOT_CHUNK);
    float* g = grads + e.w_off;
    const float is = scal[2 * i + 1];
    // <G_i, W_i> = sum of this entry's chunk partials (at most a few hundred), every workgroup in the same order
    __shared__ float dred[256];
    {
        const int nchunk = (int)((total + DOT_CHUNK - 1) / DOT_CHUNK);
        float t = 0.f;
        for (int b = threadIdx.x; b < nchunk; b += 256) t += dots[e.blk_p + b];
        dred[threadIdx.x] = t;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if ((int)threadIdx.x < st) dred[threadIdx.x] += dred[threadIdx.x + st];
            __syncthreads();
        }
    }
    const float k = dred[0] * is;
    const float* uu = u + e.u_off;
    const float* vv = v + e.v_off;
    if ((e.cols & 3) == 0) {             // a float4 never straddles a row; every index fits 32 bits (<= 21 M elements)
        float4* g4 = reinterpret_cast<float4*>(g);
        const unsigned cols = (unsigned)e.cols;
        for (unsigned t = (unsigned)lo + threadIdx.x * 4u; t < (unsigned)hi; t += 1024u) {
            const unsigned r = t / cols, c = t - r * cols;
            float4 gv = g4[t >> 2];
            if (e.u_axis == 0) {
                const float ur = uu[r] * k;
                const float4 v4 = *reinterpret_cast<const float4*>(vv + c);
                gv.x = (gv.x - ur * v4.x) * is; gv.y = (gv.y - ur * v4.y) * is;
                gv.z = (gv.z - ur * v4.z) * is; gv.w = (gv.w - ur * v4.w) * is;
            } else {
                const float vr = vv[r] * k;
                const float4 u4 = *reinterpret_cast<const float4*>(uu + c);
                gv.x = (gv.x - vr * u4.x) * is; gv.y = (gv.y - vr * u4.y) * is;
                gv.z = (gv.z - vr * u4.z) * is; gv.w = (gv.w - vr * u4.w) * is;
            }
            g4[t >> 2] = gv;
        }
        return;
    }
    for (long long t = lo + threadIdx.x; t < hi; t += 256) {
        const int r = (int)(t / e.cols), c = (int)(t - (long long)r * e.cols);
        const float uv = e.u_axis == 0 ? uu[r] * vv[c] : uu[c] * vv[r];
        g[t] = (g[t] - k * uv) * is;
    }
}


// ---- round 4: prepared weights that are a pure cast of W (1 / sigma rides in xmc_conv_desc.alpha_dev) -----------------------
// With W / sigma folded into the convolution's alpha, the activation-dtype copies no longer wait for the power iteration, so
// ONE pass over each weight serves everything the forward pass needs from it: a workgroup loads a 32 (cout) x 32 (cin) tile of
// all taps of the float32 master into LDS and emits
//   * the fragment-ordered forward and data-gradient copies (what sn_prep_kernel / prep_weight_kernel wrote, tap by tap),
//   * the 16-tap phase copies of the layers next to a 2x resampling (what phase_weight_kernel wrote from a second read),
//   * its 32 rows' share of v_raw = W^T u0 (the first matvec of the power iteration: sn_matvec_kernel's "cols" pass, a third
//     read, at 1.9 TB/s on 128-byte row segments) as one partial row per row tile; sn_colsum_kernel adds the row tiles in a
//     fixed order (no atomics: bit-reproducible).
// Spectral pass of D per half step: 3 reads of the 352 MB arena + the copies -> 2 reads + the copies.
struct WprepEntry {              // mirrors xmc_wprep_entry
    long long w_off;             // floats into the parameter arena
    long long wf_off, wd_off;    // bf16 elements into the plain forward / data-gradient buffers
    long long pf_off, pd_off;    // bf16 elements into the phase forward / data-gradient buffers
    long long part_off;          // floats into the partial-sum buffer: part[part_off + rowtile * (taps * cin) + col]
    int cout, cin, taps;         // cout % 32 == cin % 32 == 0, taps 1 or 9
    int blk0;                    // first workgroup of this entry in the tile grid ((cout / 32) x (cin / 32) workgroups)
    int flags;                   // bit 0 / 1: plain forward / data-gradient copy; bits 2-3: 0 none, 1 "ups", 2 "pool" phase copies;
                                 // bit 4: W^T u0 partials
    int u_off, v_off;            // slices of the flat u0 / v buffers (floats)
    int blk_c;                   // first workgroup of this entry in the column-sum grid (256 columns per workgroup)
};

__device__ __forceinline__ int find_wprep(const WprepEntry* __restrict__ tab, int n, int bid, int which) {
    const int lane = threadIdx.x & 63;
    int start = 0x7fffffff;
    if (lane < n) start = which == 0 ? tab[lane].blk0 : tab[lane].blk_c;
    return __popcll(__ballot(start <= bid)) - 1;
}

// Everything a row x column tile of a weight emits once its 32 x 32 x taps float32 values sit in LDS (t9) and, for a spectral
// entry, the 32 u0 rows in `us`: W^T u0 partial row, fragment-ordered plain copies, 16-tap phase copies.
__device__ __forceinline__ void wprep_emit(const WprepEntry& e, float (&t9)[9][32][33], const float (&us)[32], int n0, int c0, int tid,
                                           bf16_t* __restrict__ wf_buf, bf16_t* __restrict__ wd_buf, bf16_t* __restrict__ pf_buf,
                                           bf16_t* __restrict__ pd_buf, float* __restrict__ part) {
    const int cout = e.cout, cin = e.cin, taps = e.taps;
    const int tc = cin >> 5;
    if (e.flags & 16) {                              // this row tile's share of v_raw[col] = sum_r u0[r] W[r][col]
        const int cols = taps * cin;
        float* __restrict__ pr = part + e.part_off + (size_t)(n0 >> 5) * cols;
        for (int k = tid; k < taps * 32; k += 256) {
            const int tap = k >> 5, c = k & 31;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
            for (int r = 0; r < 32; r += 4) {
                s0 += us[r] * t9[tap][r][c]; s1 += us[r + 1] * t9[tap][r + 1][c];
                s2 += us[r + 2] * t9[tap][r + 2][c]; s3 += us[r + 3] * t9[tap][r + 3][c];
            }
            pr[tap * cin + c0 + c] = (s0 + s1) + (s2 + s3);
        }
    }
    // ---- plain copies in fragment order: per tap one 2 KiB block each (see prep_weight_tile)
    if (e.flags & 3) {
        const int v16 = tid & 127, kk = v16 >> 6, lhi = (v16 >> 5) & 1, l31 = v16 & 31;
        const int kb = kk * 16 + lhi * 8;
        bf16_t* __restrict__ wf = wf_buf + e.wf_off;
        bf16_t* __restrict__ wd = wd_buf + e.wd_off;
        for (int tap = 0; tap < taps; ++tap) {
            float f[8];
            if (tid < 128) {
                if (e.flags & 1) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) f[q] = t9[tap][l31][kb + q];
                    Vec<bf16_t> o; o.set(f);
                    const long long blk = ((long long)(n0 >> 5) * tc + (c0 >> 5)) * taps + tap;
                    o.store(wf + blk * 1024 + v16 * 8);
                }
            } else if (e.flags & 2) {
#pragma unroll
                for (int q = 0; q < 8; ++q) f[q] = t9[tap][kb + q][l31];
                Vec<bf16_t> o; o.set(f);
                const long long blk = ((long long)(c0 >> 5) * (cout >> 5) + (n0 >> 5)) * taps + (taps - 1 - tap);
                o.store(wd + blk * 1024 + v16 * 8);
            }
        }
    }
    // ---- 16-tap phase copies (conv_stream.hip: phase_weight_kernel; fwd_mode 0 = "ups" layer, 1 = "pool" layer)
    const int pm = (e.flags >> 2) & 3;
    if (pm && taps == 9) {
        const int fwd_mode = pm - 1;
        bf16_t* __restrict__ pf = pf_buf + e.pf_off;
        bf16_t* __restrict__ pd = pd_buf + e.pd_off;
        auto lo_of = [](int a, int tu) { return a == 0 ? (tu == 0 ? 0 : 1) : (tu == 0 ? 0 : 2); };
        auto hi_of = [](int a, int tu) { return a == 0 ? (tu == 0 ? 0 : 2) : (tu == 0 ? 1 : 2); };
        const int item = tid & 127, r = item & 31, k8 = item >> 5;
        for (int t = tid >> 7; t < 16; t += 2) {
            const int ph = t >> 2, tu = (t >> 1) & 1, tv = t & 1;
            const int a = fwd_mode == 0 ? (ph >> 1) : 1 - (ph >> 1), bb = fwd_mode == 0 ? (ph & 1) : 1 - (ph & 1);
            {
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int dy = lo_of(a, tu); dy <= hi_of(a, tu); ++dy)
                    for (int dx = lo_of(bb, tv); dx <= hi_of(bb, tv); ++dx)
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += t9[dy * 3 + dx][r][k8 * 8 + q];
                *reinterpret_cast<uint4*>(pf + packed_w_index(n0 + r, t, c0 + k8 * 8, 16, cin >> 5)) =
                    make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            }
            {
                const int ru = 1 - tu, rv = 1 - tv;
                float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                for (int dy = lo_of(a, ru); dy <= hi_of(a, ru); ++dy)
                    for (int dx = lo_of(bb, rv); dx <= hi_of(bb, rv); ++dx)
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] += t9[dy * 3 + dx][k8 * 8 + q][r];
                *reinterpret_cast<uint4*>(pd + packed_w_index(c0 + r, t, n0 + k8 * 8, 16, cout >> 5)) =
                    make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
            }
        }
    }
}

__global__ __launch_bounds__(256) void wprep_kernel(const WprepEntry* __restrict__ tab, int n, const float* __restrict__ params,
                                                    const float* __restrict__ u0, bf16_t* __restrict__ wf_buf,
                                                    bf16_t* __restrict__ wd_buf, bf16_t* __restrict__ pf_buf,
                                                    bf16_t* __restrict__ pd_buf, float* __restrict__ part) {
    __shared__ float t9[9][32][33];
    __shared__ float us[32];
    const WprepEntry e = tab[find_wprep(tab, n, blockIdx.x, 0)];
    const int cin = e.cin, taps = e.taps;
    const int tc = cin >> 5, b = blockIdx.x - e.blk0;
    const int n0 = (b / tc) * 32, c0 = (b % tc) * 32;
    const float* __restrict__ w = params + e.w_off;
    const int tid = threadIdx.x;
    {   // every load of the tile in flight before the first LDS write (no bounds: cout, cin are multiples of 32)
        const int row = tid >> 3, c4 = (tid & 7) * 4;
        if (taps == 9) {
            float4 v[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) v[tap] = *reinterpret_cast<const float4*>(w + ((size_t)(n0 + row) * 9 + tap) * cin + c0 + c4);
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                t9[tap][row][c4] = v[tap].x; t9[tap][row][c4 + 1] = v[tap].y; t9[tap][row][c4 + 2] = v[tap].z; t9[tap][row][c4 + 3] = v[tap].w;
            }
        } else {
            const float4 v = *reinterpret_cast<const float4*>(w + (size_t)(n0 + row) * cin + c0 + c4);
            t9[0][row][c4] = v.x; t9[0][row][c4 + 1] = v.y; t9[0][row][c4 + 2] = v.z; t9[0][row][c4 + 3] = v.w;
        }
        if ((e.flags & 16) && tid < 32) us[tid] = u0[e.u_off + n0 + tid];
    }
    __syncthreads();
    wprep_emit(e, t9, us, n0, c0, tid, wf_buf, wd_buf, pf_buf, pd_buf, part);
}

// ---- round 5: the optimiser emits the prepared weights ------------------------------------------------------------------------
// W-bar's copies are a pure function of W (xmcgan/libml/layers.py:209-221), and the optimiser is the kernel that holds the NEW W
// in registers: for every weight of the batched preparation table this kernel IS the Adam (+ EMA) update of its 32 x 32 x taps
// tile -- same arithmetic, element for element, as adam_kernel (pointwise.hip), including the gradient through sigma on the way
// in (FIX) and the zeroing / keeping of the consumed gradient -- followed by wprep_kernel's emission from the updated tile:
// the next forward pass finds its copies (and the first product of its power iteration, taken with THIS half step's new u) ready
// and never reads the float32 masters.  The flat Adam kernel skips these tensors (map value -2).
template <bool FIX>
__global__ __launch_bounds__(256) void adam_wprep_kernel(const WprepEntry* __restrict__ tab, int n, float* __restrict__ p,
                                                         float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                                         float* __restrict__ ema, float lr, float b1, float b2, float eps,
                                                         const float* __restrict__ corr, float gs, float d, int zero_g,
                                                         const float* __restrict__ kvec, const float* __restrict__ scal,
                                                         const float* __restrict__ uvec, const float* __restrict__ vvec,
                                                         bf16_t* __restrict__ wf_buf, bf16_t* __restrict__ wd_buf,
                                                         bf16_t* __restrict__ pf_buf, bf16_t* __restrict__ pd_buf,
                                                         float* __restrict__ part) {
    __shared__ float t9[9][32][33];
    __shared__ float us[32];
    const WprepEntry e = tab[find_wprep(tab, n, blockIdx.x, 0)];
    const int cin = e.cin, taps = e.taps;
    const int tc = cin >> 5, b = blockIdx.x - e.blk0;
    const int n0 = (b / tc) * 32, c0 = (b % tc) * 32;
    const int tid = threadIdx.x;
    const int row = tid >> 3, c4 = (tid & 7) * 4;
    const float ic1 = corr[1], ic2 = corr[2];        // device-side step counter, already advanced (adam_advance_kernel)
    const bool spectral = FIX && (e.flags & 16);
    float ur = 0.f, is = 1.f;
    if (spectral) {
        const int ent = e.flags >> 8;                // index of this weight in the spectral bank (kvec / scal)
        is = scal[2 * ent + 1];
        ur = uvec[e.u_off + n0 + row] * kvec[ent];
        if (tid < 32) us[tid] = uvec[e.u_off + n0 + tid];
    }
    const size_t base = (size_t)e.w_off + (size_t)(n0 + row) * taps * cin + c0 + c4;
    constexpr int TG = 3;                            // taps per trip: 12-15 16-byte loads in flight per thread
    for (int t0 = 0; t0 < taps; t0 += TG) {
        float4 pp[TG], gg[TG], mm[TG], vv[TG], ee[TG], v4[TG];
#pragma unroll
        for (int k = 0; k < TG; ++k) {
            const int tap = min(t0 + k, taps - 1);   // (taps == 1: the two spare slots re-read tap 0 and are not stored)
            const size_t o = base + (size_t)tap * cin;
            pp[k] = *reinterpret_cast<const float4*>(p + o);
            gg[k] = *reinterpret_cast<const float4*>(g + o);
            mm[k] = *reinterpret_cast<const float4*>(m + o);
            vv[k] = *reinterpret_cast<const float4*>(v + o);
            if (ema) ee[k] = *reinterpret_cast<const float4*>(ema + o);
            if (spectral) v4[k] = *reinterpret_cast<const float4*>(vvec + e.v_off + tap * cin + c0 + c4);
        }
#pragma unroll
        for (int k = 0; k < TG; ++k) {
            const int tap = t0 + k;
            if (tap >= taps) break;
            const size_t o = base + (size_t)tap * cin;
            float* pf = reinterpret_cast<float*>(&pp[k]);
            float* gf = reinterpret_cast<float*>(&gg[k]);
            float* mf = reinterpret_cast<float*>(&mm[k]);
            float* vf = reinterpret_cast<float*>(&vv[k]);
            if (spectral) {
                gf[0] = (gf[0] - ur * v4[k].x) * is; gf[1] = (gf[1] - ur * v4[k].y) * is;
                gf[2] = (gf[2] - ur * v4[k].z) * is; gf[3] = (gf[3] - ur * v4[k].w) * is;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float gr = gf[q] * gs;
                mf[q] = b1 * mf[q] + (1.f - b1) * gr;
                vf[q] = b2 * vf[q] + (1.f - b2) * gr * gr;
                pf[q] -= lr * (mf[q] * ic1) / (sqrtf(vf[q] * ic2) + eps);
            }
            *reinterpret_cast<float4*>(p + o) = pp[k];
            *reinterpret_cast<float4*>(m + o) = mm[k];
            *reinterpret_cast<float4*>(v + o) = vv[k];
            if (zero_g == 1) *reinterpret_cast<float4*>(g + o) = make_float4(0.f, 0.f, 0.f, 0.f);
            else if (zero_g == 2 && spectral) *reinterpret_cast<float4*>(g + o) = gg[k];
            if (ema) {
                float* ef = reinterpret_cast<float*>(&ee[k]);
#pragma unroll
                for (int q = 0; q < 4; ++q) ef[q] = ef[q] * d + (1.f - d) * pf[q];
                *reinterpret_cast<float4*>(ema + o) = ee[k];
            }
            t9[tap][row][c4] = pf[0]; t9[tap][row][c4 + 1] = pf[1]; t9[tap][row][c4 + 2] = pf[2]; t9[tap][row][c4 + 3] = pf[3];
        }
    }
    __syncthreads();
    wprep_emit(e, t9, us, n0, c0, tid, wf_buf, wd_buf, pf_buf, pd_buf, part);
}

// v_raw[col] = sum over the row tiles of wprep_kernel's partial rows, in row-tile order (256 columns per workgroup)
__global__ __launch_bounds__(256) void sn_colsum_kernel(const WprepEntry* __restrict__ tab, int n, const float* __restrict__ part,
                                                        float* __restrict__ v) {
    const WprepEntry e = tab[find_wprep(tab, n, blockIdx.x, 1)];
    if (!(e.flags & 16)) return;
    const int cols = e.taps * e.cin, col = (blockIdx.x - e.blk_c) * 256 + threadIdx.x;
    if (col >= cols) return;
    const float* __restrict__ pr = part + e.part_off + col;
    const int nrt = e.cout >> 5;
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    int rt = 0;
    for (; rt + 3 < nrt; rt += 4) {                  // four loads in flight; the order of the sum is fixed
        const float a0 = pr[(size_t)rt * cols], a1 = pr[(size_t)(rt + 1) * cols], a2 = pr[(size_t)(rt + 2) * cols], a3 = pr[(size_t)(rt + 3) * cols];
        s[0] += a0; s[1] += a1; s[2] += a2; s[3] += a3;
    }
    for (; rt < nrt; ++rt) s[0] += pr[(size_t)rt * cols];
    v[e.v_off + col] = (s[0] + s[1]) + (s[2] + s[3]);
}

// k_i = <G_i, W_i> / (sigma_i + eps): the entry's chunk partials of sn_dot_kernel added in chunk order by ONE workgroup per
// entry (sn_fix_kernel re-added them in every workgroup); read by the fused Adam kernel (pointwise.hip)
__global__ __launch_bounds__(256) void sn_dot_finish_kernel(const SnEntry* __restrict__ tab, const float* __restrict__ dots,
                                                            const float* __restrict__ scal, float* __restrict__ kvec) {
    __shared__ float dred[256];
    const SnEntry e = tab[blockIdx.x];
    const long long total = (long long)e.rows * e.cols;
    const int nchunk = (int)((total + DOT_CHUNK - 1) / DOT_CHUNK);
    float t = 0.f;
    for (int b = threadIdx.x; b < nchunk; b += 256) t += dots[e.blk_p + b];
    dred[threadIdx.x] = t;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) dred[threadIdx.x] += dred[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) kvec[blockIdx.x] = dred[0] * scal[2 * blockIdx.x + 1];
}

}  // namespace

// The table is an array of n xmc_sn_entry (include/xmcgan_hip.h) in DEVICE memory; `grid_*` are the
// launch sizes the host computed together with the blk_* prefix fields.
extern "C" int xmc_sn_batched_power_iter(const void* table, int32_t n, const float* params, const float* u0,
                                         float* u_new, float* v, float* u_raw, float* scal, int32_t blocks_a,
                                         int32_t blocks_b, int32_t nu_total, int32_t nv_total, float eps,
                                         void* stream) {
    XMC_REQUIRE(table && params && u0 && u_new && v && u_raw && scal && n > 0 && n <= 64);
    XMC_REQUIRE(sizeof(SnEntry) == sizeof(xmc_sn_entry));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SnEntry* tab = static_cast<const SnEntry*>(table);
    (void)nu_total; (void)nv_total;          // every element of v / u_raw is written exactly once: nothing to zero
    const dim3 grid(blocks_a + blocks_b);
    hipLaunchKernelGGL(sn_matvec_kernel, grid, dim3(256), 0, s, tab, n, params, u0, (const float*)nullptr, v,
                       (float*)nullptr, 0, blocks_a);
    hipLaunchKernelGGL(sn_norm_v_kernel, dim3(n), dim3(1024), 0, s, tab, v, eps);
    hipLaunchKernelGGL(sn_matvec_kernel, grid, dim3(256), 0, s, tab, n, params, (const float*)nullptr, v,
                       (float*)nullptr, u_raw, 1, blocks_a);
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(n), dim3(1024), 0, s, tab, u_raw, u_new, scal, eps);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_sn_batched_prep(const void* table, int32_t n, const float* params, const float* scal,
                                   void* wf_buf, void* wd_buf, int32_t blocks_p, int32_t dtype, void* stream) {
    XMC_REQUIRE(table && params && scal && wf_buf && n > 0 && n <= 64 && blocks_p > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SnEntry* tab = static_cast<const SnEntry*>(table);
    const int skip_phase = (dtype >> 8) & 1;         // bit 8 of dtype: leave the 3x3 copies of the phase sites (packed bit 2) unwritten
    dtype &= 255;
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((sn_prep_kernel<bf16_t>), dim3(blocks_p), dim3(256), 0, s, tab, n, params, scal,
                           static_cast<bf16_t*>(wf_buf), static_cast<bf16_t*>(wd_buf), skip_phase);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((sn_prep_kernel<float>), dim3(blocks_p), dim3(256), 0, s, tab, n, params, scal,
                           static_cast<float*>(wf_buf), static_cast<float*>(wd_buf), 0);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

// `table` here must carry the DOT-chunk prefix in blk_p (the host keeps a second copy of the table).
extern "C" int xmc_sn_batched_grad_fix(const void* table, int32_t n, const float* params, float* grads,
                                       const float* u, const float* v, const float* scal, float* dots,
                                       int32_t blocks, void* stream) {
    XMC_REQUIRE(table && params && grads && u && v && scal && dots && n > 0 && n <= 64 && blocks > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SnEntry* tab = static_cast<const SnEntry*>(table);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(blocks), dim3(256), 0, s, tab, n, params, grads, dots);
    hipLaunchKernelGGL(sn_fix_kernel, dim3(blocks), dim3(256), 0, s, tab, n, grads, u, v, scal, dots);
    XMC_LAUNCH_RET();
}

// ---- round 4 entry points (see the kernels above) --------------------------------------------------------------------------
extern "C" int xmc_wprep_batched(const void* table, int32_t n, const float* params, const float* u0, void* wf_buf, void* wd_buf,
                                 void* pf_buf, void* pd_buf, float* part, int32_t blocks, void* stream) {
    XMC_REQUIRE(table && params && n > 0 && n <= 64 && blocks > 0);
    XMC_REQUIRE(sizeof(WprepEntry) == sizeof(xmc_wprep_entry));
    hipLaunchKernelGGL(wprep_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), static_cast<const WprepEntry*>(table), n,
                       params, u0, static_cast<bf16_t*>(wf_buf), static_cast<bf16_t*>(wd_buf), static_cast<bf16_t*>(pf_buf),
                       static_cast<bf16_t*>(pd_buf), part);
    XMC_LAUNCH_RET();
}

// The Adam (+ EMA) update of the `table` weights fused with their preparation (adam_wprep_kernel).  `step_state` must ALREADY be
// advanced for this update: call xmc_adam_ema_dev_sn on the same arena first -- with a map that carries -2 on these tensors,
// so that the flat kernel leaves them alone -- then this.  kvec / scal / u / vv (spectral arenas: the gradient through sigma of
// the entries whose flags carry bit 4, bank index in flags >> 8; u = the NEW u of this half step, which is also the u0 of the
// next power iteration -- `part` receives W_new^T u) may be NULL for a plain arena.
extern "C" int xmc_adam_wprep_tiles(const void* table, int32_t n, int32_t blocks, float* p, float* g, float* m, float* v, float* ema,
                                    float lr, double beta1, double beta2, float eps, const float* step_state, float grad_scale,
                                    float ema_decay, int32_t zero_grads, const float* kvec, const float* scal, const float* u,
                                    const float* vv, void* wf_buf, void* wd_buf, void* pf_buf, void* pd_buf, float* part,
                                    void* stream) {
    XMC_REQUIRE(table && p && g && m && v && step_state && n > 0 && n <= 64 && blocks > 0);
    XMC_REQUIRE(sizeof(WprepEntry) == sizeof(xmc_wprep_entry));
    XMC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 && ((uintptr_t)v % 16) == 0 &&
                (ema == nullptr || ((uintptr_t)ema % 16) == 0));
    const bool fix = kvec != nullptr;
    XMC_REQUIRE(!fix || (scal && u && vv && part && ((uintptr_t)vv % 16) == 0));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const WprepEntry* tab = static_cast<const WprepEntry*>(table);
    if (fix)
        hipLaunchKernelGGL((adam_wprep_kernel<true>), dim3(blocks), dim3(256), 0, s, tab, n, p, g, m, v, ema, lr, (float)beta1, (float)beta2,
                           eps, step_state, grad_scale, ema_decay, zero_grads, kvec, scal, u, vv, static_cast<bf16_t*>(wf_buf),
                           static_cast<bf16_t*>(wd_buf), static_cast<bf16_t*>(pf_buf), static_cast<bf16_t*>(pd_buf), part);
    else
        hipLaunchKernelGGL((adam_wprep_kernel<false>), dim3(blocks), dim3(256), 0, s, tab, n, p, g, m, v, ema, lr, (float)beta1, (float)beta2,
                           eps, step_state, grad_scale, ema_decay, zero_grads, (const float*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, static_cast<bf16_t*>(wf_buf), static_cast<bf16_t*>(wd_buf),
                           static_cast<bf16_t*>(pf_buf), static_cast<bf16_t*>(pd_buf), part);
    XMC_LAUNCH_RET();
}

// Power iteration whose first matvec arrives as partial rows from xmc_wprep_batched for the `wtab` entries (`part`); the
// `irr` table (may be empty: n_irr = 0) lists the remaining weights (dense kernels, <= 3-channel convolutions), whose
// first matvec runs here.  `table` = all n weights (second matvec, normalisation, sigma).
extern "C" int xmc_sn_power_iter_fused(const void* table, int32_t n, int32_t blocks_a, int32_t blocks_b, const void* irr, int32_t n_irr,
                                       int32_t irr_blocks_a, int32_t irr_blocks_b, const void* wtab, int32_t n_w, int32_t blocks_c,
                                       const float* params, const float* u0, const float* part, float* u_new, float* v, float* u_raw,
                                       float* scal, float eps, void* stream) {
    XMC_REQUIRE(table && params && u0 && u_new && v && u_raw && scal && n > 0 && n <= 64 && n_irr >= 0 && n_irr <= 64 && n_w >= 0 && n_w <= 64);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SnEntry* tab = static_cast<const SnEntry*>(table);
    if (n_irr > 0) {
        XMC_REQUIRE(irr);
        hipLaunchKernelGGL(sn_matvec_kernel, dim3(irr_blocks_a + irr_blocks_b), dim3(256), 0, s, static_cast<const SnEntry*>(irr), n_irr, params,
                           u0, (const float*)nullptr, v, (float*)nullptr, 0, irr_blocks_a);
    }
    if (n_w > 0) {
        XMC_REQUIRE(wtab && part && blocks_c > 0);
        hipLaunchKernelGGL(sn_colsum_kernel, dim3(blocks_c), dim3(256), 0, s, static_cast<const WprepEntry*>(wtab), n_w, part, v);
    }
    hipLaunchKernelGGL(sn_norm_v_kernel, dim3(n), dim3(1024), 0, s, tab, v, eps);
    hipLaunchKernelGGL(sn_matvec_kernel, dim3(blocks_a + blocks_b), dim3(256), 0, s, tab, n, params, (const float*)nullptr, v,
                       (float*)nullptr, u_raw, 1, blocks_a);
    hipLaunchKernelGGL(sn_finalize_kernel, dim3(n), dim3(1024), 0, s, tab, u_raw, u_new, scal, eps);
    XMC_LAUNCH_RET();
}

// kvec[i] = <G_i, W_i> / (sigma_i + eps) for every table entry (the dot-chunk table, as xmc_sn_batched_grad_fix): the scalar of
// the gradient through sigma, consumed by xmc_adam_ema_dev_sn
extern "C" int xmc_sn_batched_dot(const void* table, int32_t n, const float* params, const float* grads, const float* scal,
                                  float* dots, float* kvec, int32_t blocks, void* stream) {
    XMC_REQUIRE(table && params && grads && scal && dots && kvec && n > 0 && n <= 64 && blocks > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const SnEntry* tab = static_cast<const SnEntry*>(table);
    hipLaunchKernelGGL(sn_dot_kernel, dim3(blocks), dim3(256), 0, s, tab, n, params, grads, dots);
    hipLaunchKernelGGL(sn_dot_finish_kernel, dim3(n), dim3(256), 0, s, tab, dots, scal, kvec);
    XMC_LAUNCH_RET();
}
