// Attention and contrastive-loss kernels (float32 arithmetic throughout: the reference promotes
// these paths to fp32 via its float32 masks).  The GEMM-shaped parts of
// word_loss / contrastive_loss run on xmc_gemm_f32; the kernels here are the row / column
// softmax, log-sum-exp, normalisation and cross-entropy stages around them, with wave64
// shuffle reductions.
#include "common.h"

namespace {

constexpr int EMAX = 16;   // per-lane feature registers: feature dim <= 64 * EMAX

// ---------------------------------------------------------------------------- l2_normalize rows
template <typename T>
__global__ __launch_bounds__(256) void l2norm_fwd_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                         float* __restrict__ inv, long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + row * cols;
    float ss = 0.f;
    for (int c = lane; c < cols; c += 64) { const float v = to_f<T>(xr[c]); ss += v * v; }
    ss = wave_sum(ss);
    const float iv = rsqrtf(fmaxf(ss, 1e-12f));
    for (int c = lane; c < cols; c += 64) y[row * cols + c] = to_f<T>(xr[c]) * iv;
    if (lane == 0) inv[row] = iv;
}

template <typename T>
__global__ __launch_bounds__(256) void l2norm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                         const float* __restrict__ inv, T* __restrict__ dx,
                                                         long long rows, int cols) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float iv = inv[row];
    const bool clamped = iv >= 999999.0f;          // sum x^2 <= 1e-12: y = x * 1e6, no norm term
    float dot = 0.f;
    if (!clamped)
        for (int c = lane; c < cols; c += 64) dot += dy[row * cols + c] * y[row * cols + c];
    dot = wave_sum(dot);
    for (int c = lane; c < cols; c += 64) {
        const long long k = row * cols + c;
        dx[k] = from_f<T>(iv * (dy[k] - (clamped ? 0.f : y[k] * dot)));
    }
}


// ---- round 5: contrastive_loss (xmcgan/libml/attention_lib.py:46-79) in two launches per direction ---------------------------------
// Rounds 1-4: l2-normalise a, l2-normalise b, one float32 GEMM for the B x B logits (K = 1536: split K + its reduction), xent_sym
// -- five launches of a few microseconds each, four heads per step; backward: two GEMMs, two normalisation adjoints, two adds.
// Forward here: workgroup i owns row i of `a`: |a_i|, and for every j the dot product <a_i, b_j> and |b_j| (each wave takes every
// fourth j; b stays in L2: B x D floats) -> logits[i][j] = <a_i, b_j> / (|a_i| |b_j| T) with the reference's clamp
// rsqrt(max(sum x^2, 1e-12)); ainv / binv are kept for the backward pass (the normalised copies are never written).
__global__ __launch_bounds__(256) void cl_logits_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ logits,
                                                        float* __restrict__ ainv, float* __restrict__ binv, int B, int D, float inv_t) {
    extern __shared__ float ar[];                    // a_i (D floats, zero-padded to a multiple of 512)
    __shared__ float red[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int Dp = (D + 511) & ~511;
    float ss = 0.f;
    for (int c = threadIdx.x; c < Dp; c += 256) { const float v = c < D ? a[(size_t)i * D + c] : 0.f; ar[c] = v; ss += v * v; }
    ss = wave_sum(ss);
    if (lane == 0) red[wave] = ss;
    __syncthreads();
    const float aiv = rsqrtf(fmaxf((red[0] + red[1]) + (red[2] + red[3]), 1e-12f));
    if (threadIdx.x == 0 && blockIdx.y == 0) ainv[i] = aiv;
    // blockIdx.y = one of gridDim.y slices of the rows j; inside it each wave takes every fourth j, eight loads in flight per lane
    for (int j = blockIdx.y * 4 + wave; j < B; j += 4 * gridDim.y) {
        const float* __restrict__ br = b + (size_t)j * D;
        float dot = 0.f, bq = 0.f;
        for (int c0 = 0; c0 < Dp; c0 += 512) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) { const int c = c0 + u * 64 + lane; v[u] = br[c < D ? c : D - 1]; }
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int c = c0 + u * 64 + lane;
                const float w = c < D ? v[u] : 0.f;
                dot += ar[c] * w;
                bq += w * w;
            }
        }
        dot = wave_sum(dot);
        bq = wave_sum(bq);
        if (lane == 0) {
            const float biv = rsqrtf(fmaxf(bq, 1e-12f));
            logits[(size_t)i * B + j] = dot * aiv * biv * inv_t;
            if (i == 0) binv[j] = biv;
        }
    }
}

// Backward onto one operand: workgroup i owns row i of x (x = a: coefficients dl[i][j]; x = b, TRANS: dl[j][i]):
//   dxn_i = (1 / T) sum_j dl(i, j) yinv_j y_j,   dx_i = xinv_i (dxn_i - xn_i <xn_i, dxn_i>)   (xn_i = x_i xinv_i; clamped rows: no norm term)
// written to out[i] (accumulate: added to it -- the discriminator adds these terms to the pooled-feature gradient).
__global__ __launch_bounds__(256) void cl_bwd_kernel(const float* __restrict__ dl, const float* __restrict__ x, const float* __restrict__ y,
                                                     const float* __restrict__ xinv, const float* __restrict__ yinv, float* __restrict__ out,
                                                     int B, int D, float inv_t, int trans, int accumulate) {
    extern __shared__ float coef[];                  // dl(i, j) yinv_j / T  (B floats)
    __shared__ float red[4];
    const int i = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = threadIdx.x; j < B; j += 256) coef[j] = (trans ? dl[(size_t)j * B + i] : dl[(size_t)i * B + j]) * yinv[j] * inv_t;
    __syncthreads();
    const float xiv = xinv[i];
    const bool clamped = xiv >= 999999.0f;
    constexpr int CMAX = 8;                          // columns per thread: D <= 2048
    float acc[CMAX], xn[CMAX];
    float dot = 0.f;
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
        const int c = threadIdx.x + k * 256;
        acc[k] = 0.f;
        xn[k] = c < D ? x[(size_t)i * D + c] * xiv : 0.f;
    }
    for (int j0 = 0; j0 < B; j0 += 4) {              // four rows of y per trip: up to 32 loads in flight per thread
        float yv[4][CMAX], cj[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int j = min(j0 + u, B - 1);
            cj[u] = j0 + u < B ? coef[j] : 0.f;
            const float* __restrict__ yr = y + (size_t)j * D;
#pragma unroll
            for (int k = 0; k < CMAX; ++k) {
                const int c = threadIdx.x + k * 256;
                yv[u][k] = yr[c < D ? c : D - 1];
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int k = 0; k < CMAX; ++k) acc[k] += cj[u] * yv[u][k];      // (columns >= D accumulate garbage that is never stored)
    }
#pragma unroll
    for (int k = 0; k < CMAX; ++k) dot += acc[k] * xn[k];
    dot = wave_sum(dot);
    if (lane == 0) red[wave] = dot;
    __syncthreads();
    const float tot = clamped ? 0.f : (red[0] + red[1]) + (red[2] + red[3]);
#pragma unroll
    for (int k = 0; k < CMAX; ++k) {
        const int c = threadIdx.x + k * 256;
        if (c < D) {
            const float v = xiv * (acc[k] - xn[k] * tot);
            float* o = out + (size_t)i * D + c;
            *o = accumulate ? *o + v : v;
        }
    }
}

// --------------------------------------------------------------------------------- attention_for_g
// One wave per region.  Lane t (< T) owns word t's score / probability.
template <typename T>
__global__ __launch_bounds__(256) void attn_g_fwd_kernel(const T* __restrict__ region,
                                                         const float* __restrict__ words_n,
                                                         const float* __restrict__ max_len, T* __restrict__ ctx,
                                                         float* __restrict__ attn, float* __restrict__ rinv, int B,
                                                         int R, int Tn, int E, float gamma) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);     // b*R + r
    if (row >= (long long)B * R) return;
    const int b = (int)(row / R);
    const T* rr = region + row * E;
    float rv[EMAX];
    float ss = 0.f;
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const int e = lane + 64 * i;
        const float val = to_f<T>(rr[min(e, E - 1)]);            // branch-free (see the note above wb_at)
        rv[i] = e < E ? val : 0.f;
        ss += rv[i] * rv[i];
    }
    ss = wave_sum(ss);
    const float iv = rsqrtf(fmaxf(ss, 1e-12f));
#pragma unroll
    for (int i = 0; i < EMAX; ++i) rv[i] *= iv;
    const float ml = max_len[b];
    const float* wb = words_n + (long long)b * Tn * E;
    // Every load of a word element is UNCONDITIONAL at a clamped index, and the tail lanes (e >= E) multiply it by a
    // zero of their own: `if (e < E) d += ... wb[...]` compiled to EMAX serial (branch, load, s_waitcnt vmcnt(0))
    // sequences per word -- T x EMAX exposed L2 latencies per region.
    auto wb_at = [&](int t, int i) { return wb[(long long)t * E + min(lane + 64 * i, E - 1)]; };
    float mine = -INFINITY;
    for (int t = 0; t < Tn; ++t) {
        float d = 0.f, wv[EMAX];
#pragma unroll
        for (int i = 0; i < EMAX; ++i) wv[i] = wb_at(t, i);               // all loads first: the single accumulator chain below
        asm volatile("" ::: "memory");                                    // otherwise pulled each load next to its FMA (vmcnt(0) x EMAX)
#pragma unroll
        for (int i = 0; i < EMAX; ++i) d += rv[i] * wv[i];                // rv is zero past E
        d = wave_sum(d);
        float s = d * gamma;
        s = s + (((float)t >= ml) ? 1.0f : 0.0f) * (-1e9f);      // mask * (-1e9), fp32 rounding kept
        if (lane == t) mine = s;
    }
    const float mx = wave_max(mine);
    const float ex = lane < Tn ? expf(mine - mx) : 0.f;
    const float p = ex / wave_sum(ex);
    if (lane < Tn) attn[row * Tn + lane] = p;
    float out[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) out[i] = 0.f;
    for (int t = 0; t < Tn; ++t) {
        const float pt = __shfl(p, t);
#pragma unroll
        for (int i = 0; i < EMAX; ++i) out[i] += pt * wb_at(t, i);        // lanes past E are never stored
    }
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const int e = lane + 64 * i;
        if (e < E) ctx[row * E + e] = from_f<T>(out[i]);
    }
    if (lane == 0) rinv[row] = iv;
}

template <typename T>
__global__ __launch_bounds__(256) void attn_g_bwd_kernel(const T* __restrict__ dctx, const T* __restrict__ region,
                                                         const float* __restrict__ words_n,
                                                         const float* __restrict__ attn,
                                                         const float* __restrict__ rinv, T* __restrict__ dregion,
                                                         int B, int R, int Tn, int E, float gamma) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long long)B * R) return;
    const int b = (int)(row / R);
    const float* wb = words_n + (long long)b * Tn * E;
    const float iv = rinv[row];
    float dc[EMAX], rh[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const int e = lane + 64 * i, ec = min(e, E - 1);
        const float dv = to_f<T>(dctx[row * E + ec]), rg = to_f<T>(region[row * E + ec]);
        dc[i] = e < E ? dv : 0.f;
        rh[i] = e < E ? rg * iv : 0.f;
    }
    const float pa = attn[row * Tn + min(lane, Tn - 1)];
    const float p = lane < Tn ? pa : 0.f;
    auto wb_at = [&](int t, int i) { return wb[(long long)t * E + min(lane + 64 * i, E - 1)]; };   // as in the forward kernel
    float dp = 0.f;
    for (int t = 0; t < Tn; ++t) {
        float d = 0.f, wv[EMAX];
#pragma unroll
        for (int i = 0; i < EMAX; ++i) wv[i] = wb_at(t, i);
        asm volatile("" ::: "memory");
#pragma unroll
        for (int i = 0; i < EMAX; ++i) d += dc[i] * wv[i];                // dc is zero past E
        d = wave_sum(d);
        if (lane == t) dp = d;
    }
    const float pdp = wave_sum(p * dp);
    const float ds = p * (dp - pdp) * gamma;          // d loss / d (r_hat . w_hat_t)
    float dr[EMAX];
#pragma unroll
    for (int i = 0; i < EMAX; ++i) dr[i] = 0.f;
    for (int t = 0; t < Tn; ++t) {
        const float dst = __shfl(ds, t);
#pragma unroll
        for (int i = 0; i < EMAX; ++i) dr[i] += dst * wb_at(t, i);        // lanes past E: multiplied by rh = 0, never stored
    }
    float dot = 0.f;
#pragma unroll
    for (int i = 0; i < EMAX; ++i) dot += dr[i] * rh[i];
    dot = wave_sum(dot);
    const bool clamped = iv >= 999999.0f;
#pragma unroll
    for (int i = 0; i < EMAX; ++i) {
        const int e = lane + 64 * i;
        if (e < E) dregion[row * E + e] = from_f<T>(iv * (dr[i] - (clamped ? 0.f : rh[i] * dot)));
    }
}

// ------------------------------------------------------------------------------------ word_loss
// Column stage: workgroup = (image j, 64 columns (i,t)); the R x 64 tile of S is staged in LDS.
__global__ __launch_bounds__(256) void wl_softmax_kernel(const float* __restrict__ S,
                                                         const float* __restrict__ max_len,
                                                         float* __restrict__ A, float* __restrict__ NN, int B, int R,
                                                         int Tn, float gamma1) {
    extern __shared__ __attribute__((aligned(16))) float sm[];
    float* tile = sm;                  // [R][64]
    float* red = sm + R * 64;          // [4][64] x 2
    const int ld = B * Tn;
    const int j = blockIdx.y, c0 = blockIdx.x * 64;
    const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int col = c0 + cl;
    const bool live = col < ld;
    const float* Sj = S + (long long)j * R * ld;
    float mx = -INFINITY;
    // eight rows per trip, loaded together (a dead column reads column 0; rows past R re-read the last row and are dropped):
    // one guarded load per trip made the R / 4 = 64 trips 64 dependent memory latencies -- 71 us per launch
    const int colc = live ? col : 0;
    for (int r = rg; r < R; r += 32) {
        float v8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v8[u] = Sj[(long long)min(r + 4 * u, R - 1) * ld + colc];
        asm volatile("" ::: "memory");               // keeps the loads above their (otherwise conditional) consumers
#pragma unroll
        for (int u = 0; u < 8; ++u) {                // a clamped row rewrites row R - 1 with its own value
            const float v = live ? v8[u] : 0.f;
            tile[min(r + 4 * u, R - 1) * 64 + cl] = v;
            mx = fmaxf(mx, v);
        }
    }
    red[rg * 64 + cl] = mx;
    __syncthreads();
    mx = fmaxf(fmaxf(red[cl], red[64 + cl]), fmaxf(red[128 + cl], red[192 + cl]));
    const int i = live ? col / Tn : 0, t = live ? col - i * Tn : 0;
    const bool masked = live && ((float)t >= max_len[i]);
    float se = 0.f, ses = 0.f;
    for (int r = rg; r < R; r += 4) {
        const float v = tile[r * 64 + cl];
        const float e = masked ? 1.f : expf(gamma1 * (v - mx));     // masked: s - 1e9 == -1e9 for all r
        se += e;
        ses += e * v;
    }
    __syncthreads();
    red[rg * 64 + cl] = se;
    red[256 + rg * 64 + cl] = ses;
    __syncthreads();
    se = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
    ses = red[256 + cl] + red[320 + cl] + red[384 + cl] + red[448 + cl];
    if (!live) return;
    const float inv = 1.f / se;
    float* Aj = A + (long long)j * R * ld;
    for (int r = rg; r < R; r += 4) {
        const float v = tile[r * 64 + cl];
        Aj[(long long)r * ld + col] = (masked ? 1.f : expf(gamma1 * (v - mx))) * inv;
    }
    if (rg == 0) NN[(long long)j * ld + col] = ses * inv;
}

__global__ __launch_bounds__(256) void wl_qdot_kernel(const float* __restrict__ A, const float* __restrict__ H,
                                                      float* __restrict__ Q, int B, int R, int Tn) {
    __shared__ float red[256];
    const int ld = B * Tn;
    const int j = blockIdx.y, col = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
    float s = 0.f;
    if (col < ld) {
        const long long base = (long long)j * R * ld + col;
        for (int r = rg; r < R; r += 32) {               // eight rows of both operands in flight (same order of the sum)
            float a8[8], h8[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const long long o = base + (long long)min(r + 4 * u, R - 1) * ld;
                a8[u] = A[o]; h8[u] = H[o];
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u = 0; u < 8; ++u) s += (r + 4 * u < R) ? a8[u] * h8[u] : 0.f;
        }
    }
    red[threadIdx.x] = s;
    __syncthreads();
    if (rg == 0 && col < ld)
        Q[(long long)j * ld + col] = red[threadIdx.x] + red[threadIdx.x + 64] + red[threadIdx.x + 128] +
                                     red[threadIdx.x + 192];
}

// one thread per (caption i, image j)
__global__ void wl_rows_kernel(const float* __restrict__ NN, const float* __restrict__ Q,
                               const float* __restrict__ max_len, float* __restrict__ sim_t,
                               float* __restrict__ PI, int B, int Tn, float g2, float g3) {
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= B * B) return;
    const int i = idx / B, j = idx - i * B;
    const long long base = (long long)j * B * Tn + (long long)i * Tn;
    const float ml = max_len[i];
    float mx = -INFINITY;
    for (int t = 0; t < Tn; ++t) {
        float row = g2 * (NN[base + t] * rsqrtf(Q[base + t]));
        row = row + (((float)t >= ml) ? 1.0f : 0.0f) * (-1e9f);
        mx = fmaxf(mx, row);
    }
    float se = 0.f;
    for (int t = 0; t < Tn; ++t) {
        float row = g2 * (NN[base + t] * rsqrtf(Q[base + t]));
        row = row + (((float)t >= ml) ? 1.0f : 0.0f) * (-1e9f);
        se += expf(row - mx);
    }
    const float lse = mx + logf(se);
    sim_t[i * B + j] = lse / g2 * g3;
    for (int t = 0; t < Tn; ++t) {
        float row = g2 * (NN[base + t] * rsqrtf(Q[base + t]));
        row = row + (((float)t >= ml) ? 1.0f : 0.0f) * (-1e9f);
        PI[base + t] = expf(row - mx) / se;
    }
}

// dS (in place over H) and alpha * dq, elementwise with per-(j, col) scalars
__global__ __launch_bounds__(256) void wl_bwd_cols_kernel(const float* __restrict__ S, const float* __restrict__ A,
                                                          float* __restrict__ HdS, const float* __restrict__ NN,
                                                          const float* __restrict__ Q, const float* __restrict__ PI,
                                                          const float* __restrict__ dsim_t, float* __restrict__ AS,
                                                          int B, int R, int Tn, float g1, float g3) {
    const int ld = B * Tn;
    const long long total = (long long)B * R * ld;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
        const long long row = k / ld;
        const int col = (int)(k - row * ld);
        const int j = (int)(row / R);
        const int i = col / Tn;
        const long long sc = (long long)j * ld + col;
        const float q = Q[sc], nn = NN[sc];
        const float dcos = g3 * dsim_t[i * B + j] * PI[sc];
        const float rq = rsqrtf(q);
        const float dn = dcos * rq;
        const float dq = -0.5f * dcos * nn * rq * rq * rq;
        const float a = A[k];
        const float dal = dn * S[k] + 2.f * dq * HdS[k];
        HdS[k] = a * (dn + g1 * (dal - dn * nn - 2.f * dq * q));
        AS[k] = a * dq;
    }
}

// ---------------------------------------------------------------- symmetric cross-entropy, b x b
// Also produces get_statistics (attention_lib.py:36-43) of both directions: accuracy = mean(argmax == i),
// entropy = -mean(sum p log(p + 1e-8)); stats[0] = 0.5*(acc_rows + acc_cols), stats[1] = 0.5*(ent_r + ent_c).
// ST: the B x B logits are staged in LDS first (B <= 104: all 256 threads load, coalesced) -- only B threads walk the
// rows / columns afterwards, and from global memory every element of their three passes was a dependent L2 round trip
// (22 us for B = 56, eight launches per step).
template <bool ST>
__global__ __launch_bounds__(256) void xent_sym_kernel(const float* __restrict__ Lg, int B, float weight,
                                                       float* __restrict__ loss, float* __restrict__ dL,
                                                       float* __restrict__ stats) {
    extern __shared__ float sm[];
    float* rlse = sm;          // [B]
    float* clse = sm + B;      // [B]
    __shared__ float part[256], pacc[256], pent[256];
    const float* L = Lg;
    if constexpr (ST) {
        float* tile = sm + 2 * B;
        for (int k = threadIdx.x; k < B * B; k += 256) tile[k] = Lg[k];
        __syncthreads();
        L = tile;
    }
    // Q adjacent lanes per row / column i, each over the j = q, q + Q, ... (round 5: one lane per i left 200 of the 256 lanes idle
    // through three 56-step loops of expf -- 16 us for B = 56, seven launches per step); combined by lane shuffles in a fixed order
    const int Q = 4 * B <= 256 ? 4 : (2 * B <= 256 ? 2 : 1);
    const int q = threadIdx.x % Q;
    float acc = 0.f, hits = 0.f, ent = 0.f;
    for (int i = threadIdx.x / Q; i < B; i += 256 / Q) {
        float mr = -INFINITY, mc = -INFINITY;
        int ar = B, ac = B;
        for (int j = q; j < B; j += Q) {
            const float a = L[i * B + j], c = L[j * B + i];
            if (a > mr) { mr = a; ar = j; }         // first maximum, like jnp.argmax
            if (c > mc) { mc = c; ac = j; }
        }
        for (int o = 1; o < Q; o <<= 1) {
            const float omr = __shfl_xor(mr, o), omc = __shfl_xor(mc, o);
            const int oar = __shfl_xor(ar, o), oac = __shfl_xor(ac, o);
            if (omr > mr || (omr == mr && oar < ar)) { mr = omr; ar = oar; }
            if (omc > mc || (omc == mc && oac < ac)) { mc = omc; ac = oac; }
        }
        float sr = 0.f, sc = 0.f;
        for (int j = q; j < B; j += Q) {
            sr += expf(L[i * B + j] - mr);
            sc += expf(L[j * B + i] - mc);
        }
        for (int o = 1; o < Q; o <<= 1) { sr += __shfl_xor(sr, o); sc += __shfl_xor(sc, o); }
        const float rl = mr + logf(sr), cl = mc + logf(sc);
        if (q == 0) {
            rlse[i] = rl;
            clse[i] = cl;
            acc += (rl - L[i * B + i]) + (cl - L[i * B + i]);
        }
        if (stats) {
            if (q == 0) hits += (ar == i ? 0.5f : 0.f) + (ac == i ? 0.5f : 0.f);
            for (int j = q; j < B; j += Q) {
                const float pr = expf(L[i * B + j] - rl), pc = expf(L[j * B + i] - cl);
                ent -= 0.5f * (pr * logf(pr + 1e-8f) + pc * logf(pc + 1e-8f));
            }
        }
    }
    // block sums: lanes of a wave by shuffles, the four waves through LDS (fixed order)
    for (int o = 32; o > 0; o >>= 1) { acc += __shfl_xor(acc, o); hits += __shfl_xor(hits, o); ent += __shfl_xor(ent, o); }
    if ((threadIdx.x & 63) == 0) { part[threadIdx.x >> 6] = acc; pacc[threadIdx.x >> 6] = hits; pent[threadIdx.x >> 6] = ent; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float s = (part[0] + part[1]) + (part[2] + part[3]), h = (pacc[0] + pacc[1]) + (pacc[2] + pacc[3]);
        const float e = (pent[0] + pent[1]) + (pent[2] + pent[3]);
        atomicAdd(loss, weight * s / (float)B);
        if (stats) {
            stats[0] = h / (float)B;
            stats[1] = e / (float)B;
        }
    }
    if (dL) {
        const float sc = weight / (float)B;
        for (int k = threadIdx.x; k < B * B; k += 256) {
            const int i = k / B, j = k - i * B;
            dL[k] = sc * (expf(L[k] - rlse[i]) + expf(L[k] - clse[j]) - (i == j ? 2.f : 0.f));
        }
    }
}

__global__ __launch_bounds__(256) void hinge_kernel(const float* __restrict__ logit, int B, float* d_loss,
                                                    float* g_loss, float* dld, float* dlg) {
    __shared__ float pd[256], pg[256];
    float sd = 0.f, sg = 0.f;
    const float ib = 1.f / (float)B;
    for (int i = threadIdx.x; i < B; i += 256) {
        const float r = logit[i], f = logit[B + i];
        sd += fmaxf(1.f - r, 0.f) + fmaxf(1.f + f, 0.f);
        sg -= f;
        if (dld) {
            dld[i] = (1.f - r > 0.f) ? -ib : 0.f;
            dld[B + i] = (1.f + f > 0.f) ? ib : 0.f;
        }
        if (dlg) {
            dlg[i] = 0.f;
            dlg[B + i] = -ib;
        }
    }
    pd[threadIdx.x] = sd;
    pg[threadIdx.x] = sg;
    __syncthreads();
    if (threadIdx.x == 0) {
        float a = 0.f, b = 0.f;
        for (int k = 0; k < 256; ++k) { a += pd[k]; b += pg[k]; }
        if (d_loss) atomicAdd(d_loss, a * ib);
        if (g_loss) atomicAdd(g_loss, b * ib);
    }
}

// ------------------------------------------------------------------------------ projection head
__global__ __launch_bounds__(256) void proj_fwd_kernel(const float* __restrict__ pool, const float* __restrict__ w,
                                                       const float* __restrict__ inv_sigma,
                                                       const float* __restrict__ bias, const float* __restrict__ emb,
                                                       float* __restrict__ out, int N2, int B, int C) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N2) return;
    const float is = inv_sigma ? *inv_sigma : 1.f;
    const float* pr = pool + (long long)n * C;
    const float* er = emb + (long long)(n % B) * C;
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += pr[c] * (w[c] * is + er[c]);
    s = wave_sum(s);
    if (lane == 0) out[n] = s + (bias ? bias[0] : 0.f);
}

__global__ __launch_bounds__(256) void proj_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ pool,
                                                       const float* __restrict__ w,
                                                       const float* __restrict__ inv_sigma,
                                                       const float* __restrict__ emb, float* __restrict__ dpool,
                                                       float* __restrict__ demb, int N2, int B, int C, int accum) {
    const float is = inv_sigma ? *inv_sigma : 1.f;
    const long long total = (long long)N2 * C;
    for (long long k = (long long)blockIdx.x * 256 + threadIdx.x; k < total; k += (long long)gridDim.x * 256) {
        const int n = (int)(k / C), c = (int)(k - (long long)n * C);
        const float v = dout[n] * (w[c] * is + emb[(long long)(n % B) * C + c]);
        dpool[k] = accum ? dpool[k] + v : v;
        if (demb && n < B) {
            float s = 0.f;
            for (int m = n; m < N2; m += B) s += dout[m] * pool[(long long)m * C + c];
            const long long ke = (long long)n * C + c;
            demb[ke] = accum ? demb[ke] + s : s;
        }
    }
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int xmc_l2norm_rows_fwd(const void* x, float* y, float* inv, int64_t rows, int32_t cols,
                                   int32_t dtype_in, void* stream) {
    XMC_REQUIRE(x && y && inv && rows > 0 && cols > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype_in == XMC_BF16)
        hipLaunchKernelGGL((l2norm_fwd_kernel<bf16_t>), grid, block, 0, s, static_cast<const bf16_t*>(x), y, inv,
                           (long long)rows, cols);
    else if (dtype_in == XMC_F32)
        hipLaunchKernelGGL((l2norm_fwd_kernel<float>), grid, block, 0, s, static_cast<const float*>(x), y, inv,
                           (long long)rows, cols);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_l2norm_rows_bwd(const float* dy, const float* y, const float* inv, void* dx, int64_t rows,
                                   int32_t cols, int32_t dtype_out, void* stream) {
    XMC_REQUIRE(dy && y && inv && dx && rows > 0 && cols > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((rows + 3) / 4)), block(256);
    if (dtype_out == XMC_BF16)
        hipLaunchKernelGGL((l2norm_bwd_kernel<bf16_t>), grid, block, 0, s, dy, y, inv, static_cast<bf16_t*>(dx),
                           (long long)rows, cols);
    else if (dtype_out == XMC_F32)
        hipLaunchKernelGGL((l2norm_bwd_kernel<float>), grid, block, 0, s, dy, y, inv, static_cast<float*>(dx),
                           (long long)rows, cols);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_attn_g_fwd(const void* region, const float* words_n, const float* max_len, void* ctx,
                              float* attn, float* rinv, int32_t b, int32_t r, int32_t t, int32_t e, float gamma,
                              int32_t dtype, void* stream) {
    XMC_REQUIRE(region && words_n && max_len && ctx && attn && rinv);
    XMC_REQUIRE(b > 0 && r > 0 && t > 0 && t <= 64 && e > 0 && e <= 64 * EMAX);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)(((long long)b * r + 3) / 4)), block(256);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((attn_g_fwd_kernel<bf16_t>), grid, block, 0, s, static_cast<const bf16_t*>(region),
                           words_n, max_len, static_cast<bf16_t*>(ctx), attn, rinv, b, r, t, e, gamma);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((attn_g_fwd_kernel<float>), grid, block, 0, s, static_cast<const float*>(region), words_n,
                           max_len, static_cast<float*>(ctx), attn, rinv, b, r, t, e, gamma);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_attn_g_bwd(const void* dctx, const void* region, const float* words_n, const float* attn,
                              const float* rinv, void* dregion, int32_t b, int32_t r, int32_t t, int32_t e,
                              float gamma, int32_t dtype, void* stream) {
    XMC_REQUIRE(dctx && region && words_n && attn && rinv && dregion);
    XMC_REQUIRE(b > 0 && r > 0 && t > 0 && t <= 64 && e > 0 && e <= 64 * EMAX);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)(((long long)b * r + 3) / 4)), block(256);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((attn_g_bwd_kernel<bf16_t>), grid, block, 0, s, static_cast<const bf16_t*>(dctx),
                           static_cast<const bf16_t*>(region), words_n, attn, rinv, static_cast<bf16_t*>(dregion), b,
                           r, t, e, gamma);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((attn_g_bwd_kernel<float>), grid, block, 0, s, static_cast<const float*>(dctx),
                           static_cast<const float*>(region), words_n, attn, rinv, static_cast<float*>(dregion), b, r,
                           t, e, gamma);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_internal_optin_losses(void) {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(wl_softmax_kernel)}, 160 * 1024) ? XMC_OK : XMC_EINVAL;
}

extern "C" int xmc_wl_softmax(const float* sm, const float* max_len, float* alpha, float* nn, int32_t b, int32_t r,
                              int32_t t, float gamma1, void* stream) {
    XMC_REQUIRE(sm && max_len && alpha && nn && b > 0 && r > 0 && t > 0);
    const size_t lds = sizeof(float) * ((size_t)r * 64 + 512);
    XMC_REQUIRE(lds <= 160 * 1024);
    dim3 grid((unsigned)((b * t + 63) / 64), (unsigned)b), block(256);
    if (lds > 64 * 1024 && xmc_internal_optin_losses() != XMC_OK) return XMC_EINVAL;
    hipLaunchKernelGGL(wl_softmax_kernel, grid, block, lds, static_cast<hipStream_t>(stream), sm, max_len, alpha, nn,
                       b, r, t, gamma1);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_qdot(const float* alpha, const float* h, float* q, int32_t b, int32_t r, int32_t t,
                           void* stream) {
    XMC_REQUIRE(alpha && h && q && b > 0 && r > 0 && t > 0);
    dim3 grid((unsigned)((b * t + 63) / 64), (unsigned)b), block(256);
    hipLaunchKernelGGL(wl_qdot_kernel, grid, block, 0, static_cast<hipStream_t>(stream), alpha, h, q, b, r, t);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_rows(const float* nn, const float* q, const float* max_len, float* sim_t, float* pi,
                           int32_t b, int32_t t, float gamma2, float gamma3, void* stream) {
    XMC_REQUIRE(nn && q && max_len && sim_t && pi && b > 0 && t > 0);
    hipLaunchKernelGGL(wl_rows_kernel, dim3((unsigned)((b * b + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), nn, q, max_len, sim_t, pi, b, t, gamma2, gamma3);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_wl_bwd_cols(const float* sm, const float* alpha, float* h_ds, const float* nn, const float* q,
                               const float* pi, const float* dsim_t, float* alpha_scaled, int32_t b, int32_t r,
                               int32_t t, float gamma1, float gamma3, void* stream) {
    XMC_REQUIRE(sm && alpha && h_ds && nn && q && pi && dsim_t && alpha_scaled && b > 0 && r > 0 && t > 0);
    const long long total = (long long)b * r * b * t;
    hipLaunchKernelGGL(wl_bwd_cols_kernel, dim3(grid_for(total)), dim3(256), 0, static_cast<hipStream_t>(stream), sm,
                       alpha, h_ds, nn, q, pi, dsim_t, alpha_scaled, b, r, t, gamma1, gamma3);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_xent_sym(const float* logits, int32_t b, float weight, float* loss, float* dlogits,
                            float* stats, void* stream) {
    XMC_REQUIRE(logits && loss && b > 0 && b <= 4096);
    if (b <= 104)
        hipLaunchKernelGGL(xent_sym_kernel<true>, dim3(1), dim3(256), sizeof(float) * (2 * b + b * b), static_cast<hipStream_t>(stream),
                           logits, b, weight, loss, dlogits, stats);
    else
        hipLaunchKernelGGL(xent_sym_kernel<false>, dim3(1), dim3(256), sizeof(float) * 2 * b, static_cast<hipStream_t>(stream),
                           logits, b, weight, loss, dlogits, stats);
    XMC_LAUNCH_RET();
}

// out[0..3] = {d_loss, g_loss, c_loss_d, c_loss_g} of xmc_gan.py:58-71,146-154 from the five contrastive terms (LOSS_SLOTS order:
// fake word, real word, fake sentence, real sentence, image) and the two hinge terms: one thread instead of ten scalar adds
__global__ void loss_assemble_kernel(const float* __restrict__ lv, const float* __restrict__ hinge, float* __restrict__ out) {
    const float cd = lv[1] + lv[3];
    const float cg = (lv[0] + lv[2]) + lv[4];
    out[0] = hinge[0] + cd; out[1] = hinge[1] + cg; out[2] = cd; out[3] = cg;
}

extern "C" int xmc_loss_assemble(const float* loss_vec, const float* hinge, float* out, void* stream) {
    XMC_REQUIRE(loss_vec && hinge && out);
    hipLaunchKernelGGL(loss_assemble_kernel, dim3(1), dim3(1), 0, static_cast<hipStream_t>(stream), loss_vec, hinge, out);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cl_logits(const float* a, const float* b, float* logits, float* ainv, float* binv, int32_t n, int32_t d,
                             float inv_temperature, void* stream) {
    XMC_REQUIRE(a && b && logits && ainv && binv && n > 0 && d > 0 && d <= 2048);       // as xmc_cl_bwd; one row = (d rounded up to 512) floats of LDS
    const unsigned js = n >= 32 ? 4u : 1u;                        // slices of the j range: 4 n workgroups fill the chip's latency slots
    hipLaunchKernelGGL(cl_logits_kernel, dim3((unsigned)n, js), dim3(256), sizeof(float) * ((d + 511) & ~511), static_cast<hipStream_t>(stream),
                       a, b, logits, ainv, binv, n, d, inv_temperature);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cl_bwd(const float* dlogits, const float* x, const float* y, const float* xinv, const float* yinv, float* out,
                          int32_t n, int32_t d, float inv_temperature, int32_t trans, int32_t accumulate, void* stream) {
    XMC_REQUIRE(dlogits && x && y && xinv && yinv && out && n > 0 && d > 0 && d <= 2048 && n <= 4096);
    hipLaunchKernelGGL(cl_bwd_kernel, dim3((unsigned)n), dim3(256), sizeof(float) * n, static_cast<hipStream_t>(stream), dlogits, x, y,
                       xinv, yinv, out, n, d, inv_temperature, trans, accumulate);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_hinge(const float* logit, int32_t b, float* d_loss, float* g_loss, float* dlogit_d,
                         float* dlogit_g, void* stream) {
    XMC_REQUIRE(logit && b > 0);
    hipLaunchKernelGGL(hinge_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), logit, b, d_loss,
                       g_loss, dlogit_d, dlogit_g);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_proj_head_fwd(const float* pool, const float* w, const float* inv_sigma, const float* bias,
                                 const float* emb, float* out, int32_t n2, int32_t b, int32_t c, void* stream) {
    XMC_REQUIRE(pool && w && emb && out && n2 > 0 && b > 0 && c > 0);
    hipLaunchKernelGGL(proj_fwd_kernel, dim3((unsigned)((n2 + 3) / 4)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), pool, w, inv_sigma, bias, emb, out, n2, b, c);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_proj_head_bwd(const float* dout, const float* pool, const float* w, const float* inv_sigma,
                                 const float* emb, float* dpool, float* demb, int32_t n2, int32_t b, int32_t c,
                                 int32_t accumulate, void* stream) {
    XMC_REQUIRE(dout && pool && w && emb && dpool && n2 > 0 && b > 0 && c > 0);
    hipLaunchKernelGGL(proj_bwd_kernel, dim3(grid_for((long long)n2 * c)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), dout, pool, w, inv_sigma, emb, dpool, demb, n2, b, c,
                       accumulate);
    XMC_LAUNCH_RET();
}
