// Small-batch dense layers (flax nn.Dense / SpectralDense, xmcgan/libml/layers.py:49-113; xmcgan/nets/xmc_net.py:213-216,100): at the
// step's batch sizes (M = 56 rows) these are a handful of sub-10-microsecond products -- sentence -> z_dim, z -> 4 x 4 x 1536, the four
// global conditional-BatchNorm projections, the discriminator's sentence projection.  Rounds 1-4 ran each as bias broadcast (a torch
// copy) + strided GEMM (+ split-K reduction), and each gradient as GEMM (+ reduction) + a row reduction for the bias: ~70 launches
// per step at the launch floor.  Here: ONE launch forward (bias, optional device scale, optional bf16 rounding of the operands =
// what nn.Dense(dtype=bfloat16) multiplies) and ONE launch for the kernel + bias gradient.  float32 accumulation, fixed summation
// order (k ascending / m ascending): bit-reproducible, no atomics.
#include "common.h"

namespace {

constexpr int DM = 64;           // rows (batch) per launch
constexpr int DKC = 128;         // k per LDS chunk

__device__ __forceinline__ float rnd(float v, int bf) { return bf ? bf2f(f2bf(v)) : v; }

// y[m][n] = bias[n] + alpha * sum_k x[m][k] w[k][n]     x (M, K) row pitch ldx, w (K, N) row-major, y (M, N) row pitch ldy
// workgroup = 64 columns; thread = (column, row group of 16)
__global__ __launch_bounds__(256) void dense_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                                        const float* __restrict__ alpha_dev, float* __restrict__ y, int M, int K, int N,
                                                        int ldx, int ldy, int bf) {
    __shared__ float xs[DM][DKC + 1];
    const int col = threadIdx.x & 63, rg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col;
    const bool live = n < N;
    float acc[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int k0 = 0; k0 < K; k0 += DKC) {
        const int kc = min(DKC, K - k0);
        __syncthreads();
        for (int i = threadIdx.x; i < DM * DKC; i += 256) {
            const int m = i / DKC, k = i - m * DKC;
            xs[m][k] = (m < M && k < kc) ? rnd(x[(size_t)m * ldx + k0 + k], bf) : 0.f;
        }
        __syncthreads();
        const float* __restrict__ wp = w + (size_t)k0 * N + (live ? n : 0);
        int k = 0;
        for (; k + 3 < kc; k += 4) {                 // four weight loads in flight
            float wv[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) wv[u] = rnd(wp[(size_t)(k + u) * N], bf);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] += xs[rg * 16 + r][k + u] * wv[u];
        }
        for (; k < kc; ++k) {
            const float wv = rnd(wp[(size_t)k * N], bf);
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] += xs[rg * 16 + r][k] * wv;
        }
    }
    if (!live) return;
    const float a = alpha_dev ? *alpha_dev : 1.f;
    const float bv = bias ? bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = rg * 16 + r;
        if (m < M) y[(size_t)m * ldy + n] = bv + a * acc[r];
    }
}

// dw[k][n] (=, +=) sum_m x[m][k] dy[m][n];  db[n] (=, +=) sum_m dy[m][n]     workgroup = 64 columns x 16 k; thread = (column, 4 k)
__global__ __launch_bounds__(256) void dense_bwd_w_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ dw,
                                                          float* __restrict__ db, int M, int K, int N, int ldx, int lddy, int bf,
                                                          int accumulate) {
    __shared__ float xs[DM][17];
    const int col = threadIdx.x & 63, kg = threadIdx.x >> 6;
    const int n = blockIdx.x * 64 + col, k0 = blockIdx.y * 16;
    for (int i = threadIdx.x; i < DM * 16; i += 256) {
        const int m = i >> 4, k = i & 15;
        xs[m][k] = (m < M && k0 + k < K) ? rnd(x[(size_t)m * ldx + k0 + k], bf) : 0.f;
    }
    __syncthreads();
    if (n >= N) return;
    float acc[4] = {0.f, 0.f, 0.f, 0.f}, bs = 0.f;
    int m = 0;
    for (; m + 3 < M; m += 4) {
        float g[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) g[u] = dy[(size_t)(m + u) * lddy + n];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            bs += g[u];
            const float gr = rnd(g[u], bf);
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] += xs[m + u][kg * 4 + q] * gr;
        }
    }
    for (; m < M; ++m) {
        const float g = dy[(size_t)m * lddy + n];
        bs += g;
        const float gr = rnd(g, bf);
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[q] += xs[m][kg * 4 + q] * gr;
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int k = k0 + kg * 4 + q;
        if (k < K) {
            float* o = dw + (size_t)k * N + n;
            *o = accumulate ? *o + acc[q] : acc[q];
        }
    }
    if (db && blockIdx.y == 0 && kg == 0) db[n] = accumulate ? db[n] + bs : bs;
}

}  // namespace

extern "C" int xmc_dense_fwd(const float* x, const float* w, const float* bias, const float* alpha_dev, float* y, int32_t m, int32_t k,
                             int32_t n, int32_t ldx, int32_t ldy, int32_t bf16_operands, void* stream) {
    XMC_REQUIRE(x && w && y && m > 0 && m <= DM && k > 0 && n > 0 && ldx >= k && ldy >= n);
    hipLaunchKernelGGL(dense_fwd_kernel, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, bias, alpha_dev,
                       y, m, k, n, ldx, ldy, bf16_operands);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_dense_bwd_w(const float* x, const float* dy, float* dw, float* db, int32_t m, int32_t k, int32_t n, int32_t ldx,
                               int32_t lddy, int32_t bf16_operands, int32_t accumulate, void* stream) {
    XMC_REQUIRE(x && dy && dw && m > 0 && m <= DM && k > 0 && n > 0 && ldx >= k && lddy >= n);
    hipLaunchKernelGGL(dense_bwd_w_kernel, dim3((unsigned)((n + 63) / 64), (unsigned)((k + 15) / 16)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, dy, dw, db, m, k, n, ldx, lddy, bf16_operands, accumulate);
    XMC_LAUNCH_RET();
}
