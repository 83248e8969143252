// Spectral normalisation (one power-iteration step per forward), weight preparation for the
// implicit-GEMM convolutions, and the gradient through sigma.  All HBM-bound matrix-vector
// passes over float32 master weights: W is read with consecutive lanes on consecutive columns
// (coalesced), row dots use wave64 shuffles, column sums use one atomic per column per block.
#include "common.h"

namespace {

// y[c] += sum_{r in block rows} x[r] * W[r][c]
__global__ __launch_bounds__(256) void matvec_cols_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                          float* __restrict__ y, int rows, int cols,
                                                          int rows_per_block) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int rb = blockIdx.y * rows_per_block;
    const int re = min(rows, rb + rows_per_block);
    if (c >= cols) return;
    float s = 0.f;
    for (int r = rb; r < re; ++r) s += x[r] * W[(long long)r * cols + c];
    atomicAdd(&y[c], s);
}

// y[r] = sum_c W[r][c] * x[c]   (one wave per row)
__global__ __launch_bounds__(256) void matvec_rows_kernel(const float* __restrict__ W, const float* __restrict__ x,
                                                          float* __restrict__ y, int rows, int cols) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const float* wr = W + (long long)r * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += wr[c] * x[c];
    s = wave_sum(s);
    if (lane == 0) y[r] = s;
}

__device__ __forceinline__ float block_sum_1024(float v, float* red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) red[w] = v;
    __syncthreads();
    float t = 0.f;
    for (int k = 0; k < (int)(blockDim.x >> 6); ++k) t += red[k];
    __syncthreads();
    return t;
}

// in place: y <- y * rsqrt(sum y^2 + eps)          (layers._l2_normalize, layers.py:31-46)
__global__ __launch_bounds__(1024) void sn_normalize_kernel(float* __restrict__ y, int n, float eps) {
    __shared__ float red[16];
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss += y[i] * y[i];
    ss = block_sum_1024(ss, red);
    const float inv = rsqrtf(ss + eps);
    for (int i = threadIdx.x; i < n; i += blockDim.x) y[i] *= inv;
}

// u_new = u_raw * rsqrt(|u_raw|^2 + eps); sigma = u_raw . u_new; scal = {sigma, 1/(sigma+eps)}
__global__ __launch_bounds__(1024) void sn_finalize_kernel(const float* __restrict__ u_raw, float* __restrict__ u_new,
                                                           float* __restrict__ scal, int n, float eps) {
    __shared__ float red[16];
    float ss = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) ss += u_raw[i] * u_raw[i];
    ss = block_sum_1024(ss, red);
    const float inv = rsqrtf(ss + eps);
    float sg = 0.f;
    for (int i = threadIdx.x; i < n; i += blockDim.x) {
        const float u = u_raw[i] * inv;
        u_new[i] = u;
        sg += u_raw[i] * u;
    }
    sg = block_sum_1024(sg, red);
    if (threadIdx.x == 0) {
        scal[0] = sg;
        scal[1] = 1.f / (sg + eps);
    }
}

// master [cout][taps][cin] -> fwd [cout][taps][cin], dgrad [cin][taps-1-tap][cout]; 32x32 LDS
// transpose per tap so both the read and the two writes are coalesced.  packed bit 0 / 1: write the
// forward / dgrad copy in MFMA-fragment order instead (common.h: packed_w_index; rows padded to 32 with zeros).
template <typename T>
__global__ __launch_bounds__(256) void prep_weight_kernel(const float* __restrict__ w,
                                                          const float* __restrict__ inv_sigma, T* __restrict__ wf,
                                                          T* __restrict__ wd, int cout, int taps, int cin, int packed) {
    __shared__ float tile[32][33];
    prep_weight_tile<T>(tile, w, inv_sigma ? *inv_sigma : 1.f, wf, wd, cout, taps, cin, blockIdx.z, blockIdx.y * 32,
                        blockIdx.x * 32, packed);
}

__global__ __launch_bounds__(256) void dot_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                  float* __restrict__ out, long long n) {
    __shared__ float red[4];
    float s = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        s += a[i] * b[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, red[0] + red[1] + red[2] + red[3]);
}

// g <- (g - (dot * inv_s) * u[.] * v[.]) * inv_s
__global__ __launch_bounds__(256) void sn_grad_fix_kernel(float* __restrict__ g, const float* __restrict__ u,
                                                          const float* __restrict__ v,
                                                          const float* __restrict__ scal,
                                                          const float* __restrict__ dot, int rows, int cols,
                                                          int u_axis) {
    const float is = scal[1];
    const float k = dot[0] * is;
    const long long n = (long long)rows * cols;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int r = (int)(i / cols), c = (int)(i - (long long)r * cols);
        const float uv = u_axis == 0 ? u[r] * v[c] : u[c] * v[r];
        g[i] = (g[i] - k * uv) * is;
    }
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 4096) b = 4096;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int xmc_spectral_power_iter(const float* w, const float* u0, float* u_new, float* v, float* scal,
                                       float* tmp, int32_t rows, int32_t cols, int32_t u_axis, float eps,
                                       void* stream) {
    XMC_REQUIRE(w && u0 && u_new && v && scal && tmp && rows > 0 && cols > 0);
    XMC_REQUIRE(u_axis == 0 || u_axis == 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    auto mv_cols = [&](const float* x, float* y) -> int {
        hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * cols, s);
        if (e != hipSuccess) return xmc_hip_err(e);
        int rpb = (rows + 63) / 64;
        if (rpb < 16) rpb = 16;
        dim3 grid((unsigned)((cols + 255) / 256), (unsigned)((rows + rpb - 1) / rpb));
        hipLaunchKernelGGL(matvec_cols_kernel, grid, dim3(256), 0, s, w, x, y, rows, cols, rpb);
        return xmc_hip_err(hipGetLastError());
    };
    auto mv_rows = [&](const float* x, float* y) -> int {
        hipLaunchKernelGGL(matvec_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, w, x, y, rows, cols);
        return xmc_hip_err(hipGetLastError());
    };
    int rc;
    if (u_axis == 0) {                       // u over rows (cout), v over cols (K)
        if ((rc = mv_cols(u0, v)) != XMC_OK) return rc;
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, v, cols, eps);
        if ((rc = mv_rows(v, tmp)) != XMC_OK) return rc;
        hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(1024), 0, s, tmp, u_new, scal, rows, eps);
    } else {                                 // u over cols (out), v over rows (in)
        if ((rc = mv_rows(u0, v)) != XMC_OK) return rc;
        hipLaunchKernelGGL(sn_normalize_kernel, dim3(1), dim3(1024), 0, s, v, rows, eps);
        if ((rc = mv_cols(v, tmp)) != XMC_OK) return rc;
        hipLaunchKernelGGL(sn_finalize_kernel, dim3(1), dim3(1024), 0, s, tmp, u_new, scal, cols, eps);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_prep_conv_weight(const float* w, const float* inv_sigma, void* w_fwd, void* w_dgrad,
                                    int32_t cout, int32_t taps, int32_t cin, int32_t dtype, int32_t packed,
                                    void* stream) {
    XMC_REQUIRE(w && (w_fwd || w_dgrad) && cout > 0 && taps > 0 && taps < 65536 && cin > 0);
    XMC_REQUIRE(!(packed & 1) || (dtype == XMC_BF16 && cin % 32 == 0));
    XMC_REQUIRE(!(packed & 2) || (dtype == XMC_BF16 && cout % 32 == 0));
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid((unsigned)((cin + 31) / 32), (unsigned)((cout + 31) / 32), (unsigned)taps), block(256);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((prep_weight_kernel<bf16_t>), grid, block, 0, s, w, inv_sigma,
                           static_cast<bf16_t*>(w_fwd), static_cast<bf16_t*>(w_dgrad), cout, taps, cin, packed);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((prep_weight_kernel<float>), grid, block, 0, s, w, inv_sigma, static_cast<float*>(w_fwd),
                           static_cast<float*>(w_dgrad), cout, taps, cin, 0);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_spectral_grad_fix(float* g, const float* w, const float* u, const float* v, const float* scal,
                                     float* tmp, int32_t rows, int32_t cols, int32_t u_axis, void* stream) {
    XMC_REQUIRE(g && w && u && v && scal && tmp && rows > 0 && cols > 0);
    XMC_REQUIRE(u_axis == 0 || u_axis == 1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(tmp, 0, sizeof(float), s);
    if (e != hipSuccess) return xmc_hip_err(e);
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(dot_kernel, dim3(grid_for(n)), dim3(256), 0, s, g, w, tmp, n);
    hipLaunchKernelGGL(sn_grad_fix_kernel, dim3(grid_for(n)), dim3(256), 0, s, g, u, v, scal, tmp, rows, cols, u_axis);
    XMC_LAUNCH_RET();
}
