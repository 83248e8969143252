// Streaming pointwise / resampling kernels (HBM-bound; 16-byte vectors, grid-stride) and the
// fused Adam(+EMA) arena update.
#include "common.h"

namespace {

template <typename T, int VE> struct Acc {
    static __device__ __forceinline__ void load(const T* p, float* f) { Vec<T> v; v.load(p); v.get(f); }
    static __device__ __forceinline__ void store(T* p, const float* f) { Vec<T> v; v.set(f); v.store(p); }
};
template <typename T> struct Acc<T, 1> {
    static __device__ __forceinline__ void load(const T* p, float* f) { f[0] = to_f<T>(*p); }
    static __device__ __forceinline__ void store(T* p, const float* f) { *p = from_f<T>(f[0]); }
};

// y[n][oy][ox][c] = scale * sum_{2x2} x + res;  XR: also xr = max(x, 0) at full resolution -- the pass already holds every
// element of x in registers, and the discriminator block that pools its input for the shortcut (nets/common.py:74-78 of the
// reference) reads relu(x) in its first convolution AND in that convolution's weight gradient, where an in-LDS ReLU pass
// over the DMA-staged patch costs 20-28 % of the kernel (tools/relu_cost.py)
template <typename T, int VE, bool XR>
__global__ __launch_bounds__(256) void pool2_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                    T* __restrict__ y, T* __restrict__ xr, int H, int W, int C, float scale,
                                                    long long nvec) {
    const int CV = C / VE, Wo = W >> 1, Ho = H >> 1;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long opix = v / CV;
        const int c = (int)(v - opix * CV) * VE;
        const int ox = (int)(opix % Wo);
        const long long t = opix / Wo;
        const int oy = (int)(t % Ho);
        const long long n = t / Ho;
        const long long o00 = (((n * H + 2 * oy) * W) + 2 * ox) * C + c, o10 = o00 + (long long)W * C;
        float a[VE], b[VE], d[VE], e2[VE], s[VE];
        Acc<T, VE>::load(x + o00, a);
        Acc<T, VE>::load(x + o00 + C, b);
        Acc<T, VE>::load(x + o10, d);
        Acc<T, VE>::load(x + o10 + C, e2);
#pragma unroll
        for (int e = 0; e < VE; ++e) s[e] = ((a[e] + b[e]) + d[e] + e2[e]) * scale;
        if (res) {
            float r[VE];
            Acc<T, VE>::load(res + opix * C + c, r);
#pragma unroll
            for (int e = 0; e < VE; ++e) s[e] += r[e];
        }
        Acc<T, VE>::store(y + opix * C + c, s);
        if constexpr (XR) {
#pragma unroll
            for (int e = 0; e < VE; ++e) { a[e] = fmaxf(a[e], 0.f); b[e] = fmaxf(b[e], 0.f); d[e] = fmaxf(d[e], 0.f); e2[e] = fmaxf(e2[e], 0.f); }
            Acc<T, VE>::store(xr + o00, a);
            Acc<T, VE>::store(xr + o00 + C, b);
            Acc<T, VE>::store(xr + o10, d);
            Acc<T, VE>::store(xr + o10 + C, e2);
        }
    }
}

// out[n][y][x][tap * C + c] = in[n][y + s*dy(tap)][x + s*dx(tap)][c]  (0 outside the image, 0 for k >= taps*C),
// 32 output channels, C <= 3: turns a 3x3 (or 1x1) convolution on an RGB-like tensor into a 1x1 convolution
// on a 32-channel tensor, so it runs on the MFMA patch kernels instead of the scalar-gather path.
// s = +1 gathers the forward im2col; s = -1 gathers dy(p - d(tap)) for the weight gradient of a Cout = 3 conv.
// One thread writes Vec<T>::N consecutive k of one pixel (a 16-byte store: the 32-channel output is the traffic, the
// RGB-like input is cache-resident).
template <typename T>
__global__ __launch_bounds__(256) void expand_taps_kernel(const T* __restrict__ x, T* __restrict__ y, int H, int W,
                                                          int C, int ks, int sign, long long npix) {
    constexpr int V = Vec<T>::N, G = 32 / V;                       // vectors per pixel
    const int taps = ks * ks, half = ks >> 1, kmax = taps * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < npix * G; i += (long long)gridDim.x * 256) {
        const long long pix = i / G;
        const int k0 = (int)(i - pix * G) * V;
        const int px = (int)(pix % W);
        const long long t = pix / W;
        const int py = (int)(t % H);
        const long long n = t / H;
        float v[V];
        int tap = k0 / C, c = k0 - tap * C;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            // branch-free: an out-of-range element reads x[0] and is replaced by zero afterwards (a conditional load per
            // element compiled to V serial (branch, load, s_waitcnt vmcnt(0)) sequences: V exposed memory latencies)
            const int ty = tap / ks, tx = tap - ty * ks;
            const int iy = py + sign * (ty - half), ix = px + sign * (tx - half);
            const bool ok = k0 + e < kmax && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            const long long idx = ok ? ((n * H + iy) * W + ix) * C + c : 0;
            const float val = to_f<T>(x[idx]);
            v[e] = ok ? val : 0.f;
            if (++c == C) { c = 0; ++tap; }
        }
        Vec<T> o;
        o.set(v);
        o.store(y + pix * 32 + k0);
    }
}

// The 3x3 expansion of a 3-channel bf16 tensor, one workgroup per image row: the three input rows are staged in LDS with
// coalesced 16-byte loads (8 zero elements either side = the left / right padding, zero rows outside the image), every
// thread then assembles 16-byte output vectors from 2-byte LDS reads.  (The generic kernel: eight 2-byte global loads per
// vector, 50 us for 117 MB.)  W * 3 % 8 == 0, W <= 512, 16-byte aligned tensors.
constexpr int ET_W_MAX = 512, ET_ROW = 8 + ET_W_MAX * 3 + 8;
__global__ __launch_bounds__(256) void expand_taps3_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ y, int H, int W, int sign) {
    __shared__ __attribute__((aligned(16))) bf16_t rows[3][ET_ROW];
    const int py = blockIdx.x % H, n = blockIdx.x / H;
    const int nvec = (8 + W * 3 + 8) / 8, vvec = W * 3 / 8;
    for (int it = threadIdx.x; it < 3 * nvec; it += 256) {
        const int r = it / nvec, v = it - r * nvec;
        const int iy = py + sign * (r - 1);
        uint4 q = make_uint4(0, 0, 0, 0);
        if ((unsigned)iy < (unsigned)H && v >= 1 && v - 1 < vvec)
            q = *reinterpret_cast<const uint4*>(x + ((long long)n * H + iy) * W * 3 + (v - 1) * 8);
        *reinterpret_cast<uint4*>(&rows[r][v * 8]) = q;
    }
    __syncthreads();
    bf16_t* out = y + ((long long)n * H + py) * W * 32;
    for (int it = threadIdx.x; it < W * 4; it += 256) {
        const int px = it >> 2, k0 = (it & 3) * 8;
        unsigned short h[8];
        int tap = k0 / 3, c = k0 - tap * 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int ty = tap / 3, tx = tap - ty * 3;
            h[e] = tap < 9 ? rows[ty][8 + (px + sign * (tx - 1)) * 3 + c] : (unsigned short)0;
            if (++c == 3) { c = 0; ++tap; }
        }
        *reinterpret_cast<uint4*>(out + (long long)px * 32 + k0) =
            make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bcast_relu_bwd_kernel(const float* __restrict__ dpool,
                                                             const T* __restrict__ x, T* __restrict__ dx,
                                                             long long R, long long C, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const long long a = i / (R * C), c = i % C;
        dx[i] = from_f<T>(to_f<T>(x[i]) > 0.f ? dpool[a * C + c] : 0.f);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void tanh_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = from_f<T>((tanhf(to_f<T>(x[i])) + 1.f) * 0.5f);
}
template <typename T>
__global__ __launch_bounds__(256) void tanh_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                       T* __restrict__ dx, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float t = 2.f * to_f<T>(y[i]) - 1.f;
        dx[i] = from_f<T>(to_f<T>(dy[i]) * 0.5f * (1.f - t * t));
    }
}

template <typename TI, typename TO>
__global__ __launch_bounds__(256) void cast_kernel(const TI* __restrict__ x, TO* __restrict__ y, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        y[i] = from_f<TO>(to_f<TI>(x[i]));
}

template <typename T>
__global__ __launch_bounds__(256) void add_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                  T* __restrict__ o, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        o[i] = from_f<T>(to_f<T>(a[i]) + to_f<T>(b[i]));
}

// flax.optim.Adam + EMA over a flat arena, float4 per lane.
// Round 4: FIX -- the gradient through sigma of the spectrally-normalised tensors (xmcgan/libml/layers.py:217-219:
// G <- (G - k u (x) v) / (sigma + eps), k = <G, W> / (sigma + eps)) is applied to the gradient on its way into the moments
// instead of by a separate read-modify-write pass over the gradient arena (sn_fix_kernel: 2 reads + 1 write of 352 MB per
// half step).  `map` has one int16 per 64 arena elements (every tensor starts 64-aligned: ParamArena.ALIGN): the index of the
// spectral table entry that owns them, or -1.  zero_g: the consumed gradient is overwritten with zeros (the next half step
// accumulates into a clean arena without a separate fill).
template <bool FIX>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, float* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v,
                                                   float* __restrict__ ema, long long n, float lr, float b1,
                                                   float b2, float eps, float ic1, float ic2, float gs, float d,
                                                   const float* __restrict__ corr, int zero_g, const short* __restrict__ map,
                                                   const xmc_sn_entry* __restrict__ tab, const float* __restrict__ kvec,
                                                   const float* __restrict__ scal, const float* __restrict__ uvec,
                                                   const float* __restrict__ vvec) {
    if (corr) { ic1 = corr[1]; ic2 = corr[2]; }      // device-side step counter (hipGraph replay): see adam_advance_kernel
    const long long n4 = n >> 2;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 pp = reinterpret_cast<float4*>(p)[i];
        float4 gg = reinterpret_cast<const float4*>(g)[i];
        float4 mm = reinterpret_cast<float4*>(m)[i];
        float4 vv = reinterpret_cast<float4*>(v)[i];
        float* pf = reinterpret_cast<float*>(&pp);
        float* gf = reinterpret_cast<float*>(&gg);
        float* mf = reinterpret_cast<float*>(&mm);
        float* vf = reinterpret_cast<float*>(&vv);
        bool fixed = false;
        if constexpr (!FIX) {
            if (map && map[i >> 4] == -2) continue;  // a tensor the fused optimiser + preparation kernel updates (adam_wprep_kernel)
        }
        if constexpr (FIX) {
            const int ent = map[i >> 4];
            if (ent == -2) continue;
            fixed = ent >= 0;
            if (ent >= 0) {
                const xmc_sn_entry e = tab[ent];
                const float k = kvec[ent], is = scal[2 * ent + 1];
                const unsigned t = (unsigned)((i << 2) - e.w_off), cols = (unsigned)e.cols;     // <= 21 M elements per tensor
                const float* uu = uvec + e.u_off;
                const float* vw = vvec + e.v_off;
                if ((cols & 3u) == 0 && t >= (unsigned)e.rows * cols) {
                    // the tensor's 64-element alignment padding (rows * cols % 64 != 0): it carries no gradient and must not
                    // index u / v past the entry's slices (round-4 advisor finding; no current shape has padding)
                } else if ((cols & 3u) == 0) {       // a float4 never straddles a row
                    const unsigned r = t / cols, c = t - r * cols;
                    if (e.u_axis == 0) {
                        const float ur = uu[r] * k;
                        const float4 v4 = *reinterpret_cast<const float4*>(vw + c);
                        gf[0] = (gf[0] - ur * v4.x) * is; gf[1] = (gf[1] - ur * v4.y) * is;
                        gf[2] = (gf[2] - ur * v4.z) * is; gf[3] = (gf[3] - ur * v4.w) * is;
                    } else {
                        const float vr = vw[r] * k;
                        const float4 u4 = *reinterpret_cast<const float4*>(uu + c);
                        gf[0] = (gf[0] - vr * u4.x) * is; gf[1] = (gf[1] - vr * u4.y) * is;
                        gf[2] = (gf[2] - vr * u4.z) * is; gf[3] = (gf[3] - vr * u4.w) * is;
                    }
                } else {
                    const unsigned total = (unsigned)e.rows * cols;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const unsigned tq = t + q;
                        if (tq < total) {            // (the tensor's 64-element padding carries no gradient)
                            const unsigned r = tq / cols, c = tq - r * cols;
                            const float uv = e.u_axis == 0 ? uu[r] * vw[c] : uu[c] * vw[r];
                            gf[q] = (gf[q] - k * uv) * is;
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gr = gf[e] * gs;
            mf[e] = b1 * mf[e] + (1.f - b1) * gr;
            vf[e] = b2 * vf[e] + (1.f - b2) * gr * gr;
            pf[e] -= lr * (mf[e] * ic1) / (sqrtf(vf[e] * ic2) + eps);
        }
        reinterpret_cast<float4*>(p)[i] = pp;
        reinterpret_cast<float4*>(m)[i] = mm;
        reinterpret_cast<float4*>(v)[i] = vv;
        if (zero_g == 1) reinterpret_cast<float4*>(g)[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        else if (zero_g == 2 && fixed) reinterpret_cast<float4*>(g)[i] = gg;      // keep the FINAL gradient readable (tests, tools)
        if (ema) {
            float4 ee = reinterpret_cast<float4*>(ema)[i];
            float* ef = reinterpret_cast<float*>(&ee);
#pragma unroll
            for (int e = 0; e < 4; ++e) ef[e] = ef[e] * d + (1.f - d) * pf[e];
            reinterpret_cast<float4*>(ema)[i] = ee;
        }
    }
    // tail (arena sizes are padded to 64 by the host, kept for safety; never part of a spectral tensor)
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        const long long i = (n4 << 2) + threadIdx.x;
        const float gr = g[i] * gs;
        m[i] = b1 * m[i] + (1.f - b1) * gr;
        v[i] = b2 * v[i] + (1.f - b2) * gr * gr;
        p[i] -= lr * (m[i] * ic1) / (sqrtf(v[i] * ic2) + eps);
        if (zero_g == 1) g[i] = 0.f;
        if (ema) ema[i] = ema[i] * d + (1.f - d) * p[i];
    }
}

inline unsigned grid_for(long long n) {
    long long b = (n + 255) / 256;
    if (b > 8192) b = 8192;
    if (b < 1) b = 1;
    return (unsigned)b;
}

}  // namespace

extern "C" int xmc_pool2_relu(const void* x, const void* res, void* y, void* xr, int32_t n, int32_t h, int32_t w, int32_t c,
                              float scale, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0 && h >= 2 && w >= 2 && (h % 2) == 0 && (w % 2) == 0 && c > 0);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int ve0 = dtype == XMC_BF16 ? 8 : 4;
    const bool vec = (c % ve0) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                     (res == nullptr || ((uintptr_t)res % 16) == 0) && (xr == nullptr || ((uintptr_t)xr % 16) == 0);
    const int ve = vec ? ve0 : 1;
    const long long nvec = (long long)n * (h / 2) * (w / 2) * (c / ve);
    dim3 grid(grid_for(nvec)), block(256);
#define XMC_POOL2(T, VE_)                                                                                                       \
    do {                                                                                                                        \
        if (xr) hipLaunchKernelGGL((pool2_kernel<T, VE_, true>), grid, block, 0, s, static_cast<const T*>(x), static_cast<const T*>(res), \
                                   static_cast<T*>(y), static_cast<T*>(xr), h, w, c, scale, nvec);                             \
        else hipLaunchKernelGGL((pool2_kernel<T, VE_, false>), grid, block, 0, s, static_cast<const T*>(x), static_cast<const T*>(res),  \
                                static_cast<T*>(y), static_cast<T*>(nullptr), h, w, c, scale, nvec);                           \
    } while (0)
    if (dtype == XMC_BF16) {
        if (vec) XMC_POOL2(bf16_t, 8);
        else XMC_POOL2(bf16_t, 1);
    } else {
        if (vec) XMC_POOL2(float, 4);
        else XMC_POOL2(float, 1);
    }
#undef XMC_POOL2
    XMC_LAUNCH_RET();
}

extern "C" int xmc_pool2(const void* x, const void* res, void* y, int32_t n, int32_t h, int32_t w, int32_t c,
                         float scale, int32_t dtype, void* stream) {
    return xmc_pool2_relu(x, res, y, nullptr, n, h, w, c, scale, dtype, stream);
}

extern "C" int xmc_expand_taps(const void* x, void* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t ks,
                               int32_t sign, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0 && h > 0 && w > 0 && c > 0 && (ks == 1 || ks == 3) && ks * ks * c <= 32);
    XMC_REQUIRE(sign == 1 || sign == -1);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long npix = (long long)n * h * w;
    if (dtype == XMC_BF16 && ks == 3 && c == 3 && w <= ET_W_MAX && (w * 3) % 8 == 0 && (long long)n * h < (1ll << 31) &&
        (reinterpret_cast<uintptr_t>(x) & 15) == 0 && (reinterpret_cast<uintptr_t>(y) & 15) == 0)
        hipLaunchKernelGGL(expand_taps3_rows_kernel, dim3((unsigned)((long long)n * h)), dim3(256), 0, s, static_cast<const bf16_t*>(x),
                           static_cast<bf16_t*>(y), h, w, sign);
    else if (dtype == XMC_BF16)
        hipLaunchKernelGGL((expand_taps_kernel<bf16_t>), dim3(grid_for(npix * 4)), dim3(256), 0, s,
                           static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), h, w, c, ks, sign, npix);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((expand_taps_kernel<float>), dim3(grid_for(npix * 8)), dim3(256), 0, s,
                           static_cast<const float*>(x), static_cast<float*>(y), h, w, c, ks, sign, npix);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_bcast_relu_bwd(const float* dpool, const void* x, void* dx, int64_t a, int64_t r, int64_t c,
                                  int32_t dtype, void* stream) {
    XMC_REQUIRE(dpool && x && dx && a > 0 && r > 0 && c > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long n = (long long)a * r * c;
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((bcast_relu_bwd_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, dpool,
                           static_cast<const bf16_t*>(x), static_cast<bf16_t*>(dx), (long long)r, (long long)c, n);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((bcast_relu_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, dpool,
                           static_cast<const float*>(x), static_cast<float*>(dx), (long long)r, (long long)c, n);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_tanh_out_fwd(const void* x, void* y, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((tanh_fwd_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s,
                           static_cast<const bf16_t*>(x), static_cast<bf16_t*>(y), (long long)n);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((tanh_fwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s,
                           static_cast<const float*>(x), static_cast<float*>(y), (long long)n);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_tanh_out_bwd(const void* dy, const void* y, void* dx, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(dy && y && dx && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((tanh_bwd_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s,
                           static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(y), static_cast<bf16_t*>(dx),
                           (long long)n);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((tanh_bwd_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s,
                           static_cast<const float*>(dy), static_cast<const float*>(y), static_cast<float*>(dx),
                           (long long)n);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cast(const void* x, int32_t dtype_in, void* y, int32_t dtype_out, int64_t n, void* stream) {
    XMC_REQUIRE(x && y && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 grid(grid_for(n)), block(256);
    if (dtype_in == XMC_F32 && dtype_out == XMC_BF16)
        hipLaunchKernelGGL((cast_kernel<float, bf16_t>), grid, block, 0, s, static_cast<const float*>(x),
                           static_cast<bf16_t*>(y), (long long)n);
    else if (dtype_in == XMC_BF16 && dtype_out == XMC_F32)
        hipLaunchKernelGGL((cast_kernel<bf16_t, float>), grid, block, 0, s, static_cast<const bf16_t*>(x),
                           static_cast<float*>(y), (long long)n);
    else if (dtype_in == XMC_F32 && dtype_out == XMC_F32)
        hipLaunchKernelGGL((cast_kernel<float, float>), grid, block, 0, s, static_cast<const float*>(x),
                           static_cast<float*>(y), (long long)n);
    else if (dtype_in == XMC_BF16 && dtype_out == XMC_BF16)
        hipLaunchKernelGGL((cast_kernel<bf16_t, bf16_t>), grid, block, 0, s, static_cast<const bf16_t*>(x),
                           static_cast<bf16_t*>(y), (long long)n);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_add(const void* a, const void* b, void* out, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(a && b && out && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == XMC_BF16)
        hipLaunchKernelGGL((add_kernel<bf16_t>), dim3(grid_for(n)), dim3(256), 0, s, static_cast<const bf16_t*>(a),
                           static_cast<const bf16_t*>(b), static_cast<bf16_t*>(out), (long long)n);
    else if (dtype == XMC_F32)
        hipLaunchKernelGGL((add_kernel<float>), dim3(grid_for(n)), dim3(256), 0, s, static_cast<const float*>(a),
                           static_cast<const float*>(b), static_cast<float*>(out), (long long)n);
    else return XMC_EINVAL;
    XMC_LAUNCH_RET();
}

extern "C" int xmc_adam_ema(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr,
                            float beta1, float beta2, float eps, float c1, float c2, float grad_scale,
                            float ema_decay, void* stream) {
    XMC_REQUIRE(p && g && m && v && n > 0 && c1 > 0.f && c2 > 0.f);
    XMC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                ((uintptr_t)v % 16) == 0 && (ema == nullptr || ((uintptr_t)ema % 16) == 0));
    hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, static_cast<hipStream_t>(stream), p, const_cast<float*>(g),
                       m, v, ema, (long long)n, lr, beta1, beta2, eps, 1.f / c1, 1.f / c2, grad_scale, ema_decay,
                       static_cast<const float*>(nullptr), 0, (const short*)nullptr, (const xmc_sn_entry*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    XMC_LAUNCH_RET();
}

// step counter of one optimiser kept in DEVICE memory: state[0] = t (int32 bits), state[1] = 1 / (1 - beta1^t),
// state[2] = 1 / (1 - beta2^t).  One thread advances t and refreshes the two bias corrections in double precision;
// the Adam kernel launched right behind it reads them -- so a captured hipGraph replays the right step.
__global__ void adam_advance_kernel(float* state, double b1, double b2) {
    const int t = __float_as_int(state[0]) + 1;
    state[0] = __int_as_float(t);
    state[1] = (float)(1.0 / (1.0 - pow(b1, (double)t)));
    state[2] = (float)(1.0 / (1.0 - pow(b2, (double)t)));
}

extern "C" int xmc_adam_ema_dev(float* p, const float* g, float* m, float* v, float* ema, int64_t n, float lr,
                                double beta1, double beta2, float eps, float* step_state, float grad_scale,
                                float ema_decay, void* stream) {
    XMC_REQUIRE(p && g && m && v && n > 0 && step_state);
    XMC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                ((uintptr_t)v % 16) == 0 && (ema == nullptr || ((uintptr_t)ema % 16) == 0));
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, step_state, beta1, beta2);
    hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, p, const_cast<float*>(g), m, v, ema, (long long)n, lr,
                       (float)beta1, (float)beta2, eps, 1.f, 1.f, grad_scale, ema_decay, static_cast<const float*>(step_state),
                       0, (const short*)nullptr, (const xmc_sn_entry*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (const float*)nullptr, (const float*)nullptr);
    XMC_LAUNCH_RET();
}

// xmc_adam_ema_dev with (a) the consumed gradient zeroed in place (zero_grads) and (b) the gradient through sigma of the
// spectrally-normalised tensors applied on the fly (map != NULL: one int16 per 64 arena elements naming the owning entry of
// `table` -- the dot-chunk table of xmc_sn_batched_dot, whose kvec this reads -- or -1; u / v / scal of the forward's power
// iteration).  Replaces xmc_sn_batched_grad_fix's second kernel + a fill + xmc_adam_ema_dev.
extern "C" int xmc_adam_ema_dev_sn(float* p, float* g, float* m, float* v, float* ema, int64_t n, float lr, double beta1,
                                   double beta2, float eps, float* step_state, float grad_scale, float ema_decay,
                                   int32_t zero_grads, const void* map, const void* table, int32_t n_entries,
                                   const float* kvec, const float* scal, const float* u, const float* vv, void* stream) {
    XMC_REQUIRE(p && g && m && v && n > 0 && step_state);
    XMC_REQUIRE(((uintptr_t)p % 16) == 0 && ((uintptr_t)g % 16) == 0 && ((uintptr_t)m % 16) == 0 &&
                ((uintptr_t)v % 16) == 0 && (ema == nullptr || ((uintptr_t)ema % 16) == 0));
    XMC_REQUIRE(!table || (map && n_entries > 0 && kvec && scal && u && vv));     // (map without table: skip marks only)
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, s, step_state, beta1, beta2);
    if (table)
        hipLaunchKernelGGL((adam_kernel<true>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, p, g, m, v, ema, (long long)n, lr,
                           (float)beta1, (float)beta2, eps, 1.f, 1.f, grad_scale, ema_decay, static_cast<const float*>(step_state),
                           zero_grads, static_cast<const short*>(map), static_cast<const xmc_sn_entry*>(table), kvec, scal, u, vv);
    else
        hipLaunchKernelGGL((adam_kernel<false>), dim3(grid_for(n / 4 + 1)), dim3(256), 0, s, p, g, m, v, ema, (long long)n, lr,
                           (float)beta1, (float)beta2, eps, 1.f, 1.f, grad_scale, ema_decay, static_cast<const float*>(step_state),
                           zero_grads, static_cast<const short*>(map), (const xmc_sn_entry*)nullptr, (const float*)nullptr,
                           (const float*)nullptr, (const float*)nullptr, (const float*)nullptr);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_abi_version(void) { return XMC_ABI_VERSION; }
