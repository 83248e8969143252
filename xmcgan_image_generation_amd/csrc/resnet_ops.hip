// Pointwise / gather kernels of the frozen ResNet-50 feature path (xmcgan/xmc_gan.py:74-90,
// xmcgan/utils/pretrained_model_utils.py:102-127, xmcgan/utils/resnet_v1.py:60-186).
//
// ResNet's feature maps are 112^2, 56^2, 28^2, 14^2, 7^2 -- not powers of two, which every tiled convolution kernel of
// this library assumes.  They are therefore kept on power-of-two CANVASES (128^2, 64^2, 32^2, 16^2, 8^2; the valid
// region in the top-left corner, 1.31x the pixels): a stride-1 SAME convolution on a canvas whose margin is ZERO
// equals the SAME convolution of the valid region, so all of ResNet's convolutions run on the existing kernels.
// What is left is here: bilinear resize to 224^2 (+ adjoint), the 7x7 stride-2 stem as im2col (+ col2im), 3x3
// stride-2 max pooling (+ adjoint), margin zeroing, stride-2 sub-sampling (+ zero-insertion adjoint), and the
// post-activation residual add + ReLU (+ its backward mask).  All HBM-bound streaming kernels; T = float or bf16.
#include "common.h"

namespace {

inline unsigned rn_grid(long long n) {
    long long b = (n + 255) / 256;
    if (b > 16384) b = 16384;
    if (b < 1) b = 1;
    return (unsigned)b;
}

// ---- jax.image.resize(..., "bilinear") when up-sampling == half-pixel-centre bilinear, edges clamped, no anti-aliasing
// (pretrained_model_utils.py:118-122).  x (N, Hs, Ws, C) -> y canvas (N, Hc, Wc, C): valid region Hd x Wd, margin zero.
template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hs, int Ws, int C,
                                                        int Hd, int Wd, int Hc, int Wc) {
    const long long total = (long long)N * Hc * Wc * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ox = (int)(t % Wc); t /= Wc;
        const int oy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        float v = 0.f;
        if (oy < Hd && ox < Wd) {
            const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
            const float fly = floorf(fy), flx = floorf(fx);
            const float ly = fy - fly, lx = fx - flx;
            const int y0 = max((int)fly, 0), y1 = min(max((int)ceilf(fy), 0), Hs - 1);
            const int x0 = max((int)flx, 0), x1 = min(max((int)ceilf(fx), 0), Ws - 1);
            const T* xn = x + (long long)n * Hs * Ws * C;
            const float a = to_f<T>(xn[((long long)y0 * Ws + x0) * C + c]), b = to_f<T>(xn[((long long)y0 * Ws + x1) * C + c]);
            const float d = to_f<T>(xn[((long long)y1 * Ws + x0) * C + c]), e = to_f<T>(xn[((long long)y1 * Ws + x1) * C + c]);
            const float top = a + (b - a) * lx, bot = d + (e - d) * lx;
            v = top + (bot - top) * ly;
        }
        y[i] = from_f<T>(v);
    }
}

// adjoint: dx[n][sy][sx][c] = sum over output pixels that read (sy, sx) of weight * dy.  One thread per INPUT element
// gathers from the (few) outputs whose 2x2 footprint contains it -- no atomics, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int Hs, int Ws, int C,
                                                        int Hd, int Wd, int Hc, int Wc) {
    const long long total = (long long)N * Hs * Ws * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const float iy = (float)Hd / (float)Hs, ix = (float)Wd / (float)Ws;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int px = (int)(t % Ws); t /= Ws;
        const int py = (int)(t % Hs);
        const int n = (int)(t / Hs);
        // outputs o with floor(f(o)) in {p-1, p} (or clamped onto p): a conservative window, exact weights recomputed
        const int oy_lo = max((int)floorf(((float)py - 1.f + 0.5f) * iy - 0.5f) - 1, 0), oy_hi = min((int)ceilf(((float)py + 1.f + 0.5f) * iy - 0.5f) + 1, Hd - 1);
        const int ox_lo = max((int)floorf(((float)px - 1.f + 0.5f) * ix - 0.5f) - 1, 0), ox_hi = min((int)ceilf(((float)px + 1.f + 0.5f) * ix - 0.5f) + 1, Wd - 1);
        float acc = 0.f;
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float fy = ((float)oy + 0.5f) * sy - 0.5f, fly = floorf(fy), ly = fy - fly;
            const int y0 = max((int)fly, 0), y1 = min(max((int)ceilf(fy), 0), Hs - 1);
            const float wy = (y0 == py ? 1.f - ly : 0.f) + (y1 == py ? ly : 0.f);
            if (wy == 0.f) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                const float fx = ((float)ox + 0.5f) * sx - 0.5f, flx = floorf(fx), lx = fx - flx;
                const int x0 = max((int)flx, 0), x1 = min(max((int)ceilf(fx), 0), Ws - 1);
                const float wx = (x0 == px ? 1.f - lx : 0.f) + (x1 == px ? lx : 0.f);
                if (wx == 0.f) continue;
                acc += wy * wx * to_f<T>(dy[(((long long)n * Hc + oy) * Wc + ox) * C + c]);
            }
        }
        dx[i] = from_f<T>(acc);
    }
}

// ---- stem: 7x7 stride-2 SAME convolution of the (N, Hc, Wc, 3) image canvas (valid Hv x Wv = 224^2) as im2col:
// col[n][oy][ox][tap * 3 + ch] (K = 147 padded to KP = 160), output canvas (N, Ho, Wo) with valid Hov x Wov (112^2),
// margin rows zero.  SAME padding for k = 7, s = 2 on 224: 2 before, 3 after (flax nn.Conv, resnet_v1.py:148-154).
template <typename T>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int Hc, int Wc, int Hv,
                                                         int Wv, int Ho, int Wo, int Hov, int Wov, int KP, int pad) {
    const long long total = (long long)N * Ho * Wo * KP;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k = (int)(i % KP);
        long long t = i / KP;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float v = 0.f;
        if (k < 147 && oy < Hov && ox < Wov) {
            const int tap = k / 3, ch = k - tap * 3;
            const int yy = 2 * oy + tap / 7 - pad, xx = 2 * ox + tap % 7 - pad;
            if ((unsigned)yy < (unsigned)Hv && (unsigned)xx < (unsigned)Wv) v = to_f<T>(x[(((long long)n * Hc + yy) * Wc + xx) * 3 + ch]);
        }
        col[i] = from_f<T>(v);
    }
}

// adjoint: dx[n][y][x][ch] = sum over taps of dcol[n][(y + pad - ky) / 2][(x + pad - kx) / 2][tap * 3 + ch]
template <typename T>
__global__ __launch_bounds__(256) void stem_col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int Hc, int Wc, int Hv,
                                                         int Wv, int Ho, int Wo, int Hov, int Wov, int KP, int pad) {
    const long long total = (long long)N * Hc * Wc * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int ch = (int)(i % 3);
        long long t = i / 3;
        const int xx = (int)(t % Wc); t /= Wc;
        const int yy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        float acc = 0.f;
        if (yy < Hv && xx < Wv) {
            for (int ky = 0; ky < 7; ++ky) {
                const int ty = yy + pad - ky;
                if (ty < 0 || (ty & 1)) continue;
                const int oy = ty >> 1;
                if (oy >= Hov) continue;
                for (int kx = 0; kx < 7; ++kx) {
                    const int tx = xx + pad - kx;
                    if (tx < 0 || (tx & 1)) continue;
                    const int ox = tx >> 1;
                    if (ox >= Wov) continue;
                    acc += to_f<T>(dcol[(((long long)n * Ho + oy) * Wo + ox) * KP + (ky * 7 + kx) * 3 + ch]);
                }
            }
        }
        dx[i] = from_f<T>(acc);
    }
}

// ---- nn.max_pool(x, (3, 3), strides=(2, 2), padding="SAME") (resnet_v1.py:156): input canvas (N, Hc, Wc, C) valid
// Hv x Wv, output canvas (N, Hc/2, Wc/2, C) valid ceil(Hv/2) x ceil(Wv/2); windows rows 2o .. 2o+2 clipped to the
// valid region (SAME padding for even sizes: 0 before, 1 after, padded with -inf).  Margin of the output is zero.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hc, int Wc, int C, int Hv,
                                                         int Wv) {
    const int Ho = Hc >> 1, Wo = Wc >> 1, Hov = (Hv + 1) >> 1, Wov = (Wv + 1) >> 1;
    const long long total = (long long)N * Ho * Wo * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m = 0.f;
        if (oy < Hov && ox < Wov) {
            m = -INFINITY;
            for (int dy = 0; dy < 3; ++dy) {
                const int yy = 2 * oy + dy;
                if (yy >= Hv) break;
                for (int dx = 0; dx < 3; ++dx) {
                    const int xx = 2 * ox + dx;
                    if (xx >= Wv) break;
                    m = fmaxf(m, to_f<T>(x[(((long long)n * Hc + yy) * Wc + xx) * C + c]));
                }
            }
        }
        y[i] = from_f<T>(m);
    }
}

// adjoint: the gradient of every window goes to its FIRST maximum (row-major scan, as XLA's select-and-scatter);
// one thread per input element gathers from the <= 4 windows that contain it.
template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x, const T* __restrict__ y,
                                                         T* __restrict__ dx, int N, int Hc, int Wc, int C, int Hv, int Wv) {
    const int Ho = Hc >> 1, Wo = Wc >> 1, Hov = (Hv + 1) >> 1, Wov = (Wv + 1) >> 1;
    const long long total = (long long)N * Hc * Wc * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int xx = (int)(t % Wc); t /= Wc;
        const int yy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        float acc = 0.f;
        if (yy < Hv && xx < Wv) {
            const float v = to_f<T>(x[i]);
            for (int oy = max((yy - 2 + 1) >> 1, 0); oy <= min(yy >> 1, Hov - 1); ++oy)
                for (int ox = max((xx - 2 + 1) >> 1, 0); ox <= min(xx >> 1, Wov - 1); ++ox) {
                    const long long o = (((long long)n * Ho + oy) * Wo + ox) * C + c;
                    if (to_f<T>(y[o]) != v) continue;
                    // first maximum of the window in scan order?
                    bool first = true;
                    for (int dyy = 0; dyy < 3 && first; ++dyy) {
                        const int y2 = 2 * oy + dyy;
                        if (y2 >= Hv) break;
                        for (int dxx = 0; dxx < 3; ++dxx) {
                            const int x2 = 2 * ox + dxx;
                            if (x2 >= Wv) break;
                            if (y2 == yy && x2 == xx) { dyy = 3; break; }
                            if (to_f<T>(x[(((long long)n * Hc + y2) * Wc + x2) * C + c]) == v) { first = false; break; }
                        }
                    }
                    if (first) acc += to_f<T>(dy[o]);
                }
        }
        dx[i] = from_f<T>(acc);
    }
}

// ---- x[n][y][x][:] = 0 outside the valid Hv x Wv region of a canvas (in place; touches only the margin)
template <typename T>
__global__ __launch_bounds__(256) void zero_margin_kernel(T* __restrict__ x, int N, int Hc, int Wc, int C, int Hv, int Wv) {
    const int mrow = Wc - Wv;                              // margin pixels of a valid row; rows >= Hv are all margin
    const long long per_img = (long long)Hv * mrow + (long long)(Hc - Hv) * Wc;
    const long long total = (long long)N * per_img * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const long long k = t % per_img;
        const int n = (int)(t / per_img);
        int yy, xx;
        if (k < (long long)Hv * mrow) { yy = (int)(k / mrow); xx = Wv + (int)(k % mrow); }
        else { const long long r = k - (long long)Hv * mrow; yy = Hv + (int)(r / Wc); xx = (int)(r % Wc); }
        x[(((long long)n * Hc + yy) * Wc + xx) * C + c] = from_f<T>(0.f);
    }
}

// ---- y[n][oy][ox] = x[n][2 oy + off][2 ox + off]  (stride-2 view of a stride-1 result; off = 1 for the 3x3 and 0 for the
// 1x1 stride-2 SAME convolutions of flax) and its adjoint (zero insertion)
template <typename T>
__global__ __launch_bounds__(256) void subsample2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hc, int Wc, int C, int off,
                                                        int scatter) {
    const int Ho = Hc >> 1, Wo = Wc >> 1;
    const long long total = (long long)N * Hc * Wc * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C);
        long long t = i / C;
        const int xx = (int)(t % Wc); t /= Wc;
        const int yy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        const bool hit = ((yy - off) & 1) == 0 && ((xx - off) & 1) == 0 && yy >= off && xx >= off;
        const int oy = (yy - off) >> 1, ox = (xx - off) >> 1;
        const long long o = (((long long)n * Ho + oy) * Wo + ox) * C + c;
        if (scatter) y[i] = hit ? x[o] : from_f<T>(0.f);          // y is the LARGE tensor here: adjoint
        else if (hit) y[o] = x[i];                                // plain sub-sampling: x large, y small
    }
}

// ---- out = relu(a + b)  /  g = dy * (out > 0) + (acc ? acc : 0)
template <typename T>
__global__ __launch_bounds__(256) void add_relu_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        o[i] = from_f<T>(fmaxf(to_f<T>(a[i]) + (b ? to_f<T>(b[i]) : 0.f), 0.f));
}
template <typename T>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ out,
                                                      T* __restrict__ g, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const float d = to_f<T>(dy[i]) + (dy2 ? to_f<T>(dy2[i]) : 0.f);
        g[i] = from_f<T>(to_f<T>(out[i]) > 0.f ? d : 0.f);
    }
}

}  // namespace

// launch KERNEL<T> for T = bf16_t / float according to `dtype` (a, b: the first two pointer arguments)
#define XMC_RN_LAUNCH(KERNEL, total, dtype, s, ...)                                                                    \
    do {                                                                                                               \
        if ((dtype) == XMC_BF16) { typedef bf16_t T_; hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(rn_grid(total)), dim3(256), 0, s, __VA_ARGS__); } \
        else if ((dtype) == XMC_F32) { typedef float T_; hipLaunchKernelGGL((KERNEL<float>), dim3(rn_grid(total)), dim3(256), 0, s, __VA_ARGS__); } \
        else return XMC_EINVAL;                                                                                        \
    } while (0)
#define CP(p) static_cast<const T_*>(p)
#define MP(p) static_cast<T_*>(p)

extern "C" int xmc_resize_bilinear(const void* x, void* y, int32_t n, int32_t hs, int32_t ws, int32_t c, int32_t hd, int32_t wd,
                                   int32_t hc, int32_t wc, int32_t backward, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0 && hs > 0 && ws > 0 && c > 0 && hd > 0 && wd > 0 && hc >= hd && wc >= wd);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!backward) {            // x (n, hs, ws, c) -> y canvas (n, hc, wc, c)
        XMC_RN_LAUNCH(resize_fwd_kernel, (long long)n * hc * wc * c, dtype, s, CP(x), MP(y), n, hs, ws, c, hd, wd, hc, wc);
    } else {                    // x = dy canvas (n, hc, wc, c) -> y = dx (n, hs, ws, c)
        XMC_RN_LAUNCH(resize_bwd_kernel, (long long)n * hs * ws * c, dtype, s, CP(x), MP(y), n, hs, ws, c, hd, wd, hc, wc);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_stem_im2col(const void* x, void* col, int32_t n, int32_t hc, int32_t wc, int32_t hv, int32_t wv, int32_t ho,
                               int32_t wo, int32_t kp, int32_t backward, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && col && n > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv && kp >= 147 && (hv % 2) == 0 && (wv % 2) == 0);
    const int hov = hv / 2, wov = wv / 2;
    XMC_REQUIRE(ho >= hov && wo >= wov);
    const int pad = 2;                                   // SAME, k = 7, s = 2, even input: total 5 = 2 before + 3 after
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!backward) {
        XMC_RN_LAUNCH(stem_im2col_kernel, (long long)n * ho * wo * kp, dtype, s, CP(x), MP(col), n, hc, wc, hv, wv, ho, wo, hov, wov, kp, pad);
    } else {                    // x = dcol (n, ho, wo, kp) -> col = dx canvas (n, hc, wc, 3)
        XMC_RN_LAUNCH(stem_col2im_kernel, (long long)n * hc * wc * 3, dtype, s, CP(x), MP(col), n, hc, wc, hv, wv, ho, wo, hov, wov, kp, pad);
    }
    XMC_LAUNCH_RET();
}

// forward (dy == NULL): y = maxpool(x);  backward: dx = adjoint(dy) given x and y of the forward pass
extern "C" int xmc_maxpool3x3s2(const void* x, void* y, const void* dy, void* dx, int32_t n, int32_t hc, int32_t wc, int32_t c,
                                int32_t hv, int32_t wv, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0 && c > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv && (hc % 2) == 0 && (wc % 2) == 0);
    XMC_REQUIRE((dy == nullptr) == (dx == nullptr));
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!dy) {
        XMC_RN_LAUNCH(maxpool_fwd_kernel, (long long)n * (hc / 2) * (wc / 2) * c, dtype, s, CP(x), MP(y), n, hc, wc, c, hv, wv);
    } else {
        XMC_RN_LAUNCH(maxpool_bwd_kernel, (long long)n * hc * wc * c, dtype, s, CP(dy), CP(x), CP(y), MP(dx), n, hc, wc, c, hv, wv);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_zero_margin(void* x, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv, int32_t wv, int32_t dtype,
                               void* stream) {
    XMC_REQUIRE(x && n > 0 && c > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv);
    const long long total = (long long)n * ((long long)hv * (wc - wv) + (long long)(hc - hv) * wc) * c;
    if (total == 0) return XMC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH(zero_margin_kernel, total, dtype, s, MP(x), n, hc, wc, c, hv, wv);
    XMC_LAUNCH_RET();
}

// scatter == 0: small (n, hc/2, wc/2, c) <- large (n, hc, wc, c) at positions (2o + off);  scatter == 1: the adjoint
// (large <- small with zeros elsewhere).  `large` / `small` are always passed in that order.
extern "C" int xmc_subsample2(void* large, void* small, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t off, int32_t scatter,
                              int32_t dtype, void* stream) {
    XMC_REQUIRE(large && small && n > 0 && c > 0 && (hc % 2) == 0 && (wc % 2) == 0 && (off == 0 || off == 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const long long total = (long long)n * hc * wc * c;
    if (scatter) XMC_RN_LAUNCH(subsample2_kernel, total, dtype, s, CP(small), MP(large), n, hc, wc, c, off, 1);
    else XMC_RN_LAUNCH(subsample2_kernel, total, dtype, s, CP(large), MP(small), n, hc, wc, c, off, 0);
    XMC_LAUNCH_RET();
}

// o = relu(a + b)   (b may be NULL)
extern "C" int xmc_add_relu(const void* a, const void* b, void* o, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(a && o && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH(add_relu_kernel, (long long)n, dtype, s, CP(a), CP(b), MP(o), (long long)n);
    XMC_LAUNCH_RET();
}

// g = (dy + dy2) * (out > 0)   (dy2 may be NULL)
extern "C" int xmc_relu_bwd(const void* dy, const void* dy2, const void* out, void* g, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(dy && out && g && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH(relu_bwd_kernel, (long long)n, dtype, s, CP(dy), CP(dy2), CP(out), MP(g), (long long)n);
    XMC_LAUNCH_RET();
}
