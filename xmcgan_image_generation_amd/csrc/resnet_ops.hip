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

// ---- jax.image.resize(images, (N, 224, 224, 3), "bilinear") (pretrained_model_utils.py:118-122): a separable triangle
// filter on half-pixel centres, f(o) = (o + 0.5) * in / out - 0.5, weight(i, o) = max(0, 1 - |f(o) - i| / k) normalised
// over the IN-BOUNDS taps, with k = max(in / out, 1): jax.image.resize anti-aliases by default, so the filter widens
// when the image shrinks (256 px -> 224); when it grows (128 px -> 224) this is the 2-tap bilinear with clamped edges.
// x (N, Hs, Ws, C) -> y canvas (N, Hc, Wc, C): valid region Hd x Wd, margin zero.
struct Taps { int lo, hi; float inv_total; };

__device__ __forceinline__ float tri(float f, int i, float inv_k) { return fmaxf(0.f, 1.f - fabsf(f - (float)i) * inv_k); }

__device__ __forceinline__ Taps taps_of(int o, float scale, float k, float inv_k, int in) {
    const float f = ((float)o + 0.5f) * scale - 0.5f;
    Taps t;
    t.lo = max((int)ceilf(f - k), 0);
    t.hi = min((int)floorf(f + k), in - 1);
    float tot = 0.f;
    for (int i = t.lo; i <= t.hi; ++i) tot += tri(f, i, inv_k);
    t.inv_total = tot > 0.f ? 1.f / tot : 0.f;
    return t;
}

template <typename T>
__global__ __launch_bounds__(256) void resize_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int Hs, int Ws, int C,
                                                        int Hd, int Wd, int Hc, int Wc) {
    const long long total = (long long)N * Hc * Wc * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const float ky = fmaxf(sy, 1.f), kx = fmaxf(sx, 1.f), iky = 1.f / ky, ikx = 1.f / kx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int c, ox, oy, n;
        if (total < (1ll << 31)) {                     // 32-bit index split (the 64-bit divisions were most of this kernel's instructions)
            const unsigned iu = (unsigned)i, t1 = iu / (unsigned)C, t2 = t1 / (unsigned)Wc;
            c = (int)(iu - t1 * C); ox = (int)(t1 - t2 * Wc); n = (int)(t2 / (unsigned)Hc); oy = (int)(t2 - (unsigned)n * Hc);
        } else {
            c = (int)(i % C);
            long long t = i / C;
            ox = (int)(t % Wc); t /= Wc;
            oy = (int)(t % Hc);
            n = (int)(t / Hc);
        }
        float v = 0.f;
        if (oy < Hd && ox < Wd && sy <= 1.f && sx <= 1.f) {
            // enlarging (128 px -> 224): two taps per axis with weights (1 - l, l); a tap outside the image drops out and
            // the normalisation hands its weight to the other one == clamping both indices
            const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
            const float fly = floorf(fy), flx = floorf(fx);
            const float ly = fy - fly, lx = fx - flx;
            const int y0 = max((int)fly, 0), y1 = min((int)fly + 1, Hs - 1);
            const int x0 = max((int)flx, 0), x1 = min((int)flx + 1, Ws - 1);
            const T* xn = x + (long long)n * Hs * Ws * C;
            const float a = to_f<T>(xn[((long long)y0 * Ws + x0) * C + c]), b = to_f<T>(xn[((long long)y0 * Ws + x1) * C + c]);
            const float d = to_f<T>(xn[((long long)y1 * Ws + x0) * C + c]), e = to_f<T>(xn[((long long)y1 * Ws + x1) * C + c]);
            const float top = a + (b - a) * lx, bot = d + (e - d) * lx;
            v = top + (bot - top) * ly;
        } else if (oy < Hd && ox < Wd) {
            const Taps ty = taps_of(oy, sy, ky, iky, Hs), tx = taps_of(ox, sx, kx, ikx, Ws);
            const float fy = ((float)oy + 0.5f) * sy - 0.5f, fx = ((float)ox + 0.5f) * sx - 0.5f;
            const T* xn = x + (long long)n * Hs * Ws * C;
            for (int yy = ty.lo; yy <= ty.hi; ++yy) {
                float row = 0.f;
                for (int xx = tx.lo; xx <= tx.hi; ++xx) row += tri(fx, xx, ikx) * to_f<T>(xn[((long long)yy * Ws + xx) * C + c]);
                v += tri(fy, yy, iky) * row;
            }
            v *= ty.inv_total * tx.inv_total;
        }
        y[i] = from_f<T>(v);
    }
}

// adjoint: dx[n][py][px][c] = sum over the output pixels whose filter footprint contains (py, px) of weight * dy.  One
// thread per INPUT element gathers them (a conservative output window, exact weights recomputed) -- no atomics, fixed order.
template <typename T>
__global__ __launch_bounds__(256) void resize_bwd_kernel(const T* __restrict__ dy, T* __restrict__ dx, int N, int Hs, int Ws, int C,
                                                        int Hd, int Wd, int Hc, int Wc) {
    const long long total = (long long)N * Hs * Ws * C;
    const float sy = (float)Hs / (float)Hd, sx = (float)Ws / (float)Wd;
    const float ky = fmaxf(sy, 1.f), kx = fmaxf(sx, 1.f), iky = 1.f / ky, ikx = 1.f / kx;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        int c, px, py, n;
        if (total < (1ll << 31)) {
            const unsigned iu = (unsigned)i, t1 = iu / (unsigned)C, t2 = t1 / (unsigned)Ws;
            c = (int)(iu - t1 * C); px = (int)(t1 - t2 * Ws); n = (int)(t2 / (unsigned)Hs); py = (int)(t2 - (unsigned)n * Hs);
        } else {
            c = (int)(i % C);
            long long t = i / C;
            px = (int)(t % Ws); t /= Ws;
            py = (int)(t % Hs);
            n = (int)(t / Hs);
        }
        // |f(o) - p| < k  <=>  (p - k + 0.5) / s - 0.5 < o < (p + k + 0.5) / s - 0.5
        const int oy_lo = max((int)floorf(((float)py - ky + 0.5f) / sy - 0.5f) - 1, 0);
        const int oy_hi = min((int)ceilf(((float)py + ky + 0.5f) / sy - 0.5f) + 1, Hd - 1);
        const int ox_lo = max((int)floorf(((float)px - kx + 0.5f) / sx - 0.5f) - 1, 0);
        const int ox_hi = min((int)ceilf(((float)px + kx + 0.5f) / sx - 0.5f) + 1, Wd - 1);
        float acc = 0.f;
        // the column weights (triangle x the output column's normalisation) do not depend on the row: once per thread, not once
        // per (row, column) -- windows of up to WMAX columns, wider ones take the general loop
        constexpr int WMAX = 8;
        float wxn[WMAX];
        const bool narrow = ox_hi - ox_lo < WMAX;
        if (narrow) {
#pragma unroll
            for (int k = 0; k < WMAX; ++k) {
                const int ox = min(ox_lo + k, ox_hi);
                const float wx = ox_lo + k <= ox_hi ? tri(((float)ox + 0.5f) * sx - 0.5f, px, ikx) : 0.f;
                wxn[k] = wx == 0.f ? 0.f : wx * taps_of(ox, sx, kx, ikx, Ws).inv_total;
            }
        }
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            const float wy = tri(((float)oy + 0.5f) * sy - 0.5f, py, iky);
            if (wy == 0.f) continue;
            const float wyn = wy * taps_of(oy, sy, ky, iky, Hs).inv_total;
            float row = 0.f;
            if (narrow) {
#pragma unroll
                for (int k = 0; k < WMAX; ++k) {
                    const int ox = min(ox_lo + k, ox_hi);
                    if (wxn[k] != 0.f) row += wxn[k] * to_f<T>(dy[(((long long)n * Hc + oy) * Wc + ox) * C + c]);
                }
            } else {
                for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                    const float wx = tri(((float)ox + 0.5f) * sx - 0.5f, px, ikx);
                    if (wx == 0.f) continue;
                    row += wx * taps_of(ox, sx, kx, ikx, Ws).inv_total * to_f<T>(dy[(((long long)n * Hc + oy) * Wc + ox) * C + c]);
                }
            }
            acc += wyn * row;
        }
        dx[i] = from_f<T>(acc);
    }
}

// V channels per thread: V = Vec<T>::N (one 16-byte access; channel counts divisible by it) or 1 (any channel count)
template <typename T, int V> __device__ __forceinline__ void ldv(const T* p, float* f) {
    if constexpr (V == 1) f[0] = to_f<T>(p[0]);
    else { Vec<T> v; v.load(p); v.get(f); }
}
template <typename T, int V> __device__ __forceinline__ void stv(T* p, const float* f) {
    if constexpr (V == 1) p[0] = from_f<T>(f[0]);
    else { Vec<T> v; v.set(f); v.store(p); }
}

// ---- stem: 7x7 stride-2 SAME convolution of the (N, Hc, Wc, 3) image canvas (valid Hv x Wv = 224^2) as im2col:
// col[n][oy][ox][tap * 3 + ch] (K = 147 padded to KP = 160), output canvas (N, Ho, Wo) with valid Hov x Wov (112^2),
// margin rows zero.  SAME padding for k = 7, s = 2 on 224: 2 before, 3 after (flax nn.Conv, resnet_v1.py:148-154).
// One thread writes V consecutive k (one 16-byte store: the 590 MB of columns are the traffic; the image is cache-resident).
template <typename T, int V>
__global__ __launch_bounds__(256) void stem_im2col_kernel(const T* __restrict__ x, T* __restrict__ col, int N, int Hc, int Wc, int Hv,
                                                         int Wv, int Ho, int Wo, int Hov, int Wov, int KP, int pad) {
    const int KV = KP / V;
    const long long total = (long long)N * Ho * Wo * KV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int k0 = (int)(i % KV) * V;
        long long t = i / KV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        // k = ky * 21 + j, and the 21 values j = kx * 3 + ch of one ky are CONTIGUOUS in the image row 2 oy + ky - pad,
        // starting at element (2 ox - pad) * 3
        float v[V];
        int ky = k0 / 21, j = k0 - ky * 21;
        const bool livepix = oy < Hov && ox < Wov;
        const int e0 = (2 * ox - pad) * 3;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int yy = 2 * oy + ky - pad, el = e0 + j;           // element index inside the image row (3 per pixel)
            // branch-free (all V loads in flight): an element outside the image reads x[0] and becomes zero afterwards
            const bool ok = livepix && ky < 7 && (unsigned)yy < (unsigned)Hv && el >= 0 && el < 3 * Wv;
            const long long idx = ok ? ((long long)n * Hc + yy) * Wc * 3 + el : 0;
            const float val = to_f<T>(x[idx]);
            v[e] = ok ? val : 0.f;
            if (++j == 21) { j = 0; ++ky; }
        }
        stv<T, V>(col + i * V, v);
    }
}

// The same columns, one workgroup per OUTPUT ROW (bf16, 16-byte aligned tensors, Wc <= 512, Wv % 8 == 0): the seven image
// rows the output row reads are staged in LDS with coalesced 16-byte loads (8 leading zeros = the left padding, zeros past
// the valid width and for rows outside the image), then every thread assembles 16-byte column vectors from 2-byte LDS reads.
// The generic kernel gathers with eight 2-byte GLOBAL loads per vector: 290 us for 587 MB of columns, 2 TB/s.
constexpr int IM2COL_WC_MAX = 512, IM2COL_ROW = 8 + IM2COL_WC_MAX * 3 + 24;
__global__ __launch_bounds__(256) void stem_im2col_rows_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int N, int Hc, int Wc,
                                                              int Hv, int Wv, int Ho, int Wo, int Hov, int Wov, int KP, int pad) {
    __shared__ __attribute__((aligned(16))) bf16_t rows[7][IM2COL_ROW];
    const int oy = blockIdx.x % Ho, n = blockIdx.x / Ho;
    const int KV = KP / 8, items = Wo * KV;
    bf16_t* out = col + ((long long)n * Ho + oy) * Wo * KP;
    if (oy >= Hov) {                                 // margin row of the output canvas
        for (int it = threadIdx.x; it < items; it += 256) *reinterpret_cast<uint4*>(out + (long long)it * 8) = make_uint4(0, 0, 0, 0);
        return;
    }
    const int rowlen = 8 + Wc * 3 + 24, nvec = rowlen / 8, vvec = Wv * 3 / 8;       // uint4 per staged row; valid ones start at vector 1
    for (int it = threadIdx.x; it < 7 * nvec; it += 256) {
        const int r = it / nvec, v = it - r * nvec;
        const int yy = 2 * oy + r - pad;
        uint4 q = make_uint4(0, 0, 0, 0);
        if ((unsigned)yy < (unsigned)Hv && v >= 1 && v - 1 < vvec)
            q = *reinterpret_cast<const uint4*>(x + ((long long)n * Hc + yy) * Wc * 3 + (v - 1) * 8);
        *reinterpret_cast<uint4*>(&rows[r][v * 8]) = q;
    }
    __syncthreads();
    for (int it = threadIdx.x; it < items; it += 256) {
        const int ox = it / KV, k0 = (it - ox * KV) * 8;
        unsigned short h[8];
        int ky = k0 / 21, j = k0 - ky * 21;
        const int e0 = 8 + (2 * ox - pad) * 3;                   // >= 2: inside the leading zeros for ox = 0
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h[e] = (ox < Wov && ky < 7) ? rows[ky][e0 + j] : (unsigned short)0;
            if (++j == 21) { j = 0; ++ky; }
        }
        *reinterpret_cast<uint4*>(out + (long long)ox * KP + k0) =
            make_uint4(h[0] | ((unsigned)h[1] << 16), h[2] | ((unsigned)h[3] << 16), h[4] | ((unsigned)h[5] << 16), h[6] | ((unsigned)h[7] << 16));
    }
}

// adjoint: dx[n][y][x][ch] = sum over taps of dcol[n][(y + pad - ky) / 2][(x + pad - kx) / 2][tap * 3 + ch]; one thread
// per image pixel (its three channels are adjacent in every column vector it reads)
template <typename T>
__global__ __launch_bounds__(256) void stem_col2im_kernel(const T* __restrict__ dcol, T* __restrict__ dx, int N, int Hc, int Wc, int Hv,
                                                         int Wv, int Ho, int Wo, int Hov, int Wov, int KP, int pad) {
    const long long total = (long long)N * Hc * Wc;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i;
        const int xx = (int)(t % Wc); t /= Wc;
        const int yy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        float acc[3] = {0.f, 0.f, 0.f};
        if (yy < Hv && xx < Wv) {
            // The taps that reach this pixel have ky = (yy + pad) % 2 + 2a, kx likewise: at most 4 x 4 of the 49.  All 48 loads
            // are issued before the first add (a tap outside the kernel / the output reads dcol[0] and is dropped); with a
            // `continue` per tap every load waited for the one before (233 us for 73 MB).  Same order of the sum.
            const int ky0 = (yy + pad) & 1, kx0 = (xx + pad) & 1;
            float v[16][3];
            bool ok[16];
#pragma unroll
            for (int ab = 0; ab < 16; ++ab) {
                const int ky = ky0 + 2 * (ab >> 2), kx = kx0 + 2 * (ab & 3);
                const int ty = yy + pad - ky, tx = xx + pad - kx;
                const int oy = ty >> 1, ox = tx >> 1;
                ok[ab] = ky < 7 && kx < 7 && ty >= 0 && tx >= 0 && oy < Hov && ox < Wov;
                const T* q = dcol + (ok[ab] ? (((long long)n * Ho + oy) * Wo + ox) * KP + (ky * 7 + kx) * 3 : 0);
                v[ab][0] = to_f<T>(q[0]); v[ab][1] = to_f<T>(q[1]); v[ab][2] = to_f<T>(q[2]);
            }
#pragma unroll
            for (int ab = 0; ab < 16; ++ab) {
                acc[0] += ok[ab] ? v[ab][0] : 0.f; acc[1] += ok[ab] ? v[ab][1] : 0.f; acc[2] += ok[ab] ? v[ab][2] : 0.f;
            }
        }
        T* o = dx + i * 3;
        o[0] = from_f<T>(acc[0]); o[1] = from_f<T>(acc[1]); o[2] = from_f<T>(acc[2]);
    }
}

// ---- nn.max_pool(x, (3, 3), strides=(2, 2), padding="SAME") (resnet_v1.py:156): input canvas (N, Hc, Wc, C) valid
// Hv x Wv, output canvas (N, Hc/2, Wc/2, C) valid ceil(Hv/2) x ceil(Wv/2); windows rows 2o .. 2o+2 clipped to the
// valid region (SAME padding for even sizes: 0 before, 1 after, padded with -inf).  Margin of the output is zero.
// `idx` (optional, uint8, same shape as y) receives the position 0..8 (row * 3 + column inside the window) of each
// window's FIRST maximum in row-major scan order -- XLA's select-and-scatter picks the same element -- for the adjoint.
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, uint8_t* __restrict__ idx, int N, int Hc,
                                                         int Wc, int C, int Hv, int Wv) {
    const int Ho = Hc >> 1, Wo = Wc >> 1, Hov = (Hv + 1) >> 1, Wov = (Wv + 1) >> 1, CV = C / V;
    const long long total = (long long)N * Ho * Wo * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float m[V];
        uint8_t pos[V];
#pragma unroll
        for (int e = 0; e < V; ++e) { m[e] = 0.f; pos[e] = 255; }
        if (oy < Hov && ox < Wov) {
#pragma unroll
            for (int e = 0; e < V; ++e) m[e] = -INFINITY;
            // the nine window loads are issued together (positions past the valid region clamp to its last row / column and
            // are skipped in the scan): with `break`s between them every load waited for the one before
            float f[9][V];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int yy = min(2 * oy + k / 3, Hv - 1), xx = min(2 * ox + k % 3, Wv - 1);
                ldv<T, V>(x + (((long long)n * Hc + yy) * Wc + xx) * C + c, f[k]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const bool valid = 2 * oy + k / 3 < Hv && 2 * ox + k % 3 < Wv;
#pragma unroll
                for (int e = 0; e < V; ++e)
                    if (valid && (f[k][e] > m[e] || pos[e] == 255)) { m[e] = f[k][e]; pos[e] = (uint8_t)k; }
            }
        }
        stv<T, V>(y + i * V, m);
        if (idx) {
            if constexpr (V == 8) { uint2 r; __builtin_memcpy(&r, pos, 8); *reinterpret_cast<uint2*>(idx + i * V) = r; }
            else if constexpr (V == 4) { uint32_t r; __builtin_memcpy(&r, pos, 4); *reinterpret_cast<uint32_t*>(idx + i * V) = r; }
            else idx[i] = pos[0];
        }
    }
}

// adjoint: the gradient of every window goes to the element `idx` names; one thread per input pixel x V channels
// gathers from the <= 4 windows that contain it (no atomics).
template <typename T, int V>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ idx, T* __restrict__ dx, int N,
                                                         int Hc, int Wc, int C, int Hv, int Wv) {
    const int Ho = Hc >> 1, Wo = Wc >> 1, Hov = (Hv + 1) >> 1, Wov = (Wv + 1) >> 1, CV = C / V;
    const long long total = (long long)N * Hc * Wc * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int xx = (int)(t % Wc); t /= Wc;
        const int yy = (int)(t % Hc);
        const int n = (int)(t / Hc);
        float acc[V];
#pragma unroll
        for (int e = 0; e < V; ++e) acc[e] = 0.f;
        if (yy < Hv && xx < Wv) {
            for (int oy = max((yy - 1) >> 1, 0); oy <= min(yy >> 1, Hov - 1); ++oy)
                for (int ox = max((xx - 1) >> 1, 0); ox <= min(xx >> 1, Wov - 1); ++ox) {
                    const long long o = (((long long)n * Ho + oy) * Wo + ox) * C + c;
                    const int pos = (yy - 2 * oy) * 3 + (xx - 2 * ox);
                    uint8_t p[V];
                    if constexpr (V == 8) { const uint2 r = *reinterpret_cast<const uint2*>(idx + o); __builtin_memcpy(p, &r, 8); }
                    else if constexpr (V == 4) { const uint32_t r = *reinterpret_cast<const uint32_t*>(idx + o); __builtin_memcpy(p, &r, 4); }
                    else p[0] = idx[o];
                    float g[V];
                    ldv<T, V>(dy + o, g);
#pragma unroll
                    for (int e = 0; e < V; ++e) if (p[e] == pos) acc[e] += g[e];
                }
        }
        stv<T, V>(dx + i * V, acc);
    }
}

// ---- x[n][y][x][:] = 0 outside the valid Hv x Wv region of a canvas (in place; touches only the margin)
template <typename T, int V>
__global__ __launch_bounds__(256) void zero_margin_kernel(T* __restrict__ x, int N, int Hc, int Wc, int C, int Hv, int Wv) {
    const int mrow = Wc - Wv, CV = C / V;                  // margin pixels of a valid row; rows >= Hv are all margin
    const long long per_img = (long long)Hv * mrow + (long long)(Hc - Hv) * Wc;
    const long long total = (long long)N * per_img * CV;
    float z[V];
#pragma unroll
    for (int e = 0; e < V; ++e) z[e] = 0.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const long long k = t % per_img;
        const int n = (int)(t / per_img);
        int yy, xx;
        if (k < (long long)Hv * mrow) { yy = (int)(k / mrow); xx = Wv + (int)(k % mrow); }
        else { const long long r = k - (long long)Hv * mrow; yy = Hv + (int)(r / Wc); xx = (int)(r % Wc); }
        stv<T, V>(x + (((long long)n * Hc + yy) * Wc + xx) * C + c, z);
    }
}

// ---- small[n][oy][ox] = large[n][2 oy + off][2 ox + off]  (stride-2 view of a stride-1 result; off = 1 for the 3x3 and 0
// for the 1x1 stride-2 SAME convolutions of flax) and its adjoint (zero insertion).  Gather form in both directions:
// SCATTER == 0 iterates over `small`, SCATTER == 1 over `large`.
template <typename T, int V, int SCATTER>
__global__ __launch_bounds__(256) void subsample2_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int Hc, int Wc, int C, int off) {
    const int Ho = Hc >> 1, Wo = Wc >> 1, CV = C / V;
    const int Hd = SCATTER ? Hc : Ho, Wd = SCATTER ? Wc : Wo;
    const long long total = (long long)N * Hd * Wd * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % CV) * V;
        long long t = i / CV;
        const int xx = (int)(t % Wd); t /= Wd;
        const int yy = (int)(t % Hd);
        const int n = (int)(t / Hd);
        float f[V];
        if (SCATTER) {
            const bool hit = ((yy - off) & 1) == 0 && ((xx - off) & 1) == 0 && yy >= off && xx >= off;
            if (hit) ldv<T, V>(src + (((long long)n * Ho + ((yy - off) >> 1)) * Wo + ((xx - off) >> 1)) * C + c, f);
            else {
#pragma unroll
                for (int e = 0; e < V; ++e) f[e] = 0.f;
            }
        } else {
            ldv<T, V>(src + (((long long)n * Hc + 2 * yy + off) * Wc + 2 * xx + off) * C + c, f);
        }
        stv<T, V>(dst + i * V, f);
    }
}

// ---- out = relu(a + b)  /  g = (dy + dy2) * (out > 0)
template <typename T, int V>
__global__ __launch_bounds__(256) void add_relu_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ o, long long nv) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float f[V], g[V];
        ldv<T, V>(a + i * V, f);
        if (b) {
            ldv<T, V>(b + i * V, g);
#pragma unroll
            for (int e = 0; e < V; ++e) f[e] += g[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) f[e] = fmaxf(f[e], 0.f);
        stv<T, V>(o + i * V, f);
    }
}
template <typename T, int V>
__global__ __launch_bounds__(256) void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ dy2, const T* __restrict__ out,
                                                      T* __restrict__ g, long long nv) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long long)gridDim.x * 256) {
        float d[V], d2[V], o[V];
        ldv<T, V>(dy + i * V, d);
        ldv<T, V>(out + i * V, o);
        if (dy2) {
            ldv<T, V>(dy2 + i * V, d2);
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] += d2[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = o[e] > 0.f ? d[e] : 0.f;
        stv<T, V>(g + i * V, d);
    }
}

}  // namespace

// launch KERNEL<T, ...> for T = bf16_t / float according to `dtype`
#define XMC_RN_LAUNCH(KERNEL, total, dtype, s, ...)                                                                    \
    do {                                                                                                               \
        if ((dtype) == XMC_BF16) { typedef bf16_t T_; hipLaunchKernelGGL((KERNEL<bf16_t>), dim3(rn_grid(total)), dim3(256), 0, s, __VA_ARGS__); } \
        else if ((dtype) == XMC_F32) { typedef float T_; hipLaunchKernelGGL((KERNEL<float>), dim3(rn_grid(total)), dim3(256), 0, s, __VA_ARGS__); } \
        else return XMC_EINVAL;                                                                                        \
    } while (0)
// the same with V channels per thread: 16-byte vectors when `c` divides and every pointer in `ptrs` is 16-byte aligned
#define XMC_RN_LAUNCH_V(KERNEL, EXTRA, c, total_elems, vec_ok, dtype, s, ...)                                           \
    do {                                                                                                               \
        if ((dtype) == XMC_BF16) {                                                                                     \
            typedef bf16_t T_;                                                                                         \
            if ((vec_ok) && (c) % 8 == 0) hipLaunchKernelGGL((KERNEL<bf16_t, 8 EXTRA>), dim3(rn_grid((total_elems) / 8)), dim3(256), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<bf16_t, 1 EXTRA>), dim3(rn_grid(total_elems)), dim3(256), 0, s, __VA_ARGS__);       \
        } else if ((dtype) == XMC_F32) {                                                                               \
            typedef float T_;                                                                                          \
            if ((vec_ok) && (c) % 4 == 0) hipLaunchKernelGGL((KERNEL<float, 4 EXTRA>), dim3(rn_grid((total_elems) / 4)), dim3(256), 0, s, __VA_ARGS__); \
            else hipLaunchKernelGGL((KERNEL<float, 1 EXTRA>), dim3(rn_grid(total_elems)), dim3(256), 0, s, __VA_ARGS__);        \
        } else return XMC_EINVAL;                                                                                      \
    } while (0)
#define XMC_COMMA ,
#define CP(p) static_cast<const T_*>(p)
#define MP(p) static_cast<T_*>(p)

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

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
    if (!backward && dtype == XMC_BF16 && al16(x) && al16(col) && (kp % 8) == 0 && wc <= IM2COL_WC_MAX && (wc % 8) == 0 && (wv % 8) == 0 &&
        (2 * (wov - 1) - pad) * 3 + 20 + 8 < 8 + wc * 3 + 24) {
        hipLaunchKernelGGL(stem_im2col_rows_kernel, dim3((unsigned)(n * ho)), dim3(256), 0, s, static_cast<const bf16_t*>(x),
                           static_cast<bf16_t*>(col), n, hc, wc, hv, wv, ho, wo, hov, wov, kp, pad);
    } else if (!backward) {
        XMC_RN_LAUNCH_V(stem_im2col_kernel, , kp, (long long)n * ho * wo * kp, al16(col), dtype, s, CP(x), MP(col), n, hc, wc, hv, wv, ho,
                        wo, hov, wov, kp, pad);
    } else {                    // x = dcol (n, ho, wo, kp) -> col = dx canvas (n, hc, wc, 3)
        XMC_RN_LAUNCH(stem_col2im_kernel, (long long)n * hc * wc, dtype, s, CP(x), MP(col), n, hc, wc, hv, wv, ho, wo, hov, wov, kp, pad);
    }
    XMC_LAUNCH_RET();
}

// y = maxpool(x); idx (uint8, shape of y; may be NULL) receives each window's arg-max position for xmc_maxpool3x3s2_bwd
extern "C" int xmc_maxpool3x3s2(const void* x, void* y, void* idx, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv, int32_t wv,
                                int32_t dtype, void* stream) {
    XMC_REQUIRE(x && y && n > 0 && c > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv && (hc % 2) == 0 && (wc % 2) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH_V(maxpool_fwd_kernel, , c, (long long)n * (hc / 2) * (wc / 2) * c, al16(x) && al16(y), dtype, s, CP(x), MP(y),
                    static_cast<uint8_t*>(idx), n, hc, wc, c, hv, wv);
    XMC_LAUNCH_RET();
}

// dx (n, hc, wc, c) = adjoint of the pooling applied to dy (n, hc/2, wc/2, c), given the forward pass's idx
extern "C" int xmc_maxpool3x3s2_bwd(const void* dy, const void* idx, void* dx, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv,
                                    int32_t wv, int32_t dtype, void* stream) {
    XMC_REQUIRE(dy && idx && dx && n > 0 && c > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv && (hc % 2) == 0 && (wc % 2) == 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH_V(maxpool_bwd_kernel, , c, (long long)n * hc * wc * c, al16(dy) && al16(dx) && (reinterpret_cast<uintptr_t>(idx) & 7) == 0,
                    dtype, s, CP(dy), static_cast<const uint8_t*>(idx), MP(dx), n, hc, wc, c, hv, wv);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_zero_margin(void* x, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t hv, int32_t wv, int32_t dtype,
                               void* stream) {
    XMC_REQUIRE(x && n > 0 && c > 0 && hv > 0 && wv > 0 && hc >= hv && wc >= wv);
    const long long total = (long long)n * ((long long)hv * (wc - wv) + (long long)(hc - hv) * wc) * c;
    if (total == 0) return XMC_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    XMC_RN_LAUNCH_V(zero_margin_kernel, , c, total, al16(x), dtype, s, MP(x), n, hc, wc, c, hv, wv);
    XMC_LAUNCH_RET();
}

// scatter == 0: small (n, hc/2, wc/2, c) <- large (n, hc, wc, c) at positions (2o + off);  scatter == 1: the adjoint
// (large <- small with zeros elsewhere).  `large` / `small` are always passed in that order.
extern "C" int xmc_subsample2(void* large, void* small, int32_t n, int32_t hc, int32_t wc, int32_t c, int32_t off, int32_t scatter,
                              int32_t dtype, void* stream) {
    XMC_REQUIRE(large && small && n > 0 && c > 0 && (hc % 2) == 0 && (wc % 2) == 0 && (off == 0 || off == 1));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool ok = al16(large) && al16(small);
    if (scatter) XMC_RN_LAUNCH_V(subsample2_kernel, XMC_COMMA 1, c, (long long)n * hc * wc * c, ok, dtype, s, CP(small), MP(large), n, hc, wc, c, off);
    else XMC_RN_LAUNCH_V(subsample2_kernel, XMC_COMMA 0, c, (long long)n * (hc / 2) * (wc / 2) * c, ok, dtype, s, CP(large), MP(small), n, hc, wc, c, off);
    XMC_LAUNCH_RET();
}

// o = relu(a + b)   (b may be NULL)
extern "C" int xmc_add_relu(const void* a, const void* b, void* o, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(a && o && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = al16(a) && al16(b) && al16(o) && n % 8 == 0;
    const long long nv = vec ? n / (dtype == XMC_BF16 ? 8 : 4) : n;
    XMC_RN_LAUNCH_V(add_relu_kernel, , (vec ? 8 : 1), (long long)n, vec, dtype, s, CP(a), CP(b), MP(o), nv);
    XMC_LAUNCH_RET();
}

// g = (dy + dy2) * (out > 0)   (dy2 may be NULL)
extern "C" int xmc_relu_bwd(const void* dy, const void* dy2, const void* out, void* g, int64_t n, int32_t dtype, void* stream) {
    XMC_REQUIRE(dy && out && g && n > 0);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = al16(dy) && al16(dy2) && al16(out) && al16(g) && n % 8 == 0;
    const long long nv = vec ? n / (dtype == XMC_BF16 ? 8 : 4) : n;
    XMC_RN_LAUNCH_V(relu_bwd_kernel, , (vec ? 8 : 1), (long long)n, vec, dtype, s, CP(dy), CP(dy2), CP(out), MP(g), nv);
    XMC_LAUNCH_RET();
}
