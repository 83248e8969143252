// Implicit-GEMM NHWC convolution (forward and, with dgrad-layout weights, data gradient) for
// gfx950.  One 256-thread workgroup (4 wave64) computes a 128-pixel x 128-channel output tile:
//   D[cout][pixel] += W[cout][k] * A[pixel][k],  k = (tap, cin)
// The A operand is gathered on the fly (SAME padding, optional nearest-2x upsample and ReLU
// fused into the gather -- the upsampled / activated tensor is never materialised), staged
// through LDS in 64-byte K rows (double buffered, register-staged so loads of tile t+1 fly
// under the MFMAs of tile t), and fed to v_mfma_f32_32x32x16_bf16 (bf16) or
// v_mfma_f32_32x32x2_f32 (exact-fp32 parity mode).  Weights are the MFMA "A" operand and
// pixels the "B" operand, so each lane ends up holding 4 consecutive output channels of one
// pixel per accumulator group: NHWC stores are 8-byte (bf16) / 16-byte (f32) vectors.
#include <cstdlib>

#include "common.h"

namespace {

constexpr int BM = 128;   // pixels per tile
constexpr int BN = 128;   // output channels per tile

template <typename T> struct CT;
template <> struct CT<bf16_t> {
    static constexpr int VE = 8;      // elements per 16-byte vector
    static constexpr int BK = 32;     // K elements per LDS tile row (64 bytes)
    static constexpr int PITCH = 40;  // 80-byte rows: ds_read_b128 conflict-free (5*i mod 16)
    using VT = uint4;
};
template <> struct CT<float> {
    static constexpr int VE = 4;
    static constexpr int BK = 16;
    static constexpr int PITCH = 17;  // odd pitch: ds_read_b32 of 32 consecutive rows conflict-free
    using VT = float4;
};

struct ConvArgs {
    const void* x; const void* w; const float* bias; const void* mask; const void* res; void* y;
    int N, Hi, Wi, Cin, Ho, Wo, Cout;
    int ks, ups, relu_in, res_ups, out_f32;
    int relu_out, mask_after, valid_h, valid_w;
    int log2_wo, log2_howo;
    int M, cchunks, ktiles, tiles_m, tiles_n;
    int packed;              // taps*Cin <= BK: all (tap, c) pairs share ONE K tile (RGB input: 27 of 32)
    float alpha, res_scale;
    const float* alpha_dev;
};

template <typename T> struct Stage;   // per-thread staging registers for one K tile
template <> struct Stage<bf16_t> { uint4 a[2], b[2]; };
template <> struct Stage<float> { float4 a[2], b[2]; };

__device__ __forceinline__ uint4 relu_vec(uint4 v) {
    return make_uint4(relu_bf2(v.x), relu_bf2(v.y), relu_bf2(v.z), relu_bf2(v.w));
}
__device__ __forceinline__ float4 relu_vec(float4 v) {
    return make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
}
__device__ __forceinline__ void zero_vec(uint4& v) { v = make_uint4(0, 0, 0, 0); }
__device__ __forceinline__ void zero_vec(float4& v) { v = make_float4(0.f, 0.f, 0.f, 0.f); }

// scalar-path element insert (Cin not a multiple of the vector width, e.g. RGB input)
__device__ __forceinline__ void set_elem(uint4& v, int e, bf16_t x) {
    uint32_t* w = reinterpret_cast<uint32_t*>(&v);
    w[e >> 1] |= ((uint32_t)x) << ((e & 1) * 16);
}
__device__ __forceinline__ void set_elem(float4& v, int e, float x) { reinterpret_cast<float*>(&v)[e] = x; }

template <typename T, bool VEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvArgs p) {
    using C = CT<T>;
    using VT = typename C::VT;
    constexpr int VE = C::VE, BK = C::BK, PITCH = C::PITCH;
    __shared__ __attribute__((aligned(16))) T lds[2 * 2 * 128 * PITCH];   // [buf][A|B][row][PITCH]
    T* const As = lds;                       // pixels
    T* const Bs = lds + 2 * 128 * PITCH;     // weights

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tile = xcd_remap(blockIdx.x, p.tiles_m * p.tiles_n);
    const int tn = tile / p.tiles_m, tm = tile - tn * p.tiles_m;
    const int m0 = tm * BM, n0 = tn * BN;

    const T* __restrict__ x = static_cast<const T*>(p.x);
    const T* __restrict__ w = static_cast<const T*>(p.w);

    // ---- loader geometry: thread -> (row, 16-byte slot) of the 64-byte K row
    const int lrow = tid >> 2, kv = tid & 3;
    int oy[2], ox[2], nb[2];
    bool pv[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int pix = m0 + lrow + 64 * r;
        pv[r] = pix < p.M;
        const int pp = pv[r] ? pix : 0;
        const int n = pp >> p.log2_howo, rem = pp & ((1 << p.log2_howo) - 1);
        oy[r] = rem >> p.log2_wo;
        ox[r] = rem & (p.Wo - 1);
        nb[r] = n * p.Hi * p.Wi;
    }
    const int taps = p.ks * p.ks, half = p.ks >> 1;

    auto load_tile = [&](int kt, Stage<T>& s) {
        if (!VEC && p.packed) {       // K index j = tap * Cin + c, contiguous in the [cout][tap][cin] weights
            const int kmax = taps * p.Cin;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                zero_vec(s.a[r]);
                zero_vec(s.b[r]);
                const int nrow = n0 + lrow + 64 * r;
                for (int e = 0; e < VE; ++e) {
                    const int j = kv * VE + e;
                    if (j >= kmax) break;
                    const int tp = j / p.Cin, cc = j - tp * p.Cin;
                    const int iy = oy[r] + tp / p.ks - half, ix = ox[r] + (tp - (tp / p.ks) * p.ks) - half;
                    if (pv[r] && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo) {
                        const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                        T v = x[(size_t)(nb[r] + sy * p.Wi + sx) * p.Cin + cc];
                        set_elem(s.a[r], e, v);
                    }
                    if (nrow < p.Cout) set_elem(s.b[r], e, w[(size_t)nrow * kmax + j]);
                }
                if (p.relu_in) s.a[r] = relu_vec(s.a[r]);
            }
            return;
        }
        const int tap = kt / p.cchunks;
        const int c = (kt - tap * p.cchunks) * BK + kv * VE;
        const int dy = tap / p.ks - half, dx = tap - (tap / p.ks) * p.ks - half;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            zero_vec(s.a[r]);
            const int iy = oy[r] + dy, ix = ox[r] + dx;
            const bool ok = pv[r] && (unsigned)iy < (unsigned)p.Ho && (unsigned)ix < (unsigned)p.Wo;
            if (ok && c < p.Cin) {
                const int sy = p.ups ? (iy >> 1) : iy, sx = p.ups ? (ix >> 1) : ix;
                const T* src = x + ((size_t)(nb[r] + sy * p.Wi + sx) * p.Cin + c);
                if (VEC) {
                    s.a[r] = *reinterpret_cast<const VT*>(src);
                } else {
                    for (int e = 0; e < VE; ++e)
                        if (c + e < p.Cin) set_elem(s.a[r], e, src[e]);
                }
                if (p.relu_in) s.a[r] = relu_vec(s.a[r]);
            }
            zero_vec(s.b[r]);
            const int nrow = n0 + lrow + 64 * r;
            if (nrow < p.Cout && c < p.Cin) {
                const T* src = w + ((size_t)nrow * taps + tap) * p.Cin + c;
                if (VEC) {
                    s.b[r] = *reinterpret_cast<const VT*>(src);
                } else {
                    for (int e = 0; e < VE; ++e)
                        if (c + e < p.Cin) set_elem(s.b[r], e, src[e]);
                }
            }
        }
    };
    auto store_tile = [&](int buf, const Stage<T>& s) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            T* da = As + (buf * 128 + lrow + 64 * r) * PITCH + kv * VE;
            T* db = Bs + (buf * 128 + lrow + 64 * r) * PITCH + kv * VE;
            if constexpr (sizeof(T) == 2) {
                *reinterpret_cast<uint4*>(da) = s.a[r];
                *reinterpret_cast<uint4*>(db) = s.b[r];
            } else {
                const float* fa = reinterpret_cast<const float*>(&s.a[r]);
                const float* fb = reinterpret_cast<const float*>(&s.b[r]);
#pragma unroll
                for (int e = 0; e < 4; ++e) { da[e] = fa[e]; db[e] = fb[e]; }
            }
        }
    };

    // ---- wave -> 64(cout) x 64(pixel) sub-tile, as 2x2 MFMA 32x32 blocks
    const int wp = wave & 1, wc = wave >> 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
        const T* a_base = As + (buf * 128 + wp * 64 + l31) * PITCH;
        const T* b_base = Bs + (buf * 128 + wc * 64 + l31) * PITCH;
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                bf16x8 wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wf[i] = *reinterpret_cast<const bf16x8*>(b_base + i * 32 * PITCH + kk * 16 + lhi * 8);
                    xf[i] = *reinterpret_cast<const bf16x8*>(a_base + i * 32 * PITCH + kk * 16 + lhi * 8);
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) {
                float wf[2], xf[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) {
                    wf[i] = b_base[i * 32 * PITCH + kk * 2 + lhi];
                    xf[i] = a_base[i * 32 * PITCH + kk * 2 + lhi];
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wf[i], xf[j], acc[i][j], 0, 0, 0);
            }
        }
    };

    // ---- main loop: register-staged double buffering, one barrier per K tile
    Stage<T> st;
    load_tile(0, st);
    store_tile(0, st);
    __syncthreads();
    for (int kt = 0; kt < p.ktiles; ++kt) {
        const int buf = kt & 1;
        const bool more = kt + 1 < p.ktiles;
        if (more) load_tile(kt + 1, st);
        compute(buf);
        if (more) store_tile(buf ^ 1, st);
        __syncthreads();
    }

    // ---- epilogue.  C/D map of the 32x32 MFMA: col = lane&31 (pixel), row = (reg&3) +
    // 8*(reg>>2) + 4*(lane>>5) (cout): regs 4g..4g+3 are 4 consecutive output channels.
    const T* __restrict__ mask = static_cast<const T*>(p.mask);
    const T* __restrict__ res = static_cast<const T*>(p.res);
    const bool vec_out = (p.Cout & 3) == 0;
    const float alpha = conv_alpha(p.alpha, p.alpha_dev);
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int pix = m0 + wp * 64 + j * 32 + l31;
        if (pix >= p.M) continue;
        size_t rbase = (size_t)pix * p.Cout;
        if (res && p.res_ups) {
            const int n = pix >> p.log2_howo, rem = pix & ((1 << p.log2_howo) - 1);
            const int y2 = (rem >> p.log2_wo) >> 1, x2 = (rem & (p.Wo - 1)) >> 1;
            rbase = ((size_t)(n * (p.Ho >> 1) + y2) * (p.Wo >> 1) + x2) * p.Cout;
        }
        const size_t obase = (size_t)pix * p.Cout;
        bool zero = false;
        if (p.valid_h) {
            const int rem = pix & ((1 << p.log2_howo) - 1);
            zero = (rem >> p.log2_wo) >= p.valid_h || (rem & (p.Wo - 1)) >= p.valid_w;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int c0 = n0 + wc * 64 + i * 32 + g * 8 + lhi * 4;
                if (c0 >= p.Cout) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    v[e] = acc[i][j][g * 4 + e] * alpha;
                    const int c = c0 + e;
                    if (c < p.Cout) {
                        if (p.bias) v[e] += p.bias[c];
                        if (mask && !p.mask_after && !(to_f<T>(mask[obase + c]) > 0.f)) v[e] = 0.f;
                        if (res) v[e] += p.res_scale * to_f<T>(res[rbase + c]);
                        if (mask && p.mask_after && !(to_f<T>(mask[obase + c]) > 0.f)) v[e] = 0.f;
                        if (p.relu_out) v[e] = fmaxf(v[e], 0.f);
                        if (zero) v[e] = 0.f;
                    }
                }
                if (p.out_f32 || sizeof(T) == 4) {
                    float* y = static_cast<float*>(p.y) + obase + c0;
                    if (vec_out) {
                        *reinterpret_cast<float4*>(y) = make_float4(v[0], v[1], v[2], v[3]);
                    } else {
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < p.Cout) y[e] = v[e];
                    }
                } else {
                    bf16_t* y = static_cast<bf16_t*>(p.y) + obase + c0;
                    if (vec_out) {
                        *reinterpret_cast<uint2*>(y) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                    } else {
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < p.Cout) y[e] = f2bf(v[e]);
                    }
                }
            }
        }
    }
}

}  // namespace

// conv_patch.hip: LDS-staged im2col kernel for the eligible bf16 shapes (returns 1 when not eligible)
extern "C" int xmc_conv2d_patch_try(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                    const void* mask, const void* res, void* y, void* stream);

// conv_stream.hip: weight-streaming kernel on fragment-packed weights
extern "C" int xmc_conv2d_stream(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                 const void* mask, const void* res, void* y, void* ws, const void* mask_bits, void* y_bits,
                                 void* stream);
extern "C" int xmc_conv2d_nhwc_bits(const xmc_conv_desc* d, const void* x, const void* w, const float* bias, const void* mask,
                                    const void* res, void* y, void* ws, const void* mask_bits, void* y_bits, void* stream);

extern "C" int xmc_conv2d_nhwc_ws(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                  const void* mask, const void* res, void* y, void* ws, void* stream);

extern "C" int xmc_conv2d_nhwc(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                               const void* mask, const void* res, void* y, void* stream) {
    return xmc_conv2d_nhwc_ws(d, x, w, bias, mask, res, y, nullptr, stream);
}

extern "C" int xmc_conv2d_nhwc_ws(const xmc_conv_desc* d, const void* x, const void* w, const float* bias,
                                  const void* mask, const void* res, void* y, void* ws, void* stream) {
    return xmc_conv2d_nhwc_bits(d, x, w, bias, mask, res, y, ws, nullptr, nullptr, stream);
}

extern "C" int xmc_conv2d_nhwc_bits(const xmc_conv_desc* d, const void* x, const void* w, const float* bias, const void* mask,
                                    const void* res, void* y, void* ws, const void* mask_bits, void* y_bits, void* stream) {
    XMC_REQUIRE(d && x && w && y);
    XMC_REQUIRE(d->w_packed || (!mask_bits && !y_bits));     // bit masks: kernels on fragment-packed weights only
    XMC_REQUIRE(d->ks == 1 || d->ks == 3);
    XMC_REQUIRE(d->dtype == XMC_F32 || d->dtype == XMC_BF16);
    XMC_REQUIRE(d->n > 0 && d->hi > 0 && d->wi > 0 && d->cin > 0 && d->cout > 0);
    if (d->w_packed) return xmc_conv2d_stream(d, x, w, bias, mask, res, y, ws, mask_bits, y_bits, stream);
    XMC_REQUIRE(!d->pool_out);                       // fused pooling exists only in the weight-streaming kernel
    {
        const int rc = xmc_conv2d_patch_try(d, x, w, bias, mask, res, y, stream);
        if (rc != 1) return rc;
    }
    ConvArgs a;
    a.x = x; a.w = w; a.bias = bias; a.mask = mask; a.res = res; a.y = y;
    a.N = d->n; a.Hi = d->hi; a.Wi = d->wi; a.Cin = d->cin; a.Cout = d->cout;
    a.Ho = d->ups ? 2 * d->hi : d->hi;
    a.Wo = d->ups ? 2 * d->wi : d->wi;
    a.ks = d->ks; a.ups = d->ups; a.relu_in = d->relu_in; a.res_ups = d->res_ups; a.out_f32 = d->out_f32;
    a.relu_out = d->relu_out; a.mask_after = d->mask_after_res; a.valid_h = d->valid_h; a.valid_w = d->valid_w;
    a.log2_wo = ilog2_exact(a.Wo);
    const int l2h = ilog2_exact(a.Ho);
    XMC_REQUIRE(a.log2_wo >= 0 && l2h >= 0);
    XMC_REQUIRE(!(d->res_ups) || (a.Ho >= 2 && a.Wo >= 2));
    a.log2_howo = a.log2_wo + l2h;
    const long long m = (long long)a.N * a.Ho * a.Wo;
    XMC_REQUIRE(m < (1ll << 31) && m * (long long)(a.Cout > a.Cin ? a.Cout : a.Cin) < (1ll << 40));
    a.M = (int)m;
    a.alpha = d->alpha; a.res_scale = d->res_scale; a.alpha_dev = d->alpha_dev;
    const int bk = d->dtype == XMC_BF16 ? CT<bf16_t>::BK : CT<float>::BK;
    const int ve = d->dtype == XMC_BF16 ? 8 : 4;
    a.cchunks = (a.Cin + bk - 1) / bk;
    a.ktiles = a.ks * a.ks * a.cchunks;
    a.packed = 0;
    a.tiles_m = (a.M + BM - 1) / BM;
    a.tiles_n = (a.Cout + BN - 1) / BN;
    const bool vec = (a.Cin % ve) == 0 && ((uintptr_t)x % 16) == 0 && ((uintptr_t)w % 16) == 0;
    if (!vec && a.ks * a.ks * a.Cin <= bk) {
        a.packed = 1;
        a.ktiles = 1;
    }
    XMC_REQUIRE(((uintptr_t)y % 16) == 0);
    dim3 grid(a.tiles_m * a.tiles_n), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (d->dtype == XMC_BF16) {
        if (vec) hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<bf16_t, false>), grid, block, 0, s, a);
    } else {
        if (vec) hipLaunchKernelGGL((conv_igemm_kernel<float, true>), grid, block, 0, s, a);
        else hipLaunchKernelGGL((conv_igemm_kernel<float, false>), grid, block, 0, s, a);
    }
    XMC_LAUNCH_RET();
}
