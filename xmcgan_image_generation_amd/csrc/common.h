// Shared device helpers for the XMC-GAN gfx950 kernels.  gfx950 (MI355X / CDNA4) only:
// 64-wide wavefronts, MFMA 32x32x16 bf16 / 32x32x2 f32, 160 KiB LDS per CU.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <atomic>
#include <initializer_list>

#include "../../include/xmcgan_hip.h"

// MX-fp8 scale rule (xmc_mx_scale_byte): carry constant added to the block maximum's bits before its exponent is taken
// rnd = XMC_MX_RND_NEXT_BINADE (the build's default) or XMC_MX_RND_OCP_FLOOR (the OCP MX v1.0 conversion): xmc_mx_rnd()
#define XMC_MX_RND_NEXT_BINADE 0x1fffffu
#define XMC_MX_RND_OCP_FLOOR 0u

typedef unsigned short bf16_t;

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short short4v;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

static inline int xmc_hip_err(hipError_t e) { return e == hipSuccess ? XMC_OK : -(1000 + (int)e); }
#define XMC_LAUNCH_RET() return xmc_hip_err(hipGetLastError())
#define XMC_REQUIRE(cond) \
    do {                  \
        if (!(cond)) return XMC_EINVAL; \
    } while (0)

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
// float32 -> bf16, round to nearest even: ONE v_cvt_pk_bf16_f32 per pair (gfx950 has the conversion in hardware;
// hipcc does not recognise the shift / add / select idiom of a software rounding and emitted ~7 VALU instructions
// per element for it -- 26 VALU per MFMA in the bf16 GEMM, ~1,000 in every convolution epilogue).
typedef __attribute__((ext_vector_type(2))) __bf16 xmc_bf16x2;
typedef __attribute__((ext_vector_type(2))) float xmc_f32x2;
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const xmc_f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, xmc_bf16x2));
}
__device__ __forceinline__ bf16_t f2bf(float f) { return (bf16_t)(pack_bf2(f, 0.f) & 0xffffu); }
// acc + lo + hi of two packed bf16 in ONE v_dot2c_f32_bf16 (dot with (1, 1)): the bias-gradient sums of the weight-gradient
// kernels took a shift, an and and two adds per pair
__device__ __forceinline__ float bf2_sum_acc(uint32_t v, float acc) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(xmc_bf16x2, v), __builtin_bit_cast(xmc_bf16x2, 0x3f803f80u), acc, false);
}
// relu on two packed bf16: zero a half when its sign bit is set
__device__ __forceinline__ uint32_t relu_bf2(uint32_t v) {
    uint32_t m = ((v >> 15) & 0x00010001u) * 0xffffu;
    return v & ~m;
}

template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// 8-wide (bf16) / 4-wide (f32) 16-byte vector of activations, unpacked to floats
template <typename T> struct Vec;
template <> struct Vec<bf16_t> {
    static constexpr int N = 8;
    uint4 raw;
    __device__ __forceinline__ void load(const bf16_t* p) { raw = *reinterpret_cast<const uint4*>(p); }
    __device__ __forceinline__ void store(bf16_t* p) const { *reinterpret_cast<uint4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_uint4(0, 0, 0, 0); }
    __device__ __forceinline__ void get(float* f) const {
        const uint32_t w[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            f[2 * i] = __uint_as_float(w[i] << 16);
            f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
        }
    }
    __device__ __forceinline__ void set(const float* f) {
        raw.x = pack_bf2(f[0], f[1]);
        raw.y = pack_bf2(f[2], f[3]);
        raw.z = pack_bf2(f[4], f[5]);
        raw.w = pack_bf2(f[6], f[7]);
    }
};
template <> struct Vec<float> {
    static constexpr int N = 4;
    float4 raw;
    __device__ __forceinline__ void load(const float* p) { raw = *reinterpret_cast<const float4*>(p); }
    __device__ __forceinline__ void store(float* p) const { *reinterpret_cast<float4*>(p) = raw; }
    __device__ __forceinline__ void zero() { raw = make_float4(0.f, 0.f, 0.f, 0.f); }
    __device__ __forceinline__ void get(float* f) const { f[0] = raw.x; f[1] = raw.y; f[2] = raw.z; f[3] = raw.w; }
    __device__ __forceinline__ void set(const float* f) { raw = make_float4(f[0], f[1], f[2], f[3]); }
};

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}

// Bijective XCD-aware tile remap: the dispatcher places block b on XCD b % 8 (speed only,
// never correctness); give each XCD a contiguous range of tiles so neighbouring tiles that
// share an operand panel hit the same 4 MiB L2.
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
    const int nx = 8;
    const int xcd = bid % nx, loc = bid / nx;
    const int q = nblk / nx, r = nblk % nx;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + loc;
}

// ---- LDS-DMA helpers (conv_wgrad_dma.hip, conv_stream.hip's pointwise kernel)
typedef int v4i32 __attribute__((ext_vector_type(4)));

// One `buffer_load_dwordx4 ... offen lds`: every lane fetches 16 bytes at srd.base + voff + soff (zeros when voff is
// out of range) and the wave's 1 KiB lands lane-linearly at LDS byte address lds_addr.  Inline asm on purpose: the
// compiler's waitcnt pass cannot prove that a ds_read does not alias an LDS-DMA it knows about and drains vmcnt(0)
// before EVERY LDS read, which serialises the ring; issued from asm the DMA is invisible to it and is retired by the
// counted s_waitcnt vmcnt below.  M0 (the DMA's LDS base) is compiler-reserved: saved / restored in the statement.
__device__ __forceinline__ void dma16(v4i32 srd, unsigned voff, int soff, unsigned lds_addr) {
    unsigned keep;
    soff = __builtin_amdgcn_readfirstlane(soff);      // "s" operands must be provably wave-uniform
    lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(voff), "s"(lds_addr), "s"(srd), "s"(soff)
                 : "memory");
}

__device__ __forceinline__ v4i32 make_srd(const void* ptr, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(ptr);
    v4i32 r;
    r.x = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r.y = __builtin_amdgcn_readfirstlane((int)(unsigned)((a >> 32) & 0xffffu));      // stride 0
    r.z = __builtin_amdgcn_readfirstlane((int)bytes);
    r.w = 0x00020000;
    return r;
}


// alpha of a convolution launch: the host scalar times the optional DEVICE scalar (xmc_conv_desc.alpha_dev: 1 / (sigma + eps) of
// a spectrally-normalised layer).  A scalar (SGPR) load; wave-uniform branch.
__device__ __forceinline__ float conv_alpha(float alpha, const float* alpha_dev) {
#if defined(__HIP_DEVICE_COMPILE__)
    if (alpha_dev) return alpha * *(const __attribute__((address_space(1))) float*)alpha_dev;     // global, never flat
    return alpha;
#else
    return alpha_dev ? alpha * *alpha_dev : alpha;
#endif
}

// ---- shared epilogue of the MFMA convolution kernels -------------------------------------------------------
// One 32 (cout) x 32 (pixel) accumulator block in the 32x32 C/D layout: lane (l31 = pixel, lhi) holds
// couts g * 8 + lhi * 4 + 0..3 in registers g * 4 + 0..3, i.e. four separate 4-channel runs -> 8-byte
// stores scattered over the NHWC row (measured: 38 % of the 96-channel 128^2 layers' time).  One
// v_permlane32_swap per register pair trades runs between the lane halves so that each lane ends up with 16
// CONSECUTIVE couts (block cout lhi * 16 + 0..15): mask / residual / output become 16-byte vectors.
struct ConvEpi {
    const float* bias; const bf16_t* mask; const bf16_t* res; void* y;
    int Cout, out_f32;
    float alpha, res_scale;
    int mask_after = 0;      // apply the mask to (v + res) instead of to v (ReLU backward of a post-activation residual sum)
    int relu_out = 0;        // y = max(., 0)
    int zero = 0;            // this pixel lies in the canvas margin: store zeros
    // ReLU masks as BITS (round 3): [pixel][Cout / 16] uint16, bit k = value of cout 16 * word + k is > 0.  The bf16 mask
    // read in the epilogue is latency-exposed and as large as the output (D 128^2 dgrad: 302 us with it, 204 without;
    // tools/mask_cost.py); one 2-byte word per lane replaces 32 bytes.  Cout % 16 == 0; vector path only.
    const unsigned short* mask_bits = nullptr;   // used INSTEAD of `mask` when set (same pixel / channel indexing as y)
    unsigned short* y_bits = nullptr;            // also write (stored value > 0) of the output
    int pre_bits = -1;       // >= 0: this block's mask word, loaded by the caller for ALL its blocks before the first epilogue
                             // (round 4: one exposed memory latency per wave instead of one per 32 x 32 block)
    // EMIT8 instantiations only (conv_stream_mx8.hip): also write the output as MX-fp8 packets for the NEXT convolution
    unsigned char* y8 = nullptr;   // [pixel][Cout / 64][80] (Cout % 64 == 0), nullptr: off
    int y8_relu = 0;               // the consumer's relu_in, folded into the packets
    unsigned mx_rnd = XMC_MX_RND_NEXT_BINADE;   // scale rule of the packets (xmc_mx_rnd())
    long long y8_pix = 0;          // this lane's output pixel index (set per call)
};

__device__ __forceinline__ unsigned xmc_mx_scale_byte(float amax, unsigned rnd) {      // X = 2^(floor(log2 amax) - 8) for e4m3, one binade
    const int e = (int)(((__float_as_uint(amax) + rnd) >> 23) & 0xffu) - 8;   // higher when amax would saturate (conv_stream_mx8.hip)
    return (unsigned)(e < 0 ? 0 : e);
}
__device__ __forceinline__ unsigned xmc_pack_fp8x4(float a, float b, float c, float d, float is) {
    a = fminf(fmaxf(a * is, -448.f), 448.f); b = fminf(fmaxf(b * is, -448.f), 448.f);
    c = fminf(fmaxf(c * is, -448.f), 448.f); d = fminf(fmaxf(d * is, -448.f), 448.f);
    int v = 0;
    v = __builtin_amdgcn_cvt_pk_fp8_f32(a, b, v, false);
    v = __builtin_amdgcn_cvt_pk_fp8_f32(c, d, v, true);
    return (unsigned)v;
}

// GP ("global pointers"): the epilogue's operands are addressed through the GLOBAL address space explicitly.  The
// pointwise kernel re-reads its pointers from the kernel-argument segment behind an asm barrier, so the compiler no longer
// knows where they point and its whole epilogue was flat_load / flat_store (617 + 312 instructions) -- which count in
// LGKMCNT as well as VMCNT: every s_waitcnt lgkmcnt(0) in front of the next tile's LDS fragment reads also waited for this
// tile's stores.  (For the other kernels, whose pointers are plain kernel arguments, the flag changes nothing.)
typedef unsigned int epi_u32x4 __attribute__((ext_vector_type(4)));      // plain vector types: HIP's float4 / uint4 classes
typedef float epi_f32x4 __attribute__((ext_vector_type(4)));             // cannot be dereferenced through an address space
template <bool GP, typename T> __device__ __forceinline__ T epi_ld(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (GP) return *(const __attribute__((address_space(1))) T*)p;
#endif
    return *p;
}
template <bool GP, typename T> __device__ __forceinline__ void epi_st(T* p, T v) {
#if defined(CS_ABL) && (CS_ABL & 32)                 // ablation: the epilogue's arithmetic without its store instructions
    asm volatile("" ::"v"(v), "v"(p));
    return;
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#if defined(XMC_EPI_NT)                              // experiment (tools/build_variant.sh): non-temporal epilogue stores
    if constexpr (GP) { __builtin_nontemporal_store(v, (__attribute__((address_space(1))) T*)p); return; }
#endif
    if constexpr (GP) { *(__attribute__((address_space(1))) T*)p = v; return; }
#endif
    *p = v;
}

// the mask word conv_epilogue_block(a, cb0, lhi, obase, ...) would read (or -1 when that block takes no bit mask): kernels
// call this for every block of the wave first, so the loads are in flight together
__device__ __forceinline__ int conv_epilogue_mask_word(const ConvEpi& e, int cb0, int lhi, size_t obase, bool live) {
    const int c0 = cb0 + lhi * 16;
    if (!e.mask_bits) return -1;                     // wave-uniform
#ifdef XMC_NO_MASK_PRELOAD                              // A/B build (tools/gpu_ab_mask_preload.sh): every block loads its own word
    return -1;
#endif
    // the load itself is UNCONDITIONAL (a block without a word reads word 0 and drops it): a load under a per-lane condition
    // compiles to branch + load + s_waitcnt vmcnt(0), one exposed latency per block again
    const bool ok = live && (e.Cout & 15) == 0 && c0 + 16 <= e.Cout;
    const int w = (int)e.mask_bits[ok ? (obase + c0) >> 4 : 0];
    return ok ? w : -1;
}

template <bool EMIT8 = false, bool GP = false>
__device__ __forceinline__ void conv_epilogue_block(f32x16 a, int cb0, int lhi, size_t obase, size_t rbase, const ConvEpi& e) {
    float v[16];
#pragma unroll
    for (int q = 0; q < 2; ++q)                       // (g0, g2) and (g1, g3)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(a[q * 4 + t]), __float_as_uint(a[8 + q * 4 + t]),
                                                            false, false);
            v[q * 8 + t] = __uint_as_float(r[0]);     // lo lanes keep their run, hi lanes receive the lo half's upper run
            v[q * 8 + 4 + t] = __uint_as_float(r[1]);
        }
    const int c0 = cb0 + lhi * 16;
    if (c0 >= e.Cout) return;
    if ((e.Cout & 7) == 0 && c0 + 16 <= e.Cout) {
        if (e.bias) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const epi_f32x4 b = epi_ld<GP>(reinterpret_cast<const epi_f32x4*>(e.bias + c0 + 4 * k));
                v[4 * k] = v[4 * k] * e.alpha + b.x; v[4 * k + 1] = v[4 * k + 1] * e.alpha + b.y;
                v[4 * k + 2] = v[4 * k + 2] * e.alpha + b.z; v[4 * k + 3] = v[4 * k + 3] * e.alpha + b.w;
            }
        } else {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] *= e.alpha;
        }
        auto apply_mask = [&]() {
            if (e.mask_bits) {
                const unsigned m = e.pre_bits >= 0 ? (unsigned)e.pre_bits : (unsigned)epi_ld<GP>(e.mask_bits + ((obase + c0) >> 4));
#pragma unroll
                for (int k = 0; k < 16; ++k) if (!((m >> k) & 1u)) v[k] = 0.f;
                return;
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Vec<bf16_t> m; float f[8];
                const epi_u32x4 q4 = epi_ld<GP>(reinterpret_cast<const epi_u32x4*>(e.mask + obase + c0 + 8 * h));
                m.raw = make_uint4(q4.x, q4.y, q4.z, q4.w); m.get(f);
#pragma unroll
                for (int k = 0; k < 8; ++k) if (!(f[k] > 0.f)) v[8 * h + k] = 0.f;
            }
        };
        if ((e.mask || e.mask_bits) && !e.mask_after) apply_mask();
        if (e.res) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Vec<bf16_t> m; float f[8];
                const epi_u32x4 q4 = epi_ld<GP>(reinterpret_cast<const epi_u32x4*>(e.res + rbase + c0 + 8 * h));
                m.raw = make_uint4(q4.x, q4.y, q4.z, q4.w); m.get(f);
                // an explicit fma, not `v += s * f`: left to contraction, the MX-fp8 kernel's instantiation got two of its eight
                // packed operations as v_pk_mul_f32 + v_pk_add_f32 with CROSSED halves (op_sel:[0,1] op_sel_hi:[1,0]; the
                // register allocator had the pair in swapped order) -- and exactly the two channels that went through that pair
                // lost their residual term in ~0.01 % of the outputs whenever a weight-gradient launch shared the CUs: one
                // 16-lane pass of one instruction at a time, never alone on the GPU (tools/mx8_concurrency2.py; round 4's
                // "fp8 stream race").  With the fused form every pair is one v_pk_fma_f32 and the launch is bit-stable.
                // tests/test_cabi.py keeps the crossed packed add out of the library.
#pragma unroll
                for (int k = 0; k < 8; ++k) v[8 * h + k] = fmaf(e.res_scale, f[k], v[8 * h + k]);
            }
        }
        if ((e.mask || e.mask_bits) && e.mask_after) apply_mask();
        if (e.relu_out) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = fmaxf(v[k], 0.f);
        }
        if (e.zero) {
#pragma unroll
            for (int k = 0; k < 16; ++k) v[k] = 0.f;
        }
        if (e.y_bits) {
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 16; ++k) m |= (v[k] > 0.f ? 1u : 0u) << k;
            epi_st<GP>(e.y_bits + ((obase + c0) >> 4), (unsigned short)m);
        }
        if (e.out_f32) {
            float* y = static_cast<float*>(e.y) + obase + c0;
#pragma unroll
            for (int k = 0; k < 4; ++k) epi_st<GP>(reinterpret_cast<epi_f32x4*>(y + 4 * k), epi_f32x4{v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]});
        } else {
            bf16_t* y = static_cast<bf16_t*>(e.y) + obase + c0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                Vec<bf16_t> o; o.set(v + 8 * h); epi_st<GP>(reinterpret_cast<epi_u32x4*>(y + 8 * h), epi_u32x4{o.raw.x, o.raw.y, o.raw.z, o.raw.w});
                if constexpr (EMIT8) o.get(v + 8 * h);     // the packets quantise what the bf16 tensor holds (= a separate pass)
            }
        }
        if constexpr (EMIT8) {
            // MX-fp8 twin of this 16-channel run: a 32-channel block is this lane + lane ^ 32 (same pixel, other half)
            if (e.y8) {
                float amax = 0.f;
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    if (e.y8_relu) v[k] = fmaxf(v[k], 0.f);
                    amax = fmaxf(amax, fabsf(v[k]));
                }
                amax = fmaxf(amax, __shfl_xor(amax, 32));
                const unsigned sb = xmc_mx_scale_byte(amax, e.mx_rnd);
                const float is = __uint_as_float((254u - sb) << 23);
                unsigned char* pk = e.y8 + ((size_t)e.y8_pix * (e.Cout >> 6) + (c0 >> 6)) * 80;
                *reinterpret_cast<uint4*>(pk + (c0 & 63)) =
                    make_uint4(xmc_pack_fp8x4(v[0], v[1], v[2], v[3], is), xmc_pack_fp8x4(v[4], v[5], v[6], v[7], is),
                               xmc_pack_fp8x4(v[8], v[9], v[10], v[11], is), xmc_pack_fp8x4(v[12], v[13], v[14], v[15], is));
                if (lhi == 0) pk[64 + ((c0 >> 5) & 1)] = (unsigned char)sb;
            }
        }
        return;
    }
    for (int k = 0; k < 16; ++k) {                    // ragged channel counts (3, 40, 72, 136, ...): scalar tail
        const int c = c0 + k;
        if (c >= e.Cout) break;
        float t = v[k] * e.alpha;
        if (e.bias) t += e.bias[c];
        if (e.mask && !e.mask_after && !(bf2f(e.mask[obase + c]) > 0.f)) t = 0.f;
        if (e.res) t += e.res_scale * bf2f(e.res[rbase + c]);
        if (e.mask && e.mask_after && !(bf2f(e.mask[obase + c]) > 0.f)) t = 0.f;
        if (e.relu_out) t = fmaxf(t, 0.f);
        if (e.zero) t = 0.f;
        if (e.out_f32) static_cast<float*>(e.y)[obase + c] = t;
        else static_cast<bf16_t*>(e.y)[obase + c] = f2bf(t);
    }
}

// MFMA-fragment ("packed") order of prepared conv weights, consumed by conv_stream.hip:
//   [row block of 32][K chunk of 32][tap][k16 half][lane = k8 half * 32 + row % 32][8]
// Element offset of (row r, tap, k) for a weight with `taps` taps and `kchunks` = K / 32.
__device__ __forceinline__ long long packed_w_index(int r, int tap, int k, int taps, int kchunks) {
    const long long blk = ((long long)(r >> 5) * kchunks + (k >> 5)) * taps + tap;
    return blk * 1024 + ((k >> 4) & 1) * 512 + ((((k >> 3) & 1) << 5) + (r & 31)) * 8 + (k & 7);
}

// One 32 (row) x 32 (k) tile of one tap of a float32 master weight [cout][taps][cin] -> prepared forward copy and
// (through the LDS transpose tile) dgrad copy [cin][taps-1-tap][cout]; shared by spectral.hip / spectral_batched.hip.
template <typename T>
__device__ __forceinline__ void prep_weight_tile(float (*tile)[33], const float* __restrict__ w, float is, T* __restrict__ wf,
                                                 T* __restrict__ wd, int cout, int taps, int cin, int tap, int n0, int c0,
                                                 int packed) {
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;      // 32 x 8
    // the four loads are issued together (an element outside the weight reads w[0] and becomes zero): guarded one by one they
    // were four serial memory latencies per tile
    float ld[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, c = c0 + tx;
        ld[k] = w[(n < cout && c < cin) ? ((long long)n * taps + tap) * cin + c : 0];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int n = n0 + ty + 8 * k, c = c0 + tx;
        const bool in = n < cout && c < cin;
        const long long idx = ((long long)n * taps + tap) * cin + c;
        const float v = in ? ld[k] * is : 0.f;
        if (wf && !(packed & 1) && in) wf[idx] = from_f<T>(v);
        tile[ty + 8 * k][tx] = v;
    }
    __syncthreads();
    // Fragment-order copies: the 32 x 32 tile IS one 2 KiB block [k16 half][lane = k8 half * 32 + row][8]; threads
    // 0..127 write the forward block, 128..255 the dgrad block (rows = cin, k = cout), one 16-byte vector each:
    // 2 KiB contiguous per block instead of 1024 scattered 2-byte stores.
    const int v16 = threadIdx.x & 127, kk = v16 >> 6, lhi = (v16 >> 5) & 1, l31 = v16 & 31;
    const int kb = kk * 16 + lhi * 8;
    if ((packed & 1) && wf && threadIdx.x < 128) {
        float f[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = tile[l31][kb + e];
        Vec<bf16_t> o; o.set(f);
        const long long blk = ((long long)(n0 >> 5) * (cin >> 5) + (c0 >> 5)) * taps + tap;
        o.store(reinterpret_cast<bf16_t*>(wf) + blk * 1024 + v16 * 8);
    }
    if (!wd) return;
    if (packed & 2) {
        if (threadIdx.x >= 128) {
            float f[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) f[e] = tile[kb + e][l31];
            Vec<bf16_t> o; o.set(f);
            const long long blk = ((long long)(c0 >> 5) * (cout >> 5) + (n0 >> 5)) * taps + (taps - 1 - tap);
            o.store(reinterpret_cast<bf16_t*>(wd) + blk * 1024 + v16 * 8);
        }
        return;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, n = n0 + tx;
        if (n < cout && c < cin) wd[((long long)c * taps + (taps - 1 - tap)) * cout + n] = from_f<T>(tile[tx][ty + 8 * k]);
    }
}

// Opt-in to > 64 KiB of dynamic LDS for a set of kernels, once per DEVICE (the attribute is per device; a process
// may drive several GPUs from several threads).  One static instance per launch site; lock-free, idempotent.
struct XmcLdsOptIn {
    std::atomic<uint64_t> done{0};
    bool ensure(std::initializer_list<const void*> fns, int bytes) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess) return false;
        const uint64_t bit = 1ull << (dev & 63);
        if (done.load(std::memory_order_acquire) & bit) return true;
        bool ok = true;
        for (const void* f : fns) ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
        if (ok) done.fetch_or(bit, std::memory_order_release);
        return ok;
    }
};

// Deterministic split-K of the weight-gradient kernels: with a workspace, split s of a launch writes its partial
// dW (and bias-gradient) slab to part[s][L] with plain stores (L = cout * taps * cin + cout) instead of adding it
// to dW with float atomics; xmc_internal_wgrad_reduce then adds the slabs to dW / db in a fixed order.
// `overwrite` (xmc_wgrad_desc.variant bit 12, XMC_WGRAD_OVERWRITE): dW / db = alpha * sum instead of +=.
extern "C" int xmc_internal_wgrad_reduce(const float* part, int nsplit, long long L, long long n_w, float* dw,
                                         float* db, float alpha, int overwrite, void* stream);

// Launch-heuristic knobs (include/xmcgan_hip.h: xmc_set_tuning): compiled-in defaults, overridden only through that entry point --
// the library reads no environment variable.  xmc_internal_tuning returns the current value (relaxed atomic load).
enum XmcTune {
    XMC_TUNE_KSPLIT_TARGET = 0,        // workgroups a split-K 3x3 forward / data-gradient launch aims for (256 = one per CU)
    XMC_TUNE_KSPLIT_TARGET_PHASE,      // ... the phase-decomposed kernels (384)
    XMC_TUNE_KSPLIT_TARGET_PW,         // ... the pointwise kernel (256)
    XMC_TUNE_TILE64_PCT,               // 64-cout tiles for launches of up to this % of one 128-cout tile per CU (100)
    XMC_TUNE_WGRAD_TARGET_HI,          // weight gradients, maps >= 64^2 and pointwise layers (384)
    XMC_TUNE_WGRAD_TARGET_LO,          // weight gradients below (512)
    XMC_TUNE_WGRAD_TARGET_PHASE,       // phase-decomposed weight gradients (384; round 4: 768 -- re-swept in the step after first-write gradients)
    XMC_TUNE_CBN_RUN,                  // conditional-BatchNorm run kernels (1)
    XMC_TUNE_MX8_SCALE_FLOOR,          // MX-fp8 scale rule: 0 = next binade when the block maximum would saturate (default), 1 = the OCP floor rule
    XMC_TUNE_COUNT
};
extern "C" int xmc_internal_tuning(int id);
static inline unsigned xmc_mx_rnd() { return xmc_internal_tuning(XMC_TUNE_MX8_SCALE_FLOOR) ? XMC_MX_RND_OCP_FLOOR : XMC_MX_RND_NEXT_BINADE; }

extern "C" {     // per-translation-unit LDS opt-in hooks (not part of the public header)
int xmc_internal_optin_conv_stream(void);
int xmc_internal_optin_wgrad_dma(void);
int xmc_internal_optin_wgrad_patch(void);
int xmc_internal_optin_losses(void);
}

// compute units of the current device (cached per device; 256 on MI355X) -- sizes persistent-workgroup grids
static inline int xmc_cu_count() {
    static std::atomic<int> cache[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return 256;
    int v = cache[dev & 63].load(std::memory_order_relaxed);
    if (v <= 0) {
        if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        cache[dev & 63].store(v, std::memory_order_relaxed);
    }
    return v;
}

static inline int ilog2_exact(int v) {
    if (v <= 0 || (v & (v - 1))) return -1;
    int l = 0;
    while ((1 << l) < v) ++l;
    return l;
}
