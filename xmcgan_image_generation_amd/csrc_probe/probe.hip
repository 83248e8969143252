// Hardware-layout probes used by tests/test_gpu_kernels.py: they pin the two gfx950 facts the
// MFMA kernels rely on -- the C/D register map of v_mfma_f32_32x32x16_bf16 and the lane
// transposition performed by ds_read_b64_tr_b16 -- against what the kernels assume.
#include "common.h"
#include "../../include/xmc_probe.h"

namespace {

__global__ __launch_bounds__(64) void probe_kernel(float* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) short tile[16 * 160];
    const int lane = threadIdx.x;
    typedef __attribute__((ext_vector_type(8))) short short8v;
    // ---- (a) D[i][j] = i + 1 and D[i][j] = j + 1 through the k = 0 slot
    short8v rowv = {0, 0, 0, 0, 0, 0, 0, 0}, onev = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 32) {
        rowv[0] = (short)f2bf((float)(lane + 1));
        onev[0] = (short)f2bf(1.0f);
    }
    f32x16 z;
    for (int e = 0; e < 16; ++e) z[e] = 0.f;
    const f32x16 drow = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, rowv),
                                                                __builtin_bit_cast(bf16x8, onev), z, 0, 0, 0);
    const f32x16 dcol = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, onev),
                                                                __builtin_bit_cast(bf16x8, rowv), z, 0, 0, 0);
    for (int e = 0; e < 16; ++e) {
        out[lane * 16 + e] = drow[e];
        out[1024 + lane * 16 + e] = dcol[e];
    }
    // ---- (b) transpose read: tile[row][col] = row * 256 + col
    for (int i = lane; i < 16 * 160; i += 64) tile[i] = (short)((i / 160) * 256 + (i % 160));
    __syncthreads();
    const int q = lane & 15, g = lane >> 4;
    const int krow = (g >> 1) * 8 + (q >> 2);
    const int ccol = (g & 1) * 16 + (q & 3) * 4;
    typedef __attribute__((address_space(3))) short4v* lptr;
    const short4v v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lptr)(tile + krow * 160 + ccol));
    for (int e = 0; e < 4; ++e) out[2048 + lane * 4 + e] = (float)v[e];
}

// Registers-only MFMA loop: NACC independent accumulators per wave, no memory instruction inside the loop.  What a
// full-chip launch of this sustains is the matrix-core rate the convolution kernels could reach at best at the clock
// the chip holds under that load (tools/mfma_rate_probe.py).
template <int MODE, int NACC>
__global__ __launch_bounds__(256) void mfma_rate_kernel(float* __restrict__ out, int iters, int constant) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
    typedef __attribute__((ext_vector_type(8))) short short8v;
    typedef __attribute__((ext_vector_type(8))) int int8v;
    // four operand pairs of pseudo-random values, cycled: constant operands would not toggle the multipliers and the
    // chip would hold a clock that real data does not see
    short8v a16[4], b16[4];
    int8v a8[4], b8[4];
    unsigned h = (unsigned)lane * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            a16[k][e] = (short)((h >> 16) & 0xbf7f);                 // |x| < 2, random sign and mantissa
            h = h * 1664525u + 1013904223u;
            b16[k][e] = (short)((h >> 16) & 0xbf7f);
            h = h * 1664525u + 1013904223u;
            a8[k][e] = (int)(h & 0xb7b7b7b7u);                       // e4m3, |x| < 2
            h = h * 1664525u + 1013904223u;
            b8[k][e] = (int)(h & 0xb7b7b7b7u);
            if (constant) { a16[k][e] = 0x3f80; b16[k][e] = 0x3f00; a8[k][e] = 0x38383838; b8[k][e] = 0x30303030; }
        }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (MODE == 0)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a16[i & 3]), __builtin_bit_cast(bf16x8, b16[(i + (i >> 2)) & 3]), acc[i], 0, 0, 0);
            else
                acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8[i & 3], b8[(i + (i >> 2)) & 3], acc[i], 0, 0, 0, 127, 0, 127);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[i][e];
    if (sum == -1.f) out[0] = sum;                   // never true: keeps the loop alive
}

// L2 -> CU delivery rate of the two load paths the convolution kernels use, on data every workgroup re-reads from a
// small (L2-resident) region: PATH 0 = global_load_dwordx4 into registers (bursts of 6 * D loads per wave),
// PATH 1 = buffer_load_dwordx4 ... lds (LDS-DMA, a ring of D + 1 stages of 6 instructions per wave, counted vmcnt --
// the pointwise kernel's ring).  pattern 0: every instruction reads 1 KiB contiguous; pattern 1: 16 rows x 64 bytes
// at a 2 KiB stride (the x operand of a 1024-channel pointwise layer with 32-channel stages).
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4p;
template <int PATH, int D>
__global__ __launch_bounds__(256) void load_path_kernel(const unsigned char* __restrict__ src, unsigned src_bytes, int iters,
                                                         int pattern_policy, float* __restrict__ out) {
    extern __shared__ __attribute__((aligned(1024))) unsigned char lds[];
    const int pattern = pattern_policy & 1, policy = pattern_policy >> 2;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const unsigned voff = pattern ? (unsigned)((lane >> 2) * 2048 + (lane & 3) * 16) : (unsigned)lane * 16;
    const unsigned span = pattern ? 16u * 2048u : 1024u;                  // bytes of address space one instruction covers
    const unsigned wrap = src_bytes - span - 64u * 6u * 4u;               // keep every access inside the region
    unsigned base = ((unsigned)blockIdx.x * 2654435761u) % wrap & ~1023u;
    unsigned acc = 0;
    if constexpr (PATH == 0) {
        for (int it = 0; it < iters; it += D) {
            u32x4p v[D * 6];
#pragma unroll
            for (int k = 0; k < D * 6; ++k) {
                unsigned o = base + (pattern ? (unsigned)(k & 31) * 64u + (unsigned)(k >> 5) * span : (unsigned)(wave * D * 6 + k) * 1024u);
                if (o >= wrap) o -= wrap & ~1023u;
                const u32x4p* q = reinterpret_cast<const u32x4p*>(src + o + voff);
                // cache policy of the stream (pattern bits 2-3): 0 default, 1 nt (L2: stream / evict first), 2 sc0 sc1 (system scope),
                // 3 all three -- does a neighbour's L2 footprint matter to the kernels it runs beside?  (tools/cu_contention.py)
                if (policy == 0) v[k] = *q;
                else if (policy == 1) asm volatile("global_load_dwordx4 %0, %1, off nt" : "=v"(v[k]) : "v"(q) : "memory");
                else if (policy == 2) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[k]) : "v"(q) : "memory");
                else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1 nt" : "=v"(v[k]) : "v"(q) : "memory");
            }
            if (policy != 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int k = 0; k < D * 6; ++k) acc ^= v[k].x ^ v[k].w;
            base += (unsigned)D * 24576u;
            if (base >= wrap) base -= wrap & ~1023u;
        }
    } else {
        const v4i32 sr = make_srd(src, src_bytes);
        const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);
        auto issue = [&](int slot) {
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                unsigned o = base + (pattern ? (unsigned)k * 64u + (unsigned)wave * 512u : (unsigned)(wave * 6 + k) * 1024u);
                dma16(sr, voff, (int)o, lds0 + slot * 24576 + (wave * 6 + k) * 1024);
            }
            base += pattern ? 32768u : 24576u;
            if (base >= wrap) base -= wrap & ~1023u;
        };
        int islot = 0;
#pragma unroll
        for (int d = 0; d < D; ++d) { issue(islot); islot = islot + 1 == D + 1 ? 0 : islot + 1; }
        int slot = 0;
        for (int it = 0; it < iters; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(6 * (D - 1)) : "memory");
            __builtin_amdgcn_s_barrier();
            issue(islot);
            islot = islot + 1 == D + 1 ? 0 : islot + 1;
            acc ^= *reinterpret_cast<const unsigned*>(lds + slot * 24576 + threadIdx.x * 16);
            slot = slot + 1 == D + 1 ? 0 : slot + 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0x12345679u) out[0] = 1.f;            // keeps the loads alive
}

// One wave that keeps its stream busy for `ticks` of the 100 MHz wall clock (s_sleep between reads: no memory traffic, no
// matrix-core load): everything enqueued behind it starts when it ends.
__global__ __launch_bounds__(64) void delay_kernel(long long ticks) {
    const long long t0 = (long long)wall_clock64();
    while ((long long)wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(100);
}

}  // namespace

extern "C" int xmc_delay(int32_t microseconds, void* stream) {
    XMC_REQUIRE(microseconds >= 0 && microseconds <= 2000000);
    hipLaunchKernelGGL(delay_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), (long long)microseconds * 100);
    XMC_LAUNCH_RET();
}

static int xmc_internal_optin_probe() {
    static XmcLdsOptIn opt_in;
    return opt_in.ensure({reinterpret_cast<const void*>(&load_path_kernel<1, 2>), reinterpret_cast<const void*>(&load_path_kernel<1, 3>),
                          reinterpret_cast<const void*>(&load_path_kernel<1, 5>)}, 160 * 1024) ? XMC_OK : XMC_EINVAL;
}

// mode: bit 0 = path (0 registers, 1 LDS-DMA), bit 1 = pattern, bits 4-7 = depth D in {2, 3, 5}, bits 8-9 = cache policy of the
// register path's loads (0 default, 1 nt, 2 sc0 sc1, 3 sc0 sc1 nt).  Every workgroup moves
// iters * 24 KiB; src_bytes >= 1 MiB.
extern "C" int xmc_load_path_probe(int32_t mode, int32_t blocks, int32_t iters, const void* src, int64_t src_bytes, float* out,
                                   void* stream) {
    XMC_REQUIRE(src && out && blocks > 0 && iters > 0 && src_bytes >= (1 << 20) && src_bytes < 0xfffffff0ll);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int path = mode & 1, pattern = ((mode >> 1) & 1) | (((mode >> 8) & 3) << 2), depth = (mode >> 4) & 15;
    const unsigned char* p = static_cast<const unsigned char*>(src);
#define XMC_LP(P_, D_) hipLaunchKernelGGL((load_path_kernel<P_, D_>), dim3(blocks), dim3(256), (P_) ? ((D_) + 1) * 24576 : 0, s, p, (unsigned)src_bytes, iters, pattern, out)
    if (path == 0) {
        if (depth == 2) XMC_LP(0, 2); else if (depth == 3) XMC_LP(0, 3); else if (depth == 5) XMC_LP(0, 5); else return XMC_EINVAL;
    } else {
        if (xmc_internal_optin_probe() != XMC_OK) return XMC_EINVAL;
        if (depth == 2) XMC_LP(1, 2); else if (depth == 3) XMC_LP(1, 3); else if (depth == 5) XMC_LP(1, 5); else return XMC_EINVAL;
    }
#undef XMC_LP
    XMC_LAUNCH_RET();
}

extern "C" int xmc_mfma_rate_probe(int32_t mode, int32_t blocks, int32_t iters, float* out, void* stream) {
    XMC_REQUIRE(out && blocks > 0 && iters > 0 && mode >= 0 && mode < 4);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if ((mode & 1) == 0) hipLaunchKernelGGL((mfma_rate_kernel<0, 8>), dim3(blocks), dim3(256), 0, s, out, iters, mode >> 1);
    else hipLaunchKernelGGL((mfma_rate_kernel<1, 8>), dim3(blocks), dim3(256), 0, s, out, iters, mode >> 1);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_probe_layouts(float* out, void* stream) {
    XMC_REQUIRE(out);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out);
    XMC_LAUNCH_RET();
}


// ---- does v_pk_add_f32 with CROSSED halves compute what it should beside other kernels?  (round 4: the MX-fp8 kernel's residual
// add)  Every lane runs `iters` rounds of   p = s * (x, y);  acc = (acc.x + p.y, acc.y + p.x)   in the exact two
// instructions the compiler had formed (mode 0), or with the add uncrossed and the operands swapped by hand (mode 1), next to
// the same arithmetic in scalar instructions; mismatching rounds are counted per launch.  A control for the hazard question: a
// pure VALU kernel, no memory traffic, 2 workgroups per CU like the convolution.
__global__ __launch_bounds__(256, 2) void pk_add_cross_probe_kernel(int mode, int iters, float scale, unsigned* bad) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    const unsigned seed = (blockIdx.x * 256u + threadIdx.x) * 2654435761u;
    f2 acc = {0.f, 0.f};
    float rx = 0.f, ry = 0.f;
    unsigned mism = 0;
    f2 sc = {scale, scale};
    asm volatile("" : "+s"(sc));                     // the scale in an SGPR pair, as the epilogue had it
    for (int i = 0; i < iters; ++i) {
        const unsigned h = seed ^ (i * 0x9e3779b9u);
        f2 xy = {__uint_as_float(0x3f800000u | (h & 0x7fffffu)) - 1.5f, __uint_as_float(0x3f800000u | ((h >> 9) & 0x7fffffu)) - 1.5f};
        f2 p;
        if (mode == 0) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(sc), "v"(xy));
            asm volatile("v_pk_add_f32 %0, %0, %1 op_sel:[0,1] op_sel_hi:[1,0]" : "+v"(acc) : "v"(p));
        } else if (mode == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[0,1]" : "=v"(p) : "s"(sc), "v"(xy));
            f2 q = {p.y, p.x};
            asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(acc) : "v"(q));
        } else if (mode == 2) {
            // the forms the bf16 kernels' epilogues are full of: src0 broadcast from its HIGH half (op_sel:[1,0,0]) ...
            f2 hs = {0.f, scale};
            asm volatile("" : "+v"(hs));
            f2 q = {xy.y, xy.x};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0]" : "+v"(acc) : "v"(hs), "v"(q));
        } else if (mode == 4) {
            // a CROSSED operand of the fused form (sn_matvec_kernel had two): src0 = (s.y, s.x)
            f2 cs = {scale * 2.f, scale};
            asm volatile("" : "+v"(cs));
            f2 q = {xy.y, xy.x * 2.f};                   // acc.x += cs.y * q.x = scale * xy.y;  acc.y += cs.x * q.y... see below
            q.y = xy.x * 0.5f;                           // cs.x * q.y = 2 scale * 0.5 xy.x = scale * xy.x (exact: powers of two)
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "+v"(acc) : "v"(cs), "v"(q));
        } else {
            // ... and from its LOW half (op_sel_hi:[0,1,1]), scale in an SGPR pair
            f2 q = {xy.y, xy.x};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc) : "s"(sc), "v"(q));
        }
        const float px = __fmul_rn(scale, xy.x), py = __fmul_rn(scale, xy.y);
        if (mode >= 2) { rx = fmaf(scale, xy.y, rx); ry = fmaf(scale, xy.x, ry); }
        else { rx = __fadd_rn(rx, py); ry = __fadd_rn(ry, px); }
        mism += (__float_as_uint(acc.x) != __float_as_uint(rx)) | (__float_as_uint(acc.y) != __float_as_uint(ry));
        acc.x = rx; acc.y = ry;                      // resynchronise: every round is judged on its own
    }
    if (mism) atomicAdd(bad, mism);
}

// A neighbour made of ONE instruction class per bit of `mask` (tools/pk_add_probe.py --classes): which of the things the
// weight-gradient / pointwise kernels execute makes the crossed packed add of a co-resident wave go wrong?
//   1 MFMA 32x32x16 bf16   2 v_dot2c_f32_bf16   4 ds_read_b64_tr_b16   8 ds_write_b128 + ds_read_b128   16 v_pk_max_i16
//   32 LDS-DMA (buffer_load ... lds)   64 s_barrier   128 v_permlane32_swap   256 v_cvt_pk_bf16_f32   512 global loads
//   1 = two MFMAs on ONE accumulator (a dependent chain); 1024 = four independent accumulators; 2048 = 16x16x32, dependent
__global__ __launch_bounds__(256, 2) void class_neighbour_kernel(int mask, int iters, const unsigned char* __restrict__ src, unsigned src_bytes,
                                                                 float* out) {
    __shared__ __attribute__((aligned(1024))) unsigned char lds[16384];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc, acc1, acc2, acc3;
    f32x4 acc16 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int q = 0; q < 16; ++q) { acc[q] = 0.f; acc1[q] = 0.f; acc2[q] = 0.f; acc3[q] = 0.f; }
    bf16x8 fa, fb;
    {
        const uint4 ia = make_uint4(0x3f803f80u + lane, 0x3f803f80u, 0x3f003f80u, 0x3f803f00u);
        fa = __builtin_bit_cast(bf16x8, ia); fb = fa;
    }
    float d = 0.f;
    unsigned u = 0x01020304u * (unsigned)(lane + 1), w2 = 0x00010002u;
    const v4i32 sr = make_srd(src, src_bytes);
    const unsigned lds0 = (unsigned)reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) unsigned char*)lds);
    *reinterpret_cast<uint4*>(lds + threadIdx.x * 16) = make_uint4(u, u ^ 1, u ^ 2, u ^ 3);
    __syncthreads();
    unsigned goff = (blockIdx.x * 4096u) % (src_bytes - 8192u) & ~1023u;
    for (int it = 0; it < iters; ++it) {
        if (mask & 1) { acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0); acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc, 0, 0, 0); }
        if (mask & 1024) {                           // four INDEPENDENT accumulators (no MFMA waits for the one in front of it)
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc1, 0, 0, 0);
            acc2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc2, 0, 0, 0); acc3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb, fa, acc3, 0, 0, 0);
        }
        if (mask & 2048) acc16 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa, fb, acc16, 0, 0, 0);   // the 16x16 shape, dependent chain
        if (mask & 2) asm volatile("v_dot2c_f32_bf16 %0, %1, %2" : "+v"(d) : "v"(u), "v"(w2));
        if (mask & 4) {
            unsigned long long t;
            asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(t) : "v"(lds0 + (unsigned)(threadIdx.x & 255) * 8u) : "memory");
            u ^= (unsigned)t;
        }
        if (mask & 8) {
            const uint4 t = *reinterpret_cast<const uint4*>(lds + ((threadIdx.x * 16 + it * 16) & 4095));
            *reinterpret_cast<uint4*>(lds + 4096 + threadIdx.x * 16) = t;
            u ^= t.x;
        }
        if (mask & 16) asm volatile("v_pk_max_i16 %0, %0, %1" : "+v"(u) : "v"(w2));
        if (mask & 32) { dma16(sr, (unsigned)lane * 16, (int)goff, lds0 + 8192 + wave * 1024); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
        if (mask & 64) __builtin_amdgcn_s_barrier();
        if (mask & 128) { const auto r = __builtin_amdgcn_permlane32_swap(u, w2, false, false); u = r[0]; w2 = r[1] | 1u; }
        if (mask & 256) u ^= pack_bf2(d + 1.f, (float)it);
        if (mask & 512) u ^= *reinterpret_cast<const unsigned*>(src + goff + threadIdx.x * 4);
        goff += 1024u;
        if (goff >= src_bytes - 8192u) goff = 0;
    }
    if (u == 0x12345679u || d == 1.2345f || acc[0] + acc1[0] + acc2[0] + acc3[0] + acc16[0] == 1.2345f) out[0] = 1.f;
}

extern "C" int xmc_class_neighbour(int32_t mask, int32_t blocks, int32_t iters, const void* src, int64_t src_bytes, float* out, void* stream) {
    XMC_REQUIRE(src && out && blocks > 0 && iters > 0 && src_bytes >= (1 << 20) && src_bytes < (1ll << 32));
    hipLaunchKernelGGL(class_neighbour_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mask, iters,
                       static_cast<const unsigned char*>(src), (unsigned)src_bytes, out);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_pk_add_cross_probe(int32_t mode, int32_t blocks, int32_t iters, uint32_t* bad, void* stream) {
    XMC_REQUIRE(bad && blocks > 0 && iters > 0 && mode >= 0 && mode <= 4);
    hipLaunchKernelGGL(pk_add_cross_probe_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), mode, iters, 0.25f, bad);
    XMC_LAUNCH_RET();
}
