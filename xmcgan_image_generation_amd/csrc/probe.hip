// Hardware-layout probes used by tests/test_gpu_kernels.py: they pin the two gfx950 facts the
// MFMA kernels rely on -- the C/D register map of v_mfma_f32_32x32x16_bf16 and the lane
// transposition performed by ds_read_b64_tr_b16 -- against what the kernels assume.
#include "common.h"

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

}  // namespace

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
