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

}  // namespace

extern "C" int xmc_probe_layouts(float* out, void* stream) {
    XMC_REQUIRE(out);
    hipLaunchKernelGGL(probe_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out);
    XMC_LAUNCH_RET();
}
