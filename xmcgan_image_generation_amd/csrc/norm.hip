// BatchNorm statistics and the fused (Local)ConditionalBatchNorm affine + ReLU, forward and
// backward, plus the generic middle-axis reduction.  All of these are HBM-bound streaming
// kernels: 16-byte vector loads (8 bf16 / 4 f32 per lane), float32 arithmetic, channel
// partial sums combined in LDS (ds_add_f32) and one global atomic per channel per block.
#include <type_traits>

#include <cstdlib>
#include "common.h"

namespace {

// VE-wide accessor: VE == Vec<T>::N uses 16-byte vectors, VE == 1 is the scalar fallback
// (channel counts that are not a multiple of the vector width, e.g. RGB).
template <typename T, int VE> struct Acc {
    static __device__ __forceinline__ void load(const T* p, float* f) {
        Vec<T> v;
        v.load(p);
        v.get(f);
    }
    static __device__ __forceinline__ void store(T* p, const float* f) {
        Vec<T> v;
        v.set(f);
        v.store(p);
    }
};
template <typename T> struct Acc<T, 1> {
    static __device__ __forceinline__ void load(const T* p, float* f) { f[0] = to_f<T>(*p); }
    static __device__ __forceinline__ void store(T* p, const float* f) { *p = from_f<T>(f[0]); }
};

constexpr int MAXC = 4096;   // LDS channel accumulators

// ------------------------------------------------------------------ y[a][c] += scale*sum_r f(x)
template <typename T, int VE>
__global__ __launch_bounds__(256) void reduce_mid_kernel(const T* __restrict__ x, float* __restrict__ y,
                                                         long long R, int Cfull, int relu, float scale,
                                                         int rows_per_block, int partial_rows, int cwin) {
    // channel window [c0, c0 + C) of the full row (blockIdx.z), cwin <= MAXC channels at a time
    __shared__ float acc[MAXC];
    const int tid = threadIdx.x;
    const int c0 = blockIdx.z * cwin;
    const int C = min(cwin, Cfull - c0);
    const int CV = C / VE;
    const int CVP = CV < 256 ? CV : 256;
    const int RP = 256 / CVP;
    const int r0 = tid / CVP, cv0 = tid % CVP;
    for (int c = tid; c < C; c += 256) acc[c] = 0.f;
    __syncthreads();
    const long long a = blockIdx.y;
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(R, rb + rows_per_block);
    const T* xa = x + a * R * Cfull + c0;
    if (r0 < RP) {
        for (int cv = cv0; cv < CV; cv += CVP) {
            float s[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) s[e] = 0.f;
            // eight rows per trip, their loads in flight together (rows past the end re-read row rb and are not added): one
            // load per trip left a 56-row bias gradient with 200+ dependent L2 round trips per thread
            for (long long r = rb + r0; r < re; r += 8ll * RP) {
                float f[8][VE];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const long long rr = r + (long long)u * RP;
                    Acc<T, VE>::load(xa + (rr < re ? rr : rb) * Cfull + cv * VE, f[u]);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u) {            // no control flow between the loads and the adds (a `break` here put
                    const bool ok = r + (long long)u * RP < re;          // every load back behind its own branch + vmcnt(0))
#pragma unroll
                    for (int e = 0; e < VE; ++e) s[e] += ok ? (relu ? fmaxf(f[u][e], 0.f) : f[u][e]) : 0.f;
                }
            }
            if (partial_rows && CV <= 256) {            // one owner per (row group, channel vector): fixed-order sum below
#pragma unroll
                for (int e = 0; e < VE; ++e) acc[r0 * C + cv * VE + e] = s[e];
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) atomicAdd(&acc[cv * VE + e], s[e]);   // (CV > 256: RP == 1, single writer)
            }
        }
    }
    __syncthreads();
    if (partial_rows) {
        // row blockIdx.x of y[gridDim.x][A][Cfull]: unscaled partial sums, reduced in a fixed order by reduce_rows_kernel
        for (int c = tid; c < C; c += 256) {
            float t = acc[c];
            if (CV <= 256)
                for (int r = 1; r < RP; ++r) t += acc[r * C + c];
            y[((long long)blockIdx.x * gridDim.y + a) * Cfull + c0 + c] = t;
        }
    } else {
        for (int c = tid; c < C; c += 256) atomicAdd(&y[a * Cfull + c0 + c], scale * acc[c]);
    }
}

// ------------------------------------------------------------------------- BN batch statistics
template <typename T, int VE>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ x, float* __restrict__ sums,
                                                       long long P, int C, int rows_per_block, int partial_rows) {
    __shared__ float acc[2 * MAXC];
    const int tid = threadIdx.x;
    const int CV = C / VE;
    const int CVP = CV < 256 ? CV : 256;
    const int RP = 256 / CVP;
    const int r0 = tid / CVP, cv0 = tid % CVP;
    for (int c = tid; c < 2 * C; c += 256) acc[c] = 0.f;
    __syncthreads();
    const long long rb = (long long)blockIdx.x * rows_per_block;
    const long long re = min(P, rb + rows_per_block);
    if (r0 < RP) {
        for (int cv = cv0; cv < CV; cv += CVP) {
            float s[VE], q[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) s[e] = q[e] = 0.f;
            long long r = rb + r0;
            for (; r + 3 * RP < re; r += 4 * RP) {      // 4 independent 16-byte loads in flight per thread
                float f[4][VE];
#pragma unroll
                for (int u = 0; u < 4; ++u) Acc<T, VE>::load(x + (r + u * RP) * C + cv * VE, f[u]);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int e = 0; e < VE; ++e) { s[e] += f[u][e]; q[e] += f[u][e] * f[u][e]; }
            }
            for (; r < re; r += RP) {
                float f[VE];
                Acc<T, VE>::load(x + r * C + cv * VE, f);
#pragma unroll
                for (int e = 0; e < VE; ++e) { s[e] += f[e]; q[e] += f[e] * f[e]; }
            }
            if (partial_rows && CV <= 256) {
                // each thread owns one channel vector here: park its sums in row r0 of an [RP][2C] LDS table and
                // add the rows in a fixed order below (LDS float atomics would make the result order-dependent)
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    acc[r0 * 2 * C + cv * VE + e] = s[e];
                    acc[r0 * 2 * C + C + cv * VE + e] = q[e];
                }
            } else {
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    atomicAdd(&acc[cv * VE + e], s[e]);          // RP == 1 when CV > 256: a single writer per slot
                    atomicAdd(&acc[C + cv * VE + e], q[e]);
                }
            }
        }
    }
    __syncthreads();
    // partial_rows: every workgroup writes its own row of sums[gridDim.x][2C] (no same-address global atomics:
    // 2048 workgroups adding into 2C addresses serialise in L2 -- 60 us of a 90 us launch -- and the row-wise
    // reduction in reduce_rows_kernel has a fixed order, i.e. the statistics are bit-reproducible)
    if (partial_rows) {
        for (int c = tid; c < 2 * C; c += 256) {
            float t = acc[c];
            if (CV <= 256)
                for (int r = 1; r < RP; ++r) t += acc[r * 2 * C + c];
            sums[(long long)blockIdx.x * 2 * C + c] = t;
        }
    } else {
        for (int c = tid; c < 2 * C; c += 256) atomicAdd(&sums[c], acc[c]);
    }
}

// out[c] = sum_r part[r][c] in a fixed order: 16 channels x 16 row groups per workgroup, combined through LDS.
// FINALIZE: c < C channels, part rows hold [sum, sum of squares] -> mean, rstd and the running statistics
// (flax BatchNorm, xmc_net.py:192-201); otherwise the plain column sums of `width` columns.
template <bool FINALIZE>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const float* __restrict__ part, int rows, int width,
                                                          float* __restrict__ out, float* mean, float* rstd,
                                                          float* run_mean, float* run_var, float inv_p, int C,
                                                          float eps, float momentum, int upd) {
    __shared__ float red[2][16][17];
    const int cl = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int c = blockIdx.x * 16 + cl;
    const int ncol = FINALIZE ? C : width;
    float s = 0.f, q = 0.f;
    if (c < ncol) {
        // eight rows' loads in flight, added in the SAME order as one at a time (round 5: the loop of dependent-looking loads ran at
        // one memory latency per row -- 32 rows per thread = 10 us of a 12 us launch, 47 launches per step)
        int r = rg;
        for (; r + 7 * 16 < rows; r += 8 * 16) {
            float ps[8], pq[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                ps[u] = part[(long long)(r + 16 * u) * width + c];
                pq[u] = FINALIZE ? part[(long long)(r + 16 * u) * width + C + c] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) { s += ps[u]; if (FINALIZE) q += pq[u]; }
        }
        for (; r < rows; r += 16) {
            s += part[(long long)r * width + c];
            if (FINALIZE) q += part[(long long)r * width + C + c];
        }
    }
    red[0][rg][cl] = s;
    red[1][rg][cl] = q;
    __syncthreads();
    if (rg == 0 && c < ncol) {
        float ts = 0.f, tq = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) { ts += red[0][k][cl]; tq += red[1][k][cl]; }
        if (FINALIZE) {
            const float m = ts * inv_p;
            const float var = tq * inv_p - m * m;          // biased: E[x^2] - E[x]^2 (flax 0.3.3)
            mean[c] = m;
            rstd[c] = rsqrtf(var + eps);
            if (upd) {
                run_mean[c] = momentum * run_mean[c] + (1.f - momentum) * m;
                run_var[c] = momentum * run_var[c] + (1.f - momentum) * var;
            }
        } else {
            out[c] = (upd ? out[c] : 0.f) + inv_p * ts;       // plain column sums: upd = accumulate, inv_p = scale
        }
    }
}

__global__ void bn_finalize_kernel(const float* __restrict__ sums, float* mean, float* rstd, float* run_mean,
                                   float* run_var, float inv_p, int C, float eps, float momentum, int upd) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    const float m = sums[c] * inv_p;
    const float var = sums[C + c] * inv_p - m * m;     // biased: E[x^2] - E[x]^2 (flax 0.3.3)
    mean[c] = m;
    rstd[c] = rsqrtf(var + eps);
    if (upd) {
        run_mean[c] = momentum * run_mean[c] + (1.f - momentum) * m;
        run_var[c] = momentum * run_var[c] + (1.f - momentum) * var;
    }
}

__global__ void bn_from_running_kernel(const float* rm, const float* rv, float* mean, float* rstd, int C,
                                       float eps) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    mean[c] = rm[c];
    rstd[c] = rsqrtf(rv[c] + eps);
}

// ------------------------------------------------------------ y = relu(x_hat*(gamma+1)+beta)
struct CbnGeo {
    int N, H, W, C, hc, relu;
    int log2_w, log2_hw, sh;     // sh = log2(H / hc)
    int cs;                      // floats between consecutive conditioning cells in gamma / beta (>= C):
                                 // gamma and beta are the two halves of one (cells, 2C) conv / dense output
};

__device__ __forceinline__ int cbn_cell(const CbnGeo& g, long long pix) {
    const int n = (int)(pix >> g.log2_hw), rem = (int)(pix & ((1 << g.log2_hw) - 1));
    const int y = rem >> g.log2_w, x = rem & (g.W - 1);
    return (n * g.hc + (y >> g.sh)) * g.hc + (x >> g.sh);
}

// gamma / beta (and their gradients) come as float32 or -- round 5, the LOCAL sites of the bf16 mode, as the reference's
// nn.Conv(dtype=bfloat16) produces them (xmcgan/libml/layers.py:261-273) -- as bf16: GT.  VE consecutive elements; the bf16 form
// is one 16-byte access (the launcher checks the alignment), the float32 form leaves the merging to the compiler as before.
template <typename GT, int VE>
__device__ __forceinline__ void gb_load(const GT* __restrict__ p, float* f) {
    if constexpr (std::is_same<GT, bf16_t>::value && VE == 8) {
        Vec<bf16_t> v; v.load(p); v.get(f);
    } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) f[e] = to_f<GT>(p[e]);
    }
}
template <typename GT, int VE>
__device__ __forceinline__ void gb_store(GT* __restrict__ p, const float* f) {
    if constexpr (std::is_same<GT, bf16_t>::value && VE == 8) {
        Vec<bf16_t> v; v.set(f); v.store(p);
    } else {
#pragma unroll
        for (int e = 0; e < VE; ++e) p[e] = from_f<GT>(f[e]);
    }
}

// y8 != nullptr (bf16, VE = 8, C % 64 == 0 only): also write y as MX-fp8 packets for the 3x3 convolution that consumes it
// (conv_stream_mx8.hip: [pixel][C / 64][80], 64 e4m3 elements + the two e8m0 block scales per packet) -- byte for byte what
// xmc_mx8_quantize(y, relu = 0) writes, without its pass over the tensor.  Four consecutive lanes hold one 32-channel block.
template <typename T, int VE, typename GT = float>
__global__ __launch_bounds__(256) void cbn_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                      const float* __restrict__ rstd,
                                                      const GT* __restrict__ gamma,
                                                      const GT* __restrict__ beta, T* __restrict__ y,
                                                      const CbnGeo g, long long nvec, unsigned char* __restrict__ y8 = nullptr,
                                                      unsigned mx_rnd = XMC_MX_RND_NEXT_BINADE) {
    const int CV = g.C / VE;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long pix = v / CV;
        const int c = (int)(v - pix * CV) * VE;
        const long long cb = (long long)cbn_cell(g, pix) * g.cs + c;
        float f[VE], o[VE], ga[VE], be[VE];
        Acc<T, VE>::load(x + pix * g.C + c, f);
        gb_load<GT, VE>(gamma + cb, ga);
        gb_load<GT, VE>(beta + cb, be);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float a = rstd[c + e] * (ga[e] + 1.f);
            float u = (f[e] - mean[c + e]) * a + be[e];
            o[e] = g.relu ? fmaxf(u, 0.f) : u;
        }
        Acc<T, VE>::store(y + pix * g.C + c, o);
        if constexpr (std::is_same<T, bf16_t>::value && VE == 8) {
            if (y8) {
                float amax = 0.f;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    o[e] = bf2f(f2bf(o[e]));                     // what the bf16 tensor holds
                    amax = fmaxf(amax, fabsf(o[e]));
                }
                amax = fmaxf(amax, __shfl_xor(amax, 1));
                amax = fmaxf(amax, __shfl_xor(amax, 2));
                const unsigned sb = xmc_mx_scale_byte(amax, mx_rnd);
                const float is = __uint_as_float((254u - sb) << 23);
                unsigned char* pk = y8 + (pix * (g.C >> 6) + (c >> 6)) * 80;
                *reinterpret_cast<uint2*>(pk + (c & 63)) = make_uint2(xmc_pack_fp8x4(o[0], o[1], o[2], o[3], is), xmc_pack_fp8x4(o[4], o[5], o[6], o[7], is));
                if ((threadIdx.x & 3) == 0) pk[64 + ((c >> 5) & 1)] = (unsigned char)sb;
            }
        }
    }
}

// pass 1 of backward: one thread per (conditioning cell, channel vector): it walks the f x f pixels of
// its cell with private accumulators and writes dgamma / dbeta once (exclusive, no atomics).
// Consecutive threads own consecutive channel vectors of one cell, so each pixel row is read coalesced.
template <typename T, int VE, typename GT = float>
__global__ __launch_bounds__(256) void cbn_bwd_cells_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            const GT* __restrict__ gamma,
                                                            const GT* __restrict__ beta,
                                                            GT* __restrict__ dgamma, GT* __restrict__ dbeta,
                                                            const CbnGeo g, long long nwork) {
    const int C = g.C, CV = C / VE;
    const int f = 1 << g.sh, npix = f * f;
    for (long long wk = (long long)blockIdx.x * 256 + threadIdx.x; wk < nwork; wk += (long long)gridDim.x * 256) {
        const long long cell = wk / CV;
        const int c = (int)(wk - cell * CV) * VE;
        const int cx = (int)(cell % g.hc), cy = (int)((cell / g.hc) % g.hc), n = (int)(cell / ((long long)g.hc * g.hc));
        const long long cbase = cell * g.cs + c;
        float sg[VE], sb[VE], a[VE], bt[VE], mu[VE], rs[VE];
        gb_load<GT, VE>(gamma + cbase, a);
        gb_load<GT, VE>(beta + cbase, bt);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            sg[e] = sb[e] = 0.f;
            mu[e] = mean[c + e];
            rs[e] = rstd[c + e];
            a[e] = a[e] + 1.f;
        }
        // npix is a power of four >= 1: groups of 4 pixels keep 8 independent 16-byte loads in flight per thread
        // (the global-cBN layers have only N * C / 8 threads, each walking 64-256 pixels)
        for (int q0 = 0; q0 < npix; q0 += 4) {
            float fx[4][VE], fd[4][VE];
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const int q = min(q0 + u4, npix - 1);
                const int iy = q >> g.sh, ix = q & (f - 1);
                const long long pix = ((long long)(n * g.H + (cy << g.sh) + iy) << g.log2_w) + (cx << g.sh) + ix;
                Acc<T, VE>::load(x + pix * C + c, fx[u4]);
                Acc<T, VE>::load(dy + pix * C + c, fd[u4]);
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                const bool okq = q0 + u4 < npix;
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float xh = (fx[u4][e] - mu[e]) * rs[e];
                    const float u = xh * a[e] + bt[e];
                    const float gg = (okq && (!g.relu || u > 0.f)) ? fd[u4][e] : 0.f;
                    sb[e] += gg;
                    sg[e] += gg * xh;
                }
            }
        }
        gb_store<GT, VE>(dgamma + cbase, sg);
        gb_store<GT, VE>(dbeta + cbase, sb);
    }
}

// The same sums for cells of F x F pixels (F = 4, 8: the local cBN layers at 64^2 and 128^2) with one thread per (cell, channel
// vector, ROW of the cell): the thread-per-cell kernel above walks 16-64 pixels per thread as a chain of 4-16 dependent trips
// with 2.6 workgroups per CU -- latency-bound at 3 TB/s.  Here a thread takes F pixels (all loads issued together), the F row
// sums of a cell meet in LDS and are added in row order by the row-0 thread (fixed order: bit-reproducible; no atomics).
template <typename T, int VE, int F, typename GT = float>
__global__ __launch_bounds__(256) void cbn_bwd_cells_rows_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd,
                                                                 const GT* __restrict__ gamma,
                                                                 const GT* __restrict__ beta,
                                                                 GT* __restrict__ dgamma, GT* __restrict__ dbeta,
                                                                 const CbnGeo g, long long npairs) {
    constexpr int G = 256 / F;                       // (cell, channel vector) pairs per workgroup
    __shared__ float red[F][2 * VE][G];              // pair index fastest: conflict-free columns
    const int t = threadIdx.x, j = t / G, p = t - j * G;
    const long long pair = (long long)blockIdx.x * G + p;
    const bool live = pair < npairs;
    const unsigned CV = g.C / VE;
    const unsigned pr = (unsigned)(live ? pair : npairs - 1);
    const unsigned cell = pr / CV;
    const int c = (int)(pr - cell * CV) * VE;
    const int cx = (int)(cell % g.hc), cy = (int)((cell / g.hc) % g.hc), n = (int)(cell / ((unsigned)g.hc * g.hc));
    const long long cbase = (long long)cell * g.cs + c;
    const long long pix0 = ((long long)(n * g.H + cy * F + j) << g.log2_w) + cx * F;
    float sg[VE], sb[VE], a[VE], bt[VE], mu[VE], rs[VE];
    gb_load<GT, VE>(gamma + cbase, a);
    gb_load<GT, VE>(beta + cbase, bt);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sg[e] = sb[e] = 0.f;
        mu[e] = mean[c + e];
        rs[e] = rstd[c + e];
        a[e] = a[e] + 1.f;
    }
#pragma unroll
    for (int q0 = 0; q0 < F; q0 += 4) {
        float fx[4][VE], fd[4][VE];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            Acc<T, VE>::load(x + (pix0 + q0 + u) * g.C + c, fx[u]);
            Acc<T, VE>::load(dy + (pix0 + q0 + u) * g.C + c, fd[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float xh = (fx[u][e] - mu[e]) * rs[e];
                const float uu = xh * a[e] + bt[e];
                const float gg = (!g.relu || uu > 0.f) ? fd[u][e] : 0.f;
                sb[e] += gg;
                sg[e] += gg * xh;
            }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) { red[j][e][p] = sg[e]; red[j][VE + e][p] = sb[e]; }
    __syncthreads();
    if (j == 0 && live) {
        float og[VE], ob[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float tg = red[0][e][p], tb = red[0][VE + e][p];
#pragma unroll
            for (int r = 1; r < F; ++r) { tg += red[r][e][p]; tb += red[r][VE + e][p]; }
            og[e] = tg; ob[e] = tb;
        }
        gb_store<GT, VE>(dgamma + cbase, og);
        gb_store<GT, VE>(dbeta + cbase, ob);
    }
}

// The same sums for FEW, LARGE cells (the global cBN layers: one cell per image -- 56 x C / 8 threads, each walking up to
// 1,024 pixels: 21 workgroups took 200 us for 44 MB): one workgroup = one cell x 32 channel vectors, its 8 thread rows
// take every 8th pixel and are added through LDS in a fixed order (no atomics, bit-reproducible).
template <typename T, int VE, typename GT = float>
__global__ __launch_bounds__(256) void cbn_bwd_cells_split_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd,
                                                                  const GT* __restrict__ gamma,
                                                                  const GT* __restrict__ beta,
                                                                  GT* __restrict__ dgamma, GT* __restrict__ dbeta,
                                                                  const CbnGeo g, int cgroups) {
    __shared__ float red[8][32][2 * VE];
    const int C = g.C, CV = C / VE;
    const int f = 1 << g.sh, npix = f * f;
    const long long cell = blockIdx.x / cgroups;
    const int cg = blockIdx.x - (int)(cell * cgroups);
    const int tx = threadIdx.x & 31, ps = threadIdx.x >> 5;
    const int cvi = cg * 32 + tx;
    const bool live = cvi < CV;
    const int c = (live ? cvi : 0) * VE;
    const int cx = (int)(cell % g.hc), cy = (int)((cell / g.hc) % g.hc), n = (int)(cell / ((long long)g.hc * g.hc));
    const long long cbase = cell * g.cs + c;
    float sg[VE], sb[VE], a[VE], bt[VE], mu[VE], rs[VE];
    gb_load<GT, VE>(gamma + cbase, a);
    gb_load<GT, VE>(beta + cbase, bt);
#pragma unroll
    for (int e = 0; e < VE; ++e) {
        sg[e] = sb[e] = 0.f;
        mu[e] = mean[c + e];
        rs[e] = rstd[c + e];
        a[e] = a[e] + 1.f;
    }
    // pixel q = 8 k + ps of the cell; four of them (eight 16-byte loads) in flight per thread
    for (int q0 = ps; q0 < npix; q0 += 32) {
        float fx[4][VE], fd[4][VE];
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const int q = min(q0 + 8 * u4, npix - 1);
            const int iy = q >> g.sh, ix = q & (f - 1);
            const long long pix = ((long long)(n * g.H + (cy << g.sh) + iy) << g.log2_w) + (cx << g.sh) + ix;
            Acc<T, VE>::load(x + pix * C + c, fx[u4]);
            Acc<T, VE>::load(dy + pix * C + c, fd[u4]);
        }
        asm volatile("" ::: "memory");
#pragma unroll
        for (int u4 = 0; u4 < 4; ++u4) {
            const bool okq = q0 + 8 * u4 < npix;
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float xh = (fx[u4][e] - mu[e]) * rs[e];
                const float u = xh * a[e] + bt[e];
                const float gg = (okq && (!g.relu || u > 0.f)) ? fd[u4][e] : 0.f;
                sb[e] += gg;
                sg[e] += gg * xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < VE; ++e) { red[ps][tx][e] = sg[e]; red[ps][tx][VE + e] = sb[e]; }
    __syncthreads();
    if (ps == 0 && live) {
        float og[VE], ob[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            float tg = red[0][tx][e], tb = red[0][tx][VE + e];
            for (int k = 1; k < 8; ++k) { tg += red[k][tx][e]; tb += red[k][tx][VE + e]; }
            og[e] = tg; ob[e] = tb;
        }
        gb_store<GT, VE>(dgamma + cbase, og);
        gb_store<GT, VE>(dbeta + cbase, ob);
    }
}

template <typename GT>
__global__ __launch_bounds__(256) void cbn_bwd_sums_kernel(const GT* __restrict__ gamma,
                                                           const GT* __restrict__ dgamma,
                                                           const GT* __restrict__ dbeta, float* __restrict__ s,
                                                           long long cells, int C, int cs, int cells_per_block) {
    const long long cb = (long long)blockIdx.x * cells_per_block;
    const long long ce = min(cells, cb + cells_per_block);
    for (int c = threadIdx.x; c < C; c += 256) {
        float s1 = 0.f, s2 = 0.f;
        // four cells per trip, their twelve loads in flight together (cells past the end re-read the last one and add
        // nothing); same order of the sums
        for (long long k = cb; k < ce; k += 4) {
            float ga[4], db4[4], dg4[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const long long o = min(k + u, ce - 1) * cs + c;
                ga[u] = to_f<GT>(gamma[o]); db4[u] = to_f<GT>(dbeta[o]); dg4[u] = to_f<GT>(dgamma[o]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool ok = k + u < ce;
                const float a = ga[u] + 1.f;
                s1 += ok ? a * db4[u] : 0.f;
                s2 += ok ? a * dg4[u] : 0.f;
            }
        }
        s[(long long)blockIdx.x * 2 * C + c] = s1;           // this workgroup's row of the [blocks][2C] partials
        s[(long long)blockIdx.x * 2 * C + C + c] = s2;
    }
}

template <typename T, int VE, typename GT = float>
__global__ __launch_bounds__(256) void cbn_bwd_dx_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                         const float* __restrict__ mean,
                                                         const float* __restrict__ rstd,
                                                         const GT* __restrict__ gamma,
                                                         const GT* __restrict__ beta,
                                                         const float* __restrict__ s, T* __restrict__ dx,
                                                         const CbnGeo g, long long nvec, float inv_p) {
    const int CV = g.C / VE;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long pix = v / CV;
        const int c = (int)(v - pix * CV) * VE;
        const long long cb = (long long)cbn_cell(g, pix) * g.cs + c;
        float fx[VE], fd[VE], o[VE], ga[VE], be[VE];
        Acc<T, VE>::load(x + pix * g.C + c, fx);
        Acc<T, VE>::load(dy + pix * g.C + c, fd);
        gb_load<GT, VE>(gamma + cb, ga);
        gb_load<GT, VE>(beta + cb, be);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const float rs = rstd[c + e], a = ga[e] + 1.f;
            const float xh = (fx[e] - mean[c + e]) * rs;
            const float u = xh * a + be[e];
            const float gg = (!g.relu || u > 0.f) ? fd[e] : 0.f;
            o[e] = rs * (gg * a - s[c + e] * inv_p - xh * s[g.C + c + e] * inv_p);
        }
        Acc<T, VE>::store(dx + pix * g.C + c, o);
    }
}

// "Run" forms of the two elementwise passes for cells at least R pixels wide (R = 8 or 4): one thread = one channel vector of
// R CONSECUTIVE pixels of one row -- all inside one conditioning cell, so mean / rstd / gamma / beta (and the two backward
// sums) are loaded ONCE per R pixels.  The pixel-per-thread kernels above fetch 64-128 bytes of those per 16 bytes of x
// through the vector L1 and sit at 3.7 TB/s on it (issue-stalled 62 %, profiles/r04_pmc_sq_per_kernel.txt), not on HBM.
// Same arithmetic per element, in the same order: bit-identical outputs.
template <typename T, int VE, int R, typename GT = float>
__global__ __launch_bounds__(256) void cbn_fwd_run_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd,
                                                          const GT* __restrict__ gamma,
                                                          const GT* __restrict__ beta, T* __restrict__ y,
                                                          const CbnGeo g, long long nwork) {
    const unsigned CV = g.C / VE;
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nwork; w += (long long)gridDim.x * 256) {
        const unsigned seg = (unsigned)w / CV;
        const int c = (int)((unsigned)w - seg * CV) * VE;
        const long long pix0 = (long long)seg * R;
        const long long cb = (long long)cbn_cell(g, pix0) * g.cs + c;
        float fx[R][VE], a[VE], mu[VE], bt[VE];
#pragma unroll
        for (int r = 0; r < R; ++r) Acc<T, VE>::load(x + (pix0 + r) * g.C + c, fx[r]);
        gb_load<GT, VE>(gamma + cb, a);
        gb_load<GT, VE>(beta + cb, bt);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            a[e] = rstd[c + e] * (a[e] + 1.f);
            mu[e] = mean[c + e];
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float o[VE];
#pragma unroll
            for (int e = 0; e < VE; ++e) {
                const float u = (fx[r][e] - mu[e]) * a[e] + bt[e];
                o[e] = g.relu ? fmaxf(u, 0.f) : u;
            }
            Acc<T, VE>::store(y + (pix0 + r) * g.C + c, o);
        }
    }
}

template <typename T, int VE, int R, typename GT = float>
__global__ __launch_bounds__(256) void cbn_bwd_dx_run_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                             const float* __restrict__ mean,
                                                             const float* __restrict__ rstd,
                                                             const GT* __restrict__ gamma,
                                                             const GT* __restrict__ beta,
                                                             const float* __restrict__ s, T* __restrict__ dx,
                                                             const CbnGeo g, long long nwork, float inv_p) {
    const unsigned CV = g.C / VE;
    for (long long w = (long long)blockIdx.x * 256 + threadIdx.x; w < nwork; w += (long long)gridDim.x * 256) {
        const unsigned seg = (unsigned)w / CV;
        const int c = (int)((unsigned)w - seg * CV) * VE;
        const long long pix0 = (long long)seg * R;
        const long long cb = (long long)cbn_cell(g, pix0) * g.cs + c;
        float rs[VE], a[VE], mu[VE], bt[VE], k1[VE], k2[VE];
        gb_load<GT, VE>(gamma + cb, a);
        gb_load<GT, VE>(beta + cb, bt);
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            rs[e] = rstd[c + e];
            a[e] = a[e] + 1.f;
            mu[e] = mean[c + e];
            k1[e] = s[c + e];
            k2[e] = s[g.C + c + e];
        }
#pragma unroll
        for (int r0 = 0; r0 < R; r0 += 4) {              // 8 loads in flight
            float fx[4][VE], fd[4][VE];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                Acc<T, VE>::load(x + (pix0 + r0 + r) * g.C + c, fx[r]);
                Acc<T, VE>::load(dy + (pix0 + r0 + r) * g.C + c, fd[r]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float o[VE];
#pragma unroll
                for (int e = 0; e < VE; ++e) {
                    const float xh = (fx[r][e] - mu[e]) * rs[e];
                    const float u = xh * a[e] + bt[e];
                    const float gg = (!g.relu || u > 0.f) ? fd[r][e] : 0.f;
                    o[e] = rs[e] * (gg * a[e] - k1[e] * inv_p - xh * k2[e] * inv_p);
                }
                Acc<T, VE>::store(dx + (pix0 + r0 + r) * g.C + c, o);
            }
        }
    }
}

// cells at least 4 pixels wide take the run kernels (R = 8 from 8 pixels on); XMC_CBN_RUN=0: the pixel-per-thread kernels (A/B)
inline int cbn_run_len(const CbnGeo& g, bool vec, long long nvec) {
    const bool on = xmc_internal_tuning(XMC_TUNE_CBN_RUN) != 0;
    if (!on || !vec || nvec >= (1ll << 31)) return 0;
    const int f = 1 << g.sh;
    return f >= 8 ? 8 : f >= 4 ? 4 : 0;
}

inline bool vec_ok(int c, int dtype, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr) {
    const int ve = dtype == XMC_BF16 ? 8 : 4;
    auto al = [](const void* p) { return p == nullptr || ((uintptr_t)p % 16) == 0; };
    return (c % ve) == 0 && al(p0) && al(p1) && al(p2);
}

inline int make_geo(CbnGeo& g, int n, int h, int w, int c, int hc, int relu, int cs) {
    g.N = n; g.H = h; g.W = w; g.C = c; g.hc = hc; g.relu = relu;
    g.cs = cs;
    if (cs < c) return XMC_EINVAL;
    g.log2_w = ilog2_exact(w);
    const int l2h = ilog2_exact(h), l2c = ilog2_exact(hc);
    if (g.log2_w < 0 || l2h < 0 || l2c < 0 || h != w || hc > h) return XMC_EINVAL;
    g.log2_hw = g.log2_w + l2h;
    g.sh = l2h - l2c;
    return XMC_OK;
}

}  // namespace

#define XMC_DISPATCH_VE(dtype, vec, KERNEL, grid, block, s, ...)                                     \
    do {                                                                                              \
        if ((dtype) == XMC_BF16) {                                                                    \
            if (vec) hipLaunchKernelGGL((KERNEL<bf16_t, 8>), grid, block, 0, s, __VA_ARGS__);         \
            else hipLaunchKernelGGL((KERNEL<bf16_t, 1>), grid, block, 0, s, __VA_ARGS__);             \
        } else {                                                                                      \
            if (vec) hipLaunchKernelGGL((KERNEL<float, 4>), grid, block, 0, s, __VA_ARGS__);          \
            else hipLaunchKernelGGL((KERNEL<float, 1>), grid, block, 0, s, __VA_ARGS__);              \
        }                                                                                             \
    } while (0)

static void reduce_mid_geometry(long long a, long long r, long long* rpb, long long* blocks) {
    long long b = (2048 + a - 1) / a;                         // ~2048 workgroups in total
    long long k = (r + b - 1) / b;
    if (k < 64) k = r < 64 ? r : 64;
    *rpb = k;
    *blocks = (r + k - 1) / k;
}

extern "C" int64_t xmc_reduce_mid_ws_floats(int64_t a, int64_t r, int64_t c) {
    if (a <= 0 || r <= 0 || c <= 0) return 0;
    long long rpb, blocks;
    reduce_mid_geometry(a, r, &rpb, &blocks);
    return blocks * a * c;
}

// ws == NULL: one pass, partial sums combined with float atomics (y must be zeroed / is accumulated into).
// ws != NULL (xmc_reduce_mid_ws_floats floats, no initialisation): atomic-free two-stage reduction in a fixed order
// -- bit-reproducible; this is the path the training step uses.
extern "C" int xmc_reduce_mid_ws(const void* x, float* y, float* ws, int64_t a, int64_t r, int64_t c, int32_t dtype,
                                 int32_t relu, float scale, int32_t accumulate, void* stream) {
    XMC_REQUIRE(x && y && a > 0 && r > 0 && c > 0 && a < 65536 && (c + MAXC - 1) / MAXC < 65536);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!ws && !accumulate) {
        hipError_t e = hipMemsetAsync(y, 0, sizeof(float) * a * c, s);
        if (e != hipSuccess) return xmc_hip_err(e);
    }
    const bool vec = vec_ok((int)c, dtype, x);
    long long rpb, blocks;
    reduce_mid_geometry(a, r, &rpb, &blocks);
    // channel window per workgroup: MAXC, or 1024 when rows x a give only a few workgroups (a dense layer's bias gradient:
    // 56 rows of up to 24k channels)
    const int cwin = blocks * a < 64 ? 1024 : MAXC;
    dim3 grid((unsigned)blocks, (unsigned)a, (unsigned)((c + cwin - 1) / cwin)), block(256);
    float* dst = ws ? ws : y;
    const int part = ws ? 1 : 0;
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        if (vec) hipLaunchKernelGGL((reduce_mid_kernel<bf16_t, 8>), grid, block, 0, s, xp, dst, (long long)r, (int)c, relu, scale, (int)rpb, part, cwin);
        else hipLaunchKernelGGL((reduce_mid_kernel<bf16_t, 1>), grid, block, 0, s, xp, dst, (long long)r, (int)c, relu, scale, (int)rpb, part, cwin);
    } else {
        const float* xp = static_cast<const float*>(x);
        if (vec) hipLaunchKernelGGL((reduce_mid_kernel<float, 4>), grid, block, 0, s, xp, dst, (long long)r, (int)c, relu, scale, (int)rpb, part, cwin);
        else hipLaunchKernelGGL((reduce_mid_kernel<float, 1>), grid, block, 0, s, xp, dst, (long long)r, (int)c, relu, scale, (int)rpb, part, cwin);
    }
    if (ws) {
        const long long width = (long long)a * c;
        XMC_REQUIRE(width < (1ll << 31));
        hipLaunchKernelGGL((reduce_rows_kernel<false>), dim3((unsigned)((width + 15) / 16)), dim3(256), 0, s, (const float*)ws,
                           (int)blocks, (int)width, y, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr,
                           scale, 0, 0.f, 0.f, accumulate);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_reduce_mid(const void* x, float* y, int64_t a, int64_t r, int64_t c, int32_t dtype,
                              int32_t relu, float scale, int32_t accumulate, void* stream) {
    return xmc_reduce_mid_ws(x, y, nullptr, a, r, c, dtype, relu, scale, accumulate, stream);
}

extern "C" int xmc_bn_stats(const void* x, float* sums, int64_t pixels, int32_t c, int32_t dtype, void* stream) {
    XMC_REQUIRE(x && sums && pixels > 0 && c > 0 && c <= MAXC);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(c, dtype, x);
    long long rpb = (pixels + 2047) / 2048;
    if (rpb < 64) rpb = pixels < 64 ? pixels : 64;
    const long long blocks = (pixels + rpb - 1) / rpb;
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        if (vec) hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 8>), grid, block, 0, s, xp, sums, (long long)pixels, c, (int)rpb, 0);
        else hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 1>), grid, block, 0, s, xp, sums, (long long)pixels, c, (int)rpb, 0);
    } else {
        const float* xp = static_cast<const float*>(x);
        if (vec) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), grid, block, 0, s, xp, sums, (long long)pixels, c, (int)rpb, 0);
        else hipLaunchKernelGGL((bn_stats_kernel<float, 1>), grid, block, 0, s, xp, sums, (long long)pixels, c, (int)rpb, 0);
    }
    XMC_LAUNCH_RET();
}

// Two-stage (atomic-free, bit-reproducible) batch statistics + finalize: <= 512 workgroups stream x and write
// one row of partial sums each into `ws`; a second small kernel reduces the rows in a fixed order and produces
// mean / rstd / running statistics.
static void bn_partial_geometry(long long pixels, long long* rpb, long long* blocks) {
    long long r = (pixels + 511) / 512;
    if (r < 64) r = pixels < 64 ? pixels : 64;       // (8 rows per workgroup on the small maps: level in the step, round 5)
    *rpb = r;
    *blocks = (pixels + r - 1) / r;
}

extern "C" int64_t xmc_bn_stats_ws_floats(int64_t pixels, int32_t c) {
    if (pixels <= 0 || c <= 0) return 0;
    long long rpb, blocks;
    bn_partial_geometry(pixels, &rpb, &blocks);
    return blocks * 2 * c;
}

extern "C" int xmc_bn_batch_stats(const void* x, float* ws, float* mean, float* rstd, float* run_mean, float* run_var,
                                  int64_t pixels, int32_t c, int32_t dtype, float eps, float momentum,
                                  int32_t update_running, void* stream) {
    XMC_REQUIRE(x && ws && mean && rstd && pixels > 0 && c > 0 && c <= MAXC);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    XMC_REQUIRE(!update_running || (run_mean && run_var));
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(c, dtype, x);
    long long rpb, blocks;
    bn_partial_geometry(pixels, &rpb, &blocks);
    dim3 grid((unsigned)blocks), block(256);
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        if (vec) hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 8>), grid, block, 0, s, xp, ws, (long long)pixels, c, (int)rpb, 1);
        else hipLaunchKernelGGL((bn_stats_kernel<bf16_t, 1>), grid, block, 0, s, xp, ws, (long long)pixels, c, (int)rpb, 1);
    } else {
        const float* xp = static_cast<const float*>(x);
        if (vec) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), grid, block, 0, s, xp, ws, (long long)pixels, c, (int)rpb, 1);
        else hipLaunchKernelGGL((bn_stats_kernel<float, 1>), grid, block, 0, s, xp, ws, (long long)pixels, c, (int)rpb, 1);
    }
    hipLaunchKernelGGL((reduce_rows_kernel<true>), dim3((c + 15) / 16), dim3(256), 0, s, (const float*)ws, (int)blocks,
                       2 * c, (float*)nullptr, mean, rstd, run_mean, run_var, 1.0f / (float)pixels, c, eps, momentum,
                       update_running);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_bn_finalize(const float* sums, float* mean, float* rstd, float* run_mean, float* run_var,
                               int64_t pixels, int32_t c, float eps, float momentum, int32_t update_running,
                               void* stream) {
    XMC_REQUIRE(sums && mean && rstd && pixels > 0 && c > 0);
    XMC_REQUIRE(!update_running || (run_mean && run_var));
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((c + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream),
                       sums, mean, rstd, run_mean, run_var, 1.0f / (float)pixels, c, eps, momentum, update_running);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_bn_from_running(const float* run_mean, const float* run_var, float* mean, float* rstd,
                                   int32_t c, float eps, void* stream) {
    XMC_REQUIRE(run_mean && run_var && mean && rstd && c > 0);
    hipLaunchKernelGGL(bn_from_running_kernel, dim3((c + 255) / 256), dim3(256), 0,
                       static_cast<hipStream_t>(stream), run_mean, run_var, mean, rstd, c, eps);
    XMC_LAUNCH_RET();
}

// gamma / beta element type of a cBN launch: float32, or (gb_dtype == XMC_BF16: bf16 activations, channel vectors of 8, 16-byte
// aligned rows) bf16
static bool gb_bf16_ok(int32_t gb_dtype, int32_t dtype, bool vec, const void* a, const void* b2, const void* c2, const void* d2, int cstride) {
    if (gb_dtype != XMC_BF16) return false;
    auto al = [](const void* q) { return q == nullptr || ((uintptr_t)q % 16) == 0; };
    return dtype == XMC_BF16 && vec && (cstride % 8) == 0 && al(a) && al(b2) && al(c2) && al(d2);
}

extern "C" int xmc_cbn_act_fwd(const void* x, const float* mean, const float* rstd, const void* gamma,
                               const void* beta, void* y, int32_t n, int32_t h, int32_t w, int32_t c, int32_t hc,
                               int32_t cstride, int32_t relu, int32_t dtype, int32_t gb_dtype, void* stream) {
    XMC_REQUIRE(x && mean && rstd && gamma && beta && y);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    XMC_REQUIRE(gb_dtype == XMC_F32 || gb_dtype == XMC_BF16);
    CbnGeo g;
    if (make_geo(g, n, h, w, c, hc, relu, cstride) != XMC_OK) return XMC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(c, dtype, x, y);
    const bool g16 = gb_bf16_ok(gb_dtype, dtype, vec, gamma, beta, nullptr, nullptr, cstride);
    XMC_REQUIRE(g16 || gb_dtype == XMC_F32);
    const float* gf = static_cast<const float*>(gamma);
    const float* bf = static_cast<const float*>(beta);
    const bf16_t* gh = static_cast<const bf16_t*>(gamma);
    const bf16_t* bh = static_cast<const bf16_t*>(beta);
    const int ve = vec ? (dtype == XMC_BF16 ? 8 : 4) : 1;
    const long long nvec = (long long)n * h * w * (c / ve);
    long long blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    dim3 grid((unsigned)blocks), block(256);
    if (const int R = cbn_run_len(g, vec, nvec)) {
        const long long nwork = nvec / R;
        long long rb = (nwork + 255) / 256;
        if (rb > 8192) rb = 8192;
        dim3 rgrid((unsigned)rb);
#define XMC_CBN_RUN_FWD(T_, VE_, R_, GT_, G_, B_) hipLaunchKernelGGL((cbn_fwd_run_kernel<T_, VE_, R_, GT_>), rgrid, block, 0, s, static_cast<const T_*>(x), \
                                                                     mean, rstd, G_, B_, static_cast<T_*>(y), g, nwork)
        if (g16) { if (R == 8) XMC_CBN_RUN_FWD(bf16_t, 8, 8, bf16_t, gh, bh); else XMC_CBN_RUN_FWD(bf16_t, 8, 4, bf16_t, gh, bh); }
        else if (dtype == XMC_BF16) { if (R == 8) XMC_CBN_RUN_FWD(bf16_t, 8, 8, float, gf, bf); else XMC_CBN_RUN_FWD(bf16_t, 8, 4, float, gf, bf); }
        else { if (R == 8) XMC_CBN_RUN_FWD(float, 4, 8, float, gf, bf); else XMC_CBN_RUN_FWD(float, 4, 4, float, gf, bf); }
#undef XMC_CBN_RUN_FWD
        XMC_LAUNCH_RET();
    }
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        bf16_t* yp = static_cast<bf16_t*>(y);
        if (g16) hipLaunchKernelGGL((cbn_fwd_kernel<bf16_t, 8, bf16_t>), grid, block, 0, s, xp, mean, rstd, gh, bh, yp, g, nvec);
        else if (vec) hipLaunchKernelGGL((cbn_fwd_kernel<bf16_t, 8>), grid, block, 0, s, xp, mean, rstd, gf, bf, yp, g, nvec);
        else hipLaunchKernelGGL((cbn_fwd_kernel<bf16_t, 1>), grid, block, 0, s, xp, mean, rstd, gf, bf, yp, g, nvec);
    } else {
        const float* xp = static_cast<const float*>(x);
        float* yp = static_cast<float*>(y);
        if (vec) hipLaunchKernelGGL((cbn_fwd_kernel<float, 4>), grid, block, 0, s, xp, mean, rstd, gf, bf, yp, g, nvec);
        else hipLaunchKernelGGL((cbn_fwd_kernel<float, 1>), grid, block, 0, s, xp, mean, rstd, gf, bf, yp, g, nvec);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cbn_act_fwd_mx8(const void* x, const float* mean, const float* rstd, const float* gamma,
                                   const float* beta, void* y, void* y8, int32_t n, int32_t h, int32_t w, int32_t c,
                                   int32_t hc, int32_t cstride, int32_t relu, void* stream) {
    XMC_REQUIRE(x && mean && rstd && gamma && beta && y && y8 && (c % 64) == 0);
    XMC_REQUIRE(((uintptr_t)x % 16) == 0 && ((uintptr_t)y % 16) == 0 && ((uintptr_t)y8 % 16) == 0);
    CbnGeo g;
    if (make_geo(g, n, h, w, c, hc, relu, cstride) != XMC_OK) return XMC_EINVAL;
    const long long nvec = (long long)n * h * w * (c / 8);
    long long blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL((cbn_fwd_kernel<bf16_t, 8>), dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<const bf16_t*>(x), mean, rstd, gamma, beta, static_cast<bf16_t*>(y), g, nvec,
                       static_cast<unsigned char*>(y8), xmc_mx_rnd());
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cbn_act_bwd_cells(const void* dy, const void* x, const float* mean, const float* rstd,
                                     const void* gamma, const void* beta, void* dgamma, void* dbeta, int32_t n,
                                     int32_t h, int32_t w, int32_t c, int32_t hc, int32_t cstride, int32_t relu,
                                     int32_t dtype, int32_t gb_dtype, void* stream) {
    XMC_REQUIRE(dy && x && mean && rstd && gamma && beta && dgamma && dbeta && c <= MAXC);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    XMC_REQUIRE(gb_dtype == XMC_F32 || gb_dtype == XMC_BF16);
    CbnGeo g;
    if (make_geo(g, n, h, w, c, hc, relu, cstride) != XMC_OK) return XMC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(c, dtype, x, dy);
    const bool g16 = gb_bf16_ok(gb_dtype, dtype, vec, gamma, beta, dgamma, dbeta, cstride);
    XMC_REQUIRE(g16 || gb_dtype == XMC_F32);
    const float* gf = static_cast<const float*>(gamma);
    const float* bf = static_cast<const float*>(beta);
    float* dgf = static_cast<float*>(dgamma);
    float* dbf = static_cast<float*>(dbeta);
    const bf16_t* gh = static_cast<const bf16_t*>(gamma);
    const bf16_t* bh = static_cast<const bf16_t*>(beta);
    bf16_t* dgh = static_cast<bf16_t*>(dgamma);
    bf16_t* dbh = static_cast<bf16_t*>(dbeta);
    const int ve = vec ? (dtype == XMC_BF16 ? 8 : 4) : 1;
    const long long nwork = (long long)n * hc * hc * (c / ve);
    long long blocks = (nwork + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    dim3 grid((unsigned)blocks), block(256);
    // cells of 4 x 4 / 8 x 8 pixels (whatever their number): one thread per row of the cell (cbn_bwd_cells_rows_kernel);
    // tuning "cbn_run" = 0: thread per cell / the split kernel below
    const int fcell = h / hc;
    if (cbn_run_len(g, vec, nwork) && (fcell == 4 || fcell == 8)) {
        dim3 rgrid((unsigned)((nwork + 256 / fcell - 1) / (256 / fcell)));
#define XMC_CBN_ROWS(T_, VE_, F_, GT_, G_, B_, DG_, DB_) hipLaunchKernelGGL((cbn_bwd_cells_rows_kernel<T_, VE_, F_, GT_>), rgrid, block, 0, s, static_cast<const T_*>(dy), \
                                                                            static_cast<const T_*>(x), mean, rstd, G_, B_, DG_, DB_, g, nwork)
        if (g16) { if (fcell == 8) XMC_CBN_ROWS(bf16_t, 8, 8, bf16_t, gh, bh, dgh, dbh); else XMC_CBN_ROWS(bf16_t, 8, 4, bf16_t, gh, bh, dgh, dbh); }
        else if (dtype == XMC_BF16) { if (fcell == 8) XMC_CBN_ROWS(bf16_t, 8, 8, float, gf, bf, dgf, dbf); else XMC_CBN_ROWS(bf16_t, 8, 4, float, gf, bf, dgf, dbf); }
        else { if (fcell == 8) XMC_CBN_ROWS(float, 4, 8, float, gf, bf, dgf, dbf); else XMC_CBN_ROWS(float, 4, 4, float, gf, bf, dgf, dbf); }
#undef XMC_CBN_ROWS
        XMC_LAUNCH_RET();
    }
    // few large cells (the global cBN layers): 8 threads per (cell, channel vector), see cbn_bwd_cells_split_kernel
    const int npix_cell = (h / hc) * (w / hc);
    if (vec && nwork < 65536 && npix_cell >= 64) {
        const int cgroups = (c / ve + 31) / 32;
        dim3 sgrid((unsigned)((long long)n * hc * hc * cgroups));
        if (g16)
            hipLaunchKernelGGL((cbn_bwd_cells_split_kernel<bf16_t, 8, bf16_t>), sgrid, block, 0, s, static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(x),
                               mean, rstd, gh, bh, dgh, dbh, g, cgroups);
        else if (dtype == XMC_BF16)
            hipLaunchKernelGGL((cbn_bwd_cells_split_kernel<bf16_t, 8>), sgrid, block, 0, s, static_cast<const bf16_t*>(dy), static_cast<const bf16_t*>(x),
                               mean, rstd, gf, bf, dgf, dbf, g, cgroups);
        else
            hipLaunchKernelGGL((cbn_bwd_cells_split_kernel<float, 4>), sgrid, block, 0, s, static_cast<const float*>(dy), static_cast<const float*>(x),
                               mean, rstd, gf, bf, dgf, dbf, g, cgroups);
        XMC_LAUNCH_RET();
    }
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        const bf16_t* dp = static_cast<const bf16_t*>(dy);
        if (g16) hipLaunchKernelGGL((cbn_bwd_cells_kernel<bf16_t, 8, bf16_t>), grid, block, 0, s, dp, xp, mean, rstd, gh, bh, dgh, dbh, g, nwork);
        else if (vec) hipLaunchKernelGGL((cbn_bwd_cells_kernel<bf16_t, 8>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, dgf, dbf, g, nwork);
        else hipLaunchKernelGGL((cbn_bwd_cells_kernel<bf16_t, 1>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, dgf, dbf, g, nwork);
    } else {
        const float* xp = static_cast<const float*>(x);
        const float* dp = static_cast<const float*>(dy);
        if (vec) hipLaunchKernelGGL((cbn_bwd_cells_kernel<float, 4>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, dgf, dbf, g, nwork);
        else hipLaunchKernelGGL((cbn_bwd_cells_kernel<float, 1>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, dgf, dbf, g, nwork);
    }
    XMC_LAUNCH_RET();
}

static void cbn_sums_geometry(long long cells, int* cpb, long long* blocks) {
    int k = (int)((cells + 255) / 256);
    if (k < 8) k = 8;
    *cpb = k;
    *blocks = (cells + k - 1) / k;
}

extern "C" int64_t xmc_cbn_bwd_sums_ws_floats(int64_t cells, int32_t c) {
    if (cells <= 0 || c <= 0) return 0;
    int cpb;
    long long blocks;
    cbn_sums_geometry(cells, &cpb, &blocks);
    return blocks * 2 * c;
}

// s[2C] = column sums over all cells, two-stage through `ws` (xmc_cbn_bwd_sums_ws_floats floats): atomic-free,
// fixed summation order; s needs no zero-initialisation.
extern "C" int xmc_cbn_bwd_sums(const void* gamma, const void* dgamma, const void* dbeta, float* s, float* ws,
                                int64_t cells, int32_t c, int32_t cstride, int32_t gb_dtype, void* stream) {
    XMC_REQUIRE(gamma && dgamma && dbeta && s && ws && cells > 0 && c > 0 && cstride >= c);
    XMC_REQUIRE(gb_dtype == XMC_F32 || gb_dtype == XMC_BF16);
    int cpb;
    long long blocks;
    cbn_sums_geometry(cells, &cpb, &blocks);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (gb_dtype == XMC_BF16)
        hipLaunchKernelGGL(cbn_bwd_sums_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const bf16_t*>(gamma),
                           static_cast<const bf16_t*>(dgamma), static_cast<const bf16_t*>(dbeta), ws, (long long)cells, c, cstride, cpb);
    else
        hipLaunchKernelGGL(cbn_bwd_sums_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, st, static_cast<const float*>(gamma),
                           static_cast<const float*>(dgamma), static_cast<const float*>(dbeta), ws, (long long)cells, c, cstride, cpb);
    hipLaunchKernelGGL((reduce_rows_kernel<false>), dim3((2 * c + 15) / 16), dim3(256), 0, st, (const float*)ws, (int)blocks,
                       2 * c, s, (float*)nullptr, (float*)nullptr, (float*)nullptr, (float*)nullptr, 1.f, 0, 0.f, 0.f, 0);
    XMC_LAUNCH_RET();
}

extern "C" int xmc_cbn_act_bwd_dx(const void* dy, const void* x, const float* mean, const float* rstd,
                                  const void* gamma, const void* beta, const float* sarr, void* dx, int32_t n,
                                  int32_t h, int32_t w, int32_t c, int32_t hc, int32_t cstride, int32_t relu,
                                  int32_t dtype, int32_t gb_dtype, void* stream) {
    XMC_REQUIRE(dy && x && mean && rstd && gamma && beta && sarr && dx);
    XMC_REQUIRE(dtype == XMC_F32 || dtype == XMC_BF16);
    XMC_REQUIRE(gb_dtype == XMC_F32 || gb_dtype == XMC_BF16);
    CbnGeo g;
    if (make_geo(g, n, h, w, c, hc, relu, cstride) != XMC_OK) return XMC_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bool vec = vec_ok(c, dtype, x, dy, dx);
    const bool g16 = gb_bf16_ok(gb_dtype, dtype, vec, gamma, beta, nullptr, nullptr, cstride);
    XMC_REQUIRE(g16 || gb_dtype == XMC_F32);
    const float* gf = static_cast<const float*>(gamma);
    const float* bf = static_cast<const float*>(beta);
    const bf16_t* gh = static_cast<const bf16_t*>(gamma);
    const bf16_t* bh = static_cast<const bf16_t*>(beta);
    const int ve = vec ? (dtype == XMC_BF16 ? 8 : 4) : 1;
    const long long nvec = (long long)n * h * w * (c / ve);
    long long blocks = (nvec + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    const float inv_p = 1.0f / ((float)n * h * w);
    dim3 grid((unsigned)blocks), block(256);
    if (const int R = cbn_run_len(g, vec, nvec)) {
        const long long nwork = nvec / R;
        long long rb = (nwork + 255) / 256;
        if (rb > 8192) rb = 8192;
        dim3 rgrid((unsigned)rb);
#define XMC_CBN_RUN_DX(T_, VE_, R_, GT_, G_, B_) hipLaunchKernelGGL((cbn_bwd_dx_run_kernel<T_, VE_, R_, GT_>), rgrid, block, 0, s, static_cast<const T_*>(dy), \
                                                                    static_cast<const T_*>(x), mean, rstd, G_, B_, sarr, static_cast<T_*>(dx), g, nwork, inv_p)
        if (g16) { if (R == 8) XMC_CBN_RUN_DX(bf16_t, 8, 8, bf16_t, gh, bh); else XMC_CBN_RUN_DX(bf16_t, 8, 4, bf16_t, gh, bh); }
        else if (dtype == XMC_BF16) { if (R == 8) XMC_CBN_RUN_DX(bf16_t, 8, 8, float, gf, bf); else XMC_CBN_RUN_DX(bf16_t, 8, 4, float, gf, bf); }
        else { if (R == 8) XMC_CBN_RUN_DX(float, 4, 8, float, gf, bf); else XMC_CBN_RUN_DX(float, 4, 4, float, gf, bf); }
#undef XMC_CBN_RUN_DX
        XMC_LAUNCH_RET();
    }
    if (dtype == XMC_BF16) {
        const bf16_t* xp = static_cast<const bf16_t*>(x);
        const bf16_t* dp = static_cast<const bf16_t*>(dy);
        bf16_t* op = static_cast<bf16_t*>(dx);
        if (g16) hipLaunchKernelGGL((cbn_bwd_dx_kernel<bf16_t, 8, bf16_t>), grid, block, 0, s, dp, xp, mean, rstd, gh, bh, sarr, op, g, nvec, inv_p);
        else if (vec) hipLaunchKernelGGL((cbn_bwd_dx_kernel<bf16_t, 8>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, sarr, op, g, nvec, inv_p);
        else hipLaunchKernelGGL((cbn_bwd_dx_kernel<bf16_t, 1>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, sarr, op, g, nvec, inv_p);
    } else {
        const float* xp = static_cast<const float*>(x);
        const float* dp = static_cast<const float*>(dy);
        float* op = static_cast<float*>(dx);
        if (vec) hipLaunchKernelGGL((cbn_bwd_dx_kernel<float, 4>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, sarr, op, g, nvec, inv_p);
        else hipLaunchKernelGGL((cbn_bwd_dx_kernel<float, 1>), grid, block, 0, s, dp, xp, mean, rstd, gf, bf, sarr, op, g, nvec, inv_p);
    }
    XMC_LAUNCH_RET();
}
