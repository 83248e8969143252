// Batched, arbitrarily-strided float32 GEMM on v_mfma_f32_32x32x2_f32 (exact fp32: one rounding
// per product, bit-equal to an fmaf chain -- the "contrastive loss in fp32" path).
//   C[b][m][n] = alpha * (*alpha_dev) * sum_k A[b][m][k] * B[b][k][n] + beta * C[b][m][n]
// Element strides are free, so NN / NT / TN views need no transposed copies.  The loader walks
// whichever of (m|n, k) is unit-stride with consecutive lanes (coalesced 256-byte wavefront
// reads); LDS tiles are [k][m] so the one-element-per-lane MFMA fragments (A[i = lane&31]
// [k = lane>>5]) are conflict-free.
#include "common.h"

namespace {

struct GArgs {
    const float* a; const float* b; float* c;
    int M, N, K;
    long long sab, sam, sak, sbb, sbk, sbn, scb, ldc;
    float alpha; const float* alpha_dev; float beta;
    int ksplit;                      // > 1: blockIdx.y owns a K range and writes its partial product to part[blockIdx.y][batch][M][N]
    float* part;                     //      (caller's workspace); gemm_splitk_reduce_kernel adds the partials in a fixed order
    unsigned a_bytes, b_bytes;       // extent of ONE batch element of A / B in bytes (buffer-load range, < 4 GiB)
};

constexpr int GBK = 16;

// TM x TN tile, 4 waves arranged 2x2, each wave (TM/2)x(TN/2) as RM x RN 32x32 MFMA blocks.
template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GArgs p) {
    constexpr int RM = TM / 64, RN = TN / 64;
    constexpr int PA = TM + 1, PB = TN + 1;
    constexpr int EA = TM * GBK / 256, EB = TN * GBK / 256;     // elements per thread per tile
    __shared__ float lds[2 * GBK * (PA + PB)];
    float* const As = lds;
    float* const Bs = lds + 2 * GBK * PA;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + TN - 1) / TN;
    // workgroup b runs on XCD b % 8 (each with its own L2): with tiles_n == 8 every XCD would own one column of
    // tiles and fetch ALL of A (8x the HBM traffic -- measured 120 us for a 21 GFLOP product).  Give each XCD a
    // contiguous range of tile ids instead, so the tiles that share an A panel share an L2.
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wid / tiles_n, tn = wid - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const float* __restrict__ A = p.a + (long long)blockIdx.z * p.sab;
    const float* __restrict__ B = p.b + (long long)blockIdx.z * p.sbb;
    float* __restrict__ C = p.c + (long long)blockIdx.z * p.scb;

    const bool a_kfast = (p.sak == 1), b_kfast = (p.sbk == 1);
    float ra[EA], rb[EB];

    auto load = [&](int k0) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int k = a_kfast ? (idx % GBK) : (idx / TM);
            const int m = a_kfast ? (idx / GBK) : (idx % TM);
            const int gm = m0 + m, gk = k0 + k;
            // branch-free (all loads of the tile in flight): an element outside the matrix reads A[0] and becomes zero
            const bool ok = gm < p.M && gk < p.K;
            const float val = A[ok ? (long long)gm * p.sam + (long long)gk * p.sak : 0];
            ra[e] = ok ? val : 0.f;
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * 256;
            const int k = b_kfast ? (idx % GBK) : (idx / TN);
            const int n = b_kfast ? (idx / GBK) : (idx % TN);
            const int gn = n0 + n, gk = k0 + k;
            const bool ok = gn < p.N && gk < p.K;
            const float val = B[ok ? (long long)gk * p.sbk + (long long)gn * p.sbn : 0];
            rb[e] = ok ? val : 0.f;
        }
    };
    auto store = [&](int buf) {
#pragma unroll
        for (int e = 0; e < EA; ++e) {
            const int idx = tid + e * 256;
            const int k = a_kfast ? (idx % GBK) : (idx / TM);
            const int m = a_kfast ? (idx / GBK) : (idx % TM);
            As[(buf * GBK + k) * PA + m] = ra[e];
        }
#pragma unroll
        for (int e = 0; e < EB; ++e) {
            const int idx = tid + e * 256;
            const int k = b_kfast ? (idx % GBK) : (idx / TN);
            const int n = b_kfast ? (idx / GBK) : (idx % TN);
            Bs[(buf * GBK + k) * PB + n] = rb[e];
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < GBK / 2; ++kk) {
            float af[RM], bf[RN];
            const int k = kk * 2 + lhi;
#pragma unroll
            for (int i = 0; i < RM; ++i) af[i] = As[(buf * GBK + k) * PA + wm * (TM / 2) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < RN; ++j) bf[j] = Bs[(buf * GBK + k) * PB + wn * (TN / 2) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const int ktiles_all = (p.K + GBK - 1) / GBK;
    const int per = (ktiles_all + p.ksplit - 1) / p.ksplit;
    const int kt0 = blockIdx.y * per, ktiles = min(ktiles_all, kt0 + per);
    if (kt0 >= ktiles) return;
    load(kt0 * GBK);
    store(0);
    __syncthreads();
    for (int t = kt0; t < ktiles; ++t) {
        const int buf = (t - kt0) & 1;
        const bool more = t + 1 < ktiles;
        if (more) load((t + 1) * GBK);
        compute(buf);
        if (more) store(buf ^ 1);
        __syncthreads();
    }

    float alpha = p.alpha;
    if (p.alpha_dev) alpha *= *p.alpha_dev;
    // D[i][j]: col = lane&31 -> n (contiguous in C), row = (e&3) + 8*(e>>2) + 4*(lane>>5) -> m
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (TN / 2) + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (TM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (m < p.M) {
                    float* dst = C + (long long)m * p.ldc + n;
                    float v = alpha * acc[i][j][e];
                    if (p.ksplit > 1) {          // alpha-scaled partial product of this K range (plain store)
                        p.part[(((long long)blockIdx.y * gridDim.z + blockIdx.z) * p.M + m) * p.N + n] = v;
                        continue;
                    }
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                }
            }
    }
}

// ---- same interface, bf16 MFMA: the float32 operands are rounded to bf16 while they are staged into LDS and the
// products accumulate in float32 (v_mfma_f32_32x32x16_bf16: 16x the rate of the exact-fp32 MFMA).  Used by the
// bf16 training mode for the region-word similarity GEMMs of word_loss (B^2 * R * T * E products), never by the
// float32 parity mode.  LDS tiles are [row][32 k] bf16 with an 80-byte pitch (conflict-free ds_read_b128 fragments,
// as in the convolution kernels); every thread owns KPT consecutive k of one row, fetched either as float4 runs
// (k unit-stride) or as KPT lane-coalesced scalars (row index unit-stride), so both layouts write the same image.
constexpr int HBK = 32, HPITCH = 40;     // k per tile; row pitch in bf16

// AM / BM: operand layouts, chosen by the launcher (gemm_operand_mode) -- compile-time, because a run-time choice between
// load shapes puts the loaded registers behind PHI copies and the compiler then drains vmcnt(0) at every join.
template <int TM, int TN, int AM, int BM>
__global__ __launch_bounds__(256, 2) void gemm_bf16mfma_kernel(const GArgs p) {
    constexpr int RM = TM / 64, RN = TN / 64;
    constexpr int TPRA = 256 / TM, TPRB = 256 / TN;              // threads per row
    constexpr int KA = HBK / TPRA, KB = HBK / TPRB;              // consecutive k per thread (16 or 8)
    __shared__ __attribute__((aligned(16))) bf16_t lds[2 * (TM + TN) * HPITCH];
    bf16_t* const As = lds;
    bf16_t* const Bs = lds + 2 * TM * HPITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int tiles_n = (p.N + TN - 1) / TN;
    // workgroup b runs on XCD b % 8 (each with its own L2): with tiles_n == 8 every XCD would own one column of
    // tiles and fetch ALL of A (8x the HBM traffic -- measured 120 us for a 21 GFLOP product).  Give each XCD a
    // contiguous range of tile ids instead, so the tiles that share an A panel share an L2.
    const int wid = xcd_remap(blockIdx.x, gridDim.x);
    const int tm = wid / tiles_n, tn = wid - tm * tiles_n;
    const int m0 = tm * TM, n0 = tn * TN;
    const float* __restrict__ A = p.a + (long long)blockIdx.z * p.sab;
    const float* __restrict__ B = p.b + (long long)blockIdx.z * p.sbb;
    float* __restrict__ C = p.c + (long long)blockIdx.z * p.scb;

    // Operand loads are BRANCH-FREE buffer loads: every lane computes a byte offset, or an out-of-range one (the
    // hardware then returns zeros) when its row / k is outside the matrix.  The first version guarded each element
    // with a per-lane condition: hipcc turned that into ~1,300 exec-mask branches around single-dword loads with
    // s_waitcnt vmcnt(0) between them -- 120-180 TFLOP/s on the word_loss products.
    // Three layouts per operand: k unit-stride (16-byte runs along k), row unit-stride (16-byte runs along the rows,
    // transposed into the [row][k] LDS image), anything else (dword gathers).
    constexpr unsigned OOB = 0xfffffff0u;
    const __amdgpu_buffer_rsrc_t ar = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(A), 0, p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(B), 0, p.b_bytes, 0x00020000);
    const bool a_kfast = (p.sak == 1), b_kfast = (p.sbk == 1);
    const int am = a_kfast ? tid / TPRA : tid % TM, akq = a_kfast ? tid % TPRA : tid / TM;
    const int bn = b_kfast ? tid / TPRB : tid % TN, bkq = b_kfast ? tid % TPRB : tid / TN;
    constexpr int KQA = KA / 4, KQB = KB / 4;
    constexpr bool a_rvec = AM == 2, b_rvec = BM == 2;
    const int arq = tid % (TM / 4), akb = tid / (TM / 4);
    const int brq = tid % (TN / 4), bkb = tid / (TN / 4);
    // Register ring, RD tiles deep: the operand tiles of k-steps t+1 .. t+RD-1 are in flight while step t multiplies
    // (with a prefetch distance of one tile and 2 resident workgroups every k-step exposed a full load latency)
    constexpr int RD = 3;
    float ra[RD][KA], rb[RD][KB];

    typedef unsigned int u32x4v __attribute__((ext_vector_type(4)));
    // mode 0: dword gathers, 1: 16-byte runs along k, 2: 16-byte runs along the rows (r[j * 4 + e]: k = j, row = e)
    auto load_op = [&](float* r, __amdgpu_buffer_rsrc_t rs, int mode, int row, int row4, int nrows, long long srow,
                       long long sk, int k0, int k0r, int kpt) {
        if (mode == 1) {
#pragma unroll
            for (int e = 0; e < kpt; e += 4) {
                const int k = k0 + e;
                const unsigned off = (row < nrows && k < p.K) ? (unsigned)(((long long)row * srow + k) * 4) : OOB;
                const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                r[e] = __uint_as_float(v.x); r[e + 1] = __uint_as_float(v.y); r[e + 2] = __uint_as_float(v.z); r[e + 3] = __uint_as_float(v.w);
            }
        } else if (mode == 2) {
#pragma unroll
            for (int j = 0; j < kpt / 4; ++j) {
                const int k = k0r + j;
                const unsigned off = (row4 < nrows && k < p.K) ? (unsigned)(((long long)k * sk + row4) * 4) : OOB;
                const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
                r[j * 4] = __uint_as_float(v.x); r[j * 4 + 1] = __uint_as_float(v.y); r[j * 4 + 2] = __uint_as_float(v.z); r[j * 4 + 3] = __uint_as_float(v.w);
            }
        } else {
#pragma unroll
            for (int e = 0; e < kpt; ++e) {
                const int k = k0 + e;
                const unsigned off = (row < nrows && k < p.K) ? (unsigned)(((long long)row * srow + (long long)k * sk) * 4) : OOB;
                r[e] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
            }
        }
    };
    constexpr int a_mode = AM, b_mode = BM;
    auto store_rvec = [&](const float* r, bf16_t* dst, int kq) {      // dst: LDS address of (first row, first k)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            bf16_t* d = dst + e * HPITCH;
            if (kq == 4) *reinterpret_cast<uint2*>(d) = make_uint2(pack_bf2(r[e], r[4 + e]), pack_bf2(r[8 + e], r[12 + e]));
            else *reinterpret_cast<uint32_t*>(d) = pack_bf2(r[e], r[4 + e]);
        }
    };
    // k unit-stride operands (mode 1): the 8 lanes tid & 7 of one instruction read the 128 contiguous bytes of ONE row's
    // k-tile and the 64 lanes of a wave 8 whole cache lines; item e of a thread is row (tid >> 3) + 32 e.  (First version:
    // a thread owned 16 consecutive k of one row -- every b128 instruction then touched 64 pieces of 16 bytes in 32-64
    // different lines, one TA cycle each: ~2,000 cycles of address processing per k-tile, 255 TF/s on the 21-GFLOP product.)
    auto load_kvec = [&](float* r, __amdgpu_buffer_rsrc_t rs, int row0, int nrows, long long srow, int k0, int kpt) {
        const int k = k0 + (tid & 7) * 4;
#pragma unroll
        for (int e = 0; e < kpt / 4; ++e) {
            const int row = row0 + (tid >> 3) + 32 * e;
            const unsigned off = (row < nrows && k < p.K) ? (unsigned)(((long long)row * srow + k) * 4) : OOB;
            const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
            r[e * 4] = __uint_as_float(v.x); r[e * 4 + 1] = __uint_as_float(v.y); r[e * 4 + 2] = __uint_as_float(v.z); r[e * 4 + 3] = __uint_as_float(v.w);
        }
    };
    auto store_kvec = [&](const float* r, bf16_t* base, int kpt) {     // base: LDS address of the tile's (row 0, k 0)
#pragma unroll
        for (int e = 0; e < kpt / 4; ++e)
            *reinterpret_cast<uint2*>(base + ((tid >> 3) + 32 * e) * HPITCH + (tid & 7) * 4) =
                make_uint2(pack_bf2(r[e * 4], r[e * 4 + 1]), pack_bf2(r[e * 4 + 2], r[e * 4 + 3]));
    };
    auto load = [&](float* xa, float* xb, int k0) {
        if constexpr (a_mode == 1) load_kvec(xa, ar, m0, p.M, p.sam, k0, KA);
        else load_op(xa, ar, a_mode, m0 + am, m0 + arq * 4, p.M, p.sam, p.sak, k0 + akq * KA, k0 + akb * KQA, KA);
        if constexpr (b_mode == 1) load_kvec(xb, br, n0, p.N, p.sbn, k0, KB);
        else load_op(xb, br, b_mode, n0 + bn, n0 + brq * 4, p.N, p.sbn, p.sbk, k0 + bkq * KB, k0 + bkb * KQB, KB);
    };
    auto store = [&](const float* xa, const float* xb, int buf) {
        if constexpr (a_rvec) {
            store_rvec(xa, As + (buf * TM + arq * 4) * HPITCH + akb * KQA, KQA);
        } else if constexpr (a_mode == 1) {
            store_kvec(xa, As + buf * TM * HPITCH, KA);
        } else {
#pragma unroll
            for (int e = 0; e < KA; e += 8) {
                Vec<bf16_t> v; v.set(xa + e);
                v.store(As + (buf * TM + am) * HPITCH + akq * KA + e);
            }
        }
        if constexpr (b_rvec) {
            store_rvec(xb, Bs + (buf * TN + brq * 4) * HPITCH + bkb * KQB, KQB);
        } else if constexpr (b_mode == 1) {
            store_kvec(xb, Bs + buf * TN * HPITCH, KB);
        } else {
#pragma unroll
            for (int e = 0; e < KB; e += 8) {
                Vec<bf16_t> v; v.set(xb + e);
                v.store(Bs + (buf * TN + bn) * HPITCH + bkq * KB + e);
            }
        }
    };

    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    f32x16 acc[RM][RN];
#pragma unroll
    for (int i = 0; i < RM; ++i)
#pragma unroll
        for (int j = 0; j < RN; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int buf) {
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            bf16x8 af[RM], bfv[RN];
#pragma unroll
            for (int i = 0; i < RM; ++i)
                af[i] = *reinterpret_cast<const bf16x8*>(As + (buf * TM + wm * (TM / 2) + i * 32 + l31) * HPITCH + kk * 16 + lhi * 8);
#pragma unroll
            for (int j = 0; j < RN; ++j)
                bfv[j] = *reinterpret_cast<const bf16x8*>(Bs + (buf * TN + wn * (TN / 2) + j * 32 + l31) * HPITCH + kk * 16 + lhi * 8);
#pragma unroll
            for (int i = 0; i < RM; ++i)
#pragma unroll
                for (int j = 0; j < RN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfv[j], acc[i][j], 0, 0, 0);
        }
    };

    const int ktiles_all = (p.K + HBK - 1) / HBK;
    const int per = (ktiles_all + p.ksplit - 1) / p.ksplit;
    const int kt0 = blockIdx.y * per, ktiles = min(ktiles_all, kt0 + per);
    if (kt0 >= ktiles) return;
    // register slot (tile - kt0) % RD holds a tile from its load until it has been written to LDS buffer (tile - kt0) & 1
    // (loads past the last tile have out-of-range offsets: they return zeros and are never stored)
#pragma unroll
    for (int u = 0; u < RD; ++u) load(ra[u], rb[u], (kt0 + u) * HBK);
    store(ra[0], rb[0], 0);
    __syncthreads();
    // Steady state: whole groups of RD tiles with NO condition around the loads / stores -- a load the compiler cannot
    // prove was issued forces s_waitcnt vmcnt(0) in front of every LDS store (it was: the ring never had a load in flight
    // across a barrier).  Loads past K return zeros (out-of-range offsets); a tile loaded or stored past this split's last
    // one is never multiplied.
    int t = kt0;
    for (; t + RD <= ktiles; t += RD) {
#pragma unroll
        for (int u = 0; u < RD; ++u) {
            const int tt = t + u;
            const int buf = (tt - kt0) & 1;
            load(ra[u], rb[u], (tt + RD) * HBK);                     // slot u is free: tile tt is in LDS
            compute(buf);
            store(ra[(u + 1) % RD], rb[(u + 1) % RD], buf ^ 1);
            __syncthreads();
        }
    }
    // tail: the last (ktiles - kt0) % RD tiles (slots 0 .. in order again: t - kt0 is a multiple of RD)
#pragma unroll
    for (int u = 0; u < RD - 1; ++u) {
        const int tt = t + u;
        if (tt < ktiles) {                                       // workgroup-uniform
            const int buf = (tt - kt0) & 1;
            compute(buf);
            if (tt + 1 < ktiles) store(ra[(u + 1) % RD], rb[(u + 1) % RD], buf ^ 1);
            __syncthreads();
        }
    }

    float alpha = p.alpha;
    if (p.alpha_dev) alpha *= *p.alpha_dev;
#pragma unroll
    for (int j = 0; j < RN; ++j) {
        const int n = n0 + wn * (TN / 2) + j * 32 + l31;
        if (n >= p.N) continue;
#pragma unroll
        for (int i = 0; i < RM; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int m = m0 + wm * (TM / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * lhi;
                if (m < p.M) {
                    float* dst = C + (long long)m * p.ldc + n;
                    float v = alpha * acc[i][j][e];
                    if (p.ksplit > 1) {          // alpha-scaled partial product of this K range (plain store)
                        p.part[(((long long)blockIdx.y * gridDim.z + blockIdx.z) * p.M + m) * p.N + n] = v;
                        continue;
                    }
                    if (p.beta != 0.f) v += p.beta * *dst;
                    *dst = v;
                }
            }
    }
}



// C[b][m][n] = beta * C + sum_s part[s][b][m][n], splits added in a fixed order (no atomics: bit-reproducible)
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float* __restrict__ part, int ksplit, int batch, int M,
                                                                int N, float* __restrict__ c, long long scb, long long ldc,
                                                                float beta) {
    const long long total = (long long)batch * M * N;
    const long long e = (long long)blockIdx.x * 256 + threadIdx.x;
    if (e >= total) return;
    float t = 0.f;
    for (int s = 0; s < ksplit; ++s) t += part[(long long)s * total + e];
    const int n = (int)(e % N);
    const long long bm = e / N;
    const int m = (int)(bm % M), b = (int)(bm / M);
    float* dst = c + (long long)b * scb + (long long)m * ldc + n;
    *dst = (beta != 0.f ? beta * *dst : 0.f) + t;
}

}  // namespace

// Few output tiles and a long reduction (the 56 x 56 sentence logits over K = 1536, the conditioning-vector
// gradients over K = 3072: 1-4 workgroups walking 100-200 k-tiles in sequence): split K over blockIdx.y -- only when
// the caller lends a workspace for the partial products (xmc_gemm_ws_floats); never with float atomics.
static int pick_ksplit(long long tiles, int k, int bk) {
    // (k >= 256: a single 64 x 64 tile with K = 768 -- the B x B logits of the contrastive losses -- walked 48 k-tiles alone,
    //  52-68 us; twelve splits + the reduce kernel take a fraction of that)
    if (tiles >= 128 || k < 256) return 1;
    long long s = 256 / tiles;
    const long long smax = k / (4 * bk);
    if (s > smax) s = smax;
    return s < 2 ? 1 : (int)s;
}

static int gemm_geometry(int m, int n, int k, int batch, int bk, bool* big) {
    const long long work = (long long)m * n;
    *big = m > 64 && n > 64 && work * batch >= 128ll * 128 * 128;
    const int t = *big ? 128 : 64;
    const long long tiles = (long long)((m + t - 1) / t) * ((n + t - 1) / t);
    int ks = pick_ksplit(tiles * batch, k, bk);
    // no EMPTY split: the kernels give every split ceil(ktiles / ks) k-tiles, and a split that starts past the last tile
    // returns without writing its partial product (the reduce kernel would add uninitialised workspace)
    const int ktiles = (k + bk - 1) / bk, per = (ktiles + ks - 1) / ks;
    ks = (ktiles + per - 1) / per;
    return ks;
}

extern "C" int64_t xmc_gemm_ws_floats(int32_t m, int32_t n, int32_t k, int32_t batch, int32_t bf16_mfma) {
    if (m <= 0 || n <= 0 || k <= 0 || batch <= 0) return 0;
    bool big;
    const int ks = gemm_geometry(m, n, k, batch, bf16_mfma ? 32 : 16, &big);
    return ks > 1 ? (int64_t)ks * batch * m * n : 0;
}

template <typename K128, typename K64>
static int launch_gemm(GArgs& p, int batch, int bk, float* ws, hipStream_t s, K128 k128, K64 k64) {
    bool big;
    p.ksplit = gemm_geometry(p.M, p.N, p.K, batch, bk, &big);
    if (!ws) p.ksplit = 1;                               // no workspace: one K range per tile (still deterministic)
    p.part = ws;
    const int t = big ? 128 : 64;
    const long long tiles = (long long)((p.M + t - 1) / t) * ((p.N + t - 1) / t);
    dim3 grid((unsigned)tiles, (unsigned)p.ksplit, (unsigned)batch);
    if (big) hipLaunchKernelGGL(k128, grid, dim3(256), 0, s, p);
    else hipLaunchKernelGGL(k64, grid, dim3(256), 0, s, p);
    if (p.ksplit > 1) {
        const long long total = (long long)batch * p.M * p.N;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float*)ws,
                           p.ksplit, batch, p.M, p.N, p.c, p.scb, p.ldc, p.beta);
    }
    XMC_LAUNCH_RET();
}

extern "C" int xmc_gemm_f32(const float* a, const float* b, float* c, int32_t m, int32_t n, int32_t k,
                            int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn,
                            int64_t scb, int64_t ldc, float alpha, const float* alpha_dev, float beta,
                            int32_t batch, float* ws, void* stream) {
    XMC_REQUIRE(a && b && c);
    XMC_REQUIRE(m > 0 && n > 0 && k > 0 && batch > 0 && batch < 65536);
    GArgs p{a, b, c, m, n, k, sab, sam, sak, sbb, sbk, sbn, scb, ldc, alpha, alpha_dev, beta, 1, nullptr, 0, 0};
    return launch_gemm(p, batch, GBK, ws, static_cast<hipStream_t>(stream), gemm_f32_kernel<128, 128>, gemm_f32_kernel<64, 64>);
}

extern "C" int xmc_gemm_f32_bf16mfma(const float* a, const float* b, float* c, int32_t m, int32_t n, int32_t k,
                                     int64_t sab, int64_t sam, int64_t sak, int64_t sbb, int64_t sbk, int64_t sbn,
                                     int64_t scb, int64_t ldc, float alpha, const float* alpha_dev, float beta,
                                     int32_t batch, float* ws, void* stream) {
    XMC_REQUIRE(a && b && c);
    XMC_REQUIRE(m > 0 && n > 0 && k > 0 && batch > 0 && batch < 65536);
    GArgs p{a, b, c, m, n, k, sab, sam, sak, sbb, sbk, sbn, scb, ldc, alpha, alpha_dev, beta, 1, nullptr, 0, 0};
    XMC_REQUIRE(sam >= 0 && sak >= 0 && sbk >= 0 && sbn >= 0);
    const long long ea = ((long long)(m - 1) * sam + (long long)(k - 1) * sak + 1) * 4;
    const long long eb = ((long long)(k - 1) * sbk + (long long)(n - 1) * sbn + 1) * 4;
    XMC_REQUIRE(ea < 0xfffffff0ll && eb < 0xfffffff0ll);          // buffer loads address 32-bit byte offsets
    p.a_bytes = (unsigned)ea; p.b_bytes = (unsigned)eb;
    // operand layouts: 1 = k unit-stride in 16-byte runs, 2 = row unit-stride in 16-byte runs, 0 = dword gathers
    auto mode = [&](const float* ptr, int64_t sb, int64_t srow, int64_t sk, int rows) {
        const bool al16 = (sb % 4 == 0) && ((reinterpret_cast<uintptr_t>(ptr) & 15) == 0);
        if (sk == 1 && (srow % 4 == 0) && (k % 4 == 0) && al16) return 1;
        if (sk != 1 && srow == 1 && (sk % 4 == 0) && (rows % 4 == 0) && al16) return 2;
        return 0;
    };
    const int am = mode(a, sab, sam, sak, m), bm = mode(b, sbb, sbn, sbk, n);
    typedef void (*kern_t)(const GArgs);
#define XMC_G(T_) {{gemm_bf16mfma_kernel<T_, T_, 0, 0>, gemm_bf16mfma_kernel<T_, T_, 0, 1>, gemm_bf16mfma_kernel<T_, T_, 0, 2>}, \
                   {gemm_bf16mfma_kernel<T_, T_, 1, 0>, gemm_bf16mfma_kernel<T_, T_, 1, 1>, gemm_bf16mfma_kernel<T_, T_, 1, 2>}, \
                   {gemm_bf16mfma_kernel<T_, T_, 2, 0>, gemm_bf16mfma_kernel<T_, T_, 2, 1>, gemm_bf16mfma_kernel<T_, T_, 2, 2>}}
    static const kern_t k128[3][3] = XMC_G(128), k64[3][3] = XMC_G(64);
#undef XMC_G
    return launch_gemm(p, batch, HBK, ws, static_cast<hipStream_t>(stream), k128[am][bm], k64[am][bm]);
}
