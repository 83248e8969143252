/* Host-side helpers of the input pipeline: the byte-level pieces of reading the
 * reference's TFRecord / PNG dataset (preprocess_data.py:76-96, xmcgan/libml/coco_dataset.py:85-111,127-167) that are
 * sequential per byte and therefore hopeless in Python -- CRC-32C of the TFRecord framing, the PNG scanline
 * un-filter, and the bilinear resize + left-right flip + [0,1] conversion of tf.image.resize (half-pixel centres,
 * no anti-aliasing).  Plain C, built with gcc into libxmc_io.so next to the package; bound with ctypes
 * (xmcgan_image_generation_amd/libml/_io.py).  No GPU code here: the step's input is handed over in HBM by the
 * caller (pinned host buffer -> one cudaMemcpyAsync per batch tensor). */
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include <zlib.h>

/* ---- CRC-32C (Castagnoli, reflected 0x82F63B78), slice-by-8 */
static uint32_t T[8][256];
static int t_ready = 0;
static void crc_init(void) {
    for (uint32_t i = 0; i < 256; ++i) {
        uint32_t c = i;
        for (int k = 0; k < 8; ++k) c = (c & 1) ? (c >> 1) ^ 0x82F63B78u : c >> 1;
        T[0][i] = c;
    }
    for (uint32_t i = 0; i < 256; ++i)
        for (int s = 1; s < 8; ++s) T[s][i] = (T[s - 1][i] >> 8) ^ T[0][T[s - 1][i] & 0xff];
    t_ready = 1;
}

#if defined(__x86_64__)
#include <nmmintrin.h>
/* SSE4.2 has the Castagnoli polynomial in hardware (crc32 r64, r/m64: 3 cycles of latency per 8 bytes; the table walk above
 * is ~1.3 cycles per BYTE).  Three independent streams would triple it again; one is already 0.1 ms per 0.85 MB record.
 * Chosen at run time: the library is built on one machine and run on another. */
__attribute__((target("sse4.2"))) static uint32_t crc32c_hw(const uint8_t* p, size_t n) {
    uint64_t c = 0xffffffffu;
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        c = _mm_crc32_u64(c, v);
        p += 8;
        n -= 8;
    }
    uint32_t c32 = (uint32_t)c;
    while (n--) c32 = _mm_crc32_u8(c32, *p++);
    return c32 ^ 0xffffffffu;
}
#endif

uint32_t xmc_crc32c_table(const uint8_t* p, size_t n);

uint32_t xmc_crc32c(const uint8_t* p, size_t n) {
#if defined(__x86_64__)
    static int hw = -1;
    if (hw < 0) hw = __builtin_cpu_supports("sse4.2") ? 1 : 0;
    if (hw) return crc32c_hw(p, n);
#endif
    return xmc_crc32c_table(p, n);
}

/* the portable path (and the reference the hardware path is tested against) */
uint32_t xmc_crc32c_table(const uint8_t* p, size_t n) {
    if (!t_ready) crc_init();
    uint32_t c = 0xffffffffu;
    while (n >= 8) {
        uint32_t lo, hi;
        memcpy(&lo, p, 4);
        memcpy(&hi, p + 4, 4);
        lo ^= c;
        c = T[7][lo & 0xff] ^ T[6][(lo >> 8) & 0xff] ^ T[5][(lo >> 16) & 0xff] ^ T[4][lo >> 24] ^
            T[3][hi & 0xff] ^ T[2][(hi >> 8) & 0xff] ^ T[1][(hi >> 16) & 0xff] ^ T[0][hi >> 24];
        p += 8;
        n -= 8;
    }
    while (n--) c = (c >> 8) ^ T[0][(c ^ *p++) & 0xff];
    return c ^ 0xffffffffu;
}

/* TFRecord's "masked" CRC: rotate right by 15 and add a constant */
uint32_t xmc_masked_crc32c(const uint8_t* p, size_t n) {
    const uint32_t c = xmc_crc32c(p, n);
    return ((c >> 15) | (c << 17)) + 0xa282ead8u;
}

/* ---- PNG scanline un-filter (PNG spec section 9).  `raw` = inflated IDAT stream of a non-interlaced image:
 * h rows of (1 filter byte + rowbytes); `out` receives h * rowbytes pixels.  bpp = bytes per complete pixel
 * (>= 1).  Returns 0, or -1 for an unknown filter type. */
/* Round 4: one row-kernel per filter with the pixel stride a compile-time constant (3 / 4 / 1 / 2 after inlining) and the
 * "no left neighbour" pixels peeled off -- the generic loop re-tested `x >= bpp` and `up != NULL` for every byte and ran at
 * ~16 cycles per byte (15 ms of the 36 ms a 640 x 480 COCO image costs on this container's cores; the other 15 are inflate).
 * With bpp = 3 the three colour channels are three independent dependency chains. */
static inline int paeth(int a, int b, int c) {
    int pa = b - c, pb = a - c;                  /* p - a, p - b with p = a + b - c */
    int pc = pa + pb;
    pa = pa < 0 ? -pa : pa; pb = pb < 0 ? -pb : pb; pc = pc < 0 ? -pc : pc;
    /* branch-free selection (masks): the three-way choice is data dependent and unpredictable -- as branches it cost ~20
     * cycles per byte in mispredictions */
    const int use_a = -((pa <= pb) & (pa <= pc)), use_b = -(pb <= pc);
    const int t = (b & use_b) | (c & ~use_b);
    return (a & use_a) | (t & ~use_a);
}

#if defined(__SSE2__)
#include <emmintrin.h>
/* Paeth rows of 3- / 4-byte pixels, one PIXEL per step in 16-bit lanes: the scalar form runs three (four) separate ~20-cycle
 * chains per pixel (7 cycles per byte measured); here the predictor of all channels is ~10 instructions on one register:
 *   pa = |b - c|, pb = |a - c|, pc = |(b - c) + (a - c)|; the nearest of a, b, c in that order of preference (PNG spec 9.4).
 * The 4-byte loads of a 3-byte pixel read one byte beyond it (lane 3 is computed and dropped): the vector loop stops where
 * that byte would leave the row and returns the position of the first pixel it did not do.  Stores are exactly BPP bytes. */
static inline __attribute__((always_inline)) int paeth_row_sse2(const uint8_t* in, uint8_t* cur, const uint8_t* up, int n, const int bpp) {
    const __m128i zero = _mm_setzero_si128();
    __m128i a = zero, c = zero;                                  /* left and upper-left pixel, 16-bit lanes */
    int x = 0;
    for (; x + 4 <= n; x += bpp) {
        int32_t iv, uv;
        memcpy(&iv, in + x, 4);
        memcpy(&uv, up + x, 4);
        const __m128i b = _mm_unpacklo_epi8(_mm_cvtsi32_si128(uv), zero);
        const __m128i vb = _mm_sub_epi16(b, c), va = _mm_sub_epi16(a, c);        /* p - a, p - b */
        const __m128i vc = _mm_add_epi16(vb, va);
        const __m128i pa = _mm_max_epi16(vb, _mm_sub_epi16(zero, vb));
        const __m128i pb = _mm_max_epi16(va, _mm_sub_epi16(zero, va));
        const __m128i pc = _mm_max_epi16(vc, _mm_sub_epi16(zero, vc));
        const __m128i smallest = _mm_min_epi16(pc, _mm_min_epi16(pa, pb));
        const __m128i is_a = _mm_cmpeq_epi16(smallest, pa), is_b = _mm_cmpeq_epi16(smallest, pb);
        const __m128i bc = _mm_or_si128(_mm_and_si128(is_b, b), _mm_andnot_si128(is_b, c));
        const __m128i near = _mm_or_si128(_mm_and_si128(is_a, a), _mm_andnot_si128(is_a, bc));
        const __m128i d8 = _mm_add_epi8(_mm_cvtsi32_si128(iv), _mm_packus_epi16(near, near));
        const int32_t o = _mm_cvtsi128_si32(d8);
        memcpy(cur + x, &o, (size_t)bpp);
        c = b;
        a = _mm_unpacklo_epi8(d8, zero);
    }
    return x;
}

#define PAETH_STEP_V(a, b, c, rawv, d8)                                                                          \
    do {                                                                                                          \
        const __m128i vb_ = _mm_sub_epi16(b, c), va_ = _mm_sub_epi16(a, c), vc_ = _mm_add_epi16(vb_, va_);        \
        const __m128i pa_ = _mm_max_epi16(vb_, _mm_sub_epi16(zero, vb_)), pb_ = _mm_max_epi16(va_, _mm_sub_epi16(zero, va_)); \
        const __m128i pc_ = _mm_max_epi16(vc_, _mm_sub_epi16(zero, vc_));                                         \
        const __m128i sm_ = _mm_min_epi16(pc_, _mm_min_epi16(pa_, pb_));                                          \
        const __m128i ia_ = _mm_cmpeq_epi16(sm_, pa_), ib_ = _mm_cmpeq_epi16(sm_, pb_);                           \
        const __m128i bc_ = _mm_or_si128(_mm_and_si128(ib_, b), _mm_andnot_si128(ib_, c));                        \
        const __m128i nr_ = _mm_or_si128(_mm_and_si128(ia_, a), _mm_andnot_si128(ia_, bc_));                      \
        (d8) = _mm_add_epi8(rawv, _mm_packus_epi16(nr_, nr_));                                                    \
    } while (0)

#define PAETH_STEP(a, b, c, raw, d8)                                                                              \
    do {                                                                                                          \
        const __m128i vb_ = _mm_sub_epi16(b, c), va_ = _mm_sub_epi16(a, c), vc_ = _mm_add_epi16(vb_, va_);        \
        const __m128i pa_ = _mm_max_epi16(vb_, _mm_sub_epi16(zero, vb_)), pb_ = _mm_max_epi16(va_, _mm_sub_epi16(zero, va_)); \
        const __m128i pc_ = _mm_max_epi16(vc_, _mm_sub_epi16(zero, vc_));                                         \
        const __m128i sm_ = _mm_min_epi16(pc_, _mm_min_epi16(pa_, pb_));                                          \
        const __m128i ia_ = _mm_cmpeq_epi16(sm_, pa_), ib_ = _mm_cmpeq_epi16(sm_, pb_);                           \
        const __m128i bc_ = _mm_or_si128(_mm_and_si128(ib_, b), _mm_andnot_si128(ib_, c));                        \
        const __m128i nr_ = _mm_or_si128(_mm_and_si128(ia_, a), _mm_andnot_si128(ia_, bc_));                      \
        (d8) = _mm_add_epi8(_mm_cvtsi32_si128(raw), _mm_packus_epi16(nr_, nr_));                                  \
    } while (0)

static inline int32_t load_i32(const uint8_t* p) { int32_t v; memcpy(&v, p, 4); return v; }
static inline __attribute__((always_inline)) void store_px(uint8_t* p, __m128i d8, const int bpp) {
    const int32_t o = _mm_cvtsi128_si32(d8);
    memcpy(p, &o, (size_t)bpp);
}

/* Two consecutive Paeth rows as a wavefront: pixel i of the upper row and pixel i - 1 of the lower row are independent (the
 * lower pixel needs the upper row's pixels i - 1 and i - 2, both finished), so every iteration carries two predictor chains
 * instead of one -- a single row is bound by the ~13-cycle latency of one chain per pixel.  Returns the number of upper-row
 * BYTES done (the lower row is one pixel behind); the caller finishes both rows with the scalar code. */
static inline __attribute__((always_inline)) int paeth_rows2_sse2(const uint8_t* in0, const uint8_t* in1, uint8_t* cur0, uint8_t* cur1,
                                                                  const uint8_t* up0, int n, const int bpp) {
    const __m128i zero = _mm_setzero_si128();
    __m128i a0 = zero, c0 = zero;                                /* upper row: left, upper-left */
    __m128i a1 = zero, b1 = zero, c1 = zero;                     /* lower row: left; above (= upper row's previous pixel); above-left */
    int x = 0;
    for (; x + 4 <= n; x += bpp) {
        int32_t iv0, uv0, iv1 = 0;
        memcpy(&iv0, in0 + x, 4);
        memcpy(&uv0, up0 + x, 4);
        if (x) memcpy(&iv1, in1 + x - bpp, 4);
        const __m128i b0 = _mm_unpacklo_epi8(_mm_cvtsi32_si128(uv0), zero);
        __m128i d0, d1;
        PAETH_STEP(a0, b0, c0, iv0, d0);
        PAETH_STEP(a1, b1, c1, iv1, d1);                         /* (x == 0: a dummy pixel of zeros, not stored) */
        const int32_t o0 = _mm_cvtsi128_si32(d0);
        memcpy(cur0 + x, &o0, (size_t)bpp);
        if (x) {
            const int32_t o1 = _mm_cvtsi128_si32(d1);
            memcpy(cur1 + x - bpp, &o1, (size_t)bpp);
            a1 = _mm_unpacklo_epi8(d1, zero);
        }
        c1 = b1;                                                 /* the lower row's next pixel sits under THIS upper pixel ... */
        c0 = b0;
        a0 = _mm_unpacklo_epi8(d0, zero);
        b1 = a0;                                                 /* ... whose value was just finished */
    }
    return x;
}
/* Round 5: FOUR consecutive Paeth rows as a wavefront (row r works on pixel i - r in iteration i), two rows per register: the
 * two-row form used 4 of the 8 16-bit lanes and was bound by its instruction count (~25 per pixel), not by the chain's latency.
 * Rows (0, 1) share one register, rows (2, 3) another; the "above" pixel of row r is the "left" pixel of row r - 1 (both are the
 * pixel row r - 1 finished one iteration ago), so the above-vectors are two half-register shuffles of the left-vectors.  The
 * caller has done the triangle in front (row r: pixels [0, 3 - r)); returns the first iteration not done: row r then holds
 * pixels [0, i - r). */
static inline __attribute__((always_inline)) int paeth_rows4_sse2(const uint8_t* const in[4], uint8_t* const cur[4], const uint8_t* up0,
                                                                  int n, const int bpp) {
    const __m128i zero = _mm_setzero_si128();
    const int npix = n / bpp;
#define PX8(ptr) _mm_cvtsi32_si128(load_i32(ptr))
#define PAIR16(lo8, hi8) _mm_unpacklo_epi8(_mm_unpacklo_epi32(lo8, hi8), zero)
    /* state in front of iteration 3: row r is about to do pixel 3 - r; a = left, c = above-left */
    __m128i aA = PAIR16(PX8(cur[0] + 2 * bpp), PX8(cur[1] + bpp)), cA = PAIR16(PX8(up0 + 2 * bpp), PX8(cur[0] + bpp));
    __m128i aB = PAIR16(PX8(cur[2]), zero), cB = PAIR16(PX8(cur[1]), zero);
    int i = 3;
    for (; (i + 1) * bpp + 1 <= n && i < npix; ++i) {
        const int x0 = i * bpp;
        const __m128i bA = _mm_unpacklo_epi64(_mm_unpacklo_epi8(PX8(up0 + x0), zero), aA);                       /* [up | row 0's left] */
        const __m128i bB = _mm_castpd_si128(_mm_shuffle_pd(_mm_castsi128_pd(aA), _mm_castsi128_pd(aB), 1));       /* [row 1's left | row 2's left] */
        const __m128i rawA = _mm_unpacklo_epi32(PX8(in[0] + x0), PX8(in[1] + x0 - bpp));
        const __m128i rawB = _mm_unpacklo_epi32(PX8(in[2] + x0 - 2 * bpp), PX8(in[3] + x0 - 3 * bpp));
        __m128i dA, dB;
        PAETH_STEP_V(aA, bA, cA, rawA, dA);
        PAETH_STEP_V(aB, bB, cB, rawB, dB);
        store_px(cur[0] + x0, dA, bpp);
        store_px(cur[1] + x0 - bpp, _mm_srli_si128(dA, 4), bpp);
        store_px(cur[2] + x0 - 2 * bpp, dB, bpp);
        store_px(cur[3] + x0 - 3 * bpp, _mm_srli_si128(dB, 4), bpp);
        cA = bA; cB = bB;
        aA = _mm_unpacklo_epi8(dA, zero);
        aB = _mm_unpacklo_epi8(dB, zero);
    }
#undef PX8
#undef PAIR16
    return i;
}
#endif

#if defined(__x86_64__)
#include <immintrin.h>
/* EIGHT consecutive Paeth rows as a wavefront on AVX2 (run-time dispatch): rows 0-3 are the four 64-bit elements of one 256-bit
 * register (4 x 16-bit lanes per pixel), rows 4-7 of a second one -- one predictor evaluation per FOUR pixels and two independent
 * chains per iteration.  "above" of row r = "left" of row r - 1: the above-vector is the left-vector moved up by one element
 * (vpermq), element 0 filled from memory (rows 0-3) or from row 3's left pixel (rows 4-7).  The caller has done the triangle in
 * front (row r: pixels [0, 7 - r)); returns the first iteration not done: row r then holds pixels [0, i - r). */
#define PAETH_STEP_Y(a, b, c, rawv, d8)                                                                           \
    do {                                                                                                          \
        const __m256i vb_ = _mm256_sub_epi16(b, c), va_ = _mm256_sub_epi16(a, c), vc_ = _mm256_add_epi16(vb_, va_);   \
        const __m256i pa_ = _mm256_abs_epi16(vb_), pb_ = _mm256_abs_epi16(va_), pc_ = _mm256_abs_epi16(vc_);      \
        const __m256i sm_ = _mm256_min_epi16(pc_, _mm256_min_epi16(pa_, pb_));                                    \
        const __m256i ia_ = _mm256_cmpeq_epi16(sm_, pa_), ib_ = _mm256_cmpeq_epi16(sm_, pb_);                     \
        const __m256i bc_ = _mm256_or_si256(_mm256_and_si256(ib_, b), _mm256_andnot_si256(ib_, c));               \
        const __m256i nr_ = _mm256_or_si256(_mm256_and_si256(ia_, a), _mm256_andnot_si256(ia_, bc_));             \
        (d8) = _mm256_add_epi8(rawv, _mm256_packus_epi16(nr_, nr_));                                              \
    } while (0)

__attribute__((target("avx2"))) static inline __m256i px4_y(const uint8_t* p0, const uint8_t* p1, const uint8_t* p2, const uint8_t* p3) {
    /* four pixels (NULL = zeros) -> elements 0..3, 16-bit lanes */
    const __m128i z = _mm_setzero_si128();
    const __m128i q0 = p0 ? _mm_cvtsi32_si128(load_i32(p0)) : z, q1 = p1 ? _mm_cvtsi32_si128(load_i32(p1)) : z;
    const __m128i q2 = p2 ? _mm_cvtsi32_si128(load_i32(p2)) : z, q3 = p3 ? _mm_cvtsi32_si128(load_i32(p3)) : z;
    const __m128i lo = _mm_unpacklo_epi8(_mm_unpacklo_epi32(q0, q1), z), hi = _mm_unpacklo_epi8(_mm_unpacklo_epi32(q2, q3), z);
    return _mm256_inserti128_si256(_mm256_castsi128_si256(lo), hi, 1);
}

__attribute__((target("avx2"))) static inline __attribute__((always_inline)) int paeth_rows8_body(const uint8_t* const in[8], uint8_t* const cur[8],
                                                                                                  const uint8_t* up0, int n, const int bpp) {
    const __m256i zero = _mm256_setzero_si256();
    const int npix = n / bpp;
    /* state in front of iteration 7: row r is about to do pixel 7 - r; a = left, c = above-left (row 7: pixel 0, both zero) */
    __m256i aA = px4_y(cur[0] + 6 * bpp, cur[1] + 5 * bpp, cur[2] + 4 * bpp, cur[3] + 3 * bpp);
    __m256i cA = px4_y(up0 + 6 * bpp, cur[0] + 5 * bpp, cur[1] + 4 * bpp, cur[2] + 3 * bpp);
    __m256i aB = px4_y(cur[4] + 2 * bpp, cur[5] + bpp, cur[6], NULL);
    __m256i cB = px4_y(cur[3] + 2 * bpp, cur[4] + bpp, cur[5], NULL);
    int i = 7;
    for (; (i + 1) * bpp + 1 <= n && i < npix; ++i) {
        const int x0 = i * bpp;
        const __m128i z = _mm_setzero_si128();
        const __m256i upv = _mm256_castsi128_si256(_mm_unpacklo_epi8(_mm_cvtsi32_si128(load_i32(up0 + x0)), z));
        const __m256i bA = _mm256_blend_epi32(_mm256_permute4x64_epi64(aA, 0x90), upv, 0x03);                       /* [up, a0, a1, a2] */
        const __m256i bB = _mm256_blend_epi32(_mm256_permute4x64_epi64(aB, 0x90), _mm256_permute4x64_epi64(aA, 0xff), 0x03);   /* [a3, a4, a5, a6] */
#define RAW2(p, q) _mm_unpacklo_epi32(_mm_cvtsi32_si128(load_i32(p)), _mm_cvtsi32_si128(load_i32(q)))
        const __m256i rawA = _mm256_inserti128_si256(_mm256_castsi128_si256(RAW2(in[0] + x0, in[1] + x0 - bpp)),
                                                     RAW2(in[2] + x0 - 2 * bpp, in[3] + x0 - 3 * bpp), 1);
        const __m256i rawB = _mm256_inserti128_si256(_mm256_castsi128_si256(RAW2(in[4] + x0 - 4 * bpp, in[5] + x0 - 5 * bpp)),
                                                     RAW2(in[6] + x0 - 6 * bpp, in[7] + x0 - 7 * bpp), 1);
#undef RAW2
        __m256i dA, dB;
        PAETH_STEP_Y(aA, bA, cA, rawA, dA);
        PAETH_STEP_Y(aB, bB, cB, rawB, dB);
        const __m128i dAl = _mm256_castsi256_si128(dA), dAh = _mm256_extracti128_si256(dA, 1);
        const __m128i dBl = _mm256_castsi256_si128(dB), dBh = _mm256_extracti128_si256(dB, 1);
        store_px(cur[0] + x0, dAl, bpp);
        store_px(cur[1] + x0 - bpp, _mm_srli_si128(dAl, 4), bpp);
        store_px(cur[2] + x0 - 2 * bpp, dAh, bpp);
        store_px(cur[3] + x0 - 3 * bpp, _mm_srli_si128(dAh, 4), bpp);
        store_px(cur[4] + x0 - 4 * bpp, dBl, bpp);
        store_px(cur[5] + x0 - 5 * bpp, _mm_srli_si128(dBl, 4), bpp);
        store_px(cur[6] + x0 - 6 * bpp, dBh, bpp);
        store_px(cur[7] + x0 - 7 * bpp, _mm_srli_si128(dBh, 4), bpp);
        cA = bA; cB = bB;
        aA = _mm256_unpacklo_epi8(dA, zero);
        aB = _mm256_unpacklo_epi8(dB, zero);
    }
    return i;
}
__attribute__((target("avx2"))) static int paeth_rows8_avx2_3(const uint8_t* const in[8], uint8_t* const cur[8], const uint8_t* up0, int n) {
    return paeth_rows8_body(in, cur, up0, n, 3);
}
__attribute__((target("avx2"))) static int paeth_rows8_avx2_4(const uint8_t* const in[8], uint8_t* const cur[8], const uint8_t* up0, int n) {
    return paeth_rows8_body(in, cur, up0, n, 4);
}
static int have_avx2(void) {
    static int hw = -1;
    if (hw < 0) hw = __builtin_cpu_supports("avx2") ? 1 : 0;
    return hw;
}
#endif

static inline __attribute__((always_inline)) void unfilter_row(int ft, const uint8_t* in, uint8_t* cur, const uint8_t* up, int n, const int bpp) {
    switch (ft) {
        case 0: memcpy(cur, in, (size_t)n); break;
        case 1:
            for (int x = 0; x < bpp && x < n; ++x) cur[x] = in[x];
            for (int x = bpp; x < n; ++x) cur[x] = (uint8_t)(in[x] + cur[x - bpp]);
            break;
        case 2:
            for (int x = 0; x < n; ++x) cur[x] = (uint8_t)(in[x] + up[x]);
            break;
        case 3:
            for (int x = 0; x < bpp && x < n; ++x) cur[x] = (uint8_t)(in[x] + (up[x] >> 1));
            for (int x = bpp; x < n; ++x) cur[x] = (uint8_t)(in[x] + ((cur[x - bpp] + up[x]) >> 1));
            break;
        default:
        {
            int x0 = 0;
#if defined(__SSE2__)
            if ((bpp == 3 || bpp == 4) && n % bpp == 0 && n >= 2 * bpp) x0 = paeth_row_sse2(in, cur, up, n, bpp);
#endif
            for (int x = x0; x < bpp && x < n; ++x) cur[x] = (uint8_t)(in[x] + up[x]);      /* paeth(0, b, 0) = b */
            for (int x = x0 > bpp ? x0 : bpp; x < n; ++x) cur[x] = (uint8_t)(in[x] + paeth(cur[x - bpp], up[x], up[x - bpp]));
        }
            break;
    }
}

int xmc_png_unfilter(const uint8_t* raw, uint8_t* out, int32_t h, int32_t rowbytes, int32_t bpp) {
    if (h <= 0 || rowbytes <= 0 || bpp <= 0) return -1;
    uint8_t* zero = (uint8_t*)calloc((size_t)rowbytes, 1);       /* the row above the first one */
    if (!zero) return -2;
    int rc = 0;
    for (int y = 0; y < h && rc == 0; ++y) {
        const uint8_t* in = raw + (size_t)y * (rowbytes + 1);
        const int ft = in[0];
        ++in;
        uint8_t* cur = out + (size_t)y * rowbytes;
        const uint8_t* up = y ? cur - rowbytes : zero;
        if (ft > 4) { rc = -1; break; }
#if defined(__x86_64__)
        if (ft == 4 && y + 7 < h && (bpp == 3 || bpp == 4) && rowbytes % bpp == 0 && rowbytes >= 16 * bpp && have_avx2()) {
            int all = 1;
            for (int r = 1; r < 8; ++r) all &= in[(size_t)r * (rowbytes + 1) - 1] == 4;
            if (all) {
                const uint8_t* inr[8];
                uint8_t* curr[8];
                for (int r = 0; r < 8; ++r) { inr[r] = in + (size_t)r * (rowbytes + 1); curr[r] = cur + (size_t)r * rowbytes; }
                for (int r = 0; r < 7; ++r) {                    /* the triangle in front of the wavefront: row r, pixels [0, 7 - r) */
                    const uint8_t* upr = r ? curr[r - 1] : up;
                    for (int x = 0; x < bpp; ++x) curr[r][x] = (uint8_t)(inr[r][x] + upr[x]);
                    for (int x = bpp; x < (7 - r) * bpp; ++x) curr[r][x] = (uint8_t)(inr[r][x] + paeth(curr[r][x - bpp], upr[x], upr[x - bpp]));
                }
                const int it = bpp == 3 ? paeth_rows8_avx2_3(inr, curr, up, rowbytes) : paeth_rows8_avx2_4(inr, curr, up, rowbytes);
                for (int r = 0; r < 8; ++r) {                    /* row r holds pixels [0, it - r): the tails, top row first */
                    const uint8_t* upr = r ? curr[r - 1] : up;
                    for (int x = (it - r) * bpp; x < rowbytes; ++x)
                        curr[r][x] = (uint8_t)(inr[r][x] + (x < bpp ? upr[x] : paeth(curr[r][x - bpp], upr[x], upr[x - bpp])));
                }
                y += 7;
                continue;
            }
        }
#endif
#if defined(__SSE2__)
        if (ft == 4 && y + 3 < h && in[rowbytes] == 4 && in[2 * (rowbytes + 1) - 1] == 4 && in[3 * (rowbytes + 1) - 1] == 4 &&
            (bpp == 3 || bpp == 4) && rowbytes % bpp == 0 && rowbytes >= 8 * bpp) {
            const uint8_t* inr[4];
            uint8_t* curr[4];
            for (int r = 0; r < 4; ++r) { inr[r] = in + (size_t)r * (rowbytes + 1); curr[r] = cur + (size_t)r * rowbytes; }
            /* the triangle in front of the wavefront, scalar: row r, pixels [0, 3 - r) */
            for (int r = 0; r < 3; ++r) {
                const uint8_t* upr = r ? curr[r - 1] : up;
                for (int x = 0; x < bpp; ++x) curr[r][x] = (uint8_t)(inr[r][x] + upr[x]);
                for (int x = bpp; x < (3 - r) * bpp; ++x) curr[r][x] = (uint8_t)(inr[r][x] + paeth(curr[r][x - bpp], upr[x], upr[x - bpp]));
            }
            const int it = bpp == 3 ? paeth_rows4_sse2(inr, curr, up, rowbytes, 3) : paeth_rows4_sse2(inr, curr, up, rowbytes, 4);                    /* row r holds pixels [0, it - r) */
            for (int r = 0; r < 4; ++r) {
                const uint8_t* upr = r ? curr[r - 1] : up;
                for (int x = (it - r) * bpp; x < rowbytes; ++x)
                    curr[r][x] = (uint8_t)(inr[r][x] + (x < bpp ? upr[x] : paeth(curr[r][x - bpp], upr[x], upr[x - bpp])));
            }
            y += 3;
            continue;
        }
        if (ft == 4 && y + 1 < h && in[rowbytes] == 4 && (bpp == 3 || bpp == 4) && rowbytes % bpp == 0 && rowbytes >= 4 * bpp) {
            const uint8_t* in1 = in + rowbytes + 1;
            uint8_t* cur1 = cur + rowbytes;
            const int x0 = bpp == 3 ? paeth_rows2_sse2(in, in1, cur, cur1, up, rowbytes, 3) : paeth_rows2_sse2(in, in1, cur, cur1, up, rowbytes, 4);          /* upper row: bytes [0, x0); lower: [0, x0 - bpp) */
            for (int x = x0; x < rowbytes; ++x) cur[x] = (uint8_t)(in[x] + paeth(cur[x - bpp], up[x], up[x - bpp]));
            for (int x = x0 - bpp; x < rowbytes; ++x) cur1[x] = (uint8_t)(in1[x] + paeth(cur1[x - bpp], cur[x], cur[x - bpp]));
            ++y;
            continue;
        }
#endif
        switch (bpp) {
            case 3: unfilter_row(ft, in, cur, up, rowbytes, 3); break;
            case 4: unfilter_row(ft, in, cur, up, rowbytes, 4); break;
            case 1: unfilter_row(ft, in, cur, up, rowbytes, 1); break;
            case 2: unfilter_row(ft, in, cur, up, rowbytes, 2); break;
            default: unfilter_row(ft, in, cur, up, rowbytes, bpp); break;
        }
    }
    free(zero);
    return rc;
}

/* ---- whole-image PNG decode in one call (round 4): chunk walk + CRC check + zlib inflate of the IDAT stream + un-filter, no
 * Python between the steps (the caller's thread holds no GIL for the whole image).  8-bit, non-interlaced, colour types 0 / 2 /
 * 4 / 6; palette images (type 3) and anything else return 1 = "use the Python path".
 *   xmc_png_info:   -> 0 and w, h, channels (bytes per pixel of the decoded rows), ctype; 1 unsupported; < 0 malformed
 *   xmc_png_decode: px receives h * w * channels bytes; scratch must hold h * (w * channels + 1) bytes */
/* ---- CRC-32 (IEEE, the PNG chunk checksum) by carry-less multiplication: the data is folded 64 bytes at a time onto four
 * 128-bit accumulators (x^512 mod P and friends as the fold constants), then 128 -> 64 -> 32 bits with a Barrett reduction
 * ("Fast CRC Computation for Generic Polynomials Using PCLMULQDQ", Gopal et al., Intel 2009).  zlib 1.2.11's crc32 walks tables
 * at ~1.3 GB/s: 0.45 ms of a 5 ms image decode; this is ~10x that.  Run-time dispatch (pclmul + sse4.1), zlib's for the tail
 * and for short inputs; tests hold it to zlib.crc32 on every length 0 .. 300 and on megabyte buffers. */
#if defined(__x86_64__)
#include <smmintrin.h>
#include <wmmintrin.h>
__attribute__((target("pclmul,sse4.1"))) static uint32_t crc32_fold(const uint8_t* buf, size_t len, uint32_t crc) {
    /* len >= 64 and a multiple of 16; crc and the result are the INVERTED running value */
    static const uint64_t __attribute__((aligned(16))) k1k2[] = {0x0154442bd4ull, 0x01c6e41596ull};
    static const uint64_t __attribute__((aligned(16))) k3k4[] = {0x01751997d0ull, 0x00ccaa009eull};
    static const uint64_t __attribute__((aligned(16))) k5k0[] = {0x0163cd6124ull, 0x0000000000ull};
    static const uint64_t __attribute__((aligned(16))) poly[] = {0x01db710641ull, 0x01f7011641ull};
    __m128i x0, x1, x2, x3, x4, x5, x6, x7, x8, y5, y6, y7, y8;
    x1 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
    x2 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
    x3 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
    x4 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
    x1 = _mm_xor_si128(x1, _mm_cvtsi32_si128((int)crc));
    x0 = _mm_load_si128((const __m128i*)k1k2);
    buf += 64;
    len -= 64;
    while (len >= 64) {
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x6 = _mm_clmulepi64_si128(x2, x0, 0x00);
        x7 = _mm_clmulepi64_si128(x3, x0, 0x00);
        x8 = _mm_clmulepi64_si128(x4, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x2 = _mm_clmulepi64_si128(x2, x0, 0x11);
        x3 = _mm_clmulepi64_si128(x3, x0, 0x11);
        x4 = _mm_clmulepi64_si128(x4, x0, 0x11);
        y5 = _mm_loadu_si128((const __m128i*)(buf + 0x00));
        y6 = _mm_loadu_si128((const __m128i*)(buf + 0x10));
        y7 = _mm_loadu_si128((const __m128i*)(buf + 0x20));
        y8 = _mm_loadu_si128((const __m128i*)(buf + 0x30));
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x5), y5);
        x2 = _mm_xor_si128(_mm_xor_si128(x2, x6), y6);
        x3 = _mm_xor_si128(_mm_xor_si128(x3, x7), y7);
        x4 = _mm_xor_si128(_mm_xor_si128(x4, x8), y8);
        buf += 64;
        len -= 64;
    }
    x0 = _mm_load_si128((const __m128i*)k3k4);                   /* four accumulators -> one */
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x3), x5);
    x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
    x1 = _mm_xor_si128(_mm_xor_si128(x1, x4), x5);
    while (len >= 16) {                                          /* the 16-byte blocks that are left */
        x2 = _mm_loadu_si128((const __m128i*)buf);
        x5 = _mm_clmulepi64_si128(x1, x0, 0x00);
        x1 = _mm_clmulepi64_si128(x1, x0, 0x11);
        x1 = _mm_xor_si128(_mm_xor_si128(x1, x2), x5);
        buf += 16;
        len -= 16;
    }
    x2 = _mm_clmulepi64_si128(x1, x0, 0x10);                     /* 128 -> 64 bits */
    x3 = _mm_setr_epi32(~0, 0, ~0, 0);
    x1 = _mm_srli_si128(x1, 8);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_loadl_epi64((const __m128i*)k5k0);
    x2 = _mm_srli_si128(x1, 4);
    x1 = _mm_and_si128(x1, x3);
    x1 = _mm_clmulepi64_si128(x1, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    x0 = _mm_load_si128((const __m128i*)poly);                   /* Barrett reduction to 32 bits */
    x2 = _mm_and_si128(x1, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x10);
    x2 = _mm_and_si128(x2, x3);
    x2 = _mm_clmulepi64_si128(x2, x0, 0x00);
    x1 = _mm_xor_si128(x1, x2);
    return (uint32_t)_mm_extract_epi32(x1, 1);
}
#endif

/* crc32(0, p, n) of zlib, folded where the CPU can */
uint32_t xmc_crc32_ieee(const uint8_t* p, size_t n) {
    uint32_t c = (uint32_t)crc32(0L, Z_NULL, 0);
#if defined(__x86_64__)
    static int hw = -1;
    if (hw < 0) hw = (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1")) ? 1 : 0;
    if (hw && n >= 64) {
        const size_t n16 = n & ~(size_t)15;
        c = ~crc32_fold(p, n16, ~c);
        p += n16;
        n -= n16;
    }
#endif
    while (n) {                                                  /* zlib takes a 32-bit length */
        const size_t k = n < 0x40000000u ? n : 0x40000000u;
        c = (uint32_t)crc32(c, p, (uInt)k);
        p += k;
        n -= k;
    }
    return c;
}

#define XMC_PNG_MAX_BYTES ((uint64_t)1 << 30)

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

int xmc_png_info(const uint8_t* d, int64_t n, int32_t* w, int32_t* h, int32_t* channels, int32_t* ctype) {
    static const uint8_t sig[8] = {0x89, 'P', 'N', 'G', '\r', '\n', 0x1a, '\n'};
    if (n < 33 || memcmp(d, sig, 8) != 0 || be32(d + 8) != 13 || memcmp(d + 12, "IHDR", 4) != 0) return -1;
    *w = (int32_t)be32(d + 16); *h = (int32_t)be32(d + 20);
    const int depth = d[24], ct = d[25], interlace = d[28];
    *ctype = ct;
    if (*w <= 0 || *h <= 0) return -1;
    if (depth != 8 || interlace != 0) return 1;
    switch (ct) {
        case 0: *channels = 1; break;
        case 2: *channels = 3; break;
        case 4: *channels = 2; break;
        case 6: *channels = 4; break;
        default: return 1;
    }
    /* the header is untrusted: a decoded image larger than XMC_PNG_MAX_BYTES (1 GiB of filtered rows; COCO images are
     * < 1 MB) is rejected HERE, before anybody allocates h * w * channels bytes for it or truncates the row stride to int32 */
    if ((uint64_t)*h * ((uint64_t)*w * (uint64_t)*channels + 1u) > XMC_PNG_MAX_BYTES) return -2;
    return 0;
}

int xmc_inflate_zlib_pieces2(const uint8_t* const* piece, const uint32_t* piece_len, int32_t npieces, uint8_t* out, uint64_t need,
                             int32_t check_adler);   /* xmc_inflate.c */

/* flags: bit 0 = verify the chunk CRCs; bit 1 = inflate with zlib instead of xmc_inflate.c (A/B, tests).  scratch must hold
 * h * (w * channels + 1) + xmc_inflate_out_slack() bytes. */
int xmc_png_decode(const uint8_t* d, int64_t n, uint8_t* px, uint8_t* scratch, int32_t flags) {
    int32_t w, h, ch, ct;
    const int rc0 = xmc_png_info(d, n, &w, &h, &ch, &ct);
    if (rc0 != 0) return rc0;
    const int verify_crc = flags & 1, use_zlib = flags & 2;
    const size_t rowbytes = (size_t)w * ch, need = (size_t)h * (rowbytes + 1);
    z_stream zs;
    if (use_zlib) {
        memset(&zs, 0, sizeof zs);
        if (inflateInit(&zs) != Z_OK) return -3;
        zs.next_out = scratch;
        zs.avail_out = (uInt)need;
    }
    const uint8_t* piece_s[16];
    uint32_t len_s[16];
    const uint8_t** piece = piece_s;
    uint32_t* plen = len_s;
    int npieces = 0, cap = 16;
    int64_t pos = 8;
    int done = 0, rc = 0;
    while (pos + 12 <= n && !done) {
        const uint32_t ln = be32(d + pos);
        const uint8_t* typ = d + pos + 4;
        const uint8_t* body = d + pos + 8;
        if (pos + 12 + (int64_t)ln > n) { rc = -4; break; }
        if (verify_crc) {
            const uint32_t c = xmc_crc32_ieee(typ, (size_t)4 + ln);
            if (c != be32(body + ln)) { rc = -5; break; }
        }
        if (memcmp(typ, "IDAT", 4) == 0) {
            if (use_zlib) {
                zs.next_in = (Bytef*)body;
                zs.avail_in = (uInt)ln;
                const int zr = inflate(&zs, Z_NO_FLUSH);
                if (zr != Z_OK && zr != Z_STREAM_END) { rc = -6; break; }
            } else {
                if (npieces == cap) {                            /* (encoders that cut the stream into 8 KiB IDAT chunks) */
                    cap *= 2;
                    const uint8_t** np_ = (const uint8_t**)malloc((size_t)cap * sizeof *np_);
                    uint32_t* nl_ = (uint32_t*)malloc((size_t)cap * sizeof *nl_);
                    if (!np_ || !nl_) { free(np_); free(nl_); rc = -3; break; }
                    memcpy(np_, piece, (size_t)npieces * sizeof *np_);
                    memcpy(nl_, plen, (size_t)npieces * sizeof *nl_);
                    if (piece != piece_s) { free(piece); free(plen); }
                    piece = np_; plen = nl_;
                }
                piece[npieces] = body; plen[npieces] = ln; ++npieces;
            }
        } else if (memcmp(typ, "IEND", 4) == 0) {
            done = 1;
        }
        pos += 12 + (int64_t)ln;
    }
    if (use_zlib) {
        if (rc == 0 && zs.total_out != need) rc = -7;
        inflateEnd(&zs);
    } else if (rc == 0) {
        /* one integrity check always runs: the chunk CRCs above, or the stream's own Adler-32 */
        const int ir = xmc_inflate_zlib_pieces2(piece, plen, npieces, scratch, (uint64_t)need, !verify_crc);
        if (ir != 0) rc = ir == -2 ? -7 : (ir == -3 ? -3 : -6);
    }
    if (piece != piece_s) { free(piece); free(plen); }
    if (rc != 0) return rc;
    return xmc_png_unfilter(scratch, px, h, (int32_t)rowbytes, ch) == 0 ? 0 : -8;
}

/* ---- uint8 (hs, ws, 3) -> float32 (hd, wd, 3) in [0, 1]: tf.image.convert_image_dtype (x / 255) followed by
 * tf.image.resize(method="bilinear") (TF2: half-pixel centres, antialias=False, edges clamped) and an optional
 * left-right flip (tf.image.stateless_random_flip_left_right), then clip to [0, 1] (coco_dataset.py:133-137). */
/* -> 0, or -12 (ENOMEM: the tap table of an output wider than 1024 could not be allocated; dst is untouched) */
int xmc_resize_bilinear_rgb(const uint8_t* src, int32_t hs, int32_t ws, float* dst, int32_t hd, int32_t wd, int32_t flip) {
    const float sy = (float)hs / (float)hd, sx = (float)ws / (float)wd;
    /* round 5: the column taps and weights are the same for every output row -- computed once (they were re-derived with
     * floorf / ceilf for each of the hd * wd pixels: 0.45 ms of a 5.7 ms example); same expressions, same results */
    enum { XMC_RS_STACK = 1024 };
    int32_t xo0_s[XMC_RS_STACK], xo1_s[XMC_RS_STACK];
    float lx_s[XMC_RS_STACK];
    int32_t *xo0 = xo0_s, *xo1 = xo1_s;
    float* lxs = lx_s;
    void* heap = NULL;
    if (wd > XMC_RS_STACK) {
        heap = malloc((size_t)wd * (2 * sizeof(int32_t) + sizeof(float)));
        if (!heap) return -12;
        xo0 = (int32_t*)heap; xo1 = xo0 + wd; lxs = (float*)(xo1 + wd);
    }
    for (int x = 0; x < wd; ++x) {
        const float fx = ((float)x + 0.5f) * sx - 0.5f;
        const float flx = __builtin_floorf(fx);
        int x0 = (int)flx, x1 = (int)__builtin_ceilf(fx);
        lxs[x] = fx - flx;
        if (x0 < 0) x0 = 0;
        if (x1 > ws - 1) x1 = ws - 1;
        if (x1 < 0) x1 = 0;
        xo0[x] = x0 * 3; xo1[x] = x1 * 3;
    }
    const float k = 1.f / 255.f;
    for (int y = 0; y < hd; ++y) {
        const float fy = ((float)y + 0.5f) * sy - 0.5f;
        const float fl = __builtin_floorf(fy);
        int y0 = (int)fl, y1 = (int)__builtin_ceilf(fy);
        const float ly = fy - fl;
        if (y0 < 0) y0 = 0;
        if (y1 > hs - 1) y1 = hs - 1;
        if (y1 < 0) y1 = 0;
        const uint8_t* r0 = src + (size_t)y0 * ws * 3;
        const uint8_t* r1 = src + (size_t)y1 * ws * 3;
        float* orow = dst + (size_t)y * wd * 3;
        for (int x = 0; x < wd; ++x) {
            const uint8_t *p00 = r0 + xo0[x], *p01 = r0 + xo1[x], *p10 = r1 + xo0[x], *p11 = r1 + xo1[x];
            const float lx = lxs[x];
            float* o = orow + (size_t)(flip ? wd - 1 - x : x) * 3;
            for (int c = 0; c < 3; ++c) {
                const float a = p00[c] * k, b = p01[c] * k, d = p10[c] * k, e = p11[c] * k;
                const float top = a + (b - a) * lx, bot = d + (e - d) * lx;
                const float v = top + (bot - top) * ly;
                o[c] = v < 0.f ? 0.f : (v > 1.f ? 1.f : v);
            }
        }
    }
    free(heap);
    return 0;
}

int xmc_io_abi_version(void) { return 4; }
