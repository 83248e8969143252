/* A one-shot DEFLATE / zlib decoder for the PNG path of the input pipeline (the reference decodes its
 * TFRecord images with tf.io.decode_png inside tf.data, xmcgan/libml/coco_dataset.py:107-112; here the decode runs in worker
 * threads of this library).  Round 4 measured the per-example cost of a 640 x 480 COCO-sized PNG at 5.3 ms of zlib inflate +
 * 2.5 ms of un-filtering on one core -- the whole step needs 3.7 k examples/s from a 16-core quota.  This decoder does what
 * zlib's streaming inflate cannot assume: the WHOLE compressed stream and the WHOLE output buffer are in memory, both with
 * slack behind them, so the hot loop has no end-of-buffer cases at all:
 *   - a 64-bit bit buffer refilled with one unaligned 8-byte load (at most twice per length / distance pair),
 *   - two-level tables (11-bit root for literals / lengths, 10-bit for distances) whose entries carry the bits to drop, the
 *     extra-bit count and the base value; a root entry carries up to TWO literals: one lookup + one shift + one 2-byte store,
 *   - matches copied as 8-byte words advancing by min(distance, 8) (distance 1: a splatted word), writing up to 7 bytes past
 *     the match -- into the caller's slack.
 * Malformed input never reads or writes out of bounds: the input is copied behind zero padding (reads past the end decode
 * zeros and are caught by the position check of every iteration), the output check runs once per iteration against a margin
 * that covers the longest thing an iteration can write, and every table slot no code maps to is an "invalid" entry.
 * The Adler-32 trailer is verified unless the caller says its own check covers the stream (the PNG path with chunk CRCs on:
 * the CRC-32 of every IDAT chunk covers the same bytes, and Adler-32 over the 0.9 MB of output costs 6 % of a decode).
 * Written from RFC 1950 / 1951; tests/test_input_pipeline.py holds it to zlib's output on every block type. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XI_OK 0
#define XI_EINPUT (-1)   /* malformed stream */
#define XI_ESIZE (-2)    /* output does not have the announced size / input truncated */
#define XI_ENOMEM (-3)

#define XI_IN_PAD 32     /* zero bytes the input copy carries behind the stream */
#define XI_OUT_SLACK 320 /* bytes the OUTPUT buffer must have behind `need`: 8 literals + a 258-byte match + word-copy overshoot */

#define LT_BITS 11
#define DT_BITS 10
#define LT_SIZE ((1 << LT_BITS) + 288 * 16)
#define DT_SIZE ((1 << DT_BITS) + 32 * 32)

/* table entry: bits 0-7 bits to drop | bits 8-11 extra bits (or sub-table index bits) | bits 12-14 kind | bit 15 literal(s) |
 * bits 16-31 value.  A literal entry carries ONE or TWO literals (bit 14: two; the second in bits 24-31): Huffman decoding is a
 * serial chain -- index, load, shift, next index: ~10 cycles per symbol whatever the instruction count -- and on photo-like PNG
 * rows nearly every symbol is a literal of 4-7 bits, so an 11-bit root index usually determines the NEXT literal as well. */
#define K_BASE 1u
#define K_EOB 2u
#define K_SUB 3u
#define K_BAD 4u
#define F_LIT 0x8000u
#define F_TWO 0x4000u
#define ENT(drop, extra, kind, val) ((uint32_t)(drop) | ((uint32_t)(extra) << 8) | ((uint32_t)(kind) << 12) | ((uint32_t)(val) << 16))
#define ENT_LIT(drop, val) ((uint32_t)(drop) | F_LIT | ((uint32_t)(val) << 16))
#define E_DROP(e) ((e) & 63u)
#define E_ISLIT(e) ((e) & F_LIT)
#define E_EXTRA(e) (((e) >> 8) & 15u)
#define E_KIND(e) (((e) >> 12) & 7u)
#define E_VAL(e) ((e) >> 16)

static const uint16_t len_base[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t len_extra[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t dist_base[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t dist_extra[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};
static const uint8_t precode_order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};

static inline uint32_t rev_bits(uint32_t c, int n) {
    uint32_t r = 0;
    for (int i = 0; i < n; ++i) { r = (r << 1) | (c & 1u); c >>= 1; }
    return r;
}

/* what symbol `sym` of a literal/length (which = 0), distance (1) or code-length (2) alphabet decodes to, without the drop field */
static inline uint32_t sym_entry(int which, int sym) {
    if (which == 2) return ENT_LIT(0, sym);
    if (which == 1) return sym < 30 ? ENT(0, dist_extra[sym], K_BASE, dist_base[sym]) : ENT(0, 0, K_BAD, 0);
    if (sym < 256) return ENT_LIT(0, sym);
    if (sym == 256) return ENT(0, 0, K_EOB, 0);
    return sym < 286 ? ENT(0, len_extra[sym - 257], K_BASE, len_base[sym - 257]) : ENT(0, 0, K_BAD, 0);
}

/* Canonical Huffman code (RFC 1951 3.2.2) of `n` symbols with lengths lens[] (0 = unused, <= 15) -> two-level decode table with
 * a root of `root` bits; slots no code reaches are K_BAD.  Returns 0, or -1 for an over-subscribed set of lengths. */
static int build_table(const uint8_t* lens, int n, int which, int root, uint32_t* tab, int cap) {
    int count[16] = {0}, next[16];
    uint8_t submax[1 << LT_BITS];
    for (int i = 0; i < n; ++i) ++count[lens[i]];
    count[0] = 0;
    int code = 0, left = 1;
    for (int l = 1; l <= 15; ++l) {
        left = (left << 1) - count[l];
        if (left < 0) return -1;                                 /* over-subscribed */
        code = (code + count[l - 1]) << 1;
        next[l] = code;
    }
    const int rsize = 1 << root;
    int nx0[16];
    memcpy(nx0, next, sizeof nx0);
    for (int i = 0; i < rsize; ++i) { tab[i] = ENT(0, 0, K_BAD, 0); submax[i] = 0; }
    /* pass 1: the longest code behind every root prefix */
    int nx[16];
    memcpy(nx, next, sizeof nx);
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (l <= root) { if (l) ++nx[l]; continue; }
        const uint32_t r = rev_bits((uint32_t)nx[l]++, l);
        const int pre = (int)(r & (uint32_t)(rsize - 1));
        if (l - root > submax[pre]) submax[pre] = (uint8_t)(l - root);
    }
    int used = rsize;
    for (int i = 0; i < rsize; ++i)
        if (submax[i]) {
            const int sz = 1 << submax[i];
            if (used + sz > cap) return -1;
            tab[i] = ENT(root, submax[i], K_SUB, used);
            for (int k = 0; k < sz; ++k) tab[used + k] = ENT(0, 0, K_BAD, 0);
            used += sz;
        }
    /* pass 2: fill */
    for (int s = 0; s < n; ++s) {
        const int l = lens[s];
        if (!l) continue;
        const uint32_t r = rev_bits((uint32_t)next[l]++, l);
        const uint32_t e = sym_entry(which, s);
        if (l <= root) {
            for (uint32_t k = r; k < (uint32_t)rsize; k += 1u << l) tab[k] = e | (uint32_t)l;
        } else {
            const uint32_t p = tab[r & (uint32_t)(rsize - 1)];
            const int sb = (int)E_EXTRA(p);
            uint32_t* sub = tab + E_VAL(p);
            for (uint32_t k = r >> root; k < (1u << sb); k += 1u << (l - root)) sub[k] = e | (uint32_t)(l - root);
        }
    }
    if (which == 0) {                                            /* pair the root's literals (see the entry layout) */
        /* per FIRST literal (code r1 of l1 bits): the root slots r1 | (j << l1) take their second symbol from slot j of the
         * single-literal table -- j runs over a contiguous prefix (no dependent loads), valid when that entry's code lies inside
         * the root - l1 real bits */
        uint32_t single[1 << LT_BITS];
        memcpy(single, tab, sizeof single);
        int nx2[16];
        memcpy(nx2, nx0, sizeof nx2);
        const int nlit = n < 256 ? n : 256;
        for (int s1 = 0; s1 < nlit; ++s1) {
            const int l1 = lens[s1];
            if (!l1) continue;
            const uint32_t r1 = rev_bits((uint32_t)nx2[l1]++, l1);
            if (l1 >= root) continue;
            const int room = root - l1;
            const uint32_t hi = (uint32_t)s1 << 16;
            for (uint32_t j = 0; j < (1u << room); ++j) {
                const uint32_t e2 = single[j];
                if (E_ISLIT(e2) && (int)E_DROP(e2) <= room)
                    tab[r1 | (j << l1)] = (uint32_t)(l1 + (int)E_DROP(e2)) | F_LIT | F_TWO | hi | (E_VAL(e2) << 24);
            }
        }
    }
    return 0;
}

static inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }      /* little-endian hosts (x86-64) */
static inline void store64(uint8_t* p, uint64_t v) { memcpy(p, &v, 8); }
static inline void store16(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; memcpy(p, &h, 2); }

#define REFILL() do { bitbuf |= load64(in) << bitcnt; in += (63 - bitcnt) >> 3; bitcnt |= 56; } while (0)
#define DROP(n) do { bitbuf >>= (n); bitcnt -= (int)(n); } while (0)
#define PEEK(n) ((uint32_t)bitbuf & ((1u << (n)) - 1u))
#define LOOKUP(tab, root, e) do { \
        (e) = (tab)[PEEK(root)]; \
        if (E_KIND(e) == K_SUB) { DROP(root); (e) = (tab)[E_VAL(e) + PEEK(E_EXTRA(e))]; } \
        DROP(E_DROP(e)); \
    } while (0)

typedef struct {
    uint32_t lt[LT_SIZE];
    uint32_t dt[DT_SIZE];
    uint32_t pt[1 << 7];
} xi_tables;

/* raw DEFLATE stream at `src` (zero-padded by XI_IN_PAD bytes behind n) -> exactly `need` bytes at out (XI_OUT_SLACK bytes of slack
 * behind them).  *consumed receives the number of input bytes used. */
static int inflate_raw(const uint8_t* src, size_t n, uint8_t* out0, size_t need, size_t* consumed, xi_tables* T) {
    const uint8_t* in = src;
    const uint8_t* const in_lim = src + n + 8;                   /* a position past this has decoded padding */
    uint8_t* out = out0;
    uint8_t* const out_lim = out0 + need;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    int last = 0;
    while (!last) {
        if (in > in_lim) return XI_ESIZE;
        REFILL();
        last = (int)PEEK(1); DROP(1);
        const uint32_t type = PEEK(2); DROP(2);
        if (type == 0) {                                         /* stored: byte-align, LEN / NLEN, copy */
            DROP(bitcnt & 7);
            const uint8_t* p = in - (bitcnt >> 3);
            if (p + 4 > src + n) return XI_ESIZE;
            const uint32_t len = (uint32_t)p[0] | ((uint32_t)p[1] << 8), nlen = (uint32_t)p[2] | ((uint32_t)p[3] << 8);
            if ((len ^ nlen) != 0xffffu) return XI_EINPUT;
            p += 4;
            if ((size_t)(src + n - p) < len) return XI_ESIZE;
            if ((size_t)(out_lim - out) < len) return XI_ESIZE;
            memcpy(out, p, len);
            out += len;
            in = p + len;
            bitbuf = 0; bitcnt = 0;
            continue;
        }
        if (type == 3) return XI_EINPUT;
        uint8_t lens[288 + 32];
        int nlit, ndist;
        if (type == 1) {                                         /* fixed code, RFC 1951 3.2.6 */
            for (int i = 0; i < 144; ++i) lens[i] = 8;
            for (int i = 144; i < 256; ++i) lens[i] = 9;
            for (int i = 256; i < 280; ++i) lens[i] = 7;
            for (int i = 280; i < 288; ++i) lens[i] = 8;
            for (int i = 0; i < 32; ++i) lens[288 + i] = 5;
            nlit = 288; ndist = 32;
            if (build_table(lens, 288, 0, LT_BITS, T->lt, LT_SIZE) || build_table(lens + 288, 32, 1, DT_BITS, T->dt, DT_SIZE)) return XI_EINPUT;
        } else {                                                 /* dynamic code, 3.2.7 */
            nlit = (int)PEEK(5) + 257; DROP(5);
            ndist = (int)PEEK(5) + 1; DROP(5);
            const int ncode = (int)PEEK(4) + 4; DROP(4);
            if (nlit > 286 || ndist > 30) return XI_EINPUT;
            uint8_t pl[19] = {0};
            for (int i = 0; i < ncode; ++i) {
                if (bitcnt < 3) REFILL();
                pl[precode_order[i]] = (uint8_t)PEEK(3); DROP(3);
            }
            if (build_table(pl, 19, 2, 7, T->pt, 1 << 7)) return XI_EINPUT;
            int i = 0;
            while (i < nlit + ndist) {
                if (in > in_lim) return XI_ESIZE;
                REFILL();
                const uint32_t e = T->pt[PEEK(7)];
                if (!E_ISLIT(e)) return XI_EINPUT;
                DROP(E_DROP(e));
                const int sym = (int)(E_VAL(e) & 255u);
                if (sym < 16) { lens[i++] = (uint8_t)sym; continue; }
                int rep, val = 0;
                if (sym == 16) {
                    if (i == 0) return XI_EINPUT;
                    val = lens[i - 1];
                    rep = 3 + (int)PEEK(2); DROP(2);
                } else if (sym == 17) {
                    rep = 3 + (int)PEEK(3); DROP(3);
                } else {
                    rep = 11 + (int)PEEK(7); DROP(7);
                }
                if (i + rep > nlit + ndist) return XI_EINPUT;
                while (rep--) lens[i++] = (uint8_t)val;
            }
            if (lens[256] == 0) return XI_EINPUT;                /* no end-of-block code */
            if (build_table(lens, nlit, 0, LT_BITS, T->lt, LT_SIZE) || build_table(lens + nlit, ndist, 1, DT_BITS, T->dt, DT_SIZE)) return XI_EINPUT;
        }
        const uint32_t* const lt = T->lt;
        const uint32_t* const dt = T->dt;
#define EMIT_LITS(e) do { DROP(E_DROP(e)); store16(out, (e) >> 16); out += 1 + (((e) >> 14) & 1u); } while (0)
        /* `e` is always the root entry of the CURRENT bit position, looked up as early as possible (before a match is copied,
         * before the loop's checks): a refill only ORs bits in above the counted ones, so it never invalidates it, and the 64-bit
         * buffer holds real stream bits up to 64 - (bits dropped since the last refill), which every PEEK below stays inside */
        uint32_t e;
        REFILL();
        e = lt[PEEK(LT_BITS)];
        for (;;) {
            if (in > in_lim || out > out_lim) return XI_ESIZE;
            REFILL();
            if (E_ISLIT(e)) {                                    /* up to four literal entries (<= 11 bits each) per refill */
                EMIT_LITS(e);
                e = lt[PEEK(LT_BITS)];
                if (E_ISLIT(e)) {
                    EMIT_LITS(e);
                    e = lt[PEEK(LT_BITS)];
                    if (E_ISLIT(e)) {
                        EMIT_LITS(e);
                        e = lt[PEEK(LT_BITS)];
                        if (E_ISLIT(e)) { EMIT_LITS(e); e = lt[PEEK(LT_BITS)]; continue; }
                    }
                }
            }
            if (E_KIND(e) == K_SUB) {                            /* a code longer than the root */
                DROP(LT_BITS);
                e = lt[E_VAL(e) + PEEK(E_EXTRA(e))];
                if (E_ISLIT(e)) { EMIT_LITS(e); e = lt[PEEK(LT_BITS)]; continue; }
            }
            DROP(E_DROP(e));
            if (E_KIND(e) != K_BASE) {
                if (E_KIND(e) == K_EOB) break;
                return XI_EINPUT;
            }
            const uint32_t xb = E_EXTRA(e);
            const uint32_t len = E_VAL(e) + PEEK(xb);            /* at most 3 x 11 + 15 + 5 = 53 of the >= 56 bits are gone */
            DROP(xb);
            REFILL();
            uint32_t d;
            LOOKUP(dt, DT_BITS, d);
            if (E_KIND(d) != K_BASE) return XI_EINPUT;
            const uint32_t db = E_EXTRA(d);
            const uint32_t dist = E_VAL(d) + PEEK(db);
            DROP(db);
            if (dist > (size_t)(out - out0)) return XI_EINPUT;
            e = lt[PEEK(LT_BITS)];                               /* the next symbol's entry loads while the match is copied */
            const uint8_t* s = out - dist;
            uint8_t* const oe = out + len;
            if (dist == 1) {                                     /* runs: flat image areas */
                const uint64_t v = 0x0101010101010101ull * s[0];
                do { store64(out, v); out += 8; } while (out < oe);
            } else {
                /* 8-byte words advancing by min(distance, 8): with distance < 8 only the first `distance` bytes of a loaded
                 * word are final, and exactly those are what the next word reads -- the valid prefix grows word by word (the
                 * rest lands behind the match: slack).  RGB rows match at distance 3 all the time; byte loops cost 4x this. */
                const size_t step = dist < 8 ? dist : 8;
                do { store64(out, load64(s)); out += step; s += step; } while (out < oe);
            }
            out = oe;
        }
    }
    if (out != out_lim) return XI_ESIZE;
    {
        const uint8_t* p = in - (bitcnt >> 3);                   /* whole bytes still in the bit buffer were not consumed */
        if (p > src + n) return XI_ESIZE;
        *consumed = (size_t)(p - src);
    }
    return XI_OK;
}

/* zlib stream (RFC 1950: 2-byte header, DEFLATE data, Adler-32) -> exactly `need` bytes at out, which must have XI_OUT_SLACK
 * bytes of slack behind them.  The stream may be given in pieces (the IDAT chunks of a PNG): they are gathered behind zero
 * padding first.  Returns XI_OK or a negative error. */
static uint32_t adler32_of(const uint8_t* p, size_t n) {       /* RFC 1950: sums mod 65521, reduced every 5552 bytes */
    uint32_t a = 1, b = 0;
    while (n) {
        size_t k = n < 5552 ? n : 5552;
        n -= k;
        for (; k >= 8; k -= 8, p += 8) {
            a += p[0]; b += a; a += p[1]; b += a; a += p[2]; b += a; a += p[3]; b += a;
            a += p[4]; b += a; a += p[5]; b += a; a += p[6]; b += a; a += p[7]; b += a;
        }
        for (; k; --k) { a += *p++; b += a; }
        a %= 65521u; b %= 65521u;
    }
    return (b << 16) | a;
}

int xmc_inflate_zlib_pieces2(const uint8_t* const* piece, const uint32_t* piece_len, int32_t npieces, uint8_t* out, uint64_t need,
                             int32_t check_adler) {
    size_t n = 0;
    for (int i = 0; i < npieces; ++i) n += piece_len[i];
    if (n < 6) return XI_EINPUT;
    uint8_t* buf = (uint8_t*)malloc(n + XI_IN_PAD + sizeof(xi_tables) + 64);
    if (!buf) return XI_ENOMEM;
    size_t o = 0;
    for (int i = 0; i < npieces; ++i) { memcpy(buf + o, piece[i], piece_len[i]); o += piece_len[i]; }
    memset(buf + n, 0, XI_IN_PAD);
    xi_tables* T = (xi_tables*)(((uintptr_t)(buf + n + XI_IN_PAD) + 63) & ~(uintptr_t)63);
    int rc = XI_OK;
    const unsigned cmf = buf[0], flg = buf[1];
    if ((cmf & 15) != 8 || (cmf >> 4) > 7 || ((cmf << 8) | flg) % 31 != 0 || (flg & 32)) rc = XI_EINPUT;   /* deflate, window <= 32 K, no dictionary */
    size_t used = 0;
    if (rc == XI_OK) rc = inflate_raw(buf + 2, n - 2, out, (size_t)need, &used, T);
    if (rc == XI_OK && used + 2 + 4 > n) rc = XI_ESIZE;          /* the Adler-32 trailer must at least be there */
    if (rc == XI_OK && check_adler) {
        const uint8_t* t = buf + 2 + used;
        const uint32_t want = ((uint32_t)t[0] << 24) | ((uint32_t)t[1] << 16) | ((uint32_t)t[2] << 8) | t[3];
        if (adler32_of(out, (size_t)need) != want) rc = XI_EINPUT;
    }
    free(buf);
    return rc;
}

int xmc_inflate_zlib_pieces(const uint8_t* const* piece, const uint32_t* piece_len, int32_t npieces, uint8_t* out, uint64_t need) {
    return xmc_inflate_zlib_pieces2(piece, piece_len, npieces, out, need, 1);
}

int xmc_inflate_zlib(const uint8_t* src, uint64_t n, uint8_t* out, uint64_t need) {
    if (n > 0xffffffffull) return XI_EINPUT;
    const uint32_t len = (uint32_t)n;
    return xmc_inflate_zlib_pieces(&src, &len, 1, out, need);
}

int32_t xmc_inflate_out_slack(void) { return XI_OUT_SLACK; }
