/* pgzip.h -- block-parallel decompression of an ordinary (single-stream) gzip file for the host ingest (the
 * reference reads FASTA/Q through one zlib stream per file, KSeqWrapper over gzread; at ~0.4 GB/s of text that is 70 times
 * below what the device classifies).
 *
 * A deflate stream has no index, but it can be entered at any block boundary if one accepts not to know the 32 KB of history
 * ("two-pass" parallel decompression as published with pugz / rapidgzip, restated here from the DEFLATE format, RFC 1951/1952):
 *   1. the compressed file is cut into chunks; every chunk but the first searches, bit by bit, for the header of a dynamic-Huffman
 *      block whose code tables are complete and whose first symbols decode to text;
 *   2. every chunk is inflated from there with an UNKNOWN window: a back-reference into the history yields 16-bit MARKER symbols
 *      (256 + position in the unknown 32 KB) instead of bytes, and copies of markers stay markers;
 *   3. a chunk stops exactly where the next chunk started (bit position): if its block boundaries step over that position the
 *      next chunk's start was not a real one and the chunk simply goes on -- so a wrong guess in step 1 costs time, never
 *      correctness;
 *   4. in file order, every chunk's last 32 KB are resolved with the previous chunk's window (cheap), then all chunks turn their
 *      symbols into bytes in parallel; CRC-32 and length of every gzip member are checked against its trailer (per-chunk CRCs
 *      combined with crc32_combine).
 * Own inflate (table-driven Huffman decoding, stored / fixed / dynamic blocks, several members per file); zlib is used for
 * crc32 only.  Header-only, host code, no device dependency.  Text inputs only for step 1's plausibility test (bytes 9, 10, 13,
 * 32..126); a file where no block start is found this way is still decoded correctly, by one thread. */
#ifndef MTB_HOST_PGZIP_H
#define MTB_HOST_PGZIP_H
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace mtbhost {

namespace pgz {

static constexpr uint32_t WIN = 32768;
static constexpr int PRIMARY_BITS = 10;

/* LSB-first bit reader over a byte range; `pos` = absolute bit position of the next unread bit */
struct BitReader {
    const uint8_t *p; size_t n;         /* bytes */
    uint64_t pos;                       /* bits */
    BitReader(const uint8_t *d, size_t len, uint64_t bitpos) : p(d), n(len), pos(bitpos) {}
    inline bool eof(uint32_t need) const { return pos + need > (uint64_t)n * 8; }
    /* up to 57 bits starting at pos, zero-padded beyond the end */
    inline uint64_t peek() const {
        const size_t byte = (size_t)(pos >> 3);
        uint64_t v = 0;
        if (byte + 8 <= n) memcpy(&v, p + byte, 8);
        else { for (size_t i = 0; byte + i < n && i < 8; i++) v |= (uint64_t)p[byte + i] << (8 * i); }
        return v >> (pos & 7);
    }
    inline uint32_t bits(uint32_t k) { const uint32_t v = (uint32_t)(peek() & ((1ull << k) - 1ull)); pos += k; return v; }
};

/* canonical Huffman code: lookup of the next PRIMARY_BITS bits -> (symbol, length) for short codes, a canonical walk for the
 * long ones */
struct Huff {
    uint16_t count[16];          /* codes per length */
    uint16_t symbol[320];        /* symbols in canonical order */
    uint16_t fast[1 << PRIMARY_BITS];      /* symbol << 4 | length; length 0 = long code */
    int max_len = 0;
    /* returns 0 complete, 1 incomplete (under-subscribed), -1 over-subscribed / invalid */
    int build(const uint8_t *len, int n) {
        memset(count, 0, sizeof(count));
        for (int i = 0; i < n; i++) count[len[i]]++;
        max_len = 0;
        for (int l = 15; l >= 1; l--) if (count[l]) { max_len = l; break; }
        int left = 1;
        for (int l = 1; l <= 15; l++) { left <<= 1; left -= count[l]; if (left < 0) return -1; }
        uint16_t offs[16]; offs[1] = 0;
        for (int l = 1; l < 15; l++) offs[l + 1] = (uint16_t)(offs[l] + count[l]);
        for (int i = 0; i < n; i++) if (len[i]) symbol[offs[len[i]]++] = (uint16_t)i;
        /* primary table: every code of length <= PRIMARY_BITS, bit-reversed (deflate packs codes MSB-first into an LSB-first stream) */
        memset(fast, 0, sizeof(fast));
        int code = 0, idx = 0;
        for (int l = 1; l <= 15; l++) {
            for (int k = 0; k < count[l]; k++, code++, idx++) {
                if (l <= PRIMARY_BITS) {
                    int rev = 0; for (int b = 0; b < l; b++) rev |= ((code >> b) & 1) << (l - 1 - b);
                    for (int fill = rev; fill < (1 << PRIMARY_BITS); fill += 1 << l) fast[fill] = (uint16_t)((symbol[idx] << 4) | l);
                }
            }
            code <<= 1;
        }
        return left == 0 ? 0 : 1;
    }
    /* next symbol out of the low bits of w (at least max_len valid bits), *used = its length; -1 = invalid code */
    inline int decode_word(uint64_t w, uint32_t *used) const {
        const uint16_t f = fast[w & ((1u << PRIMARY_BITS) - 1u)];
        if (f & 15) { *used = f & 15; return f >> 4; }
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= max_len; l++) {
            code |= (int)((w >> (l - 1)) & 1);
            const int c = count[l];
            if (code - c < first) { *used = (uint32_t)l; return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
    /* next symbol, or -1 (invalid code / end of input) */
    inline int decode(BitReader &br) const {
        const uint64_t w = br.peek();
        const uint16_t f = fast[w & ((1u << PRIMARY_BITS) - 1u)];
        if (f & 15) { br.pos += f & 15; return f >> 4; }
        int code = 0, first = 0, index = 0;
        for (int l = 1; l <= max_len; l++) {
            code |= (int)((w >> (l - 1)) & 1);
            const int c = count[l];
            if (code - c < first) { br.pos += (uint32_t)l; return symbol[index + (code - first)]; }
            index += c; first += c; first <<= 1; code <<= 1;
        }
        return -1;
    }
};

static const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
static const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
static const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145, 8193, 12289, 16385, 24577};
static const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

struct BlockCodes { Huff lit, dist; bool dist_usable = true; };

/* header of a dynamic block (the three bits BFINAL / BTYPE already consumed).  strict: the lit/len code must be complete and the
 * distance code complete or a single code (what compressors write) -- used by the block finder; a decoder that follows a real
 * stream accepts what zlib accepts (incomplete codes only in the single-distance-code form). */
inline bool read_dynamic_header(BitReader &br, BlockCodes &bc) {
    if (br.eof(14)) return false;
    const uint32_t hlit = br.bits(5) + 257, hdist = br.bits(5) + 1, hclen = br.bits(4) + 4;
    if (hlit > 286 || hdist > 30) return false;
    static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
    uint8_t cl[19]; memset(cl, 0, sizeof(cl));
    if (br.eof(3 * hclen)) return false;
    for (uint32_t i = 0; i < hclen; i++) cl[order[i]] = (uint8_t)br.bits(3);
    {   /* Kraft sum of the code-length code first: nearly every wrong candidate of the block finder ends here, before any table is built */
        uint32_t kraft = 0;
        for (int i = 0; i < 19; i++) if (cl[i]) kraft += 128u >> cl[i];
        if (kraft != 128u) return false;
    }
    Huff clh;
    if (clh.build(cl, 19) != 0) return false;                      /* zlib requires a complete code-length code */
    uint8_t lens[286 + 30];
    uint32_t i = 0;
    while (i < hlit + hdist) {
        if (br.eof(1)) return false;
        const int s = clh.decode(br);
        if (s < 0) return false;
        if (s < 16) lens[i++] = (uint8_t)s;
        else {
            uint32_t rep; uint8_t v = 0;
            if (s == 16) { if (i == 0) return false; v = lens[i - 1]; rep = 3 + br.bits(2); }
            else if (s == 17) rep = 3 + br.bits(3);
            else rep = 11 + br.bits(7);
            if (i + rep > hlit + hdist) return false;
            while (rep--) lens[i++] = v;
        }
    }
    if (lens[256] == 0) return false;                              /* no end-of-block code */
    const int rl = bc.lit.build(lens, (int)hlit);
    if (rl < 0 || (rl > 0 && hlit - bc.lit.count[0] != 1)) return false;      /* incomplete only if a single code */
    if (rl > 0) return false;                                      /* (a lit/len code of one symbol = a block that only ends: not written by compressors; refused) */
    const int rd = bc.dist.build(lens + hlit, (int)hdist);
    if (rd < 0) return false;
    bc.dist_usable = true;
    if (rd > 0) {                                                  /* incomplete: allowed when there is at most one distance code */
        const int used = (int)hdist - bc.dist.count[0];
        if (used > 1) return false;
        if (used == 0) bc.dist_usable = false;
    }
    return true;
}

inline void fixed_codes(BlockCodes &bc) {
    uint8_t l[288];
    for (int i = 0; i < 144; i++) l[i] = 8;
    for (int i = 144; i < 256; i++) l[i] = 9;
    for (int i = 256; i < 280; i++) l[i] = 7;
    for (int i = 280; i < 288; i++) l[i] = 8;
    bc.lit.build(l, 288);
    uint8_t d[30]; for (int i = 0; i < 30; i++) d[i] = 5;
    bc.dist.build(d, 30);
    bc.dist_usable = true;
}

/* output of a chunk: 16-bit symbols, < 256 = that byte, >= 256 = the byte at position (symbol - 256) of the 32 KB that precede the
 * chunk's output.  known_window: the chunk starts where the history is known (start of a member, or the resolved end of the
 * previous work): then `window` holds it and no marker is ever produced. */
/* growable array of 16-bit symbols that is never value-initialised and keeps its storage when emptied (64 threads growing
 * zero-filled vectors page by page were the inflater's time on the GPU box) */
class SymBuf {
public:
    SymBuf() = default;
    ~SymBuf() { free(p_); }
    SymBuf(const SymBuf &) = delete; SymBuf &operator=(const SymBuf &) = delete;
    SymBuf(SymBuf &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    SymBuf &operator=(SymBuf &&o) noexcept { if (this != &o) { free(p_); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
    size_t size() const { return n_; }
    uint16_t *data() { return p_; } const uint16_t *data() const { return p_; }
    uint16_t &operator[](size_t i) { return p_[i]; } const uint16_t &operator[](size_t i) const { return p_[i]; }
    void resize(size_t n) {
        if (n > cap_) {
            const size_t c = std::max(n, cap_ + cap_ / 2);
            uint16_t *q = (uint16_t *)realloc(p_, c * sizeof(uint16_t));
            if (!q) throw std::bad_alloc();
            p_ = q; cap_ = c;
        }
        n_ = n;
    }
    void drop_front(size_t k) { if (k >= n_) { n_ = 0; return; } memmove(p_, p_ + k, (n_ - k) * sizeof(uint16_t)); n_ -= k; }
private:
    uint16_t *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};
struct Symbols {
    SymBuf s;
    bool exact = false;                  /* no markers inside */
};

inline bool texty(uint32_t c) { return c == 9 || c == 10 || c == 13 || (c >= 32 && c <= 126); }

/* Inflates blocks from br.pos on, appending to out.  Stops (returns true) at the end of the block after which `stop(bitpos)` says
 * so, or after the final block of the member (*final_seen), or -- finder only, max_symbols != 0 -- as soon as that many symbols
 * decoded cleanly; false on a malformed stream.  `hist` = bytes of history that exist before out.s[0] (0 at a member start; WIN = a
 * full, possibly unknown window): a distance that reaches beyond it is an error.  With !out.exact a reference into the history
 * becomes a marker.  text_only (block finder): fail on a literal that is not text. */
template <class StopFn>
inline bool inflate_blocks(BitReader &br, Symbols &out, uint32_t hist, bool text_only, size_t max_symbols, const StopFn &stop, bool *final_seen) {
    BlockCodes bc;
    const uint64_t end_bits = (uint64_t)br.n * 8;
    size_t pos = out.s.size();
    auto room = [&](size_t need) { if (out.s.size() < pos + need) out.s.resize(pos + need + 65536); };      /* (SymBuf grows its storage geometrically) */
    auto fail = [&]() { out.s.resize(pos); return false; };
    auto done = [&]() { out.s.resize(pos); return true; };
    for (;;) {
        if (br.eof(3)) return fail();
        const uint32_t bfinal = br.bits(1), btype = br.bits(2);
        if (btype == 3) return fail();
        if (btype == 0) {
            br.pos = (br.pos + 7) & ~7ull;
            if (br.eof(32)) return fail();
            const uint32_t len = br.bits(16), nlen = br.bits(16);
            if ((len ^ 0xFFFFu) != nlen) return fail();
            if (br.eof(8 * len)) return fail();
            const uint8_t *src = br.p + (br.pos >> 3);
            room(len);
            uint16_t *o = out.s.data();
            for (uint32_t i = 0; i < len; i++) { if (text_only && !texty(src[i])) return fail(); o[pos + i] = src[i]; }
            pos += len;
            br.pos += 8ull * len;
        } else {
            if (btype == 1) fixed_codes(bc);
            else if (!read_dynamic_header(br, bc)) return fail();
            /* fast loop: far from the end of the input and with room for a few hundred symbols, no per-symbol bounds checks; a short
             * literal leaves enough valid bits in the loaded word for the symbol behind it (57 - 9 >= 48), so common text costs one
             * load per two symbols.  Leaves to the careful loop below at the block's end, near the input's end, or in finder mode. */
            if (!text_only && !max_symbols) {
                const uint8_t *base = br.p;
                const uint64_t safe_end = br.n > 32 ? (uint64_t)(br.n - 32) * 8 : 0;
                uint64_t bp = br.pos;
                bool more = true;
                while (more && bp < safe_end) {
                    room(8192);
                    uint16_t *o = out.s.data();
                    const size_t stop_at = out.s.size() - 600;
                    while (pos < stop_at && bp < safe_end) {
                        uint64_t w; memcpy(&w, base + (bp >> 3), 8); w >>= (bp & 7);
                        uint32_t used;
                        int sym = bc.lit.decode_word(w, &used);
                        if (sym < 256) {
                            if (sym < 0) { br.pos = bp; return fail(); }
                            o[pos++] = (uint16_t)sym; bp += used;
                            if (used > 9) continue;
                            w >>= used;
                            sym = bc.lit.decode_word(w, &used);
                            if (sym < 256) { if (sym < 0) { br.pos = bp; return fail(); } o[pos++] = (uint16_t)sym; bp += used; continue; }
                        }
                        bp += used;
                        if (sym == 256) { more = false; break; }
                        if (sym > 285) { br.pos = bp; return fail(); }
                        w >>= used;
                        const uint32_t le = LEN_EXTRA[sym - 257];
                        const uint32_t len = LEN_BASE[sym - 257] + (uint32_t)(w & ((1u << le) - 1u));
                        w >>= le; bp += le;
                        if (!bc.dist_usable) { br.pos = bp; return fail(); }
                        const int ds = bc.dist.decode_word(w, &used);
                        if (ds < 0 || ds > 29) { br.pos = bp; return fail(); }
                        w >>= used;
                        const uint32_t de = DIST_EXTRA[ds];
                        const uint32_t dist = DIST_BASE[ds] + (uint32_t)(w & ((1u << de) - 1u));
                        bp += used + de;
                        if (dist > pos + hist) { br.pos = bp; return fail(); }
                        if (dist <= pos) {
                            const size_t from = pos - dist;
                            if (dist >= len) memcpy(o + pos, o + from, 2u * len);
                            else for (uint32_t i = 0; i < len; i++) o[pos + i] = o[from + i];
                        } else {
                            if (out.exact) { br.pos = bp; return fail(); }
                            for (uint32_t i = 0; i < len; i++) {
                                const int64_t src = (int64_t)pos + i - (int64_t)dist;
                                o[pos + i] = src >= 0 ? o[src] : (uint16_t)(256 + (int64_t)WIN + src);
                            }
                        }
                        pos += len;
                    }
                }
                br.pos = bp;
                if (!more) goto block_done;                          /* end-of-block symbol seen */
            }
            for (;;) {
                if (br.pos > end_bits) return fail();
                room(260);
                uint16_t *o = out.s.data();
                /* one 8-byte load serves a whole symbol: literal/length code (<= 15 bits) + length extra (<= 5) + distance code (<= 15)
                 * + distance extra (<= 13) = 48 of the >= 57 bits a load yields */
                uint64_t w = br.peek();
                uint32_t used;
                const int sym = bc.lit.decode_word(w, &used);
                if (sym < 0) return fail();
                if (sym < 256) {
                    if (text_only && !texty((uint32_t)sym)) return fail();
                    o[pos++] = (uint16_t)sym;
                    br.pos += used;
                    if (max_symbols && pos > max_symbols) return done();
                    continue;
                }
                br.pos += used;
                if (sym == 256) break;
                if (sym > 285) return fail();
                w >>= used;
                const uint32_t le = LEN_EXTRA[sym - 257];
                const uint32_t len = LEN_BASE[sym - 257] + (uint32_t)(w & ((1u << le) - 1u));
                w >>= le; br.pos += le;
                if (!bc.dist_usable) return fail();
                const int ds = bc.dist.decode_word(w, &used);
                if (ds < 0 || ds > 29) return fail();
                w >>= used;
                const uint32_t de = DIST_EXTRA[ds];
                const uint32_t dist = DIST_BASE[ds] + (uint32_t)(w & ((1u << de) - 1u));
                br.pos += used + de;
                if (dist > pos + hist) return fail();               /* before the start of the member / beyond the window */
                if (dist <= pos) {
                    const size_t from = pos - dist;
                    if (dist >= len) memcpy(o + pos, o + from, 2u * len);
                    else for (uint32_t i = 0; i < len; i++) o[pos + i] = o[from + i];      /* overlapping: forward, element by element */
                } else {
                    if (out.exact) return fail();                    /* (an exact chunk carries its window in front of its symbols) */
                    for (uint32_t i = 0; i < len; i++) {
                        const int64_t src = (int64_t)pos + i - (int64_t)dist;
                        o[pos + i] = src >= 0 ? o[src] : (uint16_t)(256 + (int64_t)WIN + src);
                    }
                }
                pos += len;
                if (max_symbols && pos > max_symbols) return done();
            }
        }
    block_done:
        if (br.pos > end_bits) return fail();
        if (bfinal) { *final_seen = true; return done(); }
        if (stop(br.pos)) return done();
    }
}

/* CRC-32 (the gzip polynomial, reflected 0xEDB88320) eight bytes per step with eight tables: zlib 1.2.11's crc32() ran at ~0.55 GB/s
 * and was a third of a thread's time; same values (so crc32_combine joins the pieces).  Little-endian hosts. */
struct Crc32Tables {
    uint32_t t[8][256];
    Crc32Tables() {
        for (uint32_t i = 0; i < 256; i++) { uint32_t c = i; for (int k = 0; k < 8; k++) c = (c & 1) ? 0xEDB88320u ^ (c >> 1) : c >> 1; t[0][i] = c; }
        for (uint32_t i = 0; i < 256; i++) for (int k = 1; k < 8; k++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xFF];
    }
};
inline uint32_t crc32_fast(const uint8_t *p, size_t n) {
    static const Crc32Tables T;
    uint32_t c = 0xFFFFFFFFu;
    while (n && ((uintptr_t)p & 7)) { c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8); n--; }
    while (n >= 8) {
        uint64_t v; memcpy(&v, p, 8);
        v ^= c;
        c = T.t[7][v & 0xFF] ^ T.t[6][(v >> 8) & 0xFF] ^ T.t[5][(v >> 16) & 0xFF] ^ T.t[4][(v >> 24) & 0xFF] ^
            T.t[3][(v >> 32) & 0xFF] ^ T.t[2][(v >> 40) & 0xFF] ^ T.t[1][(v >> 48) & 0xFF] ^ T.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = T.t[0][(c ^ *p++) & 0xFF] ^ (c >> 8);
    return c ^ 0xFFFFFFFFu;
}

/* gzip member header at byte offset `at` (RFC 1952); returns the offset of the deflate data, 0 if there is no valid header */
inline size_t member_header(const uint8_t *d, size_t n, size_t at) {
    if (at + 18 > n || d[at] != 0x1f || d[at + 1] != 0x8b || d[at + 2] != 8) return 0;
    const uint8_t flg = d[at + 3];
    size_t p = at + 10;
    if (flg & 4) { if (p + 2 > n) return 0; p += 2 + (d[p] | ((size_t)d[p + 1] << 8)); }
    if (flg & 8) { while (p < n && d[p]) p++; p++; }
    if (flg & 16) { while (p < n && d[p]) p++; p++; }
    if (flg & 2) p += 2;
    return p < n ? p : 0;
}

} // namespace pgz

/* run(n, f): executes f(0..n-1) on the caller's worker pool */
typedef std::function<void(size_t, const std::function<void(size_t)> &)> PgzRunner;

class ParallelGzip {
public:
    /* data/len: the whole gzip file (mapped).  chunk_bytes: compressed bytes per chunk. */
    ParallelGzip(const uint8_t *data, size_t len, int threads, PgzRunner run, size_t chunk_bytes = 2u << 20)
        : d_(data), n_(len), threads_(threads < 1 ? 1 : threads), run_(std::move(run)), chunk_(chunk_bytes < 65536 ? 65536 : chunk_bytes) {
        const size_t ds = pgz::member_header(d_, n_, 0);
        if (!ds) throw std::runtime_error("not a gzip file");
        bit_ = (uint64_t)ds * 8; hist_ = 0; member_crc_ = crc32(0L, Z_NULL, 0); member_len_ = 0;
    }
    bool done() const { return done_; }

    /* appends at least `want` bytes of inflated text to `out` (fewer only at the end of the file) */
    template <class Vec> void produce(Vec &out, size_t want) {
        size_t made = 0;
        while (made < want && !done_) made += wave(out, want - made);
    }

private:
    struct Chunk {
        uint64_t start = 0;            /* bit position of a block start (0 = none found) */
        uint64_t end = 0;              /* bit position where the inflation stopped */
        pgz::Symbols sym;
        bool ok = false, final_member_end = false;
        std::vector<std::pair<size_t, uint64_t>> member_ends;      /* (symbols before the end of a member, byte offset of its trailer) */
        std::string err;
        uint32_t next = 0;             /* index of the chunk whose start this one reached (n = none: ran to the end) */
        size_t out_off = 0;
        void reset() { start = end = 0; sym.s.resize(0); sym.exact = false; ok = false; final_member_end = false; member_ends.clear(); err.clear(); next = 0; out_off = 0; }
    };
    std::vector<Chunk> chunks_;

    /* block finder: first bit position >= from (and < to) where a non-final dynamic block with complete codes starts whose first
     * symbols are text */
    uint64_t find_start(uint64_t from, uint64_t to) const {
        pgz::BlockCodes bc;
        pgz::Symbols s;
        for (uint64_t b = from; b < to; b++) {
            pgz::BitReader br(d_, n_, b);
            const uint64_t w = br.peek();
            if ((w & 7u) != 4u) continue;                           /* BFINAL = 0, BTYPE = 2 (bits: 0, 0, 1) */
            /* cheap rejections before the table building: HLIT <= 29, HDIST <= 29 */
            if (((w >> 3) & 31u) > 29u || ((w >> 8) & 31u) > 29u) continue;
            br.pos = b + 3;
            if (!pgz::read_dynamic_header(br, bc)) continue;
            /* trial: the block's first symbols */
            pgz::BitReader t(d_, n_, b);
            bool fin = false;
            s.s.resize(0);
            if (!pgz::inflate_blocks(t, s, pgz::WIN, true, 8192, [](uint64_t) { return true; }, &fin)) continue;
            if (s.s.size() < 64 || fin) continue;
            return b;
        }
        return 0;
    }

    /* one wave: up to 2 x threads chunks from the current position; returns the bytes appended */
    template <class Vec> size_t wave(Vec &out, size_t want) {
        const uint64_t total_bits = (uint64_t)n_ * 8;
        /* two chunks per thread, but no more than the request is likely to need (text inflates ~3 - 4 x): the symbols of a wave take
         * four bytes of memory per byte of compressed input and two per byte of text */
        const size_t G = std::max<size_t>(2, std::min<size_t>((size_t)threads_ * 2, want / (3 * chunk_) + 2));
        /* chunk k >= 1 starts searching at byte (cur_byte + k * chunk_); chunk 0 starts exactly at bit_ */
        const size_t byte0 = (size_t)(bit_ >> 3);
        size_t nch = 1;
        while (nch < G + 1 && byte0 + nch * chunk_ + 64 < n_) nch++;
        std::vector<Chunk> &ch = chunks_;                  /* (kept between waves: their symbol buffers are grown once) */
        if (ch.size() < nch) ch.resize(nch);
        while (ch.size() > nch) ch.pop_back();
        for (auto &c : ch) c.reset();
        ch[0].start = bit_;
        /* 1. starts of the chunks 1 .. nch-1 (the last one is only a stop mark for this wave) */
        run_(nch - 1, [&](size_t k) {
            const uint64_t from = (uint64_t)(byte0 + (k + 1) * chunk_) * 8;
            const uint64_t to = std::min<uint64_t>(total_bits, (uint64_t)(byte0 + (k + 2) * chunk_) * 8);
            ch[k + 1].start = find_start(from, to);
        });
        /* 2. inflate every chunk but the stop mark */
        const size_t nwork = nch > 1 ? nch - 1 : 1;
        run_(nwork, [&](size_t k) {
            Chunk &c = ch[k];
            if (k > 0 && c.start == 0) return;                       /* no start found in its range: the chunk before it goes on through */
            inflate_chunk(ch, k, k == 0);
        });
        /* 3. chain: chunk 0, then the chunk each one reached */
        std::vector<size_t> chain;
        size_t cur = 0;
        for (;;) {
            Chunk &c = ch[cur];
            if (!c.ok) throw std::runtime_error(c.err.empty() ? "corrupt gzip stream" : c.err);
            chain.push_back(cur);
            if (c.final_member_end || c.next >= nwork) break;        /* end of the file, or the stop mark / beyond this wave */
            cur = c.next;
        }
        /* 4. windows in file order: the last 32 KB of every chained chunk, resolved */
        std::vector<std::vector<uint8_t>> win(chain.size() + 1);
        win[0] = window_;
        for (size_t i = 0; i < chain.size(); i++) {
            const Chunk &c = ch[chain[i]];
            const std::vector<uint8_t> &w = win[i];
            std::vector<uint8_t> nw(pgz::WIN, 0);
            const size_t m = c.sym.s.size();
            for (size_t j = 0; j < pgz::WIN; j++) {
                /* byte at distance (WIN - j) before the end of this chunk's output */
                const int64_t idx = (int64_t)m - (int64_t)pgz::WIN + (int64_t)j;
                uint8_t v = 0;
                if (idx >= 0) { const uint16_t s = c.sym.s[(size_t)idx]; v = s < 256 ? (uint8_t)s : (w.size() == pgz::WIN ? w[s - 256] : 0); }
                else if (w.size() == pgz::WIN) v = w[(size_t)((int64_t)pgz::WIN + idx)];
                nw[j] = v;
            }
            win[i + 1] = std::move(nw);
        }
        /* 5. symbols -> bytes, in parallel, at their places; per-chunk CRC pieces cut at member ends */
        size_t total = 0;
        for (size_t i = 0; i < chain.size(); i++) { ch[chain[i]].out_off = total; total += ch[chain[i]].sym.s.size(); }
        const size_t at = out.size();
        out.resize_uninit(at + total);
        char *dst = out.data() + at;
        struct Piece { uint32_t crc; size_t len; };
        std::vector<std::vector<Piece>> pieces(chain.size());
        run_(chain.size(), [&](size_t i) {
            const Chunk &c = ch[chain[i]];
            const std::vector<uint8_t> &w = win[i];
            char *o = dst + c.out_off;
            const size_t m = c.sym.s.size();
            const uint16_t *s = c.sym.s.data();
            {   /* eight symbols at a time where none of them is a marker (markers thin out quickly behind a chunk's first 32 KB) */
                const bool have_w = w.size() == pgz::WIN;
                size_t j = 0;
                for (; j + 8 <= m; j += 8) {
                    uint64_t a, b; memcpy(&a, s + j, 8); memcpy(&b, s + j + 4, 8);
                    if (((a | b) & 0xFF00FF00FF00FF00ull) == 0) {
                        const uint64_t lo = (a & 0xFF) | ((a >> 8) & 0xFF00) | ((a >> 16) & 0xFF0000) | ((a >> 24) & 0xFF000000ull);
                        const uint64_t hi = (b & 0xFF) | ((b >> 8) & 0xFF00) | ((b >> 16) & 0xFF0000) | ((b >> 24) & 0xFF000000ull);
                        const uint64_t v = lo | (hi << 32);
                        memcpy(o + j, &v, 8);
                    } else {
                        for (size_t k = j; k < j + 8; k++) {
                            if (s[k] < 256) o[k] = (char)s[k];
                            else { if (!have_w) throw std::runtime_error("gzip: reference before the start of the stream"); o[k] = (char)w[s[k] - 256]; }
                        }
                    }
                }
                for (; j < m; j++) {
                    if (s[j] < 256) o[j] = (char)s[j];
                    else { if (!have_w) throw std::runtime_error("gzip: reference before the start of the stream"); o[j] = (char)w[s[j] - 256]; }
                }
            }
            size_t from = 0;
            for (auto &me : c.member_ends) {
                pieces[i].push_back(Piece{pgz::crc32_fast((const uint8_t *)o + from, me.first - from), me.first - from});
                from = me.first;
            }
            pieces[i].push_back(Piece{pgz::crc32_fast((const uint8_t *)o + from, m - from), m - from});
        });
        /* 6. member checks (CRC-32 and ISIZE of RFC 1952) in file order */
        for (size_t i = 0; i < chain.size(); i++) {
            const Chunk &c = ch[chain[i]];
            for (size_t k = 0; k < pieces[i].size(); k++) {
                member_crc_ = (uint32_t)crc32_combine(member_crc_, pieces[i][k].crc, (z_off_t)pieces[i][k].len);
                member_len_ += pieces[i][k].len;
                if (k < c.member_ends.size()) {
                    const size_t t = (size_t)c.member_ends[k].second;
                    if (t + 8 > n_) throw std::runtime_error("truncated gzip file (trailer)");
                    const uint32_t want_crc = d_[t] | ((uint32_t)d_[t + 1] << 8) | ((uint32_t)d_[t + 2] << 16) | ((uint32_t)d_[t + 3] << 24);
                    const uint32_t want_len = d_[t + 4] | ((uint32_t)d_[t + 5] << 8) | ((uint32_t)d_[t + 6] << 16) | ((uint32_t)d_[t + 7] << 24);
                    if (want_crc != member_crc_ || want_len != (uint32_t)member_len_) throw std::runtime_error("corrupt gzip file (CRC or length of a member)");
                    member_crc_ = (uint32_t)crc32(0L, Z_NULL, 0); member_len_ = 0;
                }
            }
        }
        /* 7. state for the next wave */
        const Chunk &last = ch[chain.back()];
        window_ = std::move(win[chain.size()]);
        hist_ = last.final_member_end ? 0 : pgz::WIN;     /* (a window shorter than 32 KB at the very start is padded with zeros: never referenced) */
        if (last.final_member_end && last.end == 0) done_ = true;
        else { bit_ = last.end; if (last.final_member_end) window_.clear(); }
        if (bit_ >= total_bits) done_ = true;
        return total;
    }

    /* inflate chunk k of the wave from its start until it reaches the start of a later chunk (exactly), the end of the wave's range
     * or the end of the file; member ends inside are passed (trailer, next header) */
    void inflate_chunk(std::vector<Chunk> &ch, size_t k, bool first) {
        Chunk &c = ch[k];
        try {
            pgz::BitReader br(d_, n_, c.start);
            const size_t nch = ch.size();
            uint32_t next = (uint32_t)k + 1;
            /* past the wave's range any block boundary will do: what follows is the next wave's first chunk, which starts exactly there */
            const uint64_t wave_end = (uint64_t)std::min<size_t>(n_, (size_t)(ch[0].start >> 3) + nch * chunk_) * 8;
            c.sym.exact = first;                      /* chunk 0 of a wave knows its history (hist_ bytes of it, in window_) */
            uint32_t hist = first ? hist_ : pgz::WIN;
            /* an exact chunk that has a window resolves references into it at once: the window's bytes sit in front of the symbols */
            size_t lead = 0;
            if (first && hist_ == pgz::WIN && window_.size() == pgz::WIN) { c.sym.s.resize(pgz::WIN); for (size_t j = 0; j < pgz::WIN; j++) c.sym.s[j] = window_[j]; lead = pgz::WIN; hist = 0; }
            for (;;) {
                bool fin = false;
                const bool good = pgz::inflate_blocks(br, c.sym, hist, false, 0, [&](uint64_t pos) {
                    while (next < nch && (ch[next].start == 0 || ch[next].start < pos)) next++;       /* starts that the real block sequence steps over were no starts */
                    return (next < nch && ch[next].start == pos) || pos >= wave_end;
                }, &fin);
                if (!good) { c.err = "corrupt gzip stream (deflate data)"; return; }
                if (!fin) { c.next = (next < nch && ch[next].start == br.pos) ? next : (uint32_t)nch; c.end = br.pos; break; }
                /* end of a member: trailer, then another member or the end of the file */
                const size_t trailer = (size_t)((br.pos + 7) >> 3);
                if (trailer + 8 > n_) { c.err = "truncated gzip file"; return; }
                c.member_ends.push_back({c.sym.s.size() - lead, (uint64_t)trailer});
                size_t nh = trailer + 8;
                while (nh < n_ && d_[nh] == 0) nh++;                                                /* zero padding between / after members is tolerated (as gzip -d does) */
                if (nh >= n_) { c.final_member_end = true; c.end = 0; c.next = (uint32_t)nch; break; }
                const size_t ds = pgz::member_header(d_, n_, nh);
                if (!ds) { c.final_member_end = true; c.end = 0; c.next = (uint32_t)nch; break; }      /* non-gzip bytes behind a complete member end the input, as zlib's gzread (the small-file path and the reference's kseq reader) treats them */
                br.pos = (uint64_t)ds * 8;
                /* a new member starts without history */
                if (lead) { c.sym.s.drop_front(lead); lead = 0; }
                member_start_fix(c, hist);
                while (next < nch && (ch[next].start == 0 || ch[next].start < br.pos)) next++;
                if (next < nch && ch[next].start == br.pos) { c.next = next; c.end = br.pos; break; }
            }
            if (lead) c.sym.s.drop_front(lead);
            c.ok = true;
        } catch (const std::exception &e) { c.err = e.what(); }
    }
    /* after a member boundary inside a chunk the history is empty again: references may only reach back to the boundary.  The
     * inflater checks distances against (symbols so far + hist); symbols of the previous member in front of the boundary would let
     * an invalid stream slip through, which the CRC check then catches -- nothing to adjust here. */
    static void member_start_fix(Chunk &, uint32_t &) {}

    const uint8_t *d_; size_t n_; int threads_; PgzRunner run_; size_t chunk_;
    uint64_t bit_ = 0; uint32_t hist_ = 0; std::vector<uint8_t> window_;
    uint32_t member_crc_ = 0; uint64_t member_len_ = 0;
    bool done_ = false;
};

} // namespace mtbhost
#endif
