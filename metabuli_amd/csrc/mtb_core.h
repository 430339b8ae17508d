/*
 * mtb_core.h -- per-lane arithmetic of the hot path, written once and used by
 * the HIP kernels (device) and by tests/emu (host build of the very same
 * functions, to check the kernel logic against the oracle without a GPU).
 *
 * This is NOT a restatement of the reference's control flow: it is the
 * functional specification of each stage, laid out for one
 * lane of a 64-wide wavefront:
 *   - a metamer window is a pure function of 8 consecutive codons;
 *   - a query's matches are a pure function of its AA-run in the flat index;
 *   - a (species, frame) block of sorted matches yields at most one path per
 *     match, stored at the match's own slot.
 * Reference citations (src/commons/...) are given per function.
 */
#ifndef MTB_CORE_H
#define MTB_CORE_H
#include <stdint.h>
#include "../../include/mtb.h"

#if defined(__HIPCC__)
#define MTB_HD __host__ __device__ __forceinline__
#else
#define MTB_HD inline
#endif
#if defined(__HIPCC__)
#define MTB_UNROLL _Pragma("unroll")
#else
#define MTB_UNROLL
#endif

/* ------------------------------------------------------------------ */
/* qinfo helpers (Kmer.h:11-16)                                        */
/* ------------------------------------------------------------------ */
MTB_HD uint64_t mtb_qinfo(uint32_t seq_id, uint32_t pos, uint32_t frame) {
    return (uint64_t)pos | ((uint64_t)(seq_id & 0x1FFFFFFFu) << 32) | ((uint64_t)(frame & 7u) << 61);
}
MTB_HD uint32_t mtb_q_pos(uint64_t q)   { return (uint32_t)q; }
MTB_HD uint32_t mtb_q_seq(uint64_t q)   { return (uint32_t)((q >> 32) & 0x1FFFFFFFu); }
MTB_HD uint32_t mtb_q_frame(uint64_t q) { return (uint32_t)(q >> 61); }

/* LocalUtil.h:51-59 */
MTB_HD int32_t mtb_used_len(int32_t len) {
    int32_t r = len % 3;
    return r == 2 ? len - 2 : (r == 1 ? len - 4 : len - 3);
}
/* LocalUtil.h:46-48: reads with fewer than one dense slot are skipped */
MTB_HD bool mtb_read_too_short(int32_t len) { return (mtb_used_len(len) / 3 - 8 + 1) * 6 < 1; }

/* ------------------------------------------------------------------ */
/* Tables                                                              */
/* ------------------------------------------------------------------ */
/* Built on the host by mtb_build_tables(); copied to the device once. */
typedef struct {
    uint8_t  base[256];   /* ASCII -> 0..3 (A C T G classes incl. IUPAC), 7 = invalid;
                             common.cpp:13-23 + GeneticCode.h:6.  Complement = code ^ 2. */
    uint8_t  codon[64];   /* index c0*16+c1*4+c2 -> aa(5 bits) | codon id << 5;
                             GeneticCode.h:34-193                                      */
    uint32_t hamrow[8];   /* hammingLookup[a][b] as nibble b of word a;
                             KmerMatcher.h:66-70                                       */
} mtb_tables;

static inline void mtb_build_tables(mtb_tables *t) {
    for (int i = 0; i < 256; i++) t->base[i] = 7;
    const char *cls[4] = {"ARW", "CMS", "HTY", "BDGKU"};
    for (int c = 0; c < 4; c++)
        for (const char *p = cls[c]; *p; ++p) {
            t->base[(unsigned char)*p] = (uint8_t)c;
            t->base[(unsigned char)(*p | 0x20)] = (uint8_t)c;
        }
    /* amino acid per codon, our base order A C T G; index into
       "ARNDCQEGHILKMFPSTWYV", '*' = 20 */
    static const char *AA_ACTG =
        "KNNK" "TTTT" "IIIM" "RSSR"   /* A?? : AA. AC. AT. AG. */
        "QHHQ" "PPPP" "LLLL" "RRRR"   /* C?? */
        "*YY*" "SSSS" "LFFL" "*CCW"   /* T?? */
        "EDDE" "AAAA" "VVVV" "GGGG";  /* G?? */
    static const char *LET = "ARNDCQEGHILKMFPSTWYV";
    for (int i = 0; i < 64; i++) {
        char r = AA_ACTG[i];
        int aa = 20;
        if (r != '*') { aa = 0; while (LET[aa] != r) aa++; }
        int cid = i & 3;                      /* third base */
        t->codon[i] = (uint8_t)(aa | (cid << 5));
    }
    /* six-codon families and TGA: AGG4 AGA5 TTG4 TTA5 AGT6 AGC7 TGA5 */
    #define MTB_SETCID(c0, c1, c2, id) t->codon[(c0) * 16 + (c1) * 4 + (c2)] = (uint8_t)((t->codon[(c0) * 16 + (c1) * 4 + (c2)] & 31) | ((id) << 5))
    MTB_SETCID(0, 3, 3, 4); MTB_SETCID(0, 3, 0, 5);
    MTB_SETCID(2, 2, 3, 4); MTB_SETCID(2, 2, 0, 5);
    MTB_SETCID(0, 3, 2, 6); MTB_SETCID(0, 3, 1, 7);
    MTB_SETCID(2, 3, 0, 5);
    #undef MTB_SETCID
    /* codon-id Hamming distances: ids 0-3 are the third base; 4/5 are the
       second family of Arg/Leu (and TGA), 6/7 the second family of Ser. */
    static const uint8_t H[8][8] = {
        {0, 1, 1, 1, 2, 1, 3, 3}, {1, 0, 1, 1, 2, 2, 3, 2},
        {1, 1, 0, 1, 2, 2, 2, 3}, {1, 1, 1, 0, 1, 2, 3, 3},
        {2, 2, 2, 1, 0, 1, 4, 4}, {1, 2, 2, 2, 1, 0, 4, 4},
        {3, 3, 2, 3, 4, 4, 0, 1}, {3, 2, 3, 3, 4, 4, 1, 0}};
    for (int a = 0; a < 8; a++) {
        uint32_t w = 0;
        for (int b = 0; b < 8; b++) w |= (uint32_t)H[a][b] << (4 * b);
        t->hamrow[a] = w;
    }
}

/* ------------------------------------------------------------------ */
/* Extraction: one window = 8 codon bytes -> metamer + syncmer test    */
/* ------------------------------------------------------------------ */
/* Codon byte of the codon whose first base (in reading direction) sits at
 * forward coordinate ci.  Forward frames read ci, ci+1, ci+2; reverse frames
 * read the complement of ci, ci-1, ci-2 (KmerScanner.h:89-98).  0xFF if any
 * base is not A/C/G/T-like.                                                 */
/* kmer_format 1 (OldMetamerScanner, KmerScanner.h:137-160): forward frames walk the window from
 * its END (codon = bases ci-2, ci-1, ci), reverse frames from its START (complement of ci+2, ci+1, ci) */
MTB_HD uint8_t mtb_codon_byte_old(const mtb_tables *t, const char *seq, int64_t ci, bool fwd) {
    uint32_t a, b, c;
    if (fwd) { a = t->base[(uint8_t)seq[ci - 2]]; b = t->base[(uint8_t)seq[ci - 1]]; c = t->base[(uint8_t)seq[ci]]; }
    else {
        a = t->base[(uint8_t)seq[ci + 2]]; b = t->base[(uint8_t)seq[ci + 1]]; c = t->base[(uint8_t)seq[ci]];
        if ((a | b | c) < 4) { a ^= 2; b ^= 2; c ^= 2; }
    }
    if ((a | b | c) > 3) return 0xFF;
    return t->codon[a * 16 + b * 4 + c];
}
MTB_HD uint8_t mtb_codon_byte(const mtb_tables *t, const char *seq, int64_t ci, bool fwd) {
    uint32_t a, b, c;
    if (fwd) {
        a = t->base[(uint8_t)seq[ci]]; b = t->base[(uint8_t)seq[ci + 1]]; c = t->base[(uint8_t)seq[ci + 2]];
    } else {
        a = t->base[(uint8_t)seq[ci]]; b = t->base[(uint8_t)seq[ci - 1]]; c = t->base[(uint8_t)seq[ci - 2]];
        if ((a | b | c) < 4) { a ^= 2; b ^= 2; c ^= 2; }
    }
    if ((a | b | c) > 3) return 0xFF;
    return t->codon[a * 16 + b * 4 + c];
}

/* same, from bases already canonicalised to codes 0..3 / 7 (staged in LDS) */
MTB_HD uint8_t mtb_codon_byte_codes(const mtb_tables *t, const uint8_t *code, int64_t ci, bool fwd) {
    uint32_t a, b, c;
    if (fwd) { a = code[ci]; b = code[ci + 1]; c = code[ci + 2]; }
    else { a = code[ci]; b = code[ci - 1]; c = code[ci - 2]; if ((a | b | c) < 4) { a ^= 2; b ^= 2; c ^= 2; } }
    if ((a | b | c) > 3) return 0xFF;
    return t->codon[a * 16 + b * 4 + c];
}

/* Frame geometry (KmerExtractor.cpp:350-362): begin of frame f in a read of
 * length len; the scanner window is [begin, begin + used - 1].              */
MTB_HD int32_t mtb_frame_begin(int32_t len, int32_t frame) {
    if (frame < 3) return frame;
    int32_t b = (len % 3) - (frame % 3);
    return b < 0 ? b + 3 : b;
}
/* forward coordinate of the first base of codon j of frame f */
MTB_HD int64_t mtb_codon_ci(int32_t begin, int32_t used, int32_t j, bool fwd) {
    return fwd ? (int64_t)begin + 3 * (int64_t)j : (int64_t)begin + used - 1 - 3 * (int64_t)j;
}
/* kmer_format 1: scan coordinate of codon j and reported position of window p (KmerScanner.h:145-176) */
MTB_HD int64_t mtb_codon_ci_old(int32_t begin, int32_t used, int32_t j, bool fwd) {
    return fwd ? (int64_t)begin + used - 1 - 3 * (int64_t)j : (int64_t)begin + 3 * (int64_t)j;
}
MTB_HD uint32_t mtb_window_pos_old(int32_t begin, int32_t used, int32_t p, bool fwd) {
    return fwd ? (uint32_t)(begin + used - 1 - (p + 8) * 3 + 1) : (uint32_t)(begin + 3 * p);
}
/* position reported for window p (KmerScanner.h:110-114, SyncmerScanner.h:93-97) */
MTB_HD uint32_t mtb_window_pos(int32_t begin, int32_t used, int32_t p, bool fwd) {
    return fwd ? (uint32_t)(begin + 3 * p) : (uint32_t)(begin + used - 1 - (p + 8) * 3 + 1);
}

/* cod[0..7] are the codon bytes of the window in reading direction.  Returns
 * false if the window holds an invalid codon or (syncmer mode) is not a
 * closed syncmer: the leftmost minimal s-mer must sit at offset 0 or 8-s
 * (SyncmerScanner.h:58-73).  value = AA40 << 24 | DNA24 (KmerScanner.h:99-111). */
/* kmer_format 1: amino-acid part is the base-21 number of the 8 residues (first scanned = most
 * significant), no syncmer selection */
MTB_HD bool mtb_window_metamer_old(const uint8_t *cod, uint64_t *value) {
    uint64_t aa = 0, dna = 0;
    uint32_t bad = 0;
MTB_UNROLL
    for (int i = 0; i < 8; i++) {
        uint32_t b = cod[i];
        bad |= (b == 0xFFu);
        aa = aa * 21u + (b & 31u);
        dna = (dna << 3) | (b >> 5);
    }
    if (bad) return false;
    *value = (aa << 24) | (dna & 0xFFFFFFull);
    return true;
}
/* Word form: w0 = codon bytes 0..3, w1 = codon bytes 4..7 (byte 0 of a word = the earlier codon).  The four 5-bit
 * amino-acid fields and 3-bit codon ids of a word are gathered with mask / shift pairs instead of a byte loop
 * (the extractor is VALU-bound: ~45 instead of ~80 instructions per window). */
MTB_HD uint32_t mtb_pack_aa4(uint32_t w) {           /* f0<<15 | f1<<10 | f2<<5 | f3, f_i = byte_i & 31 */
    uint32_t x = ((w & 0x001F001Fu) << 5) | ((w >> 8) & 0x001F001Fu);
    return ((x & 0x3FFu) << 10) | (x >> 16);
}
MTB_HD uint32_t mtb_pack_cid4(uint32_t w) {          /* c0<<9 | c1<<6 | c2<<3 | c3, c_i = byte_i >> 5 */
    uint32_t y = (w >> 5) & 0x07070707u;
    uint32_t p = ((y & 0x00070007u) << 3) | ((y >> 8) & 0x00070007u);
    return ((p & 0x3Fu) << 6) | (p >> 16);
}
MTB_HD bool mtb_window_metamer_words(uint32_t w0, uint32_t w1, int syncmer, int smer_len, uint64_t *value) {
    /* an invalid codon is the byte 0xFF; no valid byte has the amino-acid field 31 */
    uint32_t bad = (((w0 & 0x1F1F1F1Fu) + 0x01010101u) | ((w1 & 0x1F1F1F1Fu) + 0x01010101u)) & 0x20202020u;
    if (bad) return false;
    const uint32_t a0 = mtb_pack_aa4(w0), a1 = mtb_pack_aa4(w1);           /* 20 bits each */
    const uint32_t dna = (mtb_pack_cid4(w0) << 12) | mtb_pack_cid4(w1);
    const uint64_t aa = ((uint64_t)a0 << 20) | a1;
    *value = (aa << 24) | dna;
    if (!syncmer) return true;
    if (smer_len == 5) {            /* the default: the four 25-bit s-mers from the two 20-bit halves, 32-bit arithmetic only */
        const uint32_t s0 = (a0 << 5) | (a1 >> 15), s1 = ((a0 & 0x7FFFu) << 10) | (a1 >> 10);
        const uint32_t s2 = ((a0 & 0x3FFu) << 15) | (a1 >> 5), s3 = ((a0 & 0x1Fu) << 20) | a1;
        const uint32_t mid = s1 < s2 ? s1 : s2;
        return (s0 <= mid && s0 <= s3) || (s3 < s0 && s3 < mid);      /* leftmost minimum at the first or the last position */
    }
    int ns = 8 - smer_len + 1;
    uint64_t mask = (1ull << (5 * smer_len)) - 1;
    uint64_t best = ~0ull; int arg = 0;
    for (int k = 0; k < ns; k++) {
        uint64_t s = (aa >> (5 * (8 - smer_len - k))) & mask;   /* AAs k .. k+s-1 */
        if (s < best) { best = s; arg = k; }                      /* leftmost on ties */
    }
    return arg == 0 || arg == ns - 1;
}
MTB_HD bool mtb_window_metamer(const uint8_t *cod, int syncmer, int smer_len, uint64_t *value) {
    uint32_t w0 = (uint32_t)cod[0] | ((uint32_t)cod[1] << 8) | ((uint32_t)cod[2] << 16) | ((uint32_t)cod[3] << 24);
    uint32_t w1 = (uint32_t)cod[4] | ((uint32_t)cod[5] << 8) | ((uint32_t)cod[6] << 16) | ((uint32_t)cod[7] << 24);
    return mtb_window_metamer_words(w0, w1, syncmer, smer_len, value);
}

/* ------------------------------------------------------------------ */
/* Join: Hamming arithmetic on 24-bit codon-id strings                  */
/* ------------------------------------------------------------------ */
typedef struct { uint32_t row[8]; uint32_t qdna; } mtb_qrows;   /* row[i]: hamming row of query codon i (i=0 is the LSB codon) */

MTB_HD void mtb_prepare_query_rows(const uint32_t *hamrow, uint64_t qvalue, mtb_qrows *q) {
    uint32_t d = (uint32_t)qvalue & 0xFFFFFFu;
    q->qdna = d;
MTB_UNROLL
    for (int i = 0; i < 8; i++) q->row[i] = hamrow[(d >> (3 * i)) & 7u];
}
MTB_HD void mtb_prepare_query(const mtb_tables *t, uint64_t qvalue, mtb_qrows *q) { mtb_prepare_query_rows(t->hamrow, qvalue, q); }
/* getHammingDistanceSum (KmerMatcher.h:348-360) */
MTB_HD uint32_t mtb_ham_sum(const mtb_qrows *q, uint32_t tdna) {
    uint32_t s = 0;
MTB_UNROLL
    for (int i = 0; i < 8; i++) s += (q->row[i] >> (4 * ((tdna >> (3 * i)) & 7u))) & 15u;
    return s;
}
/* getHammings / getHammings_reverse (KmerMatcher.h:386-416) with the tables
 * HAMMING_LUT0..7 (KmerMatcher.h:72-158): 2-bit field = h<4 ? h : 0, except
 * that the field at bits 14-15 (LUT7) is 1 for query id 4|5 vs target id 6|7. */
MTB_HD uint16_t mtb_hammings(const mtb_qrows *q, uint32_t tdna, bool reverse) {
    uint32_t out = 0;
MTB_UNROLL
    for (int i = 0; i < 8; i++) {
        uint32_t b = (tdna >> (3 * i)) & 7u;
        uint32_t h = (q->row[i] >> (4 * b)) & 15u;
        uint32_t f = reverse ? 7 - i : i;
        uint32_t code = h < 4 ? h : 0;
        if (f == 7 && h == 4) {
            uint32_t a = (q->qdna >> (3 * i)) & 7u;
            if ((a == 4 || a == 5) && (b == 6 || b == 7)) code = 1;
        }
        out |= code << (2 * f);
    }
    return (uint16_t)out;
}
/* orientation rule of compareDna (KmerMatcher.cpp:1140-1142) */
MTB_HD bool mtb_hammings_reversed(uint32_t frame, int kmer_format) {
    return ((frame < 3) ^ (kmer_format == 2)) != 0;
}
/* selection threshold (KmerMatcher.cpp:1136) */
MTB_HD uint32_t mtb_ham_threshold(uint32_t min_ham) { uint32_t t = 2 * min_ham; return t < 7 ? t : 7; }

/* lower bound of key in sorted v[0..n) */
MTB_HD uint64_t mtb_lower_bound(const uint64_t *v, uint64_t n, uint64_t key) {
    uint64_t lo = 0, hi = n;
    while (lo < hi) {
        uint64_t mid = lo + ((hi - lo) >> 1);
        if (v[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

/* the flat target index resident in HBM */
typedef struct {
    const uint64_t *values; const uint32_t *info; uint64_t n_targets;
    const int32_t *tax2species; int32_t max_taxid; uint32_t info_mask; int32_t kmer_format;
} mtb_index_view;

/* Candidate run of one query inside a sorted window v[0..n) of target values
 * (the whole index or an LDS-staged slice of it): first index and length of
 * the entries sharing the query's amino-acid part.                          */
MTB_HD void mtb_join_find(const uint64_t *v, uint64_t n, uint64_t qvalue, uint64_t *run_start, uint32_t *run_len) {
    uint64_t aa = qvalue & ~0xFFFFFFull;
    uint64_t s = mtb_lower_bound(v, n, aa);
    uint64_t e = s;
    while (e < n && (v[e] & ~0xFFFFFFull) == aa) e++;
    *run_start = s; *run_len = (uint32_t)(e - s);
}
/* 16-byte form of a match inside a read's slot segment (fused short-read path): the sequenceID is implied by the
 * segment, positions are < 2^12, the hamming sum of a selected candidate is <= 7, and the 5-bit epoch tag marks the
 * slots written by the current batch.  One aligned 16-byte store per match instead of three 8-byte ones.
 *   a = species_id << 32 | target_id
 *   b = epoch[59..63] ham[55..58] frame[52..54] pos[40..51] right_end_hamming[24..39] dna[0..23]                  */
typedef struct { uint64_t a, b; } mtb_slot16;
#define MTB_SLOT_MAX_POS 4096u
#define MTB_SLOT_MAX_Q 384u         /* metamers of a read that its slot segment has direct slots for (the register-resident scorer's widest instantiation) */
#define MTB_SLOT_EPOCHS 31u
MTB_HD mtb_slot16 mtb_slot_pack(uint64_t qinfo, int32_t target_id, int32_t species_id, uint32_t dna, uint32_t reh, uint32_t ham, uint32_t epoch) {
    mtb_slot16 s;
    s.a = ((uint64_t)(uint32_t)species_id << 32) | (uint32_t)target_id;
    s.b = ((uint64_t)(epoch & 31u) << 59) | ((uint64_t)(ham & 15u) << 55) | ((uint64_t)(mtb_q_frame(qinfo) & 7u) << 52) |
          ((uint64_t)(mtb_q_pos(qinfo) & 0xFFFu) << 40) | ((uint64_t)(reh & 0xFFFFu) << 24) | (uint64_t)(dna & 0xFFFFFFu);
    return s;
}
MTB_HD uint32_t mtb_slot_epoch(const mtb_slot16 &s) { return (uint32_t)(s.b >> 59); }
/* Long reads (positions < 2^16): the buffer is cleared for every batch and a written slot carries bit 63 of word a instead of an epoch.
 *   a = 1[63] species[32..62] | target id[0..31]      b = hamming[59..62] frame[56..58] position[40..55] right_end_hamming[24..39] dna[0..23] */
MTB_HD mtb_slot16 mtb_lslot_pack(uint64_t qinfo, int32_t target_id, int32_t species_id, uint32_t dna, uint32_t reh, uint32_t ham) {
    mtb_slot16 s;
    s.a = (1ull << 63) | ((uint64_t)((uint32_t)species_id & 0x7FFFFFFFu) << 32) | (uint32_t)target_id;
    s.b = ((uint64_t)(ham & 15u) << 59) | ((uint64_t)(mtb_q_frame(qinfo) & 7u) << 56) | ((uint64_t)(mtb_q_pos(qinfo) & 0xFFFFu) << 40) |
          ((uint64_t)(reh & 0xFFFFu) << 24) | (uint64_t)(dna & 0xFFFFFFu);
    return s;
}
MTB_HD bool mtb_lslot_live(const mtb_slot16 &s) { return (s.a >> 63) != 0; }
MTB_HD int32_t mtb_lslot_species(const mtb_slot16 &s) { return (int32_t)((s.a >> 32) & 0x7FFFFFFFu); }
/* (frame, position, hamming, dna): compareMatches' order inside one species, 46 bits */
MTB_HD uint64_t mtb_lslot_key(const mtb_slot16 &s) {
    return (((s.b >> 56) & 7u) << 43) | (((s.b >> 40) & 0xFFFFu) << 27) | (((s.b >> 59) & 7u) << 24) | (s.b & 0xFFFFFFull);
}
MTB_HD mtb_match mtb_lslot_unpack(const mtb_slot16 &s, uint32_t seq_id) {
    mtb_match m;
    m.qinfo = mtb_qinfo(seq_id, (uint32_t)(s.b >> 40) & 0xFFFFu, (uint32_t)(s.b >> 56) & 7u);
    m.target_id = (int32_t)(uint32_t)s.a; m.species_id = mtb_lslot_species(s);
    m.dna = (uint32_t)s.b & 0xFFFFFFu; m.right_end_hamming = (uint16_t)((s.b >> 24) & 0xFFFFu);
    m.hamming = (uint8_t)((s.b >> 59) & 15u); m.pad = 0;
    return m;
}
/* slots of a long read with d metamers: d direct ones + a tail for the further matches of its queries (tf quarters of d, + 16) */
MTB_HD uint32_t mtb_lslot_tail(uint32_t d, uint32_t tf) { return d ? (uint32_t)(((uint64_t)d * tf) >> 2) + 16u : 0u; }
MTB_HD mtb_match mtb_slot_unpack(const mtb_slot16 &s, uint32_t seq_id) {
    mtb_match m;
    m.qinfo = mtb_qinfo(seq_id, (uint32_t)(s.b >> 40) & 0xFFFu, (uint32_t)(s.b >> 52) & 7u);
    m.target_id = (int32_t)(uint32_t)s.a; m.species_id = (int32_t)(uint32_t)(s.a >> 32);
    m.dna = (uint32_t)s.b & 0xFFFFFFu; m.right_end_hamming = (uint16_t)((s.b >> 24) & 0xFFFFu);
    m.hamming = (uint8_t)((s.b >> 55) & 15u); m.pad = 0;
    return m;
}

/* Selection over a run v[s..s+len): pass 1 (out == NULL) counts the candidates
 * with ham <= min(2*minHam, 7) (compareDna, KmerMatcher.cpp:1117-1146); pass 2
 * writes them in index order.  info/tax2species are indexed with info_base + i. */
MTB_HD uint32_t mtb_join_select(const mtb_tables *t, const uint64_t *v, uint64_t s, uint32_t len, uint64_t qvalue, uint64_t qinfo,
                                const uint32_t *info, uint64_t info_base, const int32_t *tax2species, int32_t max_taxid,
                                uint32_t info_mask, int32_t kmer_format, mtb_match *out, uint32_t out_cap, uint32_t skip = 0,
                                mtb_slot16 *out16 = 0, uint32_t epoch = 0) {
    if (len == 0) return 0;
    mtb_qrows q; mtb_prepare_query(t, qvalue, &q);
    uint32_t mn = 255;
    for (uint32_t i = 0; i < len; i++) { uint32_t h = mtb_ham_sum(&q, (uint32_t)v[s + i] & 0xFFFFFFu); mn = h < mn ? h : mn; }
    uint32_t thr = mtb_ham_threshold(mn);
    uint32_t cnt = 0;
    bool rev = mtb_hammings_reversed(mtb_q_frame(qinfo), kmer_format);
    for (uint32_t i = 0; i < len; i++) {
        uint32_t td = (uint32_t)v[s + i] & 0xFFFFFFu;
        uint32_t h = mtb_ham_sum(&q, td);
        if (h <= thr) {
            if ((out || out16) && cnt >= skip && cnt - skip < out_cap) {        /* selected candidates [skip, skip + out_cap) */
                int32_t tid = (int32_t)(info[info_base + s + i] & info_mask);
                int32_t sp = (tid >= 0 && tid <= max_taxid) ? tax2species[tid] : 0;
                if (out16) out16[cnt - skip] = mtb_slot_pack(qinfo, tid, sp, td, mtb_hammings(&q, td, rev), h, epoch);
                else {
                    mtb_match m;
                    m.qinfo = qinfo; m.target_id = tid; m.species_id = sp; m.dna = td;
                    m.right_end_hamming = mtb_hammings(&q, td, rev); m.hamming = (uint8_t)h; m.pad = 0;
                    out[cnt - skip] = m;
                }
            }
            cnt++;
        }
    }
    return cnt;
}

/* One query against the whole flat index (functional form of KmerMatcher::matchKmers,
 * KmerMatcher.cpp:275-450): candidates are the targets t < T-1 (the last entry
 * of the index is never a candidate, :363/:378) with the query's AA part.   */
MTB_HD uint32_t mtb_join_query(const mtb_tables *t, const mtb_index_view *ix, uint64_t qvalue,
                               uint64_t qinfo, mtb_match *out, uint32_t out_cap, int emit,
                               uint64_t *run_start_io, uint32_t *run_len_io) {
    uint64_t limit = ix->n_targets ? ix->n_targets - 1 : 0;      /* exclude last entry */
    if (!emit) mtb_join_find(ix->values, limit, qvalue, run_start_io, run_len_io);
    return mtb_join_select(t, ix->values, *run_start_io, *run_len_io, qvalue, qinfo, ix->info, 0, ix->tax2species, ix->max_taxid,
                           ix->info_mask, ix->kmer_format, emit ? out : (mtb_match *)0, out_cap);
}

/* ------------------------------------------------------------------ */
/* Match ordering (KmerMatcher.cpp:1149-1166)                           */
/* ------------------------------------------------------------------ */
MTB_HD bool mtb_match_less(const mtb_match &a, const mtb_match &b) {
    uint32_t sa = mtb_q_seq(a.qinfo), sb = mtb_q_seq(b.qinfo);
    if (sa != sb) return sa < sb;
    if (a.species_id != b.species_id) return a.species_id < b.species_id;
    uint32_t fa = mtb_q_frame(a.qinfo), fb = mtb_q_frame(b.qinfo);
    if (fa != fb) return fa < fb;
    uint32_t pa = mtb_q_pos(a.qinfo), pb = mtb_q_pos(b.qinfo);
    if (pa != pb) return pa < pb;
    if (a.hamming != b.hamming) return a.hamming < b.hamming;
    return a.dna < b.dna;
}

/* ------------------------------------------------------------------ */
/* Scoring                                                             */
/* ------------------------------------------------------------------ */
typedef struct {
    const uint8_t *acc_leaf;    /* rank "" or "accession": dropped from the descent when accession_level == 2 (Taxonomer.cpp:256-267); may be NULL */
    const int32_t *canon;       /* by taxid: itself, the merged.dmp target, or -1 if absent    */
    const int32_t *parent;      /* by canonical taxid; parent[root] = root                     */
    const int32_t *depth;       /* by canonical taxid                                          */
    const uint8_t *under_euk;   /* IsAncestor(eukaryota, taxid)  (Taxonomer.cpp:497-500)       */
    const int32_t *sp_parent;   /* parent(getTaxIdAtRank(taxid,"species")) (Taxonomer.cpp:178-185) */
    int32_t max_taxid;
    const void *node;           /* optional: mtb_tax_node[max_taxid + 1], the fields above gathered per taxid (one 16-byte load) */
} mtb_tax_view;

/* canon / depth / parent of a taxid behind ONE load: the scorer's taxonomy walks are chains of dependent lookups in
 * separate arrays (canon -> depth -> parent ...), each an L2 round trip; with the record a strain-under-species climb needs one. */
typedef struct { int32_t canon; int32_t depth; int32_t parent; uint32_t flags; } mtb_tax_node;      /* flags: 1 = under Eukaryota, 2 = accession-level leaf; of the canonical node */

typedef struct {
    int32_t max_codon_shift, dna_shift, denominator;  /* Taxonomer.cpp:34-48 */
    int32_t min_cons_cnt, min_cons_cnt_euk, kmer_format, accession_level;
    float   min_score, min_sp_score, tie_ratio;
} mtb_score_params;

MTB_HD void mtb_make_score_params(const mtb_params *p, mtb_score_params *s) {
    if (p->syncmer) { s->dna_shift = (8 - p->smer_len) * 3; s->max_codon_shift = 8 - p->smer_len; }
    else { s->dna_shift = 3; s->max_codon_shift = 1; }
    s->denominator = (p->seq_mode == 1 || p->seq_mode == 2) ? 100 : 1000;
    s->min_cons_cnt = p->min_cons_cnt; s->min_cons_cnt_euk = p->min_cons_cnt_euk; s->kmer_format = p->kmer_format;
    s->min_score = p->min_score; s->min_sp_score = p->min_sp_score; s->tie_ratio = p->tie_ratio;
    s->accession_level = p->accession_level;
}

MTB_HD int32_t mtb_tax_canon(const mtb_tax_view *t, int32_t x) { return (x >= 0 && x <= t->max_taxid) ? t->canon[x] : -1; }
MTB_HD bool mtb_tax_exists(const mtb_tax_view *t, int32_t x) { return mtb_tax_canon(t, x) >= 0; }
/* NcbiTaxonomy::LCA(a,b): a missing node yields the other one */
MTB_HD int32_t mtb_lca(const mtb_tax_view *t, int32_t a, int32_t b) {
    if (a == b) { int32_t c = mtb_tax_canon(t, a); return c < 0 ? a : c; }
    int32_t ca = mtb_tax_canon(t, a), cb = mtb_tax_canon(t, b);
    if (ca < 0) return b;
    if (cb < 0) return a;
    a = ca; b = cb;
    if (a == b) return a;
    int32_t da = t->depth[a], db = t->depth[b];
    while (da > db) { a = t->parent[a]; da--; }
    while (db > da) { b = t->parent[b]; db--; }
    while (a != b) { a = t->parent[a]; b = t->parent[b]; }
    return a;
}

/* Match.h:32-44 / Taxonomer.cpp:650-661: score of the `n` codons starting at
 * 2-bit field `first`, walking up (dir=+1) or down (dir=-1) */
MTB_HD float mtb_codon_score(uint32_t h) { return h == 0 ? 3.0f : 2.0f - 0.5f * (float)h; }
/* the n 2-bit fields a partial score covers, packed into the low 2n bits: the n right-end codons (fields 0..n-1) or the
 * n left-end ones (fields 7, 6, ...).  0 <= n <= 8. */
MTB_HD uint32_t mtb_part_fields(uint32_t reh, int n, bool left) {
    if (n <= 0) return 0u;
    if (n > 8) n = 8;
    return left ? ((reh & 0xFFFFu) >> (16 - 2 * n)) : (reh & ((1u << (2 * n)) - 1u));
}
/* Sum of the per-codon hamming fields / scores in closed form (popcounts) instead of a loop per codon: a codon scores
 * 3 if its field is 0, else 2 - 0.5 h, so n codons score 2 n + #zero fields - 0.5 sum(h).  All terms are multiples of
 * 0.5 far below 2^24: exact in fp32 in any association, i.e. bit-identical to the reference's running sum
 * (Match.h:32-87, Taxonomer.cpp:650-669). */
MTB_HD int32_t mtb_part_ham(uint32_t reh, int n, bool left) {
    const uint32_t f = mtb_part_fields(reh, n, left);
    return (int32_t)(__builtin_popcount(f & 0x5555u) + 2 * __builtin_popcount(f & 0xAAAAu));
}
MTB_HD float mtb_part_score(uint32_t reh, int n, bool left) {
    if (n <= 0) return 0.0f;
    if (n > 8) n = 8;
    const uint32_t f = mtb_part_fields(reh, n, left);
    const int32_t ham = __builtin_popcount(f & 0x5555u) + 2 * __builtin_popcount(f & 0xAAAAu);
    const int32_t zeros = n - __builtin_popcount((f | (f >> 1)) & 0x5555u);
    return (float)(4 * n + 2 * zeros - ham) * 0.5f;
}

typedef struct {           /* MatchPath (Taxonomer.h:34-57); endMatch = own slot */
    int32_t start, end;
    float   score;
    int32_t ham;
    int32_t depth;
    int32_t start_idx;     /* slot of startMatch */
} mtb_path;

#define MTB_PF_CONNECTED 1u
#define MTB_PF_EMITTED   2u

/* Taxonomer::isConsecutive / isConsecutive2 with shift (Taxonomer.cpp:684-699) */
MTB_HD bool mtb_consecutive(uint32_t dna_cur, uint32_t dna_next, int shift, bool fwd, int kmer_format) {
    uint32_t a = fwd ? dna_cur : dna_next, b = fwd ? dna_next : dna_cur;
    uint32_t keep = (1u << (24 - 3 * shift)) - 1u;
    if (kmer_format == 2) return (a & keep) == (b >> (3 * shift));
    return (a >> (3 * shift)) == (b & keep);
}

/* Taxonomer::getMatchPaths (Taxonomer.cpp:487-648) for one (species, frame)
 * block m[s..e) of the sorted match list, e - s >= 2.  path[i]/flag[i] are
 * per-match slots.  A path is "emitted" (pushed to filteredMatchPaths in the
 * reference) when it was not extended and is deep enough; emission order in
 * the reference equals slot order.                                          */
MTB_HD void mtb_sf_block_paths(const mtb_match *m, int32_t s, int32_t e, mtb_path *path, uint8_t *flag,
                               const mtb_score_params *sp, int32_t min_depth) {
    bool fwd = mtb_q_frame(m[s].qinfo) < 3;
    for (int32_t i = s; i < e; i++) {
        path[i].start = (int32_t)mtb_q_pos(m[i].qinfo); path[i].end = path[i].start + 23;
        path[i].score = mtb_part_score(m[i].right_end_hamming, 8, false);
        path[i].ham = m[i].hamming; path[i].depth = 1; path[i].start_idx = i; flag[i] = 0;
    }
    int32_t cs = s, ce = s;
    uint32_t cur_pos = mtb_q_pos(m[s].qinfo);
    while (ce < e && mtb_q_pos(m[ce].qinfo) == cur_pos) ce++;
    int32_t i = ce;
    while (i < e) {
        uint32_t next_pos = mtb_q_pos(m[i].qinfo);
        int32_t ns = i;
        while (i < e && mtb_q_pos(m[i].qinfo) == next_pos) i++;
        int32_t ne = i;
        int32_t shift = (int32_t)(next_pos - cur_pos) / 3;
        if (shift > 0 && shift <= sp->max_codon_shift) {
            for (int32_t nx = ns; nx < ne; nx++) {
                int32_t best = -1; float best_score = 0.0f;
                for (int32_t cu = cs; cu < ce; cu++) {
                    if (mtb_consecutive(m[cu].dna, m[nx].dna, shift, fwd, sp->kmer_format)) {
                        flag[cu] |= MTB_PF_CONNECTED;
                        if (path[cu].score > best_score) { best = cu; best_score = path[cu].score; }
                    }
                }
                if (best >= 0) {
                    path[nx].start = path[best].start;
                    path[nx].score = path[best].score + mtb_part_score(m[nx].right_end_hamming, shift, false);
                    path[nx].ham = path[best].ham + mtb_part_ham(m[nx].right_end_hamming, shift, false);
                    path[nx].depth = path[best].depth + shift;
                    path[nx].start_idx = path[best].start_idx;
                }
            }
        }
        for (int32_t cu = cs; cu < ce; cu++)
            if (!(flag[cu] & MTB_PF_CONNECTED) && path[cu].depth >= min_depth) flag[cu] |= MTB_PF_EMITTED;
        if (i == e)
            for (int32_t nx = ns; nx < ne; nx++)
                if (path[nx].depth >= min_depth) flag[nx] |= MTB_PF_EMITTED;
        cs = ns; ce = ne; cur_pos = next_pos;
    }
}

/* ordering of combineMatchPaths (Taxonomer.cpp:417-426): a before b */
MTB_HD bool mtb_path_before(const mtb_path &a, const mtb_path &b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.ham != b.ham) return a.ham < b.ham;
    return a.start > b.start;
}

/* Taxonomer::combineMatchPaths (Taxonomer.cpp:410-468) for the species block
 * m[s..e): stable order of the emitted paths, greedy non-overlap selection
 * with in-place trimming (trimMatchPath, :475-485).  order[] and acc[] are
 * per-match scratch slots.  Returns Sum(score)/read_len; *n_paths = number of
 * emitted paths of the species.                                             */
MTB_HD float mtb_species_combine(const mtb_match *m, int32_t s, int32_t e, mtb_path *path, const uint8_t *flag,
                                 int32_t *order, int32_t *acc, int32_t read_len, int32_t *n_paths) {
    int32_t n = 0;
    for (int32_t i = s; i < e; i++) {
        if (!(flag[i] & MTB_PF_EMITTED)) continue;
        int32_t j = n++;                                   /* stable insertion */
        while (j > 0 && mtb_path_before(path[i], path[order[s + j - 1]])) { order[s + j] = order[s + j - 1]; j--; }
        order[s + j] = i;
    }
    *n_paths = n;
    if (n == 0) return 0.0f;
    float score = 0.0f;
    int32_t na = 0;
    for (int32_t k = 0; k < n; k++) {
        mtb_path &p = path[order[s + k]];
        int32_t pi = order[s + k];
        bool drop = false;
        for (int32_t a = 0; a < na && !drop; a++) {
            const mtb_path &c = path[acc[s + a]];
            if (!((p.end < c.start) || (c.end < p.start))) {
                int32_t ov = (p.end < c.end ? p.end : c.end) - (p.start > c.start ? p.start : c.start) + 1;
                if (ov == p.end - p.start + 1) { drop = true; break; }
                if (ov < 24) {
                    if (p.start < c.start) {
                        p.end = c.start - 1;
                        int32_t h = p.ham - mtb_part_ham(m[pi].right_end_hamming, ov / 3, false);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[pi].right_end_hamming, ov / 3, false) - (float)(ov % 3);
                    } else {
                        p.start = c.end + 1;
                        int32_t h = p.ham - mtb_part_ham(m[p.start_idx].right_end_hamming, ov / 3, true);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[p.start_idx].right_end_hamming, ov / 3, true) - (float)(ov % 3);
                    }
                } else drop = true;
            }
        }
        if (!drop) { acc[s + na++] = pi; score += p.score; }
    }
    return score / (float)read_len;
}

/* Taxonomer::filterRedundantMatches (Taxonomer.cpp:205-241) over the best
 * species block m[s..e).  b_tax/b_ham are bucket arrays of n_buckets entries
 * (pos / dna_shift); out_tax/out_cnt receive Query::taxCnt in ascending taxid
 * order (std::map).  Returns the number of entries (<= out_cap written).    */
/* one bucket q of filterRedundantMatches: the bucket's taxon is the LCA of the
 * target ids of its minimum-Hamming matches (the reference's replace / LCA-merge
 * chain in match order gives exactly that).  Returns false if the bucket is empty. */
MTB_HD bool mtb_filter_bucket(const mtb_match *m, int32_t s, int32_t e, const mtb_tax_view *tx, int32_t dna_shift,
                              int32_t q, int32_t *tax_out) {
    int32_t tax = 0; uint32_t ham = 255;
    for (int32_t i = s; i < e; i++) {
        if ((int32_t)(mtb_q_pos(m[i].qinfo) / (uint32_t)dna_shift) != q) continue;
        uint32_t h = m[i].hamming;
        if (ham == 255 || h < ham) { tax = m[i].target_id; ham = h; }
        else if (h == ham) tax = mtb_lca(tx, tax, m[i].target_id);
    }
    *tax_out = tax;
    return ham != 255;
}
/* bucket taxa (b_ham[q] != 255 marks a used bucket) -> Query::taxCnt in
 * ascending taxid order (std::map).  Returns the number of entries.        */
MTB_HD int32_t mtb_taxcnt_gather(const int32_t *b_tax, const uint8_t *b_ham, int32_t n_buckets,
                                 int32_t *out_tax, uint32_t *out_cnt, int32_t out_cap) {
    int32_t n = 0;
    for (int32_t q = 0; q < n_buckets; q++) {
        if (b_ham[q] == 255) continue;
        int32_t t = b_tax[q];
        int32_t k = 0;
        while (k < n && out_tax[k] < t) k++;
        if (k < n && out_tax[k] == t) { out_cnt[k]++; continue; }
        if (n < out_cap) {
            for (int32_t j = n; j > k; j--) { out_tax[j] = out_tax[j - 1]; out_cnt[j] = out_cnt[j - 1]; }
            out_tax[k] = t; out_cnt[k] = 1; n++;
        }
    }
    return n;
}
MTB_HD int32_t mtb_filter_redundant(const mtb_match *m, int32_t s, int32_t e, const mtb_tax_view *tx,
                                    int32_t dna_shift, int32_t *b_tax, uint8_t *b_ham, int32_t n_buckets,
                                    int32_t *out_tax, uint32_t *out_cnt, int32_t out_cap) {
    for (int32_t q = 0; q < n_buckets; q++) {
        int32_t t;
        bool used = mtb_filter_bucket(m, s, e, tx, dna_shift, q, &t);
        b_tax[q] = t; b_ham[q] = used ? 0 : 255;
    }
    return mtb_taxcnt_gather(b_tax, b_ham, n_buckets, out_tax, out_cnt, out_cap);
}
/* bucket count that covers every position of a read (Taxonomer.cpp:210) */
MTB_HD int32_t mtb_num_buckets(int32_t read_len, int32_t dna_shift) { return (read_len + 3) / dna_shift + 2; }

/* Taxonomer::lowerRankClassification + getSpeciesCladeCounts + BFS
 * (Taxonomer.cpp:252-314), accession_level != 2.  Descend from the species
 * while exactly one child holds the maximal clade count and that count is
 * >= max((len-1)/denominator, ...) in the sense of BFS's compare chain.      */
MTB_HD int32_t mtb_lower_rank(const mtb_tax_view *tx, const int32_t *tc_tax, const uint32_t *tc_cnt, int32_t n,
                              int32_t species, int32_t read_len, int32_t denominator, int32_t accession_level = 0) {
    uint32_t thr = (uint32_t)((read_len - 1) / denominator);
    int32_t root = mtb_tax_canon(tx, species);
    if (root < 0) return species;
    for (int guard = 0; guard < 64; guard++) {
        /* children of root on the paths taxon -> species, with clade counts */
        uint32_t max_cnt = thr; int32_t best = -1; int32_t n_best = 0; bool any_child = false;
        for (int32_t i = 0; i < n; i++) {
            /* child of root above tc_tax[i], if tc_tax[i] is a strict descendant of root */
            int32_t t = mtb_tax_canon(tx, tc_tax[i]);
            if (t < 0 || tx->depth[t] <= tx->depth[root]) continue;
            int32_t c = t;
            while (tx->depth[c] > tx->depth[root] + 1) c = tx->parent[c];
            if (tx->parent[c] != root) continue;
            if (accession_level == 2 && tx->acc_leaf && tx->acc_leaf[c]) continue;     /* erased from the children list */
            any_child = true;
            /* count each distinct child once: only at its first contributing entry */
            bool first = true;
            for (int32_t j = 0; j < i && first; j++) {
                int32_t u = mtb_tax_canon(tx, tc_tax[j]);
                if (u < 0 || tx->depth[u] <= tx->depth[root]) continue;
                int32_t d = u;
                while (tx->depth[d] > tx->depth[root] + 1) d = tx->parent[d];
                if (d == c) first = false;
            }
            if (!first) continue;
            uint32_t clade = 0;
            for (int32_t j = i; j < n; j++) {
                int32_t u = mtb_tax_canon(tx, tc_tax[j]);
                if (u < 0 || tx->depth[u] <= tx->depth[root]) continue;
                int32_t d = u;
                while (tx->depth[d] > tx->depth[root] + 1) d = tx->parent[d];
                if (d == c) clade += tc_cnt[j];
            }
            if (clade > max_cnt) { best = c; n_best = 1; max_cnt = clade; }
            else if (clade == max_cnt) { if (n_best == 0) best = c; n_best++; }
        }
        if (!any_child) return root;
        if (n_best == 1) root = best; else return root;
    }
    return root;
}

/* Second half of Taxonomer::getBestSpeciesMatches (Taxonomer.cpp:354-407) and
 * the early exits of Taxonomer::chooseBestTaxon (Taxonomer.cpp:130-165) for one
 * read.  sps[s] holds, at the first slot s of every species block,
 * min(combine(),1) or -1 if the species produced no path.  Returns true when
 * a single best species was chosen (then best_s..best_e bound its matches and
 * *species is its id) and the redundancy filter / sub-species descent must run. */
MTB_HD bool mtb_read_select(const mtb_match *m, int32_t n, const float *sps, const mtb_tax_view *tx,
                            const mtb_score_params *sp, mtb_result *R, int32_t *best_s_out, int32_t *best_e_out, int32_t *species) {
    R->classification = 0; R->score = 0.0f; R->is_classified = 0; R->n_taxcnt = 0;
    float best_sp = 0.0f; int32_t best_s = 0, best_e = 0; int32_t meaningful = 0;
    int32_t i = 0;
    while (i < n) {
        int32_t s = i; int32_t spc = m[i].species_id;
        while (i < n && m[i].species_id == spc) i++;
        float sc = sps[s];
        if (sc == -1.0f) continue;                 /* no path for this species          */
        if (sc < sp->min_score) continue;          /* Taxonomer.cpp:357-359             */
        if (sc > 0.0f) meaningful++;
        if (sc > best_sp) { best_sp = sc; best_s = s; best_e = i; }
    }
    if (meaningful == 0) return false;             /* score 0, unclassified (:372-375)  */
    /* ties within tie_ratio (:388-402); LCA(vector) skips unknown ids */
    float sum = 0.0f; int32_t n_max = 0; int32_t lca = -1; int32_t only = 0, first_spc = 0;
    float cut = best_sp * sp->tie_ratio;
    i = 0;
    while (i < n) {
        int32_t s = i; int32_t spc = m[i].species_id;
        while (i < n && m[i].species_id == spc) i++;
        float sc = sps[s];
        if (sc == -1.0f || sc < sp->min_score) continue;
        if (sc >= cut) {
            sum += sc; only = spc; n_max++;
            if (n_max == 1) first_spc = spc;       /* the LCA is only needed for ties */
            else {
                if (n_max == 2) lca = mtb_tax_exists(tx, first_spc) ? mtb_tax_canon(tx, first_spc) : -1;
                if (mtb_tax_exists(tx, spc)) lca = lca < 0 ? mtb_tax_canon(tx, spc) : mtb_lca(tx, lca, spc);
            }
        }
    }
    float score = n_max > 1 ? sum / (float)n_max : sum;
    R->score = score;
    if (score == 0.0f || score < sp->min_score) return false;    /* :149-156 */
    if (n_max > 1) { R->is_classified = 1; R->classification = lca < 0 ? 0 : lca; return false; }   /* :159-165 */
    *best_s_out = best_s; *best_e_out = best_e; *species = only;
    R->is_classified = 1;
    return true;
}
/* after the redundancy filter: Taxonomer.cpp:178-198 */
MTB_HD void mtb_read_finish(const mtb_tax_view *tx, const mtb_score_params *sp, int32_t species, int32_t read_len,
                            const int32_t *out_tax, const uint32_t *out_cnt, int32_t ntc, mtb_result *R) {
    R->n_taxcnt = (uint16_t)ntc;
    if (R->score < sp->min_sp_score) {
        R->classification = (species >= 0 && species <= tx->max_taxid) ? tx->sp_parent[species] : 0;
        return;
    }
    R->classification = mtb_lower_rank(tx, out_tax, out_cnt, ntc, species, read_len, sp->denominator, sp->accession_level);
}
MTB_HD void mtb_read_decide(const mtb_match *m, int32_t n, const float *sps, const mtb_tax_view *tx,
                            const mtb_score_params *sp, int32_t read_len, int32_t *b_tax, uint8_t *b_ham,
                            int32_t n_buckets, int32_t *out_tax, uint32_t *out_cnt, int32_t out_cap,
                            mtb_result *R) {
    int32_t bs, be, species;
    if (!mtb_read_select(m, n, sps, tx, sp, R, &bs, &be, &species)) return;
    int32_t ntc = mtb_filter_redundant(m, bs, be, tx, sp->dna_shift, b_tax, b_ham, n_buckets, out_tax, out_cnt, out_cap);
    mtb_read_finish(tx, sp, species, read_len, out_tax, out_cnt, ntc, R);
}

#endif /* MTB_CORE_H */
