/* kernels_extract.h -- six-frame translation, syncmer selection and 64-bit
 * metamer packing: one wavefront per read (BASELINE north_star).
 *
 * Per frame the wave first turns the read into a codon-byte string in LDS
 * (lane j -> codon j: three base loads + one 64-entry LUT), then lane p packs
 * the window of codons p..p+7 and applies the closed-syncmer test
 * (mtb_window_metamer); selected windows are compacted with a ballot /
 * popcount prefix.  Two passes (count, emit) give deterministic output in the
 * reference's emission order (read, mate, frame, window):
 *   KmerExtractor::fillQueryKmerBuffer  src/commons/KmerExtractor.cpp:342-373
 *   MetamerScanner::next               src/commons/KmerScanner.h:82-117
 *   SyncmerScanner::next               src/commons/SyncmerScanner.h:36-101
 * Algorithmic HBM bytes: sum(L) read per pass + 16 B per emitted metamer.    */
#ifndef MTB_KERNELS_EXTRACT_H
#define MTB_KERNELS_EXTRACT_H
#include "dev_util.h"
#include "mtb_core.h"

#define MTB_EXTRACT_STAGE 320        /* reads up to this length are staged in LDS once */

struct ExtractArgs {
    const char *bases; const uint64_t *offs;
    const char *bases2; const uint64_t *offs2;
    uint64_t n_reads;
    int32_t seq_mode, syncmer, smer_len, kmer_format;
    int32_t tag_ord;      /* MODE 2: bits 16-31 of qinfo's position field carry the metamer's ordinal within its read */
    int32_t dig_shift;    /* MODE 2 with dig_out: the letter pair whose base-21 digit is written next to every record (54 = the top pair: first pass of the bucket-local sort; 34 = LSD) */
};

/* MODE 0 = count pass, 1 = emit pass (deterministic reference order: the stage API), 2 = single pass for the fused
 * path: the count pass (the same arithmetic again, 13 of 30 ms) and the offset scan disappear.  The wave collects
 * metamers in an LDS buffer and copies them into a chunk of `out` it owns; chunks come from one global atomic each
 * (MTB_EXTRACT_CHUNK metamers, smaller near the end of the wave's reads; a flush per atomic on one address was
 * 20 ms slower, measured), and the unused tail of a wave's last chunk is filled with blank records (sequenceID 0,
 * skipped by the join; < 1 % of the list).  The order of the runs in `out` is arbitrary -- the radix sort that
 * follows does not care.  counter: [0] records allocated, [1] set if out_cap was too small, [2] real metamers,
 * [3] largest number of metamers of one read.  With a.tag_ord the position field of qinfo (positions < 2^16) also
 * carries the ordinal of the metamer within its read (extraction order: mate, frame, window) in bits 16-31: the
 * join uses it as the slot of the query's first match inside the read's segment.                                  */
#define MTB_EXTRACT_BUF 320          /* metamers buffered per wave in MODE 2 (5 KB) */
#define MTB_EXTRACT_CHUNK 8192
template <int MODE>
__global__ __launch_bounds__(64) void k_extract(ExtractArgs a, const mtb_tables *__restrict__ tabs,
                                                uint32_t *__restrict__ counts, const uint64_t *__restrict__ out_offs,
                                                mtb_kmer *__restrict__ out, int32_t *__restrict__ qlen,
                                                int32_t *__restrict__ qlen2, uint32_t *__restrict__ max_len,
                                                unsigned long long *__restrict__ counter, uint64_t out_cap,
                                                uint16_t *__restrict__ dig_out = nullptr, uint8_t *__restrict__ off_flags = nullptr) {
    constexpr bool EMIT = MODE != 0;
    constexpr bool STATS = MODE != 1;             /* qlen / max_len are produced by the count pass or the single pass */
    __shared__ mtb_kmer s_out[MODE == 2 ? MTB_EXTRACT_BUF : 1];
    __shared__ unsigned long long s_base;
    uint32_t n_buf = 0;
    uint64_t chunk_pos = 0, chunk_end = 0, produced = 0, reads_done = 0, cur_read = 0;
    bool overflow = false;
    auto flush = [&]() {
        if (MODE != 2 || n_buf == 0) return;
        __syncthreads();
        uint32_t done = 0;
        while (done < n_buf) {
            if (chunk_pos == chunk_end) {               /* new chunk: sized to what this wave still expects to produce */
                const uint64_t need = n_buf - done;
                const uint64_t rem_reads = (a.n_reads - 1 - cur_read) / gridDim.x + 1;
                const uint64_t avg = reads_done ? produced / reads_done + 1 : 128;
                uint64_t size = need + rem_reads * avg * 5 / 4 + 64;
                size = size < MTB_EXTRACT_CHUNK ? size : MTB_EXTRACT_CHUNK;
                size = size < need ? need : size;
                if (threadIdx.x == 0) s_base = atomicAdd(counter, (unsigned long long)size);
                __syncthreads();
                chunk_pos = (uint64_t)s_base; chunk_end = chunk_pos + size;
                __syncthreads();
                if (chunk_end > out_cap) { overflow = true; chunk_pos = chunk_end; break; }
            }
            const uint32_t room = (uint32_t)(chunk_end - chunk_pos < (uint64_t)(n_buf - done) ? chunk_end - chunk_pos : (uint64_t)(n_buf - done));
            for (uint32_t i = threadIdx.x; i < room; i += 64) {
                const mtb_kmer x = s_out[done + i];
                out[chunk_pos + i] = x;
                if (dig_out) {          /* first radix pass's digit (amino-acid letters 4,5 as a base-21 pair, kernels_sort.h) */
                    uint32_t two = (uint32_t)(x.value >> a.dig_shift) & 1023u, d = (two >> 5) * 21u + (two & 31u);
                    dig_out[chunk_pos + i] = (uint16_t)(d < 511u ? d : 511u);
                }
            }
            chunk_pos += room; done += room;
        }
        produced += n_buf;
        __syncthreads();
        n_buf = 0;
    };
    __shared__ mtb_tables s_tab;
    __shared__ __attribute__((aligned(8))) uint8_t s_cod[80];
    __shared__ __attribute__((aligned(8))) uint8_t s_cod3[MODE == 2 ? 3 : 1][MODE == 2 ? 120 : 8];      /* codon bytes of the three frames of a strand (reads <= 320 bases: <= 106 codons) */
    __shared__ uint8_t s_code[MTB_EXTRACT_STAGE];      /* a short read's bases as codes: one global load serves all six frames */
    const uint32_t lane = threadIdx.x;
    for (uint32_t i = lane; i < sizeof(mtb_tables) / 4; i += 64) ((uint32_t *)&s_tab)[i] = ((const uint32_t *)tabs)[i];
    __syncthreads();
    uint32_t my_max = 0, my_maxq = 0;
    uint32_t el_max = 0, el_maxq = 0, n_off = 0;       /* single pass with off_flags: maxima over the reads the slot segments can hold, and the others counted */
    const bool paired = a.seq_mode == 2;
    for (uint64_t r = blockIdx.x; r < a.n_reads; r += gridDim.x) {
        cur_read = r; reads_done++;
        const uint64_t o1 = a.offs[r];
        const int32_t len1 = (int32_t)(a.offs[r + 1] - o1);
        uint64_t o2 = 0; int32_t len2 = 0;
        if (paired) { o2 = a.offs2[r]; len2 = (int32_t)(a.offs2[r + 1] - o2); }
        const int32_t ql1 = mtb_used_len(len1), ql2 = paired ? mtb_used_len(len2) : 0;
        if (STATS) {
            if (lane == 0) { qlen[r] = ql1; qlen2[r] = ql2; }
            uint32_t tot_len = (uint32_t)(ql1 + ql2);
            my_max = tot_len > my_max ? tot_len : my_max;
        }
        /* pair skipped if either mate is too short (KmerExtractor.cpp:443-453) */
        const bool skip = mtb_read_too_short(len1) || (paired && mtb_read_too_short(len2));
        if (skip) { if ((MODE == 0 || (MODE == 2 && counts)) && lane == 0) counts[r] = 0; continue; }
        uint64_t wpos = MODE == 1 ? out_offs[r] : 0;
        uint32_t total = 0;
        for (int mate = 0; mate < (paired ? 2 : 1); mate++) {
            const char *seq = mate ? a.bases2 + o2 : a.bases + o1;
            const int32_t len = mate ? len2 : len1;
            const uint32_t pos_off = mate ? (uint32_t)(ql1 + 3) : 0u;     /* KmerExtractor.cpp:329 */
            const int32_t used = mtb_used_len(len);
            const int32_t n_cod = used / 3, n_win = n_cod - 7;
            const bool old_fmt = a.kmer_format == 1;          /* OldMetamerScanner geometry, base-21 amino-acid part */
            const bool staged = !old_fmt && len <= MTB_EXTRACT_STAGE;
            if (staged) {
                __syncthreads();
                for (int32_t i = (int32_t)lane; i < len; i += 64) s_code[i] = s_tab.base[(uint8_t)seq[i]];
                __syncthreads();
            }
            if (MODE == 2 && staged) {
                /* single pass, staged read: the three frames of a strand share n_cod / n_win, so their windows are
                 * laid end to end over the lanes (150 bp: 3 x 42 = 126 windows = 2 wave steps instead of 3).  Codon
                 * bytes per frame in s_cod3[frame][codon]; combined window index c -> (frame, window); the windows of
                 * a strand whose positions fall with the window index are walked backwards, so that ordinals rise
                 * with the position inside every frame. */
                for (int strand = 0; strand < 2; strand++) {
                    const bool fwd = strand == 0;
                    const bool back = !fwd;                              /* kmer_format 2 only (staged excludes the old format) */
                    __syncthreads();
                    for (int fi = 0; fi < 3; fi++) {
                        const int32_t begin = mtb_frame_begin(len, strand * 3 + fi);
                        for (int32_t j = (int32_t)lane; j < n_cod; j += 64)
                            s_cod3[fi][j] = mtb_codon_byte_codes(&s_tab, s_code, mtb_codon_ci(begin, used, j, fwd), fwd);
                    }
                    __syncthreads();
                    const int32_t n_all = 3 * n_win;
                    for (int32_t c0 = 0; c0 < n_all; c0 += 64) {
                        const int32_t cidx = c0 + (int32_t)lane;
                        const int32_t fi = (cidx >= n_win) + (cidx >= 2 * n_win);
                        const int32_t wi = cidx - fi * n_win;
                        const int32_t w = back ? n_win - 1 - wi : wi;
                        bool ok = false; uint64_t v = 0;
                        if (cidx < n_all) {
                            const uint32_t *c32 = (const uint32_t *)&s_cod3[fi][0];
                            const uint32_t q = (uint32_t)w >> 2, sh = (uint32_t)w & 3u;
                            const uint32_t x0 = c32[q], x1 = c32[q + 1], x2 = c32[q + 2];
                            ok = mtb_window_metamer_words(__builtin_amdgcn_alignbyte(x1, x0, sh), __builtin_amdgcn_alignbyte(x2, x1, sh), a.syncmer, a.smer_len, &v);
                        }
                        const uint64_t mask = __ballot(ok);
                        const uint32_t c = (uint32_t)__popcll(mask);
                        if (n_buf + c > MTB_EXTRACT_BUF) flush();        /* wave-uniform */
                        if (ok) {
                            const int f = strand * 3 + fi;
                            const int32_t begin = mtb_frame_begin(len, f);
                            const uint32_t below = (uint32_t)__popcll(mask & lanemask_lt());
                            uint32_t pf = mtb_window_pos(begin, used, w, fwd) + pos_off;
                            if (a.tag_ord) { const uint32_t ord = total + below; pf |= (ord < 0xFFFFu ? ord : 0xFFFFu) << 16; }
                            mtb_kmer k; k.value = v; k.qinfo = mtb_qinfo((uint32_t)(r + 1), pf, (uint32_t)f);
                            s_out[n_buf + below] = k;
                        }
                        n_buf += c; total += c;
                    }
                }
            } else
            for (int f = 0; f < 6; f++) {
                const bool fwd = f < 3;
                const int32_t begin = mtb_frame_begin(len, f);
                /* tagged single pass: ordinals must rise with the position inside a frame, and the positions of a reverse
                 * frame fall with the window index -> walk its chunks (and rank inside a chunk) backwards */
                const bool back = MODE == 2 && a.tag_ord && (old_fmt ? fwd : !fwd);      /* OldMetamerScanner: the forward frames fall */
                const int32_t n_chunk = (n_win + 63) / 64;
                for (int32_t cw = 0; cw < n_chunk; cw++) {
                    const int32_t w0 = (back ? n_chunk - 1 - cw : cw) * 64;
                    int32_t j = w0 + (int32_t)lane;
                    int32_t j2 = w0 + 64 + (int32_t)lane;
                    if (staged) {
                        if (j < n_cod) s_cod[lane] = mtb_codon_byte_codes(&s_tab, s_code, mtb_codon_ci(begin, used, j, fwd), fwd);
                        if (lane < 8 && j2 < n_cod) s_cod[64 + lane] = mtb_codon_byte_codes(&s_tab, s_code, mtb_codon_ci(begin, used, j2, fwd), fwd);
                    } else if (old_fmt) {
                        if (j < n_cod) s_cod[lane] = mtb_codon_byte_old(&s_tab, seq, mtb_codon_ci_old(begin, used, j, fwd), fwd);
                        if (lane < 8 && j2 < n_cod) s_cod[64 + lane] = mtb_codon_byte_old(&s_tab, seq, mtb_codon_ci_old(begin, used, j2, fwd), fwd);
                    } else {
                        if (j < n_cod) s_cod[lane] = mtb_codon_byte(&s_tab, seq, mtb_codon_ci(begin, used, j, fwd), fwd);
                        if (lane < 8 && j2 < n_cod) s_cod[64 + lane] = mtb_codon_byte(&s_tab, seq, mtb_codon_ci(begin, used, j2, fwd), fwd);
                    }
                    __syncthreads();
                    int32_t w = w0 + (int32_t)lane;
                    bool ok = false; uint64_t v = 0;
                    if (w < n_win) {
                        if (old_fmt) ok = mtb_window_metamer_old(&s_cod[lane], &v);
                        else {          /* the window's 8 codon bytes from three aligned LDS words + two byte alignments */
                            const uint32_t *c32 = (const uint32_t *)s_cod;
                            const uint32_t q = lane >> 2, sh = lane & 3u;
                            const uint32_t x0 = c32[q], x1 = c32[q + 1], x2 = c32[q + 2];
                            ok = mtb_window_metamer_words(__builtin_amdgcn_alignbyte(x1, x0, sh), __builtin_amdgcn_alignbyte(x2, x1, sh), a.syncmer, a.smer_len, &v);
                        }
                    }
                    uint64_t mask = __ballot(ok);
                    uint32_t c = (uint32_t)__popcll(mask);
                    if (MODE == 2 && n_buf + c > MTB_EXTRACT_BUF) flush();        /* wave-uniform */
                    if (EMIT && ok) {
                        mtb_kmer k;
                        k.value = v;
                        uint32_t pf = (old_fmt ? mtb_window_pos_old(begin, used, w, fwd) : mtb_window_pos(begin, used, w, fwd)) + pos_off;
                        if (MODE == 2 && a.tag_ord) {
                            const uint32_t below = (uint32_t)__popcll(mask & lanemask_lt());
                            const uint32_t ord = total + (back ? c - 1 - below : below);
                            pf |= (ord < 0xFFFFu ? ord : 0xFFFFu) << 16;
                        }
                        k.qinfo = mtb_qinfo((uint32_t)(r + 1), pf, (uint32_t)f);
                        if (MODE == 2) s_out[n_buf + (uint32_t)__popcll(mask & lanemask_lt())] = k;
                        else out[wpos + (uint64_t)__popcll(mask & lanemask_lt())] = k;
                    }
                    if (MODE == 2) n_buf += c;
                    wpos += c; total += c;
                    __syncthreads();
                }
            }
        }
        if ((MODE == 0 || (MODE == 2 && counts)) && lane == 0) counts[r] = total;      /* single pass: only the long-read slot path asks for them */
        my_maxq = total > my_maxq ? total : my_maxq;
        if (MODE == 2 && off_flags) {
            /* a read whose positions do not fit the 12-bit field of a slot record, or with more metamers than a slot segment has direct
             * slots for, is marked: the join files its matches in the overflow list, it is scored from an exact segment (mtb_api.hip) */
            const uint32_t tl = (uint32_t)(ql1 + ql2);
            if (tl + 3u >= MTB_SLOT_MAX_POS || total > MTB_SLOT_MAX_Q) { if (lane == 0) off_flags[r] = 1; n_off++; }
            else { el_max = tl > el_max ? tl : el_max; el_maxq = total > el_maxq ? total : el_maxq; }
        }
    }
    flush();
    if (MODE == 2) {
        mtb_kmer blank; blank.value = 0; blank.qinfo = 0;
        if (!overflow) for (uint64_t i = chunk_pos + threadIdx.x; i < chunk_end; i += 64) { out[i] = blank; if (dig_out) dig_out[i] = 0; }
        if (threadIdx.x == 0) { if (overflow) counter[1] = 1ull; else atomicAdd(counter + 2, (unsigned long long)produced); atomicMax(counter + 3, (unsigned long long)my_maxq);
                                if (off_flags) { atomicMax(counter + 4, (unsigned long long)el_maxq); if (n_off) atomicAdd(counter + 5, (unsigned long long)n_off); atomicMax(counter + 6, (unsigned long long)el_max); } }
    }
    if (STATS && max_len) {
        for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_down(my_max, d, 64); my_max = o > my_max ? o : my_max; }
        if (lane == 0 && my_max) atomicMax(max_len, my_max);
    }
}

#endif
