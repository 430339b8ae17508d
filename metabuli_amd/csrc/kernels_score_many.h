/* kernels_score_many.h -- the short reads that meet MANY species: reads of conserved genes, whose query metamers land in candidate
 * runs of 10^3 - 10^4 species and bring a few hundred matches each, most of them of species that appear once or twice.
 * Their tails overflow, so neither k_score_fast nor the slot mode of k_score can take them from their slots; until round 4 they were
 * copied to exact segments, sorted by a generic key sort and scored by the generic kernel out of HBM slabs (k_big_* -> k_segsort_* ->
 * k_score<.., DYN>): a fifth of the step for a twelfth of the reads.
 *
 * What the reference does with such a read (Taxonomer::getBestSpeciesMatches, src/commons/Taxonomer.cpp:316-408): the sorted match list
 * is cut into species, a species into (species, frame) groups, and ONLY groups of two or more matches reach getMatchPaths (:342,
 * `if (i - start > 1)`).  A species none of whose groups holds two matches gets no path, hence no entry in sp2score, hence can be
 * neither the best species nor part of a tie -- and the later steps (redundancy filter, taxCnt, lower-rank descent: :130-260) read
 * the matches of the best species' range only.  So the matches of such species are DEAD: dropping them before anything is ordered
 * changes no output of the read.  In a read of a conserved gene that is most of the matches.
 *
 * k_score_many, one wavefront per listed read (the reads k_score<SLOT> deferred), straight from where the join left the matches --
 * the read's slot segment (direct + tail slots) and its entries of the overflow list, grouped by read by k_ovf_group:
 *   pass 1  every live record enters its species into an open-addressing table in LDS (atomicCAS on the key) and marks its frame:
 *           bit f = "a match in frame f", bit 8 + f = "a second match in frame f" (two atomicOr);
 *   pass 2  the records are read again (L2 hits), those of species with any "second match" bit are staged in LDS in encounter order;
 *   then    the generic per-read phases (score_read_par, kernels_score.h: rank sort on the 64-bit compareMatches key, paths,
 *           combination, decision, filter, taxCnt, descent) run on the survivors -- a list of the size of an ordinary read.
 * Reads it cannot take (survivors beyond the staging, species table full, reads routed around their slots, too many position buckets)
 * are listed for the exact-segment path, which stays as the catch-all.
 * The table lives in the part of the scoring workspace that pass 2 does not write (everything behind the match records).
 * Algorithmic HBM bytes: 16 per slot + 24 per overflow entry, read twice (the second time from L2), + 24 per read result.      */
#ifndef MTB_KERNELS_SCORE_MANY_H
#define MTB_KERNELS_SCORE_MANY_H
#include "dev_util.h"
#include "mtb_core.h"
#include "kernels_join.h"
#include "kernels_score.h"
#include "kernels_score_long.h"
#include "kernels_seg_order.h"

#define MTB_MANY_CLAIM 16u               /* listed reads a wave claims per atomic on the work counter */
#define MTB_MANY_HASH 1024u              /* species table entries (8 bytes each); a read with more than 3/4 of that many species is handed on */

/* overflow entries per read: what the join pushed beyond the read's tail (cursor counts every match after the query's first one).
 * Reads routed around their slots (off[]: cursor pushed by tail_cap + 1 per match) are not grouped: they go to the exact-segment path. */
__global__ __launch_bounds__(256) void k_ovf_count(const uint32_t *__restrict__ cursor, const uint8_t *__restrict__ off, uint64_t n_reads, uint32_t tail_cap,
                                                    uint32_t *__restrict__ novf) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n_reads) return;
    const uint32_t cur = cursor[r];
    const uint32_t n = (cur > tail_cap && !(off && off[r])) ? cur - tail_cap : 0u;
    novf[r] = n <= 65535u ? n : 0u;          /* (the place of an entry in its group rides in 16 bits of the record; a read beyond that takes the exact-segment path) */
}
/* every entry of the overflow list -> its read's group.  The directory join's entries carry their place (pad & 2: bits [16, 32) of qinfo =
 * the match's tail cursor value - the tail's capacity, k_join_dir's ovf_put): a plain store; the entries of the other producers (k_join<SEG>,
 * k_slot_place) are placed by a returning atomic on the read's counter (the order inside a group is irrelevant: the scorer sorts; a read's
 * entries all come from one producer).  The grouped records carry the reference's qinfo and pad = 0 again.
 * region_cap != 0: striped list (JoinSegArgs::ovf_stripes), blockIdx.y = stripe. */
__global__ __launch_bounds__(256) void k_ovf_group(const mtb_match *__restrict__ ovf, uint64_t n_ovf, const uint64_t *__restrict__ start, const uint32_t *__restrict__ novf,
                                                    uint32_t *__restrict__ ocur, mtb_match *__restrict__ out, uint64_t region_cap, const unsigned long long *__restrict__ counters) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (region_cap) {
        const unsigned long long cnt = counters[8 * blockIdx.y];
        if (i >= cnt || i >= region_cap) return;
        i += (uint64_t)blockIdx.y * region_cap;
    } else if (i >= n_ovf) return;
    mtb_match m = ovf[i];
    const uint32_t r = mtb_q_seq(m.qinfo) - 1;
    const uint32_t cap = novf[r];
    if (!cap) return;                                     /* a read routed around its slots, or one with more entries than a group holds */
    uint32_t at;
    if (m.pad & 2u) { at = (uint32_t)(m.qinfo >> 16) & 0xFFFFu; m.qinfo &= ~0xFFFF0000ull; m.pad = 0; }
    else at = atomicAdd(&ocur[r], 1u);
    if (at < cap) out[start[r] + at] = m;
}

/* ---- the reads beyond k_score_many's staging: a read of a conserved gene whose organism is NOT in the index has no equal target in its
 * candidate runs, every query of it selects by hamming distance, and the read brings THOUSANDS of matches over a thousand species, most of
 * them with two matches in some frame (the dead-species drop barely helps).  Until round 5 such reads were copied to exact segments and
 * sorted by an HBM-resident bitonic network, then scored out of HBM slabs: 72 + 41 ms per 2 M reads of held-out organisms (163 k such
 * reads).  k_many_sort, one workgroup per listed read, from the same two sources as k_score_many (slots + grouped overflow entries):
 *   pass 1   species table in LDS (4096 entries), frame bits as above;
 *   pass 2   the survivors' source indices -> a list (one LDS atomic per wave step);
 *   pass 3   their 64-bit compareMatches keys (species, frame, position, hamming, dna: host-checked to fit) into the table's storage,
 *            bitonic sort of (key, source index) in LDS;
 *   pass 4   the records in order -> the read's exact segment in HBM (24-byte Match records), seg_cnt = survivors.
 * k_score_long<2048, 256, 256, 256> (kernels_score_long.h: a workgroup per read streaming its sorted segment) scores them.  Reads beyond the
 * budgets here or there are flagged in `todo` and take the exact-segment path. */
#define MTB_MSORT_NT 256
/* Two instantiations (round 6): <11, 2048> first -- 2048 table entries, 2048 survivors: 20 KB of LDS, eight workgroups per CU; the typical read
 * of this tier carries ~1100 records -- and <12, 4096> (40 KB, three per CU) for the reads the first one flags 2 (table beyond 3/4, survivors beyond its
 * budget); flag 1 = a read neither can take (exact-segment path).  `only_flag` != 0: only the listed reads whose flag equals it are taken (and the
 * flag is cleared first). */
template <int LOG2HASH, uint32_t MAXN>
__global__ __launch_bounds__(MTB_MSORT_NT) void k_many_sort(const mtb_slot16 *__restrict__ slots_all, uint32_t stride, uint32_t direct, uint32_t epoch,
                                                             const uint32_t *__restrict__ cursor, const uint8_t *__restrict__ off_reads,
                                                             const mtb_match *__restrict__ ovfg, const uint64_t *__restrict__ ovf_start,
                                                             const uint32_t *__restrict__ list, uint32_t n_list, const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2,
                                                             int32_t dna_shift, const uint64_t *__restrict__ big_start, mtb_match *__restrict__ big, uint32_t *__restrict__ seg_cnt,
                                                             uint8_t *__restrict__ todo, uint32_t only_flag, uint32_t beyond_flag /* what a read beyond THIS instantiation's budgets is flagged */) {
    constexpr uint32_t HASH = 1u << LOG2HASH;
    __shared__ __attribute__((aligned(16))) uint64_t s_u[HASH];          /* the species table (keys | values), then the sort keys */
    __shared__ uint16_t s_idx[MAXN];
    __shared__ uint32_t s_n, s_full;
    static_assert(MAXN * 8 <= HASH * 8, "the sort keys live in the table's storage");
    uint32_t *const h_key = (uint32_t *)s_u, *const h_val = h_key + HASH;
    const uint32_t tid = threadIdx.x, lane = tid & 63u;
    const uint64_t lt = lanemask_lt();
    const uint32_t tail_cap = stride - direct;
    for (uint32_t b = blockIdx.x; b < n_list; b += gridDim.x) {
        const uint64_t r = (uint64_t)list[b];
        if (only_flag && todo[r] != only_flag) continue;          /* (uniform over the workgroup) */
        const uint32_t cur = cursor[r], tail_n = cur < tail_cap ? cur : tail_cap;
        uint64_t o0 = 0; uint32_t n_ov = 0;
        if (ovf_start) { o0 = ovf_start[r]; n_ov = (uint32_t)(ovf_start[r + 1] - o0); }
        const int32_t nb = mtb_num_buckets(qlen[r] + qlen2[r], dna_shift);
        const bool skip = (off_reads && off_reads[r]) || cur - tail_n != n_ov || stride + n_ov > 65535u || nb > 256;      /* (256: the position buckets of the scorer that follows, k_score_long<.., 256>) */
        __syncthreads();                                  /* the previous read is through with the LDS arrays */
        if (only_flag && tid == 0) todo[r] = 0;
        if (skip) { if (tid == 0) { seg_cnt[b] = 0; todo[r] = 1; } continue; }
        for (uint32_t q = tid; q < HASH; q += MTB_MSORT_NT) { h_key[q] = 0xFFFFFFFFu; h_val[q] = 0u; }
        if (tid == 0) { s_n = 0; s_full = 0; }
        __syncthreads();
        const mtb_slot16 *slots = slots_all + r * (uint64_t)stride;
        const uint32_t n_src = stride + n_ov;
        /* record `i` of the read's sources: slot i, or overflow entry i - stride.  -> (live, species, frame) */
        auto probe = [&](uint32_t i, uint32_t *sp_, uint32_t *fr_) -> bool {
            if (i < stride) { const mtb_slot16 x = slots[i]; *sp_ = (uint32_t)(x.a >> 32); *fr_ = (uint32_t)(x.b >> 52) & 7u; return seg_slot_live(x, i, direct, tail_n, epoch); }
            const mtb_match m = ovfg[o0 + (i - stride)]; *sp_ = (uint32_t)m.species_id; *fr_ = mtb_q_frame(m.qinfo); return true;
        };
        for (uint32_t i = tid; i < n_src; i += MTB_MSORT_NT) {
            uint32_t s_, f_;
            if (!probe(i, &s_, &f_)) continue;
            if (*(volatile uint32_t *)&s_full) break;                     /* the table has filled up: the read is handed on, nobody keeps probing */
            uint32_t h = (s_ * 0x9E3779B1u) >> (32 - LOG2HASH); bool done = false;
            for (uint32_t p = 0; p < HASH && !done; p++) {
                const uint32_t old = atomicCAS(&h_key[h], 0xFFFFFFFFu, s_);
                if (old == 0xFFFFFFFFu || old == s_) { const uint32_t bit = 1u << f_; if (atomicOr(&h_val[h], bit) & bit) atomicOr(&h_val[h], bit << 8); done = true; }
                h = (h + 1u) & (HASH - 1u);
            }
            if (!done) s_full = 1;
        }
        __syncthreads();
        uint32_t used = 0;
        for (uint32_t q = tid; q < HASH; q += MTB_MSORT_NT) used += h_key[q] != 0xFFFFFFFFu ? 1u : 0u;
        if (used) atomicAdd(&s_n, used);
        __syncthreads();
        const bool over = s_full || s_n > HASH / 4u * 3u;
        __syncthreads();
        if (over) { if (tid == 0) { seg_cnt[b] = 0; todo[r] = (uint8_t)beyond_flag; } continue; }
        if (tid == 0) s_n = 0;
        __syncthreads();
        /* pass 2: survivors' source indices (whole wave steps: one LDS atomic per step) */
        for (uint32_t i0 = 0; i0 < n_src; i0 += MTB_MSORT_NT) {
            const uint32_t i = i0 + tid;
            bool keep = false;
            if (i < n_src) {
                uint32_t s_, f_;
                if (probe(i, &s_, &f_)) {
                    uint32_t h = (s_ * 0x9E3779B1u) >> (32 - LOG2HASH);
                    for (uint32_t p = 0; p < HASH; p++) { const uint32_t k = h_key[h]; if (k == s_) { keep = (h_val[h] >> 8) != 0u; break; } if (k == 0xFFFFFFFFu) break; h = (h + 1u) & (HASH - 1u); }
                }
            }
            const uint64_t km = __ballot(keep);
            if (km) {
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_n, (uint32_t)__popcll(km));
                at0 = (uint32_t)__shfl((int)at0, 0, 64);
                if (keep) { const uint32_t at = at0 + (uint32_t)__popcll(km & lt); if (at < MAXN) s_idx[at] = (uint16_t)i; }
            }
        }
        __syncthreads();
        const uint32_t n = s_n;
        __syncthreads();
        if (n > MAXN) { if (tid == 0) { seg_cnt[b] = 0; todo[r] = (uint8_t)beyond_flag; } continue; }
        /* pass 3: keys (the table is dead) */
        auto fetch = [&](uint32_t i) -> mtb_match {
            if (i < stride) return mtb_slot_unpack(slots[i], (uint32_t)r + 1);
            mtb_match m = ovfg[o0 + (i - stride)]; m.pad = 0; return m;
        };
        for (uint32_t j = tid; j < n; j += MTB_MSORT_NT) {
            const mtb_match m = fetch(s_idx[j]);
            s_u[j] = ((uint64_t)(uint32_t)m.species_id << 41) | ((uint64_t)mtb_q_frame(m.qinfo) << 38) | ((uint64_t)(mtb_q_pos(m.qinfo) & 0x7FFu) << 27) |
                     ((uint64_t)(m.hamming & 7u) << 24) | (m.dna & 0xFFFFFFu);
        }
        __syncthreads();
        so_bitonic(s_u, s_idx, n, tid);
        /* pass 4: the records in order */
        mtb_match *dst = big + big_start[b];
        for (uint32_t j = tid; j < n; j += MTB_MSORT_NT) {
            const mtb_match m = fetch(s_idx[j]);
            uint64_t *d = (uint64_t *)(dst + j); const uint64_t *q = (const uint64_t *)&m;
            d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
        }
        if (tid == 0) seg_cnt[b] = n;
    }
}
/* after k_score_long over the listed reads: the flagged ones (budgets exceeded in k_many_sort or k_score_long) -> the list of the
 * exact-segment path; the others' match counts (all records, dropped ones included) -> cnt_out (statistics) */
__global__ __launch_bounds__(256) void k_list_flagged(const uint32_t *__restrict__ list, uint32_t n_list, const uint8_t *__restrict__ todo, const uint32_t *__restrict__ big_cnt,
                                                       uint32_t *__restrict__ rest, uint32_t *__restrict__ n_rest, uint32_t *__restrict__ cnt_out) {
    const uint32_t b = blockIdx.x * 256 + threadIdx.x;
    if (b >= n_list) return;
    const uint32_t r = list[b];
    if (todo[r]) rest[atomicAdd(n_rest, 1u)] = r;
    else cnt_out[r] = big_cnt[b];
}

template <bool KEY64, int CAP>
__global__ __launch_bounds__(64, (CAP > 192 ? 2 : 3)) void k_score_many(const mtb_slot16 *__restrict__ slots_all, uint32_t stride, uint32_t direct, uint32_t epoch,
                                                       const uint32_t *__restrict__ cursor, const uint8_t *__restrict__ off_reads,
                                                       const mtb_match *__restrict__ ovfg, const uint64_t *__restrict__ ovf_start,
                                                       const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list,
                                                       const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2, mtb_tax_view tx, mtb_score_params sp,
                                                       const uint64_t *__restrict__ tc_off, mtb_result *__restrict__ results, int32_t *__restrict__ tc_tax,
                                                       uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, uint64_t tc_base,
                                                       uint32_t *__restrict__ rest_list, uint32_t *__restrict__ n_rest, uint32_t *__restrict__ cnt_out,
                                                       unsigned long long *__restrict__ work, unsigned long long *__restrict__ stats /* [0] matches seen, [1] survivors; reads handed on: [2] routed off / too many buckets / inconsistent counts, [3] species table full, [4] survivors beyond the staging */) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ws[MTB_SCORE_WS_BYTES_(CAP)];
    static_assert(MTB_SCORE_WS_BYTES_(CAP) - CAP * sizeof(mtb_match) >= MTB_MANY_HASH * 8, "the species table fits behind the match records");
    static_assert(CAP * sizeof(mtb_path) >= MTB_SCORE_BKT * 13 + (MTB_LR_MAXE + MTB_LR_MAXE * MTB_LR_K) * 4, "decide arrays must fit the path area");
    uint32_t *const h_key = (uint32_t *)(s_ws + CAP * sizeof(mtb_match));
    uint32_t *const h_val = h_key + MTB_MANY_HASH;
    __shared__ uint32_t s_tfull;                 /* the species table of the read in work has filled up: nobody keeps probing it (every further record walked all its entries) */
    const uint32_t lane = threadIdx.x;
    const uint64_t lt = lanemask_lt();
    const uint32_t tail_cap = stride - direct;
    MTB_BEGIN_ACQUIRE();
    const uint64_t n_iter = (uint64_t)*n_list;
    unsigned long long seen = 0, kept = 0;
    /* reads differ a lot (a few dozen to a thousand records): claimed from a counter, MTB_MANY_CLAIM list entries per atomic -- one
     * returning atomic per read on ONE address was ~19 ns each once 3000 waves hammer it: 15 of the kernel's 30 ms for 0.8 M reads */
    for (uint64_t c0 = (uint64_t)__shfl(lane == 0 ? atomicAdd(work, (unsigned long long)MTB_MANY_CLAIM) : 0ull, 0, 64); c0 < n_iter;
         c0 = (uint64_t)__shfl(lane == 0 ? atomicAdd(work, (unsigned long long)MTB_MANY_CLAIM) : 0ull, 0, 64))
    for (uint64_t it = c0; it < c0 + MTB_MANY_CLAIM && it < n_iter; it++) {
        const uint64_t r = (uint64_t)list[it];
        const int32_t ql1 = qlen[r], ql2 = qlen2[r];
        const int32_t read_len = ql1 + ql2;
        const int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        const uint32_t cur = cursor[r];
        const uint32_t tail_n = cur < tail_cap ? cur : tail_cap;
        uint64_t o0 = 0; uint32_t n_ov = 0;
        if (ovf_start) { o0 = ovf_start[r]; n_ov = (uint32_t)(ovf_start[r + 1] - o0); }
        bool hand_on = (off_reads && off_reads[r]) || nb > MTB_SCORE_BKT || cur - tail_n != n_ov;
        uint32_t why = 2;
        /* a read with more than three times the staging in its tail + overflow entries alone (a conserved gene of an organism that is NOT in the
         * index: ~1100 records over a thousand species, most of them with a pair of matches -- the dead-species drop barely shrinks it) would
         * fail the staging after both passes over its records: handed on at once (the next tier, k_many_sort, takes any read within ITS budgets;
         * which tier scores a read is a question of time, not of the result).  12 ms of this kernel's 12 on 2 M such reads were these passes. */
        if (!hand_on && cur > 3u * (uint32_t)CAP) { hand_on = true; why = 4; }
        const mtb_slot16 *slots = slots_all + r * (uint64_t)stride;
        mtb_sws<uint16_t> w;
        mtb_sws_carve<uint16_t>(&w, s_ws, CAP);
        uint32_t n = 0, n_all = 0;
        if (!hand_on) {
            score_sync<uint16_t>();                      /* the previous read is through with the workspace */
            for (uint32_t q = lane; q < MTB_MANY_HASH; q += 64) { h_key[q] = 0xFFFFFFFFu; h_val[q] = 0u; }
            if (lane == 0) s_tfull = 0;
            score_sync<uint16_t>();
            bool full = false;
            /* species `s` (never 0xFFFFFFFF: ids are < 2^31), frame f: find or claim the entry, mark the frame */
            auto enter = [&](uint32_t s, uint32_t f) {
                if (*(volatile uint32_t *)&s_tfull) { full = true; return; }
                uint32_t h = (s * 0x9E3779B1u) >> 22;
                for (uint32_t p = 0; p < MTB_MANY_HASH; p++) {
                    const uint32_t old = atomicCAS(&h_key[h], 0xFFFFFFFFu, s);
                    if (old == 0xFFFFFFFFu || old == s) {
                        const uint32_t bit = 1u << f;
                        if (atomicOr(&h_val[h], bit) & bit) atomicOr(&h_val[h], bit << 8);
                        return;
                    }
                    h = (h + 1u) & (MTB_MANY_HASH - 1u);
                }
                full = true; s_tfull = 1;
            };
            for (uint32_t c0 = 0; c0 < stride; c0 += 64) {
                const uint32_t i = c0 + lane;
                mtb_slot16 x; x.a = 0; x.b = 0;
                if (i < stride) x = slots[i];
                const bool live = i < stride && seg_slot_live(x, i, direct, tail_n, epoch);
                if (live) enter((uint32_t)(x.a >> 32), (uint32_t)(x.b >> 52) & 7u);
                n_all += (uint32_t)__popcll(__ballot(live));
            }
            for (uint32_t c0 = 0; c0 < n_ov; c0 += 64) {
                const uint32_t i = c0 + lane;
                if (i < n_ov) { const mtb_match m = ovfg[o0 + i]; enter((uint32_t)m.species_id, mtb_q_frame(m.qinfo)); }
            }
            n_all += n_ov;
            score_sync<uint16_t>();
            /* occupancy: a table beyond 3/4 is handed on (probe chains; and `full` must never have been hit) */
            uint32_t used = 0;
            for (uint32_t q = lane; q < MTB_MANY_HASH; q += 64) used += h_key[q] != 0xFFFFFFFFu ? 1u : 0u;
            for (int d = 32; d > 0; d >>= 1) used += (uint32_t)__shfl_xor((int)used, d, 64);
            hand_on = __any(full) || used > MTB_MANY_HASH / 4u * 3u;
            why = 3;
            if (!hand_on) {
                auto alive = [&](uint32_t s) -> bool {
                    uint32_t h = (s * 0x9E3779B1u) >> 22;
                    for (uint32_t p = 0; p < MTB_MANY_HASH; p++) {
                        const uint32_t k = h_key[h];
                        if (k == s) return (h_val[h] >> 8) != 0u;
                        if (k == 0xFFFFFFFFu) return false;          /* (cannot happen: every record was entered) */
                        h = (h + 1u) & (MTB_MANY_HASH - 1u);
                    }
                    return false;
                };
                uint64_t *dst64 = (uint64_t *)w.m;
                auto put = [&](uint32_t pos, const mtb_match &m) {
                    const uint64_t *q = (const uint64_t *)&m;
                    dst64[3 * pos] = q[0]; dst64[3 * pos + 1] = q[1]; dst64[3 * pos + 2] = q[2];
                };
                for (uint32_t c0 = 0; c0 < stride; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    mtb_slot16 x; x.a = 0; x.b = 0;
                    if (i < stride) x = slots[i];
                    const bool keep = i < stride && seg_slot_live(x, i, direct, tail_n, epoch) && alive((uint32_t)(x.a >> 32));
                    const uint64_t mask = __ballot(keep);
                    const uint32_t pos = n + (uint32_t)__popcll(mask & lt);
                    if (keep && pos < (uint32_t)CAP) put(pos, mtb_slot_unpack(x, (uint32_t)r + 1));
                    n += (uint32_t)__popcll(mask);
                }
                for (uint32_t c0 = 0; c0 < n_ov; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    mtb_match m; m.qinfo = 0; m.target_id = 0; m.species_id = 0; m.dna = 0; m.right_end_hamming = 0; m.hamming = 0; m.pad = 0;
                    bool keep = false;
                    if (i < n_ov) { m = ovfg[o0 + i]; keep = alive((uint32_t)m.species_id); }
                    const uint64_t mask = __ballot(keep);
                    const uint32_t pos = n + (uint32_t)__popcll(mask & lt);
                    if (keep && pos < (uint32_t)CAP) { m.pad = 0; put(pos, m); }
                    n += (uint32_t)__popcll(mask);
                }
                hand_on = n > (uint32_t)CAP;
                why = 4;
                score_sync<uint16_t>();                  /* the table is dead from here on: the workspace behind the records is the scorer's */
            }
        }
        if (hand_on) { if (lane == 0) { rest_list[atomicAdd(n_rest, 1u)] = (uint32_t)r; if (stats) atomicAdd(&stats[why], 1ull); } continue; }
        seen += n_all; kept += n;
        mtb_result R;
        R.classification = 0; R.score = 0.0f; R.query_length = ql1; R.query_length2 = ql2;
        R.is_classified = 0; R.reserved = 0; R.n_taxcnt = 0; R.taxcnt_off = 0;
        if (lane == 0) cnt_out[r] = n_all;
        if (n == 0) { if (lane == 0) results[r] = R; continue; }
        const uint64_t off = tc_off[r], room = tc_off[r + 1] - off;
        int32_t *s_btax = (int32_t *)w.path, *s_otax = s_btax + MTB_SCORE_BKT;
        uint32_t *s_ocnt = (uint32_t *)(s_otax + MTB_SCORE_BKT);
        int32_t *s_lev = (int32_t *)(s_ocnt + MTB_SCORE_BKT), *s_anc = s_lev + MTB_LR_MAXE;
        uint8_t *s_bham = (uint8_t *)(s_anc + MTB_LR_MAXE * MTB_LR_K);
        score_read_par<uint16_t, true, KEY64, mtb_match, true, CAP>(w.m, (int32_t)n, w, s_btax, s_bham, s_otax, s_ocnt, s_lev, s_anc, nb, read_len, tx, sp, off, room,
                                                                   tc_tax, tc_cnt, tc_cap, (mtb_match *)nullptr, R);
        if (lane == 0) { R.query_length = ql1; R.query_length2 = ql2; R.reserved = 0; R.taxcnt_off += (uint32_t)tc_base; results[r] = R; }
    }
    if (stats && lane == 0 && seen) { atomicAdd(&stats[0], seen); atomicAdd(&stats[1], kept); }
}

#endif
