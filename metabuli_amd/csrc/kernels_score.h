/* kernels_score.h -- per-read match extension and scoring, one wavefront per
 * read (Classifier::assignTaxonomy -> Taxonomer::chooseBestTaxon,
 * src/commons/Classifier.cpp:166-208, src/commons/Taxonomer.cpp:130-699).
 *
 * The read's matches are staged in LDS (slot mode: the live slots of its segment, compacted in slot order; other
 * callers: its exact segment; segments beyond the LDS capacity work out of a per-workgroup slab in HBM).  The
 * wave then runs the phases of mtb_score_par.h, every lane on its own matches / paths / species:
 *   order    nothing if the staged order already is compareMatches' (usual on the slot path), a two-run merge for
 *            read pairs, otherwise a rank sort on a 64-bit key
 *   paths    head flags + ballot prefix sums -> position groups, (species, frame) blocks, species; links; chain DP
 *            (getMatchPaths) as one prefix sum, by pointer doubling, or in rounds, depending on the link structure
 *   combine  per species: parallel stable rank of its paths, parallel certain-drop test, short serial greedy pass
 *            (combineMatchPaths)
 *   decide   lane 0: best species / ties -> LCA; match-parallel redundancy filter; wave-parallel taxCnt gather;
 *            sub-species descent on pre-climbed chains
 * Algorithmic HBM bytes: 24 per match read + 24 per read result written.     */
#ifndef MTB_KERNELS_SCORE_H
#define MTB_KERNELS_SCORE_H
#include "dev_util.h"
#include "mtb_core.h"
#include "kernels_join.h"
#include "mtb_score_par.h"

/* profiling build only (-DMTB_SCORE_PHASE_CYCLES): cycles per phase of k_score,
 * accumulated into mtb_phase_cycles[16]: 0 load+keys, 1 rank, 2 permute, 3 flags+ids, 4 starts, 5 links, 6 chain DP,
 * 7 emit, 8 combine, 9 select, 10 filter, 11 gather, 12 climb, 13 decide+output */
#ifdef MTB_SCORE_PHASE_CYCLES
#define MTB_NPHASE 16
__device__ unsigned long long mtb_phase_cycles[MTB_NPHASE];
__shared__ unsigned long long mtb_phase_lds[MTB_NPHASE];      /* per-workgroup accumulators, flushed once at kernel end */
#define MTB_PHASE_BEGIN() unsigned long long ph_t0_ = __builtin_readcyclecounter()
#define MTB_PHASE_MARK(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) mtb_phase_lds[k] += t_ - ph_t0_; ph_t0_ = t_; } while (0)
#define MTB_PHASE_KERNEL_BEGIN() do { if (threadIdx.x < MTB_NPHASE) mtb_phase_lds[threadIdx.x] = 0; __syncthreads(); } while (0)
#define MTB_PHASE_KERNEL_END() do { __syncthreads(); if (threadIdx.x < MTB_NPHASE) atomicAdd(&mtb_phase_cycles[threadIdx.x], mtb_phase_lds[threadIdx.x]); } while (0)
#else
#define MTB_PHASE_BEGIN() do {} while (0)
#define MTB_PHASE_MARK(k) do {} while (0)
#define MTB_PHASE_KERNEL_BEGIN() do {} while (0)
#define MTB_PHASE_KERNEL_END() do {} while (0)
#endif

#ifndef MTB_SCORE_LDS
#define MTB_SCORE_LDS 160        /* matches per read staged in LDS (10.8 KB/wave -> 14 waves/CU; sweep in profiles/r01_notes.md) */
#endif
#ifndef MTB_SCORE_MINWAVES
#define MTB_SCORE_MINWAVES 4      /* <= 128 VGPRs: occupancy beats the ~90 B/lane of spills (measured) */
#endif
#define MTB_SCORE_BKT 128        /* position buckets / taxCnt entries in LDS  */

/* bytes of slab one workgroup needs for a segment of n matches, nb buckets */
__host__ __device__ __forceinline__ uint64_t score_slab_bytes(uint64_t n, uint64_t nb) {
    uint64_t b = mtb_sws_bytes<uint32_t>(n);
    b += nb * 12 + (MTB_LR_MAXE + MTB_LR_MAXE * MTB_LR_K) * 4 + ((nb + 15) & ~15ull);   /* btax, otax, ocnt, lev, anc, bham */
    return (b + 63) & ~63ull;
}

__global__ __launch_bounds__(64) void k_taxcnt_bound(const uint64_t *__restrict__ seg_start, const uint32_t *__restrict__ cursor,
                                                      const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2, uint64_t n_reads,
                                                      int32_t dna_shift, uint32_t *__restrict__ bound) {
    uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t nb = (uint64_t)mtb_num_buckets(qlen[r] + qlen2[r], dna_shift);
    if (cursor) { bound[r] = (uint32_t)nb; return; }           /* slot mode: the read's match count is not known yet */
    uint64_t n = seg_start[r + 1] - seg_start[r];
    bound[r] = (uint32_t)(n < nb ? n : nb);
}

/* number of bytes equal to 1 (reads k_score_fast / k_score_long left to the generic kernel; statistics).  Flag 2 -- a short read whose tail
 * overflowed: it goes to the deferred path (k_score_many ...) and is counted there, not here */
__global__ __launch_bounds__(256) void k_count_flags(const uint8_t *__restrict__ f, uint64_t n, unsigned long long *__restrict__ out) {
    uint32_t c = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256) c += f[i] == 1 ? 1u : 0u;
    for (int d = 32; d > 0; d >>= 1) c += __shfl_down(c, d, 64);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(out, (unsigned long long)c);
}

/* reads k_score_fast flagged 2 (tail overflow) -> the list of the deferred path: one atomic per workgroup of 256 reads */
__global__ __launch_bounds__(256) void k_list_flag2(const uint8_t *__restrict__ f, uint64_t n, uint32_t *__restrict__ list, uint32_t *__restrict__ n_list) {
    __shared__ uint32_t s_w[4]; __shared__ uint32_t s_base;
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const bool on = r < n && f[r] == 2;
    const uint64_t m = __ballot(on);
    const uint32_t wv = threadIdx.x >> 6, lane = threadIdx.x & 63u;
    if (lane == 0) s_w[wv] = (uint32_t)__popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) { const uint32_t tot = s_w[0] + s_w[1] + s_w[2] + s_w[3]; s_base = tot ? atomicAdd(n_list, tot) : 0u; }
    __syncthreads();
    if (on) {
        uint32_t at = s_base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
        for (uint32_t q = 0; q < wv; q++) at += s_w[q];
        list[at] = (uint32_t)r;
    }
}

/* list the reads whose segment does not fit the LDS staging of k_score; they
 * are sorted in HBM by k_segsort_large and scored out of per-workgroup slabs */
__global__ __launch_bounds__(256) void k_list_large(const uint64_t *__restrict__ seg_start, uint64_t n_reads, uint32_t threshold,
                                                     uint32_t *__restrict__ large, uint32_t *__restrict__ n_large, uint32_t *__restrict__ max_seg) {
    uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint32_t n = 0;
    if (r < n_reads) {
        n = (uint32_t)(seg_start[r + 1] - seg_start[r]);
        if (n > threshold) large[atomicAdd(n_large, 1u)] = (uint32_t)r;
    }
    for (int d = 32; d > 0; d >>= 1) { uint32_t o = __shfl_down(n, d, 64); n = o > n ? o : n; }
    if ((threadIdx.x & 63) == 0 && n > threshold) atomicMax(max_seg, n);
}

/* Query::taxCnt from the bucket taxa, wave-parallel (same result as mtb_taxcnt_gather: ascending taxid, one
 * entry per distinct taxon of the used buckets): every round extracts the smallest taxon not yet emitted (lane-
 * strided minimum + wave reduction) and counts its buckets.  Rounds = distinct taxa (1-3 for most reads), against
 * one serial insertion per bucket on lane 0 before.  The capacity is never the limit: cap = min(matches, buckets)
 * >= used buckets >= distinct taxa (k_taxcnt_bound).                                                          */
__device__ __forceinline__ int32_t taxcnt_gather_wave(const int32_t *btax, const uint8_t *bham, int32_t nb, int32_t *otax, uint32_t *ocnt, int32_t cap) {
    const int32_t lane = (int32_t)threadIdx.x;
    int32_t ntc = 0;
    int32_t last = -1;                               /* taxids are non-negative (info & mask, canonical LCA ids) */
    while (ntc < cap) {
        int32_t mn = INT32_MAX;
        for (int32_t q = lane; q < nb; q += 64) if (bham[q] != 255) { int32_t t = btax[q]; if (t > last && t < mn) mn = t; }
        for (int d = 32; d > 0; d >>= 1) { int32_t o = __shfl_xor(mn, d, 64); mn = o < mn ? o : mn; }
        if (mn == INT32_MAX) break;
        uint32_t cnt = 0;
        for (int32_t q0 = 0; q0 < nb; q0 += 64) { int32_t q = q0 + lane; cnt += (uint32_t)__popcll(__ballot(q < nb && bham[q] != 255 && btax[q] == mn)); }
        if (lane == 0) { otax[ntc] = mn; ocnt[ntc] = cnt; }
        ntc++; last = mn;
    }
    return ntc;
}

/* Phase boundary inside one wavefront.  A k_score workgroup is ONE wave: its LDS instructions are executed in issue
 * order, so a phase that wrote LDS and the next one that reads it need no s_barrier and no s_waitcnt vmcnt(0) (which
 * __syncthreads() implies and which also drains every outstanding global load) -- only a compiler fence.  The HBM
 * slab workspace of big segments (IDX = uint32_t) keeps the full barrier: global stores followed by loads of other
 * lanes need the wait.                                                                                         */
template <typename IDX>
__device__ __forceinline__ void score_sync() {
    if (sizeof(IDX) == 2) { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }
    else __syncthreads();
}

/* One read, data-parallel (phases of mtb_score_par.h).  All storage of a call
 * site lives in ONE address space (LDS or an HBM slab) so that the compiler
 * emits ds_* / global_* instead of flat accesses.  SORT: the segment arrives
 * unordered and holds at most 64*MAXPER matches: rank sort on 12-byte keys. */
template <typename IDX, bool SORT, bool KEY64, typename REC, bool INPLACE = false, int CAP = MTB_SCORE_LDS>
__device__ __forceinline__ void score_read_par(const REC *src, int32_t n, mtb_sws<IDX> w, int32_t *btax,
                                               uint8_t *bham, int32_t *otax, uint32_t *ocnt, int32_t *lr_lev, int32_t *lr_anc,
                                               int32_t nb, int32_t read_len, const mtb_tax_view &tx, const mtb_score_params &sp,
                                               uint64_t tc_off, uint64_t tc_room, int32_t *__restrict__ tc_tax,
                                               uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, mtb_match *__restrict__ sorted_out,
                                               mtb_result &R) {
    constexpr int MAXPER = (CAP + 63) / 64;          /* register slots per lane that cover a staged segment */
    if (sizeof(IDX) == 2) __builtin_assume(n >= 1 && n <= CAP);     /* LDS workspace: lets the compiler drop the big-segment variants */
    const int32_t lane = (int32_t)threadIdx.x;
    const uint64_t lt = lanemask_lt();
    w.n = n;
    MTB_PHASE_BEGIN();
    if (SORT) {
        /* rank sort: keys in the (not yet live) path area, records in registers.
         * KEY64 (host-checked: taxids < 2^22, positions < 2^11): the whole
         * compareMatches key fits one 64-bit word. */
        uint64_t *k1 = (uint64_t *)w.path; uint32_t *k2 = (uint32_t *)(k1 + n);
        mtb_match rec[MAXPER]; uint64_t a1[MAXPER]; uint32_t a2[MAXPER]; int32_t rank[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            int32_t i = lane + 64 * k;
            rank[k] = 0; a1[k] = 0; a2[k] = 0;
            if (i < n) {
                rec[k] = rec_m(src[i]);
                if (KEY64) {
                    a1[k] = ((uint64_t)(uint32_t)rec[k].species_id << 41) | ((uint64_t)mtb_q_frame(rec[k].qinfo) << 38) |
                            ((uint64_t)(mtb_q_pos(rec[k].qinfo) & 0x7FFu) << 27) | ((uint64_t)(rec[k].hamming & 7u) << 24) | (rec[k].dna & 0xFFFFFFu);
                    k1[i] = a1[k];
                } else { a1[k] = mtb_key1(rec[k]); a2[k] = mtb_key2(rec[k]); k1[i] = a1[k]; k2[i] = a2[k]; }
            }
        }
        score_sync<IDX>();
        MTB_PHASE_MARK(0);
        const int32_t nslot = (n + 63) >> 6;          /* live register slots (wave-uniform) */
        /* already in order?  (slot mode: one match per query and one species = extraction order = compareMatches order).
         * Exactly one descent = two ordered runs (the two mates of a pair, whose slots are mate-major while the order is
         * frame-major): merge by rank, one binary search in the other run per record. */
        uint32_t n_desc = 0; int32_t split = 0;
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            int32_t i = lane + 64 * k;
            bool desc = false;
            if (i > 0 && i < n) {
                uint64_t p1 = k1[i - 1];
                if (KEY64) desc = p1 > a1[k];
                else { uint32_t p2 = k2[i - 1]; desc = p1 > a1[k] || (p1 == a1[k] && p2 > a2[k]); }
            }
            const uint64_t md = __ballot(desc);
            if (md) { if (n_desc == 0) split = 64 * k + (int32_t)__builtin_ctzll(md); n_desc += (uint32_t)__popcll(md); }
        }
        const bool sorted = n_desc == 0;
        bool merged = false;
        bool unique = true;
        if (sorted) {
#pragma unroll
            for (int k = 0; k < MAXPER; k++) rank[k] = lane + 64 * k;
        } else if (n_desc == 1) {
            merged = true;
#pragma unroll
            for (int k = 0; k < MAXPER; k++) {
                int32_t i = lane + 64 * k;
                if (i < n) {
                    /* first run [0, split): + elements of the second run strictly smaller; second run: + elements of the first run <= */
                    const bool first = i < split;
                    int32_t lo = first ? split : 0, hi = first ? n : split;
                    const int32_t base0 = lo;
                    while (lo < hi) {
                        const int32_t mid = (lo + hi) >> 1;
                        const uint64_t b1 = k1[mid];
                        bool less;                      /* key[mid] < key_i (first) or key[mid] <= key_i (second) */
                        if (KEY64) less = first ? b1 < a1[k] : b1 <= a1[k];
                        else { const uint32_t b2 = k2[mid]; less = b1 < a1[k] || (b1 == a1[k] && (first ? b2 < a2[k] : b2 <= a2[k])); }
                        if (less) lo = mid + 1; else hi = mid;
                    }
                    rank[k] = (first ? i : i - split) + (lo - base0);
                }
            }
        } else
        if (KEY64) {
            /* fast path: rank = number of strictly smaller keys (2 VALU per comparison).  Keys are
             * distinct unless the index holds duplicate entries; equal keys are detected below. */
            int32_t j = 0;
            for (; j + 4 <= n; j += 4) {
                uint64_t b1[4];
#pragma unroll
                for (int u = 0; u < 4; u++) b1[u] = k1[j + u];
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int k = 0; k < MAXPER; k++) if (k < nslot) rank[k] += (b1[u] < a1[k]);
                }
            }
            for (; j < n; j++) {
                uint64_t b1 = k1[j];
#pragma unroll
                for (int k = 0; k < MAXPER; k++) rank[k] += (b1 < a1[k]);
            }
            /* two equal keys get the same rank: then some key differs from the one stored at its rank */
            score_sync<IDX>();
            uint64_t *chk = (uint64_t *)w.m;          /* m[] is not written yet */
#pragma unroll
            for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) chk[rank[k]] = (uint64_t)i; }
            score_sync<IDX>();
#pragma unroll
            for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n && chk[rank[k]] != (uint64_t)i) unique = false; }
            unique = __all(unique);
        }
        if (!sorted && !merged && (!KEY64 || !unique)) {
#pragma unroll
            for (int k = 0; k < MAXPER; k++) rank[k] = 0;
            int32_t j = 0;
            for (; j + 4 <= n; j += 4) {                  /* 4 independent LDS reads in flight */
                uint64_t b1[4]; uint32_t b2[4];
#pragma unroll
                for (int u = 0; u < 4; u++) { b1[u] = k1[j + u]; b2[u] = KEY64 ? 0u : k2[j + u]; }
#pragma unroll
                for (int u = 0; u < 4; u++) {
#pragma unroll
                    for (int k = 0; k < MAXPER; k++) {
                        if (k < nslot) {
                            int32_t i = lane + 64 * k;
                            if (KEY64) rank[k] += (b1[u] < a1[k]) || (b1[u] == a1[k] && (j + u) < i);
                            else rank[k] += (b1[u] < a1[k]) || (b1[u] == a1[k] && (b2[u] < a2[k] || (b2[u] == a2[k] && (j + u) < i)));
                        }
                    }
                }
            }
            for (; j < n; j++) {
                uint64_t b1 = k1[j]; uint32_t b2 = KEY64 ? 0u : k2[j];
#pragma unroll
                for (int k = 0; k < MAXPER; k++) {
                    int32_t i = lane + 64 * k;
                    if (KEY64) rank[k] += (b1 < a1[k]) || (b1 == a1[k] && j < i);
                    else rank[k] += (b1 < a1[k]) || (b1 == a1[k] && (b2 < a2[k] || (b2 == a2[k] && j < i)));
                }
            }
        }
        score_sync<IDX>();
        MTB_PHASE_MARK(1);
        if (!(INPLACE && sorted)) {
#pragma unroll
            for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) w.m[rank[k]] = rec[k]; }
        }
        score_sync<IDX>();
        if (sorted_out) {
            const uint64_t *s64 = (const uint64_t *)w.m; uint64_t *d64 = (uint64_t *)sorted_out;
            for (int32_t i = lane; i < n * 3; i += 64) d64[i] = s64[i];
        }
    } else {
        for (int32_t i = lane; i < n; i += 64) w.m[i] = rec_m(src[i]);
        score_sync<IDX>();
    }
    MTB_PHASE_MARK(2);
    /* flags, ids */
    for (int32_t i = lane; i < n; i += 64) mtb_ph_flags(w, i);
    score_sync<IDX>();
    int32_t ng = 0, nbk = 0, nsp = 0;
    for (int32_t c0 = 0; c0 < n; c0 += 64) {
        int32_t i = c0 + lane;
        uint32_t f = (i < n) ? w.flag[i] : 0u;
        uint64_t mg = __ballot(f & MTB_F_GHEAD), mb = __ballot(f & MTB_F_BHEAD), ms = __ballot(f & MTB_F_SHEAD);
        if (i < n) {
            w.gid[i] = (IDX)(ng + __popcll(mg & lt) + ((f & MTB_F_GHEAD) ? 1 : 0) - 1);
            w.bid[i] = (IDX)(nbk + __popcll(mb & lt) + ((f & MTB_F_BHEAD) ? 1 : 0) - 1);
            w.sid[i] = (IDX)(nsp + __popcll(ms & lt) + ((f & MTB_F_SHEAD) ? 1 : 0) - 1);
        }
        ng += __popcll(mg); nbk += __popcll(mb); nsp += __popcll(ms);
    }
    score_sync<IDX>();
    MTB_PHASE_MARK(3);
    for (int32_t i = lane; i < n; i += 64) mtb_ph_starts(w, i, &tx);
    score_sync<IDX>();
    MTB_PHASE_MARK(4);
    int32_t maxrank = 0;
    if (ng == n) {      /* every position group is one match (the usual read): neighbours are the adjacent slots */
        for (int32_t i = lane; i < n; i += 64) { mtb_ph_links_unit(w, i, &sp); int32_t rr = w.rk[i]; maxrank = rr > maxrank ? rr : maxrank; }
    } else
    for (int32_t i = lane; i < n; i += 64) { mtb_ph_links(w, i, &tx, &sp, ng, nbk); int32_t rr = w.rk[i]; maxrank = rr > maxrank ? rr : maxrank; }
    for (int d = 32; d > 0; d >>= 1) { int32_t o = __shfl_xor(maxrank, d, 64); maxrank = o > maxrank ? o : maxrank; }
    score_sync<IDX>();
    MTB_PHASE_MARK(5);
    /* chain DP: pointer doubling when every match has <= 1 consecutive predecessor, else rounds */
    const bool small_n = n <= 64 * MAXPER;
    bool simple = small_n || sizeof(IDX) == 4;      /* big segments need the 32-bit workspace for the ping-pong array */
    if (simple) {
        for (int32_t i = lane; i < n; i += 64) simple = simple && mtb_chain_simple(w, i);
        simple = __all(simple);
    }
    /* every link goes to the slot just before (position groups of one match, the usual case): the chains are
     * contiguous runs and V[i] = P[i] - P[root(i)] with P = ONE inclusive prefix sum of the per-link increments over
     * the slots (in registers, 6 shuffle steps) -- no doubling rounds.  Scores are multiples of 0.5, hamming / depth
     * small integers: the sums are exact in any association, so the paths are bit-identical. */
    bool adjacent = simple && small_n && sizeof(IDX) == 2;
    if (adjacent) {
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            const int32_t i = lane + 64 * k;
            if (i < n) { const uint32_t sh = w.shift[i], cm = w.cmask[i]; if (sh && cm) adjacent = adjacent && ((int32_t)w.bid[i] + __builtin_ctz(cm) == i - 1); }
        }
        adjacent = __all(adjacent);
    }
    /* the same for big segments (HBM slab): chunked prefix sum over the slots, P[] and the chain roots kept in the (dead)
     * sid/rk/grp_start/blk_start block.  One pass over the segment instead of log2(longest chain) doubling rounds of
     * three slab accesses each (long reads: chains of ~1400 links, 41 % of the scoring time). */
    bool adjacent_big = simple && !small_n && sizeof(IDX) == 4;
    if (adjacent_big) {
        for (int32_t i = lane; i < n; i += 64) {
            const uint32_t sh = w.shift[i], cm = w.cmask[i];
            if (sh && cm) adjacent_big = adjacent_big && ((int32_t)w.bid[i] + __builtin_ctz(cm) == i - 1);
        }
        adjacent_big = __all(adjacent_big);
    }
    if (adjacent) {
        float ps[MAXPER]; int32_t phd[MAXPER]; int32_t root[MAXPER];
        float carry_s = 0.0f; int32_t carry_hd = 0, last_root = 0;
        const int32_t nslot = (n + 63) >> 6;
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            ps[k] = 0.0f; phd[k] = 0; root[k] = 0;
            if (k < nslot) {
                const int32_t i = lane + 64 * k;
                float is = 0.0f; int32_t ihd = 0; bool is_root = true;
                if (i < n) {
                    const uint32_t sh = w.shift[i], cm = w.cmask[i];
                    if (sh && cm) {
                        const int32_t shift = (int32_t)(sh & 0x7Fu);
                        const uint32_t reh = w.m[i].right_end_hamming;
                        is = mtb_part_score(reh, shift, false); ihd = (mtb_part_ham(reh, shift, false) << 16) | shift; is_root = false;
                    }
                }
                ps[k] = wave_inclusive_scan_dpp(is) + carry_s; phd[k] = wave_inclusive_scan_dpp(ihd) + carry_hd;
                carry_s = __shfl(ps[k], 63, 64); carry_hd = __shfl(phd[k], 63, 64);
                const uint64_t mr = __ballot(is_root && i < n);
                const uint64_t le = mr & (lt | (1ull << lane));
                root[k] = le ? 64 * k + 63 - (int32_t)__builtin_clzll(le) : last_root;
                if (mr) last_root = 64 * k + 63 - (int32_t)__builtin_clzll(mr);
            }
        }
        mtb_jump *pre = (mtb_jump *)w.path;              /* P[] where every lane can read it (the path storage is free until the end) */
#pragma unroll
        for (int k = 0; k < MAXPER; k++) { const int32_t i = lane + 64 * k; if (i < n) { mtb_jump j; j.ptr = 0; j.score = ps[k]; j.ham = phd[k]; j.depth = 0; pre[i] = j; } }
        score_sync<IDX>();
        mtb_jump fin[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            const int32_t i = lane + 64 * k;
            if (i < n) {
                const mtb_jump pr = pre[root[k]];
                const int32_t dhd = phd[k] - pr.ham;
                fin[k].ptr = root[k] == i ? -1 : root[k]; fin[k].score = ps[k] - pr.score; fin[k].ham = dhd >> 16; fin[k].depth = dhd & 0xFFFF;
            }
        }
        score_sync<IDX>();
#pragma unroll
        for (int k = 0; k < MAXPER; k++) { const int32_t i = lane + 64 * k; if (i < n) w.path[i] = mtb_ph_jump_final(w, i, fin[k]); }
        score_sync<IDX>();
    } else
    if (adjacent_big) {
        mtb_jump *pre = (mtb_jump *)w.sid;              /* 16 B x (cap + 1), as the ping-pong array of the doubling variant */
        float carry_s = 0.0f; int32_t carry_hd = 0, last_root = 0;
        for (int32_t c0 = 0; c0 < n; c0 += 64) {
            const int32_t i = c0 + lane;
            float is = 0.0f; int32_t ihd = 0; bool is_root = true;
            if (i < n) {
                const uint32_t sh = w.shift[i], cm = w.cmask[i];
                if (sh && cm) {
                    const int32_t shift = (int32_t)(sh & 0x7Fu);
                    const uint32_t reh = w.m[i].right_end_hamming;
                    is = mtb_part_score(reh, shift, false); ihd = (mtb_part_ham(reh, shift, false) << 16) | shift; is_root = false;
                }
            }
            const float ps = wave_inclusive_scan_dpp(is) + carry_s; const int32_t phd = wave_inclusive_scan_dpp(ihd) + carry_hd;
            carry_s = __shfl(ps, 63, 64); carry_hd = __shfl(phd, 63, 64);
            const uint64_t mr = __ballot(is_root && i < n);
            const uint64_t le = mr & (lt | (1ull << lane));
            const int32_t root = le ? c0 + 63 - (int32_t)__builtin_clzll(le) : last_root;
            if (mr) last_root = c0 + 63 - (int32_t)__builtin_clzll(mr);
            if (i < n) { mtb_jump j; j.ptr = root; j.score = ps; j.ham = phd; j.depth = 0; pre[i] = j; }
        }
        score_sync<IDX>();
        for (int32_t i = lane; i < n; i += 64) {
            const mtb_jump me = pre[i], pr = pre[me.ptr];
            const int32_t dhd = me.ham - pr.ham;
            mtb_jump fin; fin.ptr = me.ptr == i ? -1 : me.ptr; fin.score = me.score - pr.score; fin.ham = dhd >> 16; fin.depth = dhd & 0xFFFF;
            w.path[i] = mtb_ph_jump_final(w, i, fin);
        }
        score_sync<IDX>();
    } else
    if (simple && !small_n) {
        /* big segment (HBM slab): ping-pong between the path storage and the (now dead) sid/rk/grp_start/blk_start block */
        mtb_jump *ja = (mtb_jump *)w.path, *jb = (mtb_jump *)w.sid;
        for (int32_t i = lane; i < n; i += 64) mtb_ph_jump_init(w, i, ja);
        score_sync<IDX>();
        for (int32_t span = 1; span <= maxrank; span <<= 1) {
            for (int32_t i = lane; i < n; i += 64) jb[i] = mtb_ph_jump_step(ja, i);
            score_sync<IDX>();
            mtb_jump *t = ja; ja = jb; jb = t;
        }
        if (ja != (mtb_jump *)w.sid) {               /* final records must not sit in the path storage while paths are written */
            for (int32_t i = lane; i < n; i += 64) jb[i] = ja[i];
            score_sync<IDX>();
            ja = jb;
        }
        for (int32_t i = lane; i < n; i += 64) { mtb_jump j = ja[i]; w.path[i] = mtb_ph_jump_final(w, i, j); }
        score_sync<IDX>();
    } else
    if (simple) {
        mtb_jump *jump = (mtb_jump *)w.path;          /* fresh paths are rebuilt from the roots at the end */
        mtb_jump t[MAXPER];
        for (int32_t i = lane; i < n; i += 64) mtb_ph_jump_init(w, i, jump);
        score_sync<IDX>();
        for (int32_t span = 1; span <= maxrank; span <<= 1) {
#pragma unroll
            for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) t[k] = mtb_ph_jump_step(jump, i); }
            score_sync<IDX>();
#pragma unroll
            for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) jump[i] = t[k]; }
            score_sync<IDX>();
        }
#pragma unroll
        for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) t[k] = jump[i]; }
        score_sync<IDX>();
#pragma unroll
        for (int k = 0; k < MAXPER; k++) { int32_t i = lane + 64 * k; if (i < n) w.path[i] = mtb_ph_jump_final(w, i, t[k]); }
        score_sync<IDX>();
    } else if (n <= 64 * MAXPER) {
        /* register-cached round state: idle lanes touch no memory in a round */
        int32_t c_rk[MAXPER]; uint32_t c_sh[MAXPER], c_cm[MAXPER], c_reh[MAXPER]; int32_t c_pl[MAXPER];
#pragma unroll
        for (int k = 0; k < MAXPER; k++) {
            int32_t i = lane + 64 * k;
            c_rk[k] = -1; c_sh[k] = 0; c_cm[k] = 0; c_pl[k] = 0; c_reh[k] = 0;
            if (i < n) {
                uint32_t sh = w.shift[i];
                if (sh) { c_rk[k] = w.rk[i]; c_sh[k] = sh; c_cm[k] = w.cmask[i]; c_pl[k] = w.bid[i]; c_reh[k] = w.m[i].right_end_hamming; }
            }
        }
        for (int32_t r = 1; r <= maxrank; r++) {
#pragma unroll
            for (int k = 0; k < MAXPER; k++) {
                bool act = c_rk[k] == r;
                if (!__any(act)) continue;
                if (act) {
                    int32_t i = lane + 64 * k;
                    if (c_sh[k] & MTB_SHIFT_SLOW) mtb_ph_round(w, i, r, &sp);
                    else {
                        int32_t shift = (int32_t)(c_sh[k] & 0x7Fu);
                        int32_t best = -1; float best_score = 0.0f;
                        uint32_t cm = c_cm[k]; int32_t pl = c_pl[k];
                        for (int32_t q = 0; cm; q++, cm >>= 1)
                            if (cm & 1u) { float sc = w.path[pl + q].score; if (sc > best_score) { best = pl + q; best_score = sc; } }
                        if (best >= 0) {
                            mtb_path b = w.path[best];
                            mtb_path p;
                            p.start = b.start; p.end = w.path[i].end;
                            p.score = b.score + mtb_part_score(c_reh[k], shift, false);
                            p.ham = b.ham + mtb_part_ham(c_reh[k], shift, false);
                            p.depth = b.depth + shift; p.start_idx = b.start_idx;
                            w.path[i] = p;
                        }
                    }
                }
            }
            score_sync<IDX>();
        }
    } else {
        for (int32_t r = 1; r <= maxrank; r++) {
            for (int32_t i = lane; i < n; i += 64) mtb_ph_round(w, i, r, &sp);
            score_sync<IDX>();
        }
    }
    MTB_PHASE_MARK(6);
    /* emit + compaction: elist = gid[], exclusive emitted prefix = rk[] */
    IDX *elist = w.gid, *ec = w.rk;
    int32_t ne = 0;
    for (int32_t c0 = 0; c0 < n; c0 += 64) {
        int32_t i = c0 + lane;
        bool e = (i < n) && mtb_ph_emit(w, i, &sp);
        uint64_t me = __ballot(e);
        int32_t pre = ne + (int32_t)__popcll(me & lt);
        if (i < n) ec[i] = (IDX)pre;
        if (e) elist[pre] = (IDX)i;
        ne += (int32_t)__popcll(me);
    }
    score_sync<IDX>();
    MTB_PHASE_MARK(7);
    /* combine: stable order by parallel rank, certain drops in parallel, then one lane per species for the rest.
     * Dead arrays reused: bid -> sorted list, sid -> start of the entry's species range, shift -> pre-drop flag */
    IDX *sorted = w.bid, *elo = w.sid; uint8_t *predrop = w.shift;
    uint64_t *ckeys = (uint64_t *)w.grp_start;        /* slab workspace only: grp_start + blk_start = 8 B per match, dead until sps[] is written */
    if (sizeof(IDX) == 4) {
        for (int32_t e = lane; e < ne; e += 64) ckeys[e] = mtb_path_key(w.path[elist[e]]);
        score_sync<IDX>();
    }
    for (int32_t e = lane; e < ne; e += 64) {
        const int32_t s = mtb_ph_comb_species(w, nsp, (int32_t)elist[e]);
        const int32_t lo = ec[w.sp_start[s]], hi = (s + 1 < nsp) ? (int32_t)ec[w.sp_start[s + 1]] : ne;
        const int32_t pos = sizeof(IDX) == 4 ? mtb_ph_comb_rank_keys(w, elist, ckeys, e, lo, hi) : mtb_ph_comb_rank(w, elist, e, lo, hi);
        sorted[pos] = elist[e];
        elo[e] = (IDX)lo;
    }
    score_sync<IDX>();
    for (int32_t k = lane; k < ne; k += 64) predrop[k] = mtb_ph_comb_predrop(w, sorted, k, (int32_t)elo[k]) ? 1 : 0;
    score_sync<IDX>();
    float *sps = (float *)w.grp_start;
    if (sizeof(IDX) == 4) {
        /* slab segments (long reads): a species can have hundreds of paths and accepted paths; one lane walking the
         * accepted list per candidate was 30 % of the scoring time.  Here the species that emitted paths are taken one
         * after another by the whole wave: the lanes test a candidate against 64 accepted paths at a time, and only the
         * overlapping ones (accepted paths are disjoint, so a handful at most) are applied in acceptance order, as
         * the serial loop does -- trimming can only shrink the candidate, so no other accepted path becomes relevant. */
        for (int32_t q = lane; q < nsp; q += 64) sps[q] = -1.0f;
        score_sync<IDX>();
        for (int32_t lo = 0; lo < ne;) {
            int32_t hi = lo + 1;                                     /* range of this species in the sorted emitted list */
            while (hi < ne && (int32_t)elo[hi] == lo) hi++;
            const int32_t s = mtb_ph_comb_species(w, nsp, (int32_t)sorted[lo]);
            float score = 0.0f; int32_t na = 0;
            for (int32_t k = lo; k < hi; k++) {
                if (predrop[k]) continue;
                const int32_t pi = sorted[k];
                mtb_path p = w.path[pi];
                const mtb_path p0 = p;
                bool drop = false;
                for (int32_t a0 = 0; a0 < na && !drop; a0 += 64) {
                    const int32_t a = a0 + lane;
                    bool ov = false;
                    if (a < na) { const mtb_path c = w.path[w.acc[lo + a]]; ov = !((p0.end < c.start) || (c.end < p0.start)); }
                    uint64_t mask = __ballot(ov);
                    while (mask && !drop) {
                        const int32_t b = a0 + (int32_t)__builtin_ctzll(mask); mask &= mask - 1;
                        const mtb_path c = w.path[w.acc[lo + b]];
                        if (!((p.end < c.start) || (c.end < p.start))) {
                            const int32_t ov2 = (p.end < c.end ? p.end : c.end) - (p.start > c.start ? p.start : c.start) + 1;
                            if (ov2 == p.end - p.start + 1) { drop = true; break; }
                            if (ov2 < 24) {
                                if (p.start < c.start) {
                                    p.end = c.start - 1;
                                    const int32_t h = p.ham - mtb_part_ham(w.m[pi].right_end_hamming, ov2 / 3, false);
                                    p.ham = h > 0 ? h : 0;
                                    p.score = p.score - mtb_part_score(w.m[pi].right_end_hamming, ov2 / 3, false) - (float)(ov2 % 3);
                                } else {
                                    p.start = c.end + 1;
                                    const int32_t h = p.ham - mtb_part_ham(w.m[p.start_idx].right_end_hamming, ov2 / 3, true);
                                    p.ham = h > 0 ? h : 0;
                                    p.score = p.score - mtb_part_score(w.m[p.start_idx].right_end_hamming, ov2 / 3, true) - (float)(ov2 % 3);
                                }
                            } else drop = true;
                        }
                    }
                }
                if (!drop) {
                    if (lane == 0) { w.path[pi] = p; w.acc[lo + na] = (IDX)pi; }
                    na++; score += p.score;
                    score_sync<IDX>();
                }
            }
            float sc = score / (float)read_len;
            if (lane == 0) sps[s] = sc < 1.0f ? sc : 1.0f;
            lo = hi;
        }
    } else
    for (int32_t s0 = 0; s0 < nsp; s0 += 64) {
        int32_t s = s0 + lane;
        float sc = -1.0f;
        if (s < nsp) {
            int32_t lo = ec[w.sp_start[s]], hi = (s + 1 < nsp) ? (int32_t)ec[w.sp_start[s + 1]] : ne;
            if (hi > lo) { sc = mtb_ph_comb_greedy(w, sorted, predrop, lo, hi, read_len); sc = sc < 1.0f ? sc : 1.0f; }
        }
        score_sync<IDX>();                 /* sps[] aliases grp_start/blk_start: all reads above are done */
        if (s < nsp) sps[s] = sc;
    }
    score_sync<IDX>();
    MTB_PHASE_MARK(8);
    /* select (lane 0), then the redundancy filter over the best species' matches */
    int32_t bs = 0, be = 0, species = 0, go = 0;
    if (lane == 0) go = mtb_ph_select(w, sps, nsp, &tx, &sp, &R, &bs, &be, &species) ? 1 : 0;
    go = __shfl(go, 0, 64);
    MTB_PHASE_MARK(9);
    if (go) {
        bs = __shfl(bs, 0, 64); be = __shfl(be, 0, 64); species = __shfl(species, 0, 64);
        uint32_t *hmin = ocnt;
        for (int32_t q = lane; q < nb; q += 64) { hmin[q] = 255u; btax[q] = -1; }
        score_sync<IDX>();
        for (int32_t i = bs + lane; i < be; i += 64) mtb_ph_filter_min(w.m, i, sp.dna_shift, nb, hmin);
        score_sync<IDX>();
        for (int32_t i = bs + lane; i < be; i += 64) mtb_ph_filter_merge(w.m, i, sp.dna_shift, nb, hmin, btax, &tx);
        score_sync<IDX>();
        for (int32_t q = lane; q < nb; q += 64) bham[q] = hmin[q] == 255u ? 255 : 0;
        score_sync<IDX>();
        MTB_PHASE_MARK(10);
        const int32_t ntc = taxcnt_gather_wave(btax, bham, nb, otax, ocnt, (int32_t)tc_room);
        score_sync<IDX>();
        MTB_PHASE_MARK(11);
        /* sub-species descent: climb the (few) taxa in parallel, walk the chains on lane 0 */
        bool to_parent = R.score < sp.min_sp_score;          /* R valid on lane 0 only; recomputed below on lane 0 */
        int32_t slow = ntc > MTB_LR_MAXE ? 1 : 0;
        if (!slow && lane < ntc) {
            int32_t lv;
            mtb_lr_climb(&tx, otax[lane], species, &lv, lr_anc + lane * MTB_LR_K);
            lr_lev[lane] = lv;
            if (lv > MTB_LR_K) slow = 1;
        }
        slow = __any(slow) ? 1 : 0;
        score_sync<IDX>();
        MTB_PHASE_MARK(12);
        if (lane == 0) {
            R.n_taxcnt = (uint16_t)ntc;
            to_parent = R.score < sp.min_sp_score;
            int32_t cs = mtb_tax_canon(&tx, species);
            if (to_parent) R.classification = (species >= 0 && species <= tx.max_taxid) ? tx.sp_parent[species] : 0;
            else if (slow || cs < 0) R.classification = mtb_lower_rank(&tx, otax, ocnt, ntc, species, read_len, sp.denominator, sp.accession_level);
            else R.classification = mtb_lr_bfs(lr_lev, lr_anc, ocnt, ntc, cs, read_len, sp.denominator, &tx, sp.accession_level);
            R.taxcnt_off = (uint32_t)tc_off;
            for (int32_t k = 0; k < ntc; k++)
                if (tc_off + k < tc_cap) { tc_tax[tc_off + k] = otax[k]; tc_cnt[tc_off + k] = ocnt[k]; }
        }
    }
    score_sync<IDX>();
    MTB_PHASE_MARK(13);
}

/* SORT = true: segments arrive grouped by read but unordered (fused path): the
 * wave rank-sorts the staged segment (compareMatches order) and, if
 * sorted_out != NULL, writes it back.  Segments larger than MTB_SCORE_LDS must
 * already be sorted in HBM (k_segsort_large).                               */
#define MTB_SCORE_WS_BYTES_(C) (((C) * (24 + 24 + 3) + ((C) + 1) * 8 * 2 + 64 + 15) & ~15)
/* CAP = matches staged in LDS per read: 160 (10.8 KB per wave, 14 waves/CU) for single reads, 320 for read pairs */
/* SLOT = slot mode (segments filled by k_join<SEG>, `cursor` set): a compile-time switch so that the instantiation the
 * short-read path runs does not carry the exact-segment / slab code (and its register pressure) along. */
template <bool SORT, bool KEY64, typename REC, int CAP = MTB_SCORE_LDS, bool DYN = false, bool SLOT = false>
__global__ __launch_bounds__(64, (CAP > MTB_SCORE_LDS ? 2 : MTB_SCORE_MINWAVES)) void k_score(const REC *__restrict__ matches, const uint64_t *__restrict__ seg_start,
                                               uint64_t n_reads, const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2,
                                               mtb_tax_view tx, mtb_score_params sp, const uint64_t *__restrict__ tc_off,
                                               mtb_result *__restrict__ results, int32_t *__restrict__ tc_tax,
                                               uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, uint8_t *__restrict__ slabs,
                                               uint64_t slab_bytes, uint32_t slab_max_n, uint32_t slab_max_nb,
                                               mtb_match *__restrict__ sorted_out, uint64_t tc_base,
                                               const uint32_t *__restrict__ list, const uint32_t *__restrict__ n_list,
                                               const uint32_t *__restrict__ cursor, uint32_t stride, int seg_by_list,
                                               uint32_t direct, uint32_t epoch, uint32_t *__restrict__ big_list, uint32_t *__restrict__ n_big_out,
                                               uint32_t *__restrict__ cnt_out, unsigned long long *__restrict__ work,
                                               const uint8_t *__restrict__ only_flagged, const uint32_t *__restrict__ seg_cnt = nullptr) {
    __shared__ __attribute__((aligned(16))) uint8_t s_ws[MTB_SCORE_WS_BYTES_(CAP)];
    __shared__ uint32_t s_pf[64];                 /* landing zone of the slot prefetch (never read) */
    /* bucket / taxCnt / chain arrays of the decide phase live in the path storage,
     * which is dead once the species scores exist (keeps LDS per wave small -> occupancy) */
    static_assert(CAP * sizeof(mtb_path) >= MTB_SCORE_BKT * 13 + (MTB_LR_MAXE + MTB_LR_MAXE * MTB_LR_K) * 4, "decide arrays must fit the path area");
    const uint32_t lane = threadIdx.x;
    MTB_BEGIN_ACQUIRE();
    MTB_PHASE_KERNEL_BEGIN();
    const uint64_t n_iter = list ? (uint64_t)*n_list : n_reads;      /* optional: only the listed reads */
    /* DYN (slab launches: long reads, each worth milliseconds): the workgroups claim reads one by one from the counter
     * `work` instead of striding -- the slab pool limits the grid to the resident workgroups, and a static split of a few
     * thousand very unequal reads leaves most of them idle at the end.  A compile-time switch: as a run-time one it cost
     * the short-read instantiation 12 % (59 -> 66 ms, more spill traffic). */
    for (uint64_t it = DYN ? (uint64_t)__shfl(lane == 0 ? atomicAdd(work, 1ull) : 0ull, 0, 64) : (uint64_t)blockIdx.x; it < n_iter;
         it = DYN ? (uint64_t)__shfl(lane == 0 ? atomicAdd(work, 1ull) : 0ull, 0, 64) : it + gridDim.x) {
#ifdef MTB_SCORE_PHASE_CYCLES
        unsigned long long kt0_ = __builtin_readcyclecounter();
#endif
        const uint64_t r = list ? (uint64_t)list[it] : it;
        if ((SLOT || DYN) && only_flagged && only_flagged[r] != 1) continue;   /* already scored by k_score_fast (slot mode) / by k_score_long (slab launches); 2 = listed for the deferred path by k_list_flag2 */
        if (SLOT && !DYN && !list && !only_flagged && it + gridDim.x < n_iter) {
            /* slot mode: start pulling the NEXT read's slots towards L2 now (one dword per 128-byte line, delivered
             * straight into a dummy LDS area: no register, nothing waits for it) -- the slot loads are one dependent
             * HBM round trip per read with nothing to overlap otherwise */
            const uint32_t lines = (stride * (uint32_t)sizeof(mtb_slot16) + 127u) / 128u;
            if (lane < lines) {
                const uint8_t *pf = (const uint8_t *)((const mtb_slot16 *)matches + (it + gridDim.x) * (uint64_t)stride) + (uint64_t)lane * 128u;
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pf,
                                                 (__attribute__((address_space(3))) void *)s_pf, 4, 0, 0);
            }
        }
        /* segment of read r: record slots filled by k_join<SEG> (slot mode), or seg_start indexed by the read or by the list slot */
        uint64_t s0 = 0; int32_t n = 0;
        if (!SLOT) {
            const uint64_t si = seg_by_list ? it : r;
            s0 = seg_start[si]; n = (DYN && seg_cnt) ? (int32_t)seg_cnt[si] : (int32_t)(seg_start[si + 1] - s0);
        }
        const int32_t ql1 = qlen[r], ql2 = qlen2[r];
        const int32_t read_len = ql1 + ql2;
        mtb_result R;
        R.classification = 0; R.score = 0.0f; R.query_length = ql1; R.query_length2 = ql2;
        R.is_classified = 0; R.reserved = 0; R.n_taxcnt = 0; R.taxcnt_off = 0;
        const int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        const uint64_t off = tc_off[r], room = tc_off[r + 1] - off;
        if (SLOT) {
            /* slot mode: live records of the read's slots -> LDS (compaction keeps slot order); reads that do not fit
             * (tail overflow, more live records than the staging, too many position buckets) go to big_list */
            const uint32_t cur = cursor[r], tail_cap = stride - direct;
            mtb_sws<uint16_t> w;
            mtb_sws_carve<uint16_t>(&w, s_ws, CAP);
            bool defer = cur > tail_cap || nb > MTB_SCORE_BKT;
            if (!defer) {
                uint32_t cnt = 0;
                score_sync<uint16_t>();
#ifdef MTB_SCORE_PHASE_CYCLES
                { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) mtb_phase_lds[15] += t_ - kt0_; kt0_ = t_; }
#endif
                /* slots hold the 16-byte form (mtb_slot16): two aligned 64-bit words per lane and slot, expanded to the
                 * 24-byte record in LDS.  (An earlier struct copy with a patched byte was lowered to overlapping
                 * unaligned loads that waited for each other: 12 k cycles per read, measured.) */
                const mtb_slot16 *slots = (const mtb_slot16 *)matches + r * (uint64_t)stride;
                uint64_t *dst64 = (uint64_t *)w.m;
                auto put = [&](uint32_t pos, const mtb_slot16 &x) {
                    const mtb_match m = mtb_slot_unpack(x, (uint32_t)r + 1);
                    const uint64_t *q = (const uint64_t *)&m;
                    dst64[3 * pos] = q[0]; dst64[3 * pos + 1] = q[1]; dst64[3 * pos + 2] = q[2];
                };
                if (stride <= 192) {
                    mtb_slot16 x[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {          /* all slot loads of the lane in flight at once */
                        const uint32_t i = lane + 64 * k;
                        x[k].a = 0; x[k].b = 0;
                        if (i < stride) x[k] = slots[i];
                    }
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const uint32_t i = lane + 64 * k;
                        const bool live = i < stride && mtb_slot_epoch(x[k]) == epoch && (i < direct || i - direct < cur);
                        const uint64_t mask = __ballot(live);
                        const uint32_t pos = cnt + (uint32_t)__popcll(mask & lanemask_lt());
                        if (live && pos < CAP) put(pos, x[k]);
                        cnt += (uint32_t)__popcll(mask);
                    }
                } else
                for (uint32_t c0 = 0; c0 < stride; c0 += 64) {
                    const uint32_t i = c0 + lane;
                    mtb_slot16 x; x.a = 0; x.b = 0;
                    if (i < stride) x = slots[i];
                    const bool live = i < stride && mtb_slot_epoch(x) == epoch && (i < direct || i - direct < cur);
                    const uint64_t mask = __ballot(live);
                    const uint32_t pos = cnt + (uint32_t)__popcll(mask & lanemask_lt());
                    if (live && pos < CAP) put(pos, x);
                    cnt += (uint32_t)__popcll(mask);
                }
                n = (int32_t)cnt;
                defer = cnt > CAP;
                score_sync<uint16_t>();
#ifdef MTB_SCORE_PHASE_CYCLES
                { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) mtb_phase_lds[14] += t_ - kt0_; kt0_ = t_; }
#endif
            }
            if (defer) { if (lane == 0) big_list[atomicAdd(n_big_out, 1u)] = (uint32_t)r; continue; }
            if (lane == 0) cnt_out[r] = (uint32_t)n;
            if (n == 0) { if (lane == 0) results[r] = R; continue; }
            int32_t *s_btax = (int32_t *)w.path, *s_otax = s_btax + MTB_SCORE_BKT;
            uint32_t *s_ocnt = (uint32_t *)(s_otax + MTB_SCORE_BKT);
            int32_t *s_lev = (int32_t *)(s_ocnt + MTB_SCORE_BKT), *s_anc = s_lev + MTB_LR_MAXE;
            uint8_t *s_bham = (uint8_t *)(s_anc + MTB_LR_MAXE * MTB_LR_K);
            score_read_par<uint16_t, true, KEY64, mtb_match, true, CAP>(w.m, n, w, s_btax, s_bham, s_otax, s_ocnt, s_lev, s_anc, nb, read_len, tx, sp, off, room,
                                                                     tc_tax, tc_cnt, tc_cap, (mtb_match *)nullptr, R);
            if (lane == 0) { R.query_length = ql1; R.query_length2 = ql2; R.reserved = 0; R.taxcnt_off += (uint32_t)tc_base; results[r] = R; }
            continue;
        }
        if (n == 0) { if (lane == 0) results[r] = R; continue; }
        const bool big = n > CAP || nb > MTB_SCORE_BKT;
        if (!big) {
            mtb_sws<uint16_t> w;
            mtb_sws_carve<uint16_t>(&w, s_ws, CAP);
            int32_t *s_btax = (int32_t *)w.path, *s_otax = s_btax + MTB_SCORE_BKT;
            uint32_t *s_ocnt = (uint32_t *)(s_otax + MTB_SCORE_BKT);
            int32_t *s_lev = (int32_t *)(s_ocnt + MTB_SCORE_BKT), *s_anc = s_lev + MTB_LR_MAXE;
            uint8_t *s_bham = (uint8_t *)(s_anc + MTB_LR_MAXE * MTB_LR_K);
            score_read_par<uint16_t, SORT, KEY64, REC, false, CAP>(matches + s0, n, w, s_btax, s_bham, s_otax, s_ocnt, s_lev, s_anc, nb, read_len, tx, sp, off, room,
                                           tc_tax, tc_cnt, tc_cap, sorted_out ? sorted_out + s0 : nullptr, R);
        } else {
            if ((uint32_t)n > slab_max_n || (uint32_t)nb > slab_max_nb) {      /* cannot happen: slabs are sized from the maxima */
                if (lane == 0) { R.reserved = 0xFF; results[r] = R; }
                continue;
            }
            uint8_t *slab = slabs + (uint64_t)blockIdx.x * slab_bytes;
            mtb_sws<uint32_t> w;
            mtb_sws_carve<uint32_t>(&w, slab, slab_max_n);
            uint8_t *p = slab + mtb_sws_bytes<uint32_t>(slab_max_n);
            uint64_t B = slab_max_nb;
            int32_t *btax = (int32_t *)p; int32_t *otax = btax + B; uint32_t *ocnt = (uint32_t *)(otax + B);
            int32_t *lev = (int32_t *)(ocnt + B); int32_t *anc = lev + MTB_LR_MAXE; uint8_t *bham = (uint8_t *)(anc + MTB_LR_MAXE * MTB_LR_K);
            /* big segments are pre-sorted in HBM; a small segment of a long read still needs its sort */
            if (SORT && n <= CAP)
                score_read_par<uint32_t, true, KEY64, REC, false, CAP>(matches + s0, n, w, btax, bham, otax, ocnt, lev, anc, nb, read_len, tx, sp, off, room, tc_tax, tc_cnt,
                                               tc_cap, sorted_out ? sorted_out + s0 : nullptr, R);
            else
                score_read_par<uint32_t, false, false, REC>(matches + s0, n, w, btax, bham, otax, ocnt, lev, anc, nb, read_len, tx, sp, off, room, tc_tax, tc_cnt,
                                                tc_cap, (mtb_match *)nullptr, R);
        }
        if (lane == 0) { R.query_length = ql1; R.query_length2 = ql2; R.reserved = 0; R.taxcnt_off += (uint32_t)tc_base; results[r] = R; }
    }
    MTB_PHASE_KERNEL_END();
}

/* Large segments (long reads): one 1024-thread workgroup per read.
 *   1. chunks of MTB_SEGLDS_CHUNK matches: (key1, key2, index) triples sorted in LDS (all-ascending bitonic network
 *      with virtual +inf padding; ties by index = stable), records gathered into the scratch buffer;
 *   2. runs merged pairwise by rank: every record finds its output slot with one binary search in the sibling run
 *      (lower bound for the left run, upper bound for the right run = stable), ping-pong scratch <-> segment.
 * HBM/L2 sees each record 2 + 2*ceil(log2(n/chunk)) times instead of ~log^2(n) times as in k_segsort_large.   */
#define MTB_SEGLDS_CHUNK 8192
#define MTB_SEGLDS_THREADS 1024
MTB_HD bool mtb_key_less(uint64_t a1, uint32_t a2, uint64_t b1, uint32_t b2) { return a1 < b1 || (a1 == b1 && a2 < b2); }

template <typename REC>
__global__ __launch_bounds__(MTB_SEGLDS_THREADS) void k_segsort_lds(REC *m, const uint64_t *__restrict__ seg_start, const uint32_t *__restrict__ large,
                                                                     const uint32_t *__restrict__ n_large, REC *scratch) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_dyn[];
    uint64_t *k1 = (uint64_t *)s_dyn;
    uint32_t *k2 = (uint32_t *)(k1 + MTB_SEGLDS_CHUNK);
    uint16_t *ix = (uint16_t *)(k2 + MTB_SEGLDS_CHUNK);
    const uint32_t nl = *n_large, t = threadIdx.x, NT = MTB_SEGLDS_THREADS;
    for (uint32_t b = blockIdx.x; b < nl; b += gridDim.x) {
        const uint32_t r = large ? large[b] : b;
        const uint64_t s0 = seg_start[r];
        const uint32_t n = (uint32_t)(seg_start[r + 1] - s0);
        REC *seg = m + s0, *tmp = scratch + s0;
        for (uint32_t base = 0; base < n; base += MTB_SEGLDS_CHUNK) {
            const uint32_t cn = n - base < MTB_SEGLDS_CHUNK ? n - base : MTB_SEGLDS_CHUNK;
            const REC *src = seg + base;
            __syncthreads();
            for (uint32_t i = t; i < cn; i += NT) { const mtb_match &x = rec_m(src[i]); k1[i] = mtb_key1(x); k2[i] = mtb_key2(x); ix[i] = (uint16_t)i; }
            __syncthreads();
            uint32_t p2 = 1; while (p2 < cn) p2 <<= 1;
            for (uint32_t k = 2; k <= p2; k <<= 1) {
                for (uint32_t j = k >> 1; j > 0; j >>= 1) {
                    for (uint32_t q = t; q < (p2 >> 1); q += NT) {
                        uint32_t lo = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                        uint32_t hi = (j == (k >> 1)) ? (lo ^ ((j << 1) - 1)) : (lo + j);
                        if (hi < cn) {
                            uint64_t a1 = k1[lo], b1 = k1[hi]; uint32_t a2 = k2[lo], b2 = k2[hi];
                            uint16_t ai = ix[lo], bi = ix[hi];
                            bool sw = (b1 < a1) || (b1 == a1 && (b2 < a2 || (b2 == a2 && bi < ai)));
                            if (sw) { k1[lo] = b1; k1[hi] = a1; k2[lo] = b2; k2[hi] = a2; ix[lo] = bi; ix[hi] = ai; }
                        }
                    }
                    __syncthreads();
                }
            }
            for (uint32_t i = t; i < cn; i += NT) tmp[base + i] = src[ix[i]];
        }
        __syncthreads();
        REC *src = tmp, *dst = seg;
        for (uint32_t R = MTB_SEGLDS_CHUNK; R < n; R <<= 1) {
            for (uint32_t i = t; i < n; i += NT) {
                const uint32_t pair = i / (2 * R), pb = pair * 2 * R, off = i - pb;
                const REC rec = src[i];
                const mtb_match &x = rec_m(rec);
                const uint64_t x1 = mtb_key1(x); const uint32_t x2 = mtb_key2(x);
                uint32_t pos;
                if (off < R) {                                   /* left run: + #right elements strictly smaller */
                    uint32_t lo = pb + R < n ? pb + R : n, hi = pb + 2 * R < n ? pb + 2 * R : n;
                    const uint32_t b0 = lo;
                    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; const mtb_match &y = rec_m(src[mid]);
                                      if (mtb_key_less(mtb_key1(y), mtb_key2(y), x1, x2)) lo = mid + 1; else hi = mid; }
                    pos = off + (lo - b0);
                } else {                                         /* right run: + #left elements smaller or equal */
                    uint32_t lo = pb, hi = pb + R;
                    while (lo < hi) { uint32_t mid = (lo + hi) >> 1; const mtb_match &y = rec_m(src[mid]);
                                      if (!mtb_key_less(x1, x2, mtb_key1(y), mtb_key2(y))) lo = mid + 1; else hi = mid; }
                    pos = (off - R) + (lo - pb);
                }
                dst[pb + pos] = rec;
            }
            __syncthreads();
            REC *sw = src; src = dst; dst = sw;
        }
        if (src != seg) { for (uint32_t i = t; i < n; i += NT) seg[i] = src[i]; }
        __syncthreads();
    }
}

#endif
