/* kernels_join.h -- sorted-merge lookup of the query metamers against the flat
 * target index resident in HBM, and the regrouping of matches by read.
 *
 * k_join_bounds finds, for every tile of 512 sorted query metamers, the slice
 * of the target index that holds the tile's amino-acid range (one lane per
 * tile).  k_join stages that slice in LDS with coalesced 64-bit loads (32 KB
 * window; slices that do not fit are searched in place), every lane then
 * locates its query's amino-acid run in the window by binary search in LDS and
 * applies the two-phase selection (min Hamming, threshold min(2*min,7));
 * workgroup scan of the counts, ONE atomic per workgroup to reserve output,
 * then emit.  Functional form of
 * KmerMatcher::matchKmers (src/commons/KmerMatcher.cpp:123-481) + compareDna
 * (:1117-1146); the per-read match counters it also feeds replace the global
 * comparison sort of matches by sequenceID (sortMatches, :1071-1078).
 * Algorithmic HBM bytes: 16 per query + 12 per target of the spanned range +
 * 24 per emitted match.                                                      */
#ifndef MTB_KERNELS_JOIN_H
#define MTB_KERNELS_JOIN_H
#include "dev_util.h"
#include "mtb_core.h"

/* Optional 32-byte padded segment record.  Measured (profiles/r01_notes.md): full-sector
 * stores do NOT remove the 3.3x traffic amplification of the scattered regroup writes
 * (the L2 write-allocates whole lines), so the fused path keeps 24-byte records. */
struct __attribute__((aligned(32))) mtb_match32 { mtb_match m; uint64_t pad; };
__device__ __forceinline__ const mtb_match &rec_m(const mtb_match &r) { return r; }
__device__ __forceinline__ const mtb_match &rec_m(const mtb_match32 &r) { return r.m; }
__device__ __forceinline__ void rec_set(mtb_match &r, const mtb_match &m) { r = m; }
__device__ __forceinline__ void rec_set(mtb_match32 &r, const mtb_match &m) { r.m = m; r.pad = 0; }

#ifndef MTB_JOIN_QPT
#define MTB_JOIN_QPT 2                      /* queries per thread                              */
#endif
#define MTB_JOIN_QPB (256 * MTB_JOIN_QPT)   /* sorted queries per workgroup                     */
#ifndef MTB_JOIN_WIN
#define MTB_JOIN_WIN 2560                   /* target values staged in LDS (20 KB -> 7 workgroups/CU; sweep in profiles/r01_notes.md) */
#endif

/* Target window of every query tile, one lane per tile: the tile is sorted at
 * least on its top 32 bits, so its amino-acid range lies inside
 * [first & ~(2^32-1), (last | (2^32-1)) + 1).  Two binary searches per tile,
 * all tiles in parallel: the 33 dependent loads of a search are paid once per
 * tile here instead of serialising every workgroup of k_join.               */
__global__ __launch_bounds__(256) void k_join_bounds(const mtb_kmer *__restrict__ q, uint64_t n, const uint64_t *__restrict__ values,
                                                      uint64_t limit, uint64_t n_tiles, uint64_t *__restrict__ bounds, int low_bits) {
    uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (t >= n_tiles) return;
    uint64_t base = t * MTB_JOIN_QPB;
    uint64_t last = (base + MTB_JOIN_QPB <= n ? base + MTB_JOIN_QPB : n) - 1;
    const uint64_t low = (1ull << low_bits) - 1;       /* the query list is ordered on bits [low_bits, 64) at least */
    uint64_t ka = q[base].value & ~low;
    uint64_t kb = q[last].value | low;
    uint64_t lo = mtb_lower_bound(values, limit, ka);
    uint64_t hi = (kb == ~0ull) ? limit : mtb_lower_bound(values, limit, kb + 1);
    bounds[2 * t] = lo; bounds[2 * t + 1] = hi;
}

/* SEG mode (short reads, fused path): every read owns a segment of `stride` record slots.  The extractor tagged
 * every query with its ordinal inside the read (bits 16-31 of qinfo's position field): the first match of query
 * `ord` goes to slot `ord` (ord < direct) with NO atomic and no count -- most queries of a classifiable read have
 * exactly one match, and in slot order those matches are already in (frame, position) order, which is the
 * scorer's order when one species is involved.  Further matches of the query (and queries with ord >= direct) take
 * slots direct + t of the segment's tail, t from a returning atomic on the read's tail cursor; matches beyond the
 * tail go to an overflow list and their reads complete on the large-segment path (k_big_*).  Slots hold the 16-byte
 * form of a match (mtb_slot16); a slot is live when its epoch field equals the batch's tag: segments are never
 * cleared between batches.                                                                                   */
struct JoinSegArgs {
    mtb_slot16 *seg; uint32_t stride, direct; uint32_t *cursor;
    mtb_match *ovf; uint64_t ovf_cap; unsigned long long *ovf_counter;
    uint32_t epoch;
    /* long reads (k_join_dir<.., LONG>): read r owns slots [rb[r], rb[r + 1]) = dcnt[r] direct ones (one per metamer, by ordinal)
     * + mtb_lslot_tail(dcnt[r], tf) tail slots; matches beyond the tail are only counted (the caller retries with a larger tail) */
    const uint64_t *rb; const uint32_t *dcnt; uint32_t tf;
    uint32_t list;      /* k_join_dir<.., 2>: matches go to the dense list ovf[0 .. ovf_cap) (owner side of the partitioned index) */
    uint32_t coop_min;  /* k_join_dir: candidate runs longer than this are scanned by the whole wave (kernels_dir.h) */
    /* The overflow list of k_join_dir's slot modes is cut into ovf_stripes regions of ovf_region entries, each with a counter of its own
     * (64 bytes apart, ovf_counter[8 * stripe]; a workgroup uses stripe blockIdx.x mod ovf_stripes): ONE counter for the whole list made
     * every overflowing match a returning atomic on one address -- ~19 ns each once they no longer coalesce inside a wave, +100 ms per
     * 10 M reads of a sample in which the reads of conserved genes overflow their tails (profiles/r04_notes.md).  0: one dense list. */
    uint32_t ovf_stripes; uint64_t ovf_region;
    uint32_t retry;     /* host side only (dev_join): this is a further attempt at a batch whose overflow list was too small: no tuning of the join's variant on it */
    uint32_t dense_ovf; /* host side only (dev_join): 1 = one dense overflow list even where the list would be striped -- the last retry of a batch: which
                         * workgroup emits a match beyond a read's tail depends on the order of the atomics, so the stripes fill differently from
                         * attempt to attempt and a list sized from the failed attempt's fullest stripe can fail again; the dense list needs the total only */
    const uint8_t *off; /* k_join_dir, fixed segments: reads marked here never use their slots -- every match goes to the overflow list and the read's
                         * tail cursor is pushed beyond the tail's capacity, so that the scorers hand the read to the exact-segment path */
};

#ifdef MTB_SCORE_PHASE_CYCLES
__device__ unsigned long long mtb_join_cycles[8];     /* 0 load+window, 1 find, 2 count, 3 reserve (atomics), 4 emit */
#define MTB_JP_BEGIN() unsigned long long jp_t_ = __builtin_readcyclecounter()
#define MTB_JP_MARK(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&mtb_join_cycles[k], t_ - jp_t_); jp_t_ = t_; } while (0)
#else
#define MTB_JP_BEGIN() do {} while (0)
#define MTB_JP_MARK(k) do {} while (0)
#endif

template <bool SEG>
__global__ __launch_bounds__(256) void k_join(const mtb_kmer *__restrict__ q, uint64_t n, mtb_index_view ix,
                                               const mtb_tables *__restrict__ tabs, const uint64_t *__restrict__ bounds,
                                               mtb_match *__restrict__ out, uint64_t cap, unsigned long long *__restrict__ counter,
                                               uint32_t *__restrict__ read_cnt, uint32_t *__restrict__ overflow, JoinSegArgs sa) {
    __shared__ mtb_tables s_tab;
    __shared__ uint32_t s_tmp[8];
    __shared__ unsigned long long s_base;
    __shared__ uint64_t s_win[MTB_JOIN_WIN];
    MTB_JP_BEGIN();
    for (uint32_t i = threadIdx.x; i < sizeof(mtb_tables) / 4; i += 256) ((uint32_t *)&s_tab)[i] = ((const uint32_t *)tabs)[i];
    const uint64_t base = (uint64_t)blockIdx.x * MTB_JOIN_QPB;
    mtb_kmer k[MTB_JOIN_QPT];
    bool valid[MTB_JOIN_QPT];
#pragma unroll
    for (int u = 0; u < MTB_JOIN_QPT; u++) {
        uint64_t j = base + (uint64_t)u * 256 + threadIdx.x;
        valid[u] = j < n;
        if (valid[u]) { k[u] = q[j]; valid[u] = mtb_q_seq(k[u].qinfo) != 0; }   /* blank slots carry sequenceID 0 */
        else { k[u].value = 0; k[u].qinfo = 0; }
    }
    const uint64_t lo = bounds[2 * (uint64_t)blockIdx.x], hi = bounds[2 * (uint64_t)blockIdx.x + 1];
    const uint64_t span = hi - lo;
    const bool in_lds = span <= MTB_JOIN_WIN;
    if (in_lds) for (uint64_t i = threadIdx.x; i < span; i += 256) s_win[i] = ix.values[lo + i];      /* coalesced 64-bit loads */
    __syncthreads();
    MTB_JP_MARK(0);
    uint32_t c[MTB_JOIN_QPT]; uint64_t rs[MTB_JOIN_QPT]; uint32_t rl[MTB_JOIN_QPT];
    uint32_t csum = 0;
#ifdef MTB_SCORE_PHASE_CYCLES
    for (int u = 0; u < MTB_JOIN_QPT; u++) {
        rs[u] = 0; rl[u] = 0;
        if (valid[u]) { if (in_lds) mtb_join_find(s_win, span, k[u].value, &rs[u], &rl[u]); else mtb_join_find(ix.values + lo, span, k[u].value, &rs[u], &rl[u]); }
    }
    MTB_JP_MARK(1);
#endif
#pragma unroll
    for (int u = 0; u < MTB_JOIN_QPT; u++) {
        c[u] = 0; rs[u] = 0; rl[u] = 0;
        if (valid[u]) {
            if (in_lds) {
                mtb_join_find(s_win, span, k[u].value, &rs[u], &rl[u]);
                c[u] = mtb_join_select(&s_tab, s_win, rs[u], rl[u], k[u].value, k[u].qinfo, ix.info, lo, ix.tax2species, ix.max_taxid,
                                       ix.info_mask, ix.kmer_format, (mtb_match *)nullptr, 0);
            } else {
                mtb_join_find(ix.values + lo, span, k[u].value, &rs[u], &rl[u]);
                c[u] = mtb_join_select(&s_tab, ix.values + lo, rs[u], rl[u], k[u].value, k[u].qinfo, ix.info, lo, ix.tax2species, ix.max_taxid,
                                       ix.info_mask, ix.kmer_format, (mtb_match *)nullptr, 0);
            }
        }
        csum += c[u];
    }
    MTB_JP_MARK(2);
    if (SEG) {
        /* the (rare) returning atomics of the thread first, then the stores */
        const uint32_t tail_cap = sa.stride - sa.direct;
        uint32_t slot[MTB_JOIN_QPT];
#pragma unroll
        for (int u = 0; u < MTB_JOIN_QPT; u++) {
            const uint32_t ord = mtb_q_pos(k[u].qinfo) >> 16;
            const uint32_t ntail = c[u] - ((c[u] && ord < sa.direct) ? 1u : 0u);
            slot[u] = ntail ? atomicAdd(&sa.cursor[mtb_q_seq(k[u].qinfo) - 1], ntail) : 0u;
        }
#pragma unroll
        for (int u = 0; u < MTB_JOIN_QPT; u++) {
            if (c[u] == 0) continue;
            const uint32_t r = mtb_q_seq(k[u].qinfo) - 1;
            const uint32_t ord = mtb_q_pos(k[u].qinfo) >> 16;
            const uint64_t qinfo = k[u].qinfo & ~0xFFFF0000ull;          /* the record carries the reference's qinfo */
            const uint32_t first = ord < sa.direct ? 1u : 0u;
            mtb_slot16 *seg = sa.seg + (uint64_t)r * sa.stride;
            if (first) {
                if (in_lds) mtb_join_select(&s_tab, s_win, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                            ix.kmer_format, (mtb_match *)nullptr, 1, 0, seg + ord, sa.epoch);
                else mtb_join_select(&s_tab, ix.values + lo, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                     ix.kmer_format, (mtb_match *)nullptr, 1, 0, seg + ord, sa.epoch);
            }
            const uint32_t ntail = c[u] - first;
            if (ntail == 0) continue;
            const uint32_t s = slot[u];
            const uint32_t fit = s < tail_cap ? (ntail < tail_cap - s ? ntail : tail_cap - s) : 0u;
            if (fit) {
                if (in_lds) mtb_join_select(&s_tab, s_win, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                            ix.kmer_format, (mtb_match *)nullptr, fit, first, seg + sa.direct + s, sa.epoch);
                else mtb_join_select(&s_tab, ix.values + lo, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                     ix.kmer_format, (mtb_match *)nullptr, fit, first, seg + sa.direct + s, sa.epoch);
            }
            const uint32_t n_ovf = ntail - fit;
            if (n_ovf) {
                const unsigned long long o = atomicAdd(sa.ovf_counter, (unsigned long long)n_ovf);
                if (o + n_ovf <= sa.ovf_cap) {
                    if (in_lds) mtb_join_select(&s_tab, s_win, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                                ix.kmer_format, sa.ovf + o, n_ovf, first + fit);
                    else mtb_join_select(&s_tab, ix.values + lo, rs[u], rl[u], k[u].value, qinfo, ix.info, lo, ix.tax2species, ix.max_taxid, ix.info_mask,
                                         ix.kmer_format, sa.ovf + o, n_ovf, first + fit);
                } else *overflow = 1;
            }
        }
        MTB_JP_MARK(4);
        return;
    }
    uint32_t tot;
    uint32_t off = block256_exclusive_scan<uint32_t>(csum, s_tmp, &tot);
    if (threadIdx.x == 0) s_base = tot ? atomicAdd(counter, (unsigned long long)tot) : 0ull;
    __syncthreads();
    if (csum == 0) return;
    uint64_t dst = (uint64_t)s_base + off;
    if (dst + csum > cap) { *overflow = 1; }
#pragma unroll
    for (int u = 0; u < MTB_JOIN_QPT; u++) {
        if (c[u] == 0) continue;
        if (read_cnt) atomicAdd(&read_cnt[mtb_q_seq(k[u].qinfo) - 1], c[u]);
        if (dst + c[u] <= cap) {
            if (in_lds) mtb_join_select(&s_tab, s_win, rs[u], rl[u], k[u].value, k[u].qinfo, ix.info, lo, ix.tax2species, ix.max_taxid,
                                        ix.info_mask, ix.kmer_format, out + dst, c[u]);
            else mtb_join_select(&s_tab, ix.values + lo, rs[u], rl[u], k[u].value, k[u].qinfo, ix.info, lo, ix.tax2species, ix.max_taxid,
                                 ix.info_mask, ix.kmer_format, out + dst, c[u]);
        }
        dst += c[u];
    }
}

/* ---- completion of the reads k_score could not take from their slots (tail overflow, or more live records than
 * its LDS staging): exact segments = live direct slots + tail slots + overflow entries, sorted in HBM, scored by a
 * second launch.  `big_list` is filled by k_score.  One wavefront per listed read.                            */
__device__ __forceinline__ bool seg_slot_live(const mtb_slot16 &s, uint32_t i, uint32_t direct, uint32_t tail_n, uint32_t epoch) {
    return mtb_slot_epoch(s) == epoch && (i < direct || i - direct < tail_n);
}
__global__ __launch_bounds__(64) void k_big_count(const mtb_slot16 *__restrict__ seg, uint32_t stride, uint32_t direct, uint32_t epoch,
                                                   const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ big_list, uint32_t n_big,
                                                   uint32_t *__restrict__ big_cnt, uint32_t *__restrict__ bigidx, uint32_t *__restrict__ max_seg,
                                                   const uint8_t *__restrict__ off = nullptr /* reads routed around their slots: the join pushed their cursor by tail_cap + 1 per match */,
                                                   const uint32_t *__restrict__ novf = nullptr /* entries of the read in the GROUPED overflow list (k_ovf_count), if there is one */,
                                                   unsigned long long *__restrict__ n_ungrouped = nullptr /* += listed reads whose overflow entries are not in the grouped list */) {
    uint32_t mx = 0;
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t r = big_list[b];
        const uint32_t cur = cursor[r], tail_cap = stride - direct, tail_n = cur < tail_cap ? cur : tail_cap;
        const mtb_slot16 *s = seg + (uint64_t)r * stride;
        uint32_t n = 0;
        for (uint32_t c0 = 0; c0 < stride; c0 += 64) {
            uint32_t i = c0 + threadIdx.x;
            bool live = false;
            if (i < stride) { mtb_slot16 x = s[i]; live = seg_slot_live(x, i, direct, tail_n, epoch); }
            n += (uint32_t)__popcll(__ballot(live));
        }
        const uint32_t n_o = (off && off[r]) ? cur / (tail_cap + 1u) : cur - tail_n;       /* overflow entries */
        n += n_o;
        if (threadIdx.x == 0) { big_cnt[b] = n; bigidx[r] = b; if (n_ungrouped && n_o && !(novf && novf[r])) atomicAdd(n_ungrouped, 1ull); }
        mx = n > mx ? n : mx;
    }
    if (threadIdx.x == 0 && mx) atomicMax(max_seg, mx);
}
__global__ __launch_bounds__(64) void k_big_copy(const mtb_slot16 *__restrict__ seg, uint32_t stride, uint32_t direct, uint32_t epoch,
                                                  const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ big_list,
                                                  const uint64_t *__restrict__ big_start, uint32_t n_big, uint32_t *__restrict__ bigcur,
                                                  mtb_match *__restrict__ big,
                                                  /* the overflow list grouped by read (k_ovf_group), when the batch has one: a listed read's entries are copied from its group --
                                                   * k_big_ovf's pass over the WHOLE list (7 ms per 10 M reads of held-out genomes, for 5 334 listed reads) is only launched for
                                                   * reads that have no group (routed around their slots, more than 65535 entries) */
                                                  const mtb_match *__restrict__ ovfg = nullptr, const uint64_t *__restrict__ ostart = nullptr, const uint32_t *__restrict__ novf = nullptr) {
    for (uint32_t b = blockIdx.x; b < n_big; b += gridDim.x) {
        const uint32_t r = big_list[b];
        const uint32_t cur = cursor[r], tail_cap = stride - direct, tail_n = cur < tail_cap ? cur : tail_cap;
        const mtb_slot16 *s = seg + (uint64_t)r * stride;
        mtb_match *dst = big + big_start[b];
        uint32_t n = 0;
        for (uint32_t c0 = 0; c0 < stride; c0 += 64) {
            uint32_t i = c0 + threadIdx.x;
            mtb_slot16 x; x.a = 0; x.b = 0; bool live = false;
            if (i < stride) { x = s[i]; live = seg_slot_live(x, i, direct, tail_n, epoch); }
            uint64_t mask = __ballot(live);
            if (live) {
                const mtb_match m = mtb_slot_unpack(x, r + 1);
                uint64_t *d = (uint64_t *)(dst + n + (uint32_t)__popcll(mask & lanemask_lt()));
                const uint64_t *q = (const uint64_t *)&m;
                d[0] = q[0]; d[1] = q[1]; d[2] = q[2];
            }
            n += (uint32_t)__popcll(mask);
        }
        if (novf) {
            const uint32_t g = novf[r];
            const mtb_match *src = ovfg + ostart[r];
            for (uint32_t j = threadIdx.x; j < g; j += 64) dst[n + j] = src[j];
            n += g;
        }
        if (threadIdx.x == 0) bigcur[b] = n;
    }
}
/* region_cap != 0: the list is striped (JoinSegArgs::ovf_stripes): blockIdx.y = stripe, its entries [0, counters[8 * stripe]) */
__global__ __launch_bounds__(256) void k_big_ovf(const mtb_match *__restrict__ ovf, uint64_t n_ovf, const uint32_t *__restrict__ bigidx,
                                                  const uint64_t *__restrict__ big_start, uint32_t *__restrict__ bigcur, mtb_match *__restrict__ big,
                                                  uint64_t region_cap = 0, const unsigned long long *__restrict__ counters = nullptr, const uint32_t *__restrict__ novf = nullptr) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (region_cap) {
        const unsigned long long cnt = counters[8 * blockIdx.y];
        if (i >= cnt || i >= region_cap) return;
        i += (uint64_t)blockIdx.y * region_cap;
    } else if (i >= n_ovf) return;
    mtb_match m = ovf[i];
    if (m.pad & 2u) { m.qinfo &= ~0xFFFF0000ull; m.pad = 0; }      /* (the directory join's entries carry their place in the read's group: kernels_dir.h, ovf_put) */
    if (novf && novf[mtb_q_seq(m.qinfo) - 1]) return;      /* (copied from its read's group by k_big_copy) */
    uint32_t b = bigidx[mtb_q_seq(m.qinfo) - 1];
    if (b == 0xFFFFFFFFu) return;         /* a read that is not on the list (scored by k_score_many from the grouped overflow entries): bigidx[] is set to ~0 before k_big_count */
    uint32_t slot = atomicAdd(&bigcur[b], 1u);
    big[big_start[b] + slot] = m;
}

/* Home side of the range-partitioned index: matches that came back from the range owners (k_join_dir<.., 2>: qinfo still carries the
 * query's ordinal, pad = 1 for the query's first match) -> this rank's slot segments, exactly as the fused join fills them:
 * first match to slot `ord`, the others to the read's tail (returning atomic on its cursor), beyond that to the overflow list. */
__global__ __launch_bounds__(256) void k_slot_place(const mtb_match *__restrict__ in, uint64_t n, JoinSegArgs sa, uint64_t n_reads, uint32_t *__restrict__ bad) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const mtb_match m = in[i];
    const uint32_t seq = mtb_q_seq(m.qinfo);
    if (seq == 0 || seq > n_reads) { *bad = 1; return; }
    const uint32_t r = seq - 1, ord = mtb_q_pos(m.qinfo) >> 16;
    const uint64_t qinfo = m.qinfo & ~0xFFFF0000ull;
    mtb_slot16 *seg = sa.seg + (uint64_t)r * sa.stride;
    if ((m.pad & 1u) && ord < sa.direct) {
        const mtb_slot16 sl = mtb_slot_pack(qinfo, m.target_id, m.species_id, m.dna, m.right_end_hamming, m.hamming, sa.epoch);
        __builtin_nontemporal_store(sl.a, &seg[ord].a); __builtin_nontemporal_store(sl.b, &seg[ord].b);
        return;
    }
    const uint32_t tail_cap = sa.stride - sa.direct;
    const uint32_t at = atomicAdd(&sa.cursor[r], 1u);
    if (at < tail_cap) {
        const mtb_slot16 sl = mtb_slot_pack(qinfo, m.target_id, m.species_id, m.dna, m.right_end_hamming, m.hamming, sa.epoch);
        __builtin_nontemporal_store(sl.a, &seg[sa.direct + at].a); __builtin_nontemporal_store(sl.b, &seg[sa.direct + at].b);
    } else {
        const unsigned long long o = atomicAdd(sa.ovf_counter, 1ull);
        if (o < sa.ovf_cap) { mtb_match x = m; x.qinfo = qinfo; x.pad = 0; sa.ovf[o] = x; }
    }
}

/* Move every match into its read's segment (seg_start from a scan of the
 * per-read counters).  The order inside a segment is irrelevant: the segment
 * sort that follows imposes the total order of compareMatches.              */
template <typename REC>
__global__ __launch_bounds__(256) void k_regroup(const mtb_match *__restrict__ in, uint64_t n, const uint64_t *__restrict__ seg_start,
                                                  uint32_t *__restrict__ cursor, REC *__restrict__ out) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    mtb_match m = in[i];
    uint32_t r = mtb_q_seq(m.qinfo) - 1;
    uint32_t slot = atomicAdd(&cursor[r], 1u);
    REC o; rec_set(o, m);
    out[seg_start[r] + slot] = o;
}

/* per-read counters from an (arbitrarily ordered) match list: stage API path */
__global__ __launch_bounds__(256) void k_count_reads(const mtb_match *__restrict__ in, uint64_t n, uint32_t *__restrict__ read_cnt,
                                                      uint64_t n_reads, uint32_t *__restrict__ bad) {
    uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    uint32_t r = mtb_q_seq(in[i].qinfo) - 1;
    if (r >= n_reads) { *bad = 1; return; }          /* caller-supplied records: sequenceID outside the batch */
    atomicAdd(&read_cnt[r], 1u);
}

/* ---- segment sort: compareMatches order inside every read segment -------
 * All-ascending bitonic network (partner i^(2k-1) on the first step of a
 * merge, i^j afterwards): with virtual +inf padding behind the segment every
 * compare against an index >= n is a no-op, so n need not be a power of two.
 * Small segments: one wavefront, records staged in LDS.  Large segments
 * (listed in `large`): one 256-thread workgroup, in place in HBM/L2.        */
#define MTB_SEG_LDS 512

template <typename REC>
__device__ __forceinline__ void seg_cmpx(REC *a, uint32_t i, uint32_t j) {
    REC x = a[i], y = a[j];
    if (mtb_match_less(rec_m(y), rec_m(x))) { a[i] = y; a[j] = x; }
}

template <int NT, typename REC>
__device__ __forceinline__ void seg_bitonic(REC *a, uint32_t n, uint32_t tid) {
    uint32_t p2 = 1; while (p2 < n) p2 <<= 1;
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t t = tid; t < (p2 >> 1); t += NT) {
                /* t-th compare-exchange pair of this step */
                uint32_t lo = ((t / j) * (j << 1)) + (t % j);
                uint32_t hi = (j == (k >> 1)) ? (lo ^ ((j << 1) - 1)) : (lo + j);
                if (hi < lo) { uint32_t z = lo; lo = hi; hi = z; }
                if (hi < n) seg_cmpx(a, lo, hi);
            }
            __syncthreads();
        }
    }
}

__global__ __launch_bounds__(64) void k_segsort_small(mtb_match *__restrict__ m, const uint64_t *__restrict__ seg_start, uint64_t n_reads,
                                                       uint32_t *__restrict__ large, uint32_t *__restrict__ n_large, uint32_t *__restrict__ max_seg) {
    __shared__ mtb_match s_m[MTB_SEG_LDS];
    uint32_t my_max = 0;
    for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        uint64_t s = seg_start[r];
        uint64_t n64 = seg_start[r + 1] - s;
        uint32_t n = (uint32_t)n64;
        my_max = n > my_max ? n : my_max;
        if (n < 2) continue;
        if (n > MTB_SEG_LDS) { if (threadIdx.x == 0) large[atomicAdd(n_large, 1u)] = (uint32_t)r; continue; }
        const uint64_t *src = (const uint64_t *)(m + s);
        uint64_t *dstl = (uint64_t *)s_m;
        for (uint32_t i = threadIdx.x; i < n * 3; i += 64) dstl[i] = src[i];
        __syncthreads();
        seg_bitonic<64, mtb_match>(s_m, n, threadIdx.x);
        uint64_t *dstg = (uint64_t *)(m + s);
        for (uint32_t i = threadIdx.x; i < n * 3; i += 64) dstg[i] = dstl[i];
        __syncthreads();
    }
    if (max_seg && threadIdx.x == 0 && my_max) atomicMax(max_seg, my_max);
}

template <typename REC>
__global__ __launch_bounds__(256) void k_segsort_large(REC *__restrict__ m, const uint64_t *__restrict__ seg_start,
                                                        const uint32_t *__restrict__ large, const uint32_t *__restrict__ n_large) {
    uint32_t nl = *n_large;
    for (uint32_t b = blockIdx.x; b < nl; b += gridDim.x) {
        uint32_t r = large ? large[b] : b;
        uint64_t s = seg_start[r];
        uint32_t n = (uint32_t)(seg_start[r + 1] - s);
        seg_bitonic<256, REC>(m + s, n, threadIdx.x);
    }
}

#endif
