/* kernels_score.h -- per-read match extension and scoring, one wavefront per
 * read (Classifier::assignTaxonomy -> Taxonomer::chooseBestTaxon,
 * src/commons/Classifier.cpp:166-208, src/commons/Taxonomer.cpp:130-699).
 *
 * The read's sorted match segment is staged in LDS (or, for segments larger
 * than MTB_SCORE_LDS matches, in a per-workgroup slab in HBM).  Lanes then
 * work on independent pieces of the reference's nested loops:
 *   phase 1  one lane per (species, frame) block: chain DP (getMatchPaths);
 *            every match owns the slot of the only path it can end;
 *   phase 2  one lane per species block: stable order + greedy combination of
 *            its paths (combineMatchPaths), score at the block's first slot;
 *   phase 3  lane 0: best species / ties -> LCA, redundancy filter, sub-species
 *            descent (mtb_read_decide).
 * Algorithmic HBM bytes: 24 per match read + 24 per read result written.     */
#ifndef MTB_KERNELS_SCORE_H
#define MTB_KERNELS_SCORE_H
#include "dev_util.h"
#include "mtb_core.h"
#include "kernels_join.h"

/* profiling build only (-DMTB_SCORE_PHASE_CYCLES): cycles per phase of k_score,
 * accumulated into mtb_phase_cycles[4] = {stage+sort, paths, combine, decide} */
#ifdef MTB_SCORE_PHASE_CYCLES
__device__ unsigned long long mtb_phase_cycles[4];
#define MTB_PHASE_BEGIN() unsigned long long ph_t0_ = __builtin_readcyclecounter()
#define MTB_PHASE_MARK(k) do { unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) atomicAdd(&mtb_phase_cycles[k], t_ - ph_t0_); ph_t0_ = t_; } while (0)
#else
#define MTB_PHASE_BEGIN() do {} while (0)
#define MTB_PHASE_MARK(k) do {} while (0)
#endif

#define MTB_SCORE_LDS 192        /* matches per read staged in LDS            */
#define MTB_SCORE_BKT 128        /* position buckets / taxCnt entries in LDS  */

/* bytes of slab one workgroup needs for a segment of n matches, nb buckets */
__host__ __device__ __forceinline__ uint64_t score_slab_bytes(uint64_t n, uint64_t nb) {
    uint64_t b = n * (sizeof(mtb_match) + sizeof(mtb_path) + 4 + 4 + 4) + ((n + 7) & ~7ull);   /* m, path, order, acc, sps, flag */
    b += nb * (4 + 4 + 4) + ((nb + 7) & ~7ull);                                                /* b_tax, o_tax, o_cnt, b_ham   */
    return (b + 63) & ~63ull;
}

__global__ __launch_bounds__(64) void k_taxcnt_bound(const uint64_t *__restrict__ seg_start, const int32_t *__restrict__ qlen,
                                                      const int32_t *__restrict__ qlen2, uint64_t n_reads, int32_t dna_shift,
                                                      uint32_t *__restrict__ bound) {
    uint64_t r = (uint64_t)blockIdx.x * 64 + threadIdx.x;
    if (r >= n_reads) return;
    uint64_t n = seg_start[r + 1] - seg_start[r];
    uint64_t nb = (uint64_t)mtb_num_buckets(qlen[r] + qlen2[r], dna_shift);
    bound[r] = (uint32_t)(n < nb ? n : nb);
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

/* One read, all storage in ONE address space per call site (LDS or HBM slab)
 * so that the compiler emits ds_* / global_* instead of flat accesses.       */
template <bool SORT>
__device__ __forceinline__ void score_read_body(const mtb_match *__restrict__ src, int32_t n, mtb_match *m, mtb_path *path,
                                                int32_t *order, int32_t *acc, float *sps, uint8_t *flag, int32_t *btax,
                                                uint8_t *bham, int32_t *otax, uint32_t *ocnt, int32_t nb, int32_t read_len,
                                                const mtb_tax_view &tx, const mtb_score_params &sp, uint64_t tc_off, uint64_t tc_room,
                                                int32_t *__restrict__ tc_tax, uint32_t *__restrict__ tc_cnt, uint64_t tc_cap,
                                                mtb_match *__restrict__ sorted_out, mtb_result &R) {
    const int32_t lane = (int32_t)threadIdx.x;
    MTB_PHASE_BEGIN();
    if ((const mtb_match *)m != src) {        /* stage the segment (24-byte records as 3 x u64, coalesced) */
        const uint64_t *s64 = (const uint64_t *)src;
        uint64_t *d64 = (uint64_t *)m;
        for (int32_t i = lane; i < n * 3; i += 64) d64[i] = s64[i];
    }
    for (int32_t i = lane; i < n; i += 64) { flag[i] = 0; sps[i] = -1.0f; }
    __syncthreads();
    if (SORT) {
        seg_bitonic<64>(m, (uint32_t)n, (uint32_t)lane);
        if (sorted_out) {
            const uint64_t *s64 = (const uint64_t *)m;
            uint64_t *d64 = (uint64_t *)sorted_out;
            for (int32_t i = lane; i < n * 3; i += 64) d64[i] = s64[i];
        }
    }
    MTB_PHASE_MARK(0);
    /* phase 1: (species, frame) blocks */
    for (int32_t i = lane; i < n; i += 64) {
        int32_t spc = m[i].species_id; uint32_t fr = mtb_q_frame(m[i].qinfo);
        bool head = (i == 0) || m[i - 1].species_id != spc || mtb_q_frame(m[i - 1].qinfo) != fr;
        if (!head) continue;
        int32_t e = i + 1;
        while (e < n && m[e].species_id == spc && mtb_q_frame(m[e].qinfo) == fr) e++;
        if (e - i > 1) {      /* Taxonomer.cpp:342 */
            int32_t md = (spc >= 0 && spc <= tx.max_taxid && tx.under_euk[spc]) ? sp.min_cons_cnt_euk : sp.min_cons_cnt;
            mtb_sf_block_paths(m, i, e, path, flag, &sp, md);
        }
    }
    __syncthreads();
    MTB_PHASE_MARK(1);
    /* phase 2: species blocks */
    for (int32_t i = lane; i < n; i += 64) {
        int32_t spc = m[i].species_id;
        bool head = (i == 0) || m[i - 1].species_id != spc;
        if (!head) continue;
        int32_t e = i + 1;
        while (e < n && m[e].species_id == spc) e++;
        int32_t np = 0;
        float sc = mtb_species_combine(m, i, e, path, flag, order, acc, read_len, &np);
        if (np > 0) sps[i] = sc < 1.0f ? sc : 1.0f;        /* Taxonomer.cpp:356 */
    }
    __syncthreads();
    MTB_PHASE_MARK(2);
    /* phase 3: decision.  Lane 0 picks the species; the redundancy filter runs
     * one lane per position bucket (their LCA chains are independent). */
    int32_t bs = 0, be = 0, species = 0, go = 0;
    if (lane == 0) go = mtb_read_select(m, n, sps, &tx, &sp, &R, &bs, &be, &species) ? 1 : 0;
    go = __shfl(go, 0, 64);
    if (go) {
        bs = __shfl(bs, 0, 64); be = __shfl(be, 0, 64);
        for (int32_t q = lane; q < nb; q += 64) {
            int32_t t;
            bool used = mtb_filter_bucket(m, bs, be, &tx, sp.dna_shift, q, &t);
            btax[q] = t; bham[q] = used ? 0 : 255;
        }
        __syncthreads();
        if (lane == 0) {
            int32_t ntc = mtb_taxcnt_gather(btax, bham, nb, otax, ocnt, (int32_t)tc_room);
            mtb_read_finish(&tx, &sp, species, read_len, otax, ocnt, ntc, &R);
            R.taxcnt_off = (uint32_t)tc_off;
            for (int32_t k = 0; k < ntc; k++)
                if (tc_off + k < tc_cap) { tc_tax[tc_off + k] = otax[k]; tc_cnt[tc_off + k] = ocnt[k]; }
        }
    }
    __syncthreads();
    MTB_PHASE_MARK(3);
}

/* SORT = true: segments arrive grouped by read but unordered (fused path): the
 * wave sorts the staged segment in LDS first (compareMatches order) and, if
 * sorted_out != NULL, writes it back.  Segments larger than MTB_SCORE_LDS must
 * already be sorted in HBM (k_segsort_large).                               */
template <bool SORT>
__global__ __launch_bounds__(64) void k_score(const mtb_match *__restrict__ matches, const uint64_t *__restrict__ seg_start,
                                               uint64_t n_reads, const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2,
                                               mtb_tax_view tx, mtb_score_params sp, const uint64_t *__restrict__ tc_off,
                                               mtb_result *__restrict__ results, int32_t *__restrict__ tc_tax,
                                               uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, uint8_t *__restrict__ slabs,
                                               uint64_t slab_bytes, uint32_t slab_max_n, uint32_t slab_max_nb,
                                               mtb_match *__restrict__ sorted_out) {
    __shared__ mtb_match s_m[MTB_SCORE_LDS];
    __shared__ mtb_path s_path[MTB_SCORE_LDS];
    __shared__ int32_t s_order[MTB_SCORE_LDS];
    __shared__ int32_t s_acc[MTB_SCORE_LDS];
    __shared__ float s_sps[MTB_SCORE_LDS];
    __shared__ uint8_t s_flag[MTB_SCORE_LDS];
    __shared__ int32_t s_btax[MTB_SCORE_BKT];
    __shared__ int32_t s_otax[MTB_SCORE_BKT];
    __shared__ uint32_t s_ocnt[MTB_SCORE_BKT];
    __shared__ uint8_t s_bham[MTB_SCORE_BKT];
    const uint32_t lane = threadIdx.x;
    for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        const uint64_t s0 = seg_start[r];
        const int32_t n = (int32_t)(seg_start[r + 1] - s0);
        const int32_t ql1 = qlen[r], ql2 = qlen2[r];
        const int32_t read_len = ql1 + ql2;
        mtb_result R;
        R.classification = 0; R.score = 0.0f; R.query_length = ql1; R.query_length2 = ql2;
        R.is_classified = 0; R.reserved = 0; R.n_taxcnt = 0; R.taxcnt_off = 0;
        if (n == 0) { if (lane == 0) results[r] = R; continue; }
        const int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        const bool big = n > MTB_SCORE_LDS || nb > MTB_SCORE_BKT;
        const uint64_t off = tc_off[r], room = tc_off[r + 1] - off;
        if (!big) {
            score_read_body<SORT>(matches + s0, n, s_m, s_path, s_order, s_acc, s_sps, s_flag, s_btax, s_bham, s_otax, s_ocnt, nb,
                                  read_len, tx, sp, off, room, tc_tax, tc_cnt, tc_cap, sorted_out ? sorted_out + s0 : nullptr, R);
        } else {
            if ((uint32_t)n > slab_max_n || (uint32_t)nb > slab_max_nb) {      /* cannot happen: slabs are sized from the maxima */
                if (lane == 0) { R.reserved = 0xFF; results[r] = R; }
                continue;
            }
            uint8_t *slab = slabs + (uint64_t)blockIdx.x * slab_bytes;
            uint64_t N = slab_max_n, B = slab_max_nb;
            mtb_match *m = (mtb_match *)slab; mtb_path *path = (mtb_path *)(m + N); int32_t *order = (int32_t *)(path + N);
            int32_t *acc = order + N; float *sps = (float *)(acc + N); uint8_t *flag = (uint8_t *)(sps + N);
            uint8_t *p = slab + N * (sizeof(mtb_match) + sizeof(mtb_path) + 12) + ((N + 7) & ~7ull);
            int32_t *btax = (int32_t *)p; int32_t *otax = btax + B; uint32_t *ocnt = (uint32_t *)(otax + B); uint8_t *bham = (uint8_t *)(ocnt + B);
            /* big segments are pre-sorted in HBM; a small segment of a long read still needs its sort */
            if (SORT && n <= MTB_SCORE_LDS)
                score_read_body<true>(matches + s0, n, m, path, order, acc, sps, flag, btax, bham, otax, ocnt, nb, read_len, tx, sp,
                                      off, room, tc_tax, tc_cnt, tc_cap, sorted_out ? sorted_out + s0 : nullptr, R);
            else
                score_read_body<false>(matches + s0, n, m, path, order, acc, sps, flag, btax, bham, otax, ocnt, nb, read_len, tx, sp,
                                       off, room, tc_tax, tc_cnt, tc_cap, (mtb_match *)nullptr, R);
        }
        if (lane == 0) { R.query_length = ql1; R.query_length2 = ql2; R.reserved = 0; results[r] = R; }
    }
}

#endif
