/* kernels_score_fast.h -- register-resident scorer for the common short read (fused path, slot segments).
 *
 * k_score (kernels_score.h) runs Taxonomer::chooseBestTaxon (src/commons/Taxonomer.cpp:130-699) as generic per-element
 * phases over LDS arrays: ~2000 VALU + ~1300 SALU wave instructions per 150 bp read (rocprofv3 SQ counters,
 * profiles/r02_pmc_sq_baseline.tsv), VALU pipes 73 % busy -- it is instruction-issue bound.  Nearly all short reads have the
 * same simple structure, and for that structure the whole decision fits in registers:
 *
 *   (S1) the read's live slots, taken species by species in slot order, are in compareMatches order (species, frame, position,
 *        hamming, dna) -- the extractor numbers a read's metamers in (frame, rising position) order, so this holds whenever
 *        every metamer has at most one match per species;
 *   (S2) every position group holds one match (no two matches share species, frame and position);
 *   (S3) at most 64 paths are emitted.
 *
 * Then a match can only link to the slot before it, connectedToNext(i) is "i+1 links to i", the chain DP of
 * getMatchPaths (:487-648) is one segmented prefix sum over the slots (increments are multiples of 0.5 / small integers:
 * exact in fp32 in any association), and only the EMITTED paths are ever materialised (a handful per
 * read): they are sorted per species by an all-pairs rank over at most 64 lanes (stable: ties by emission order =
 * combineMatchPaths' insertion sort, :410-426) and combined by the reference's greedy loop executed wave-uniformly, the
 * accepted paths of the current species living one per lane (:428-468, trimMatchPath :475-485).  Species selection, the
 * redundancy filter, the taxCnt gather and the sub-species descent follow k_score (same helper functions).
 *
 * One wavefront per read, element i of the sorted list in lane i % 64, register slot i / 64.  Reads that violate S1-S3
 * (or overflow their tail, or have too many position buckets) are flagged in `slow_flag` and scored by k_score<SLOT>
 * afterwards.  Needs the 64-bit sort key (taxonomy ids < 2^22, positions < 2^11: checked by the host).
 * Algorithmic HBM bytes as k_score: 24 per match + 16 per read.                                                      */
#ifndef MTB_KERNELS_SCORE_FAST_H
#define MTB_KERNELS_SCORE_FAST_H
#include "dev_util.h"
#include "kernels_score.h"
#include "mtb_core.h"
#include "mtb_score_par.h"

#define MTB_FAST_BKT 128          /* position buckets handled in LDS */
#ifndef MTB_FAST_MERGE_SINGLES
#define MTB_FAST_MERGE_SINGLES 0  /* A/B switch (make libmtb_xmerge.so X=-DMTB_FAST_MERGE_SINGLES=1): the tail merge for single reads too */
#endif

/* debugging build only (-DMTB_FAST_DEBUG): why reads leave the fast path: 0 tail overflow / buckets, 1 > 8 species, 2 not sorted
 * (S1), 3 position group with two matches (S2), 4 > 64 paths (S3), 5 handled; 6 sum of emitted paths, 7 sum of species */
#ifdef MTB_FAST_DEBUG
__device__ unsigned long long mtb_fast_reasons[32];          /* [8..]: cycles per phase (wave 0 of every workgroup = every wave) */
#define MTB_FAST_COUNT(k, v) do { if (threadIdx.x == 0) atomicAdd(&mtb_fast_reasons[k], (unsigned long long)(v)); } while (0)
#define MTB_FAST_T0() unsigned long long ft_ = __builtin_readcyclecounter()
#define MTB_FAST_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); if (threadIdx.x == 0) s_ph[k] += t_ - ft_; ft_ = t_; } while (0)
#else
#define MTB_FAST_COUNT(k, v) do {} while (0)
#define MTB_FAST_T0() do {} while (0)
#define MTB_FAST_MARK(k) do {} while (0)
#endif

__device__ __forceinline__ int32_t rl_i(int32_t v, int32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float rl_f(float v, int32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ void wave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

/* wave reductions with DPP row shifts / broadcasts (6 VALU + 1 readlane; a __shfl_xor butterfly is 6 ds_bpermute round trips).
 * row_shr:n = 0x110+n inside rows of 16 lanes, row_bcast:15 = 0x142 (rows 1 and 3), row_bcast:31 = 0x143 (rows 2 and 3); lanes a
 * step does not reach combine their value with itself.  The result is wave-uniform. */
#define MTB_DPP_RED(v, OP) do { \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x111, 0xF, 0xF, false); (v) = OP((v), o_); } \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x112, 0xF, 0xF, false); (v) = OP((v), o_); } \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x114, 0xF, 0xF, false); (v) = OP((v), o_); } \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x118, 0xF, 0xF, false); (v) = OP((v), o_); } \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x142, 0xA, 0xF, false); (v) = OP((v), o_); } \
    { const uint32_t o_ = (uint32_t)__builtin_amdgcn_update_dpp((int)(v), (int)(v), 0x143, 0xC, 0xF, false); (v) = OP((v), o_); } } while (0)
#define MTB_MINU(a, b) ((b) < (a) ? (b) : (a))
#define MTB_MAXU(a, b) ((b) > (a) ? (b) : (a))
__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) { MTB_DPP_RED(v, MTB_MINU); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) { MTB_DPP_RED(v, MTB_MAXU); return (uint32_t)__builtin_amdgcn_readlane((int)v, 63); }

/* The compareMatches key of a match as two 32-bit words (all arithmetic on it is 32-bit: the kernel is VALU bound):
 *   hi = species[10..31] frame[7..9] position >> 4 [0..6]        lo = position & 15 [27..30] hamming[24..26] dna[0..23]
 * (hi, lo) compared lexicographically = (species, frame, position, hamming, dna).  Needs species < 2^22, positions < 2^11. */
struct FKey { uint32_t lo, hi; };
__device__ __forceinline__ uint32_t fk_species(const FKey &k) { return k.hi >> 10; }
__device__ __forceinline__ uint32_t fk_frame(const FKey &k) { return (k.hi >> 7) & 7u; }
__device__ __forceinline__ uint32_t fk_pos(const FKey &k) { return ((k.hi & 127u) << 4) | (k.lo >> 27); }
__device__ __forceinline__ uint32_t fk_ham(const FKey &k) { return (k.lo >> 24) & 7u; }
__device__ __forceinline__ uint32_t fk_dna(const FKey &k) { return k.lo & 0xFFFFFFu; }

/* emitted path, one per lane during the combination */
struct FastPath { int32_t start, end; float score; int32_t ham; uint32_t rehs; /* start_reh | end_reh << 16 */ int32_t species; };

/* K = elements per lane after the compaction (n <= 64 K matches), KL = slots loaded per lane (stride <= 64 KL).  BYFRAME: pairs.
 * The slot order of a pair is (mate, frame, position), compareMatches order is (species, frame, position) with the second mate's
 * positions behind the first mate's: the compaction then takes one (species, frame) after another instead of one species. */
template <int K, int KL = K, bool BYFRAME = false>
__global__ __launch_bounds__(64, K <= 3 ? 5 : 4) void k_score_fast(const mtb_slot16 *__restrict__ slots_all, uint64_t n_reads, const int32_t *__restrict__ qlen,
                                                       const int32_t *__restrict__ qlen2, mtb_tax_view tx, mtb_score_params sp,
                                                       const uint64_t *__restrict__ tc_off, mtb_result *__restrict__ results,
                                                       int32_t *__restrict__ tc_tax, uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, uint64_t tc_base,
                                                       const uint32_t *__restrict__ cursor, uint32_t stride, uint32_t direct, uint32_t epoch,
                                                       uint8_t *__restrict__ slow_flag, uint32_t *__restrict__ cnt_out) {
    constexpr int NMAX = 64 * K;
    __shared__ FKey s_key[NMAX];                /* sort keys of the compacted slots                     */
    __shared__ uint64_t s_aux[NMAX];            /* target id | right_end_hamming << 32                  */
    __shared__ uint64_t s_pp[NMAX];             /* prefix sums of the chain increments: score | hd << 32 */
    __shared__ FastPath s_path[64];
    __shared__ uint32_t s_pf[64];              /* landing zone of the slot prefetch (never read) */
    __shared__ uint32_t s_tl[(BYFRAME || MTB_FAST_MERGE_SINGLES) ? 64 : 1];     /* tail matches of the read: their places in the compacted list */
    /* The wave's time is memory round trips (slot records, three rounds of taxonomy lookups): what bounds the kernel is the number of
     * waves a CU holds, and LDS is what limited it (10.1 KB: 16 waves; 97 registers: 4 per SIMD).  The small tables of the phases
     * before the keys exist and after the last path is emitted therefore live inside the big arrays (a wave fence separates every two
     * phases; the arrays of one phase are distinct):
     *   compaction:            s_hcnt (matches per species hash)                     in s_pp   (written by the chain prefix sums)
     *   filter, gather:        s_hmin / s_btax (position buckets)                    in s_key  (last read by the emission)
     *   gather .. output:      s_otax / s_ocnt (the read's taxID:count list)         in s_pp   (last read by the emission)
     *   descent:               s_lev / s_anc                                         in s_aux  (last read by the emission)           */
    typedef uint32_t __attribute__((may_alias)) u32a; typedef int32_t __attribute__((may_alias)) i32a;
    static_assert(NMAX * 8 >= 2 * MTB_FAST_BKT * 4 && NMAX * 8 >= 256 * 4 && NMAX * 8 >= (MTB_LR_MAXE + MTB_LR_MAXE * MTB_LR_K) * 4, "tables fit the arrays they live in");
    u32a *const s_hcnt = (u32a *)s_pp;
    u32a *const s_hmin = (u32a *)s_key; i32a *const s_btax = (i32a *)s_key + MTB_FAST_BKT;
    i32a *const s_otax = (i32a *)s_pp; u32a *const s_ocnt = (u32a *)s_pp + MTB_FAST_BKT;
    i32a *const s_lev = (i32a *)s_aux; i32a *const s_anc = (i32a *)s_aux + MTB_LR_MAXE;
    const int32_t lane = (int32_t)threadIdx.x;
    MTB_BEGIN_ACQUIRE();
#ifdef MTB_FAST_DEBUG
    __shared__ unsigned long long s_ph[16];
    if (threadIdx.x < 16) s_ph[threadIdx.x] = 0;
#endif
    const uint64_t lt = lanemask_lt();
    const uint64_t le = lt | (1ull << lane);
    const uint32_t tail_cap = stride - direct;
    const uint32_t div_m = (65536u + (uint32_t)sp.dna_shift - 1u) / (uint32_t)sp.dna_shift;     /* pos / dna_shift = pos * div_m >> 16 for pos < 2^11 */
    const uint32_t lines = (stride * (uint32_t)sizeof(mtb_slot16) + 127u) / 128u;

    for (uint64_t r = blockIdx.x; r < n_reads; r += gridDim.x) {
        if (r + gridDim.x < n_reads && (uint32_t)lane < lines) {       /* pull the next read's slot lines towards L2 (see k_score) */
            const uint8_t *pf = (const uint8_t *)(slots_all + (r + gridDim.x) * (uint64_t)stride) + (uint64_t)lane * 128u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)pf, (__attribute__((address_space(3))) void *)s_pf, 4, 0, 0);
        }
        MTB_FAST_T0();
        const int32_t ql1 = qlen[r], ql2 = qlen2[r];
        const int32_t read_len = ql1 + ql2;
        const int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        const uint32_t cur = cursor[r];
        const mtb_slot16 *slots = slots_all + r * (uint64_t)stride;
        mtb_slot16 x[KL];
#pragma unroll
        for (int k = 0; k < KL; k++) { const uint32_t i = (uint32_t)lane + 64u * k; x[k].a = 0; x[k].b = 0; if (i < stride) x[k] = slots[i]; }
        bool slow = cur > tail_cap || nb > MTB_FAST_BKT;
        wave_fence();                                   /* previous read's LDS traffic is complete */
        MTB_FAST_MARK(0);       /* setup + slot loads issued */
        /* ---- compaction of the live slots -> keys / aux in LDS, ordered by (species, slot order).  A stray match of a foreign
         * species (a filler hit of a metamer that carries a read error) is what usually breaks the slot order: the species are
         * taken one after another in ascending order (a wave minimum per species, 1-3 of them), each species' matches keep their
         * slot order -- which is (frame, position) order when the read has one match per metamer and species.  Whether the result
         * really is compareMatches order is checked on the keys below (S1). ---- */
        int32_t n = 0, n_live = 0;
        uint32_t spc[KL]; bool live[KL]; int32_t dst[KL];
        /* A match whose species has no other match in the read cannot be part of a path (a (species, frame) block needs two position
         * groups, Taxonomer.cpp:342 and the walk of getMatchPaths), so its species never gets a score and the match is never looked
         * at again: it is dropped here.  With an index full of foreign species most stray hits are of this kind (every hit of a
         * read of an unknown organism, typically), and they are what breaks the slot order.  Lonely = alone in its bucket of a
         * 256-entry species hash (a collision only keeps a droppable match). */
        uint32_t smin = 0xFFFFFFFFu, smax = 0u;
#pragma unroll
        for (int k = 0; k < KL; k++) {
            const uint32_t i = (uint32_t)lane + 64u * k;
            live[k] = i < stride && mtb_slot_epoch(x[k]) == epoch && (i < direct || i - direct < cur);
            spc[k] = live[k] ? (uint32_t)(x[k].a >> 32) : 0xFFFFFFFFu;
            dst[k] = 0;
            n_live += (int32_t)__popcll(__ballot(live[k]));
            smin = spc[k] < smin ? spc[k] : smin;
            if (live[k]) smax = spc[k] > smax ? spc[k] : smax;
        }
        smin = wave_min_u32(smin); smax = wave_max_u32(smax);
        uint32_t todo_min = 0xFFFFFFFFu;
        if (smin != smax) {                         /* a read of one species (about half of them) needs neither the hash nor more than one round */
            for (int32_t q = lane; q < 256; q += 64) s_hcnt[q] = 0;
            wave_fence();
#pragma unroll
            for (int k = 0; k < KL; k++) if (live[k]) atomicAdd(&s_hcnt[(spc[k] * 0x9E3779B1u) >> 24], 1u);
            wave_fence();
#pragma unroll
            for (int k = 0; k < KL; k++) if (live[k] && s_hcnt[(spc[k] * 0x9E3779B1u) >> 24] < 2u) { live[k] = false; spc[k] = 0xFFFFFFFFu; }
        } else if (n_live < 2) {
#pragma unroll
            for (int k = 0; k < KL; k++) { live[k] = false; spc[k] = 0xFFFFFFFFu; }        /* a single match is lonely too */
        }
        if (BYFRAME) {                              /* round key: species, frame (species < 2^22 on this path) */
#pragma unroll
            for (int k = 0; k < KL; k++) if (live[k]) spc[k] = (spc[k] << 3) | ((uint32_t)(x[k].b >> 52) & 7u);
        }
#pragma unroll
        for (int k = 0; k < KL; k++) todo_min = spc[k] < todo_min ? spc[k] : todo_min;
        for (int round = 0; ; round++) {
            const uint32_t m = wave_min_u32(todo_min);
            if (m == 0xFFFFFFFFu) break;
            if (round == 24) { slow = true; break; }   /* many species: the generic kernel sorts */
            todo_min = 0xFFFFFFFFu;
#pragma unroll
            for (int k = 0; k < KL; k++) {
                const bool mine = spc[k] == m;
                const uint64_t mask = __ballot(mine);
                if (mine) { dst[k] = n + (int32_t)__popcll(mask & lt); spc[k] = 0xFFFFFFFFu; }
                n += (int32_t)__popcll(mask);
                todo_min = spc[k] < todo_min ? spc[k] : todo_min;
            }
        }
        if (KL > K && n > NMAX) slow = true;        /* more matches than the register-resident part holds */
#pragma unroll
        for (int k = 0; k < KL; k++) {
            if (live[k] && !slow) {
                const uint64_t b = x[k].b;
                const uint32_t bh = (uint32_t)(b >> 32), ps_ = (bh >> 8) & 0x7FFu;          /* slot word b: epoch ham[23..26] frame[20..22] pos[8..19] | reh dna */
                FKey fk; fk.hi = ((uint32_t)(x[k].a >> 32) << 10) | (((bh >> 20) & 7u) << 7) | (ps_ >> 4);
                fk.lo = ((ps_ & 15u) << 27) | (((bh >> 23) & 7u) << 24) | ((uint32_t)b & 0xFFFFFFu);
                s_key[dst[k]] = fk;
                s_aux[dst[k]] = (x[k].a & 0xFFFFFFFFull) | (((b >> 24) & 0xFFFFull) << 32) | ((uint32_t)lane + 64u * k >= direct ? (1ull << 63) : 0ull);      /* bit 63: from a tail slot */
            }
        }
        MTB_FAST_MARK(1);       /* species order + keys to LDS (includes the wait for the slot loads) */
        mtb_result R;
        R.classification = 0; R.score = 0.0f; R.query_length = ql1; R.query_length2 = ql2; R.is_classified = 0; R.reserved = 0; R.n_taxcnt = 0; R.taxcnt_off = (uint32_t)tc_base;
        if (!slow && n == 0) { if (lane == 0) { cnt_out[r] = (uint32_t)n_live; results[r] = R; } continue; }
        wave_fence();
        /* ---- tail matches into their places.  The further matches of a multi-match metamer sit in the read's tail slots, i.e. at the
         * END of their run (species; for pairs species and frame) in the order above, whereas compareMatches wants them by (frame,)
         * position among the run's direct matches -- which are in order among themselves.  So only the tail matches move: a direct
         * match shifts behind the tail matches of its run with a smaller key, a tail match goes to (direct matches of its run with a
         * smaller key) + (tail matches of its run with a smaller key).  A handful of tail matches per read at most (the tail's
         * capacity); reads without any skip this.  Before, 5.2 % of the pairs left the fast path for this reason alone. ---- */
        {
            uint32_t nt = 0;
            bool any_tail = false;
#pragma unroll
            for (int k = 0; k < KL; k++) any_tail |= live[k] && (uint32_t)lane + 64u * k >= direct;
            /* pairs only: there the generic kernel costs 20 ns per read it takes over (320-record staging, 2 waves per SIMD), and the merge
             * pays (12.5 M pairs: generic 21.5 -> 6.8 ms, this kernel 72 -> 77 ms); single reads lose (generic 2.9 -> 0.9 ms, this kernel
             * 22.5 -> 26.5 ms: the merge costs about what the read's whole scoring does) and keep handing such reads over */
            if ((BYFRAME || MTB_FAST_MERGE_SINGLES) && !slow && __any(any_tail)) {
                FKey ke[K]; uint64_t ax[K]; int32_t np[K]; bool tl[K];
                const int32_t nsl = (n + 63) >> 6;
#pragma unroll
                for (int k = 0; k < K; k++) {
                    const int32_t i = lane + 64 * k;
                    ke[k].lo = 0; ke[k].hi = 0; ax[k] = 0; np[k] = i; tl[k] = false;
                    if (k < nsl && i < n) { ke[k] = s_key[i]; ax[k] = s_aux[i]; tl[k] = (ax[k] >> 63) != 0; }
                    if (k < nsl) {
                        const uint64_t tm = __ballot(tl[k]);
                        if (tl[k]) { const uint32_t at = nt + (uint32_t)__popcll(tm & lt); if (at < 64u) s_tl[at] = (uint32_t)i; }
                        nt += (uint32_t)__popcll(tm);
                    }
                }
                if (nt > 64u) slow = true;
                wave_fence();
                if (!slow) {
                    auto same_run = [&](const FKey &a, const FKey &b) { return BYFRAME ? ((a.hi ^ b.hi) >> 7) == 0 : ((a.hi ^ b.hi) >> 10) == 0; };
                    auto less = [&](const FKey &a, const FKey &b) { return a.hi < b.hi || (a.hi == b.hi && a.lo < b.lo); };
                    for (uint32_t t = 0; t < nt; t++) {
                        const int32_t ti = (int32_t)s_tl[t];
                        const FKey kt = s_key[ti];
                        /* direct matches of the run: those with a greater key move one place back; counts for the tail match itself */
                        uint32_t n_dir_less = 0, n_dir_run = 0;
#pragma unroll
                        for (int k = 0; k < K; k++) {
                            if (k < nsl) {
                                const int32_t i = lane + 64 * k;
                                const bool dir = i < n && !tl[k] && same_run(ke[k], kt);
                                const bool smaller = dir && less(ke[k], kt);
                                if (dir && !smaller) np[k]++;
                                n_dir_run += (uint32_t)__popcll(__ballot(dir)); n_dir_less += (uint32_t)__popcll(__ballot(smaller));
                            }
                        }
                        uint32_t n_tl_less = 0, n_tl_before = 0;
                        for (uint32_t u = 0; u < nt; u++) {
                            if (u == t) continue;
                            const FKey ku = s_key[s_tl[u]];
                            if (!same_run(ku, kt)) continue;
                            n_tl_less += (less(ku, kt) || (!less(kt, ku) && u < t)) ? 1u : 0u; n_tl_before += u < t ? 1u : 0u;      /* equal keys (one metamer value in several targets of a species): tail order breaks the tie, two records never share a place */
                        }
                        const int32_t run_start = ti - (int32_t)n_dir_run - (int32_t)n_tl_before;
#pragma unroll
                        for (int k = 0; k < K; k++) if (lane + 64 * k == ti) np[k] = run_start + (int32_t)(n_dir_less + n_tl_less);
                    }
                    wave_fence();
#pragma unroll
                    for (int k = 0; k < K; k++) { const int32_t i = lane + 64 * k; if (k < nsl && i < n) { s_key[np[k]] = ke[k]; s_aux[np[k]] = ax[k] & ~(1ull << 63); } }
                    wave_fence();
                }
            }
        }
        /* ---- own elements, neighbours, structure checks ---- */
        FKey key[K]; uint32_t tid[K], reh[K];
        int32_t tcanon[K]; uint8_t euk[K];
        const mtb_tax_node *nodes = (const mtb_tax_node *)tx.node;
        bool bhead[K], linked[K];
        int32_t shv[K];
        uint64_t bmask[K + 1], lmask[K + 1], rmask[K];
        const int32_t nslot = (n + 63) >> 6;
        bool bad = false, bad2 = false;
#ifdef MTB_FAST_DEBUG
        if (slow) MTB_FAST_COUNT(cur > tail_cap || nb > MTB_FAST_BKT ? 0 : 1, 1);
#endif
#pragma unroll
        for (int k = 0; k < K; k++) {
            const int32_t i = lane + 64 * k;
            key[k].lo = 0; key[k].hi = 0; tid[k] = 0; reh[k] = 0; bhead[k] = true; linked[k] = false; shv[k] = 0; tcanon[k] = -1; euk[k] = 0;
            bmask[k] = ~0ull; lmask[k] = 0; rmask[k] = 0;
            if (k < nslot && !slow) {
                FKey pk; pk.lo = 0; pk.hi = 0;
                if (i < n) {
                    key[k] = s_key[i]; const uint64_t a = s_aux[i]; tid[k] = (uint32_t)a; reh[k] = (uint32_t)(a >> 32) & 0xFFFFu; if (i > 0) pk = s_key[i - 1];
                    /* taxonomy lookups this match may need later, issued now and all at once: its target's canonical id (redundancy
                     * filter) and whether its species sits under Eukaryota (minimum path depth) -- the scorer's wall time is L2
                     * round trips of such lookups, so they must not queue up behind each other */
                }
                {   /* unconditional loads from clamped indices (a load under a branch is waited for at the branch's end) */
                    const int32_t t_ = (int32_t)tid[k], s_ = (int32_t)fk_species(key[k]);
                    const bool tv = i < n && t_ >= 0 && t_ <= tx.max_taxid, sv = i < n && s_ >= 0 && s_ <= tx.max_taxid;
                    const int32_t tc_ = nodes[tv ? t_ : 0].canon; const uint8_t eu_ = tx.under_euk[sv ? s_ : 0];
                    tcanon[k] = tv ? tc_ : -1; euk[k] = sv ? eu_ : 0;
                }
                const bool first = i == 0;
                if (i < n && !first) {
                    if (pk.hi > key[k].hi || (pk.hi == key[k].hi && pk.lo > key[k].lo)) bad = true;      /* S1: not in compareMatches order */
                    if (pk.hi == key[k].hi && ((pk.lo ^ key[k].lo) >> 27) == 0) bad2 = true;              /* S2: two matches in one position group */
                }
                bhead[k] = i >= n || first || ((pk.hi ^ key[k].hi) >> 7) != 0;
                if (i < n && !bhead[k]) {
                    const int32_t pos = (int32_t)fk_pos(key[k]), ppos = (int32_t)fk_pos(pk);
                    const int32_t s = (pos - ppos) / 3;
                    if (s > 0 && s <= sp.max_codon_shift) {
                        shv[k] = s;
                        const bool fwd = fk_frame(key[k]) < 3u;
                        linked[k] = mtb_consecutive(fk_dna(pk), fk_dna(key[k]), s, fwd, sp.kmer_format);
                    }
                }
                bmask[k] = __ballot(bhead[k]);
                lmask[k] = __ballot(linked[k]);
                rmask[k] = __ballot(i < n && !linked[k]);
            }
        }
        bmask[K] = ~0ull; lmask[K] = 0;
#ifdef MTB_FAST_DEBUG
        if (!slow) { if (__any(bad)) MTB_FAST_COUNT(2, 1); else if (__any(bad2)) MTB_FAST_COUNT(3, 1); }
#endif
        slow = slow || __any(bad || bad2);
        /* flag 2 = the read's tail overflowed: it is listed for the deferred path (k_list_flag2) and the generic kernel does not look at it.
         * (No shared counter here: millions of returning atomics on one address cost tens of ms -- the generic kernel appended 1.1 M such
         * reads to its list one atomic each, 11 of its 12.7 ms.) */
        if (slow) { if (lane == 0) slow_flag[r] = cur > tail_cap ? 2 : 1; continue; }
        if (lane == 0) cnt_out[r] = (uint32_t)n_live;
        MTB_FAST_MARK(2);       /* own elements, flags, links */
        /* ---- chain DP as one segmented prefix sum; candidates for emission ---- */
        float ps[K]; int32_t phd[K]; int32_t root[K];
        bool cand[K];
        {
            float carry_s = 0.0f; int32_t carry_hd = 0, last_root = 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                ps[k] = 0.0f; phd[k] = 0; root[k] = 0; cand[k] = false;
                if (k < nslot) {
                    const int32_t i = lane + 64 * k;
                    float is = 0.0f; int32_t ihd = 0;
                    if (linked[k]) { is = mtb_part_score(reh[k], shv[k], false); ihd = (mtb_part_ham(reh[k], shv[k], false) << 16) | shv[k]; }
                    ps[k] = wave_inclusive_scan_dpp(is) + carry_s; phd[k] = wave_inclusive_scan_dpp(ihd) + carry_hd;
                    carry_s = rl_f(ps[k], 63); carry_hd = rl_i(phd[k], 63);
                    const uint64_t mr = rmask[k];
                    const uint64_t lo_roots = mr & le;
                    root[k] = lo_roots ? 64 * k + 63 - (int32_t)__builtin_clzll(lo_roots) : last_root;
                    if (mr) last_root = 64 * k + 63 - (int32_t)__builtin_clzll(mr);
                    if (i < n) s_pp[i] = (uint64_t)__float_as_uint(ps[k]) | ((uint64_t)(uint32_t)phd[k] << 32);
                    /* MULTI (block with more than one position group) and not connectedToNext */
                    const uint64_t bnext = (bmask[k] >> 1) | ((bmask[k + 1] & 1ull) << 63);
                    const uint64_t lnext = (lmask[k] >> 1) | ((lmask[k + 1] & 1ull) << 63);
                    const bool has_next = !((bnext >> lane) & 1ull);
                    const bool conn = (lnext >> lane) & 1ull;
                    cand[k] = i < n && (!bhead[k] || has_next) && !conn;
                }
            }
        }
        wave_fence();
        MTB_FAST_MARK(3);       /* chain prefix sums */
        /* ---- emitted paths -> one per lane ---- */
        int32_t ne = 0;
        bool too_many = false;
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k < nslot) {
                bool e = false;
                FastPath P; P.start = 0; P.end = 0; P.score = 0.0f; P.ham = 0; P.rehs = 0; P.species = 0;
                if (cand[k]) {
                    const int32_t spc = (int32_t)fk_species(key[k]);
                    const int32_t md = euk[k] ? sp.min_cons_cnt_euk : sp.min_cons_cnt;                  /* IsAncestor(eukaryota, species), Taxonomer.cpp:497-500 */
                    const int32_t rt = root[k];
                    const uint64_t pr = s_pp[rt]; const FKey rkey = s_key[rt];
                    const uint32_t rreh = (uint32_t)(s_aux[rt] >> 32);
                    const int32_t dhd = phd[k] - (int32_t)(uint32_t)(pr >> 32);
                    const int32_t depth = 1 + (dhd & 0xFFFF);
                    if (depth >= md) {
                        e = true;
                        P.start = (int32_t)fk_pos(rkey);
                        P.end = (int32_t)fk_pos(key[k]) + 23;
                        P.score = mtb_part_score(rreh, 8, false) + (rt == lane + 64 * k ? 0.0f : ps[k] - __uint_as_float((uint32_t)pr));
                        P.ham = (int32_t)fk_ham(rkey) + (rt == lane + 64 * k ? 0 : (dhd >> 16));
                        P.rehs = rreh | (reh[k] << 16);
                        P.species = spc;
                    }
                }
                const uint64_t me = __ballot(e);
                const int32_t at = ne + (int32_t)__popcll(me & lt);
                if (e && at < 64) s_path[at] = P;
                ne += (int32_t)__popcll(me);
            }
        }
        too_many = ne > 64;
        if (too_many) MTB_FAST_COUNT(4, 1); else { MTB_FAST_COUNT(5, 1); MTB_FAST_COUNT(6, ne); }
        if (too_many) { if (lane == 0) slow_flag[r] = 1; continue; }       /* S3 (cnt_out is rewritten by k_score) */
        if (ne == 0) { if (lane == 0) results[r] = R; continue; }           /* no species produced a path: unclassified, score 0 (:372-375) */
        wave_fence();
        MTB_FAST_MARK(4);       /* emission */
        /* ---- combination: lane e owns emitted path e ---- */
        FastPath P = s_path[lane < ne ? lane : 0];
        const bool have = lane < ne;
        int32_t prev_spc = __shfl_up(P.species, 1, 64);
        const bool shead = have && (lane == 0 || prev_spc != P.species);
        const uint64_t hmask = __ballot(shead);
        const int32_t nsp = (int32_t)__popcll(hmask);
        MTB_FAST_COUNT(7, nsp);

        /* Species by species (one, for nearly every read).  The greedy loop of combineMatchPaths (:428-468) accepts the species' best
         * path first and never trims it, so a path that overlaps the best one by its whole length or by >= 24 is dropped at the
         * loop's first test whatever follows: those are decided in parallel, and only the survivors -- a handful -- are ranked
         * (stable: score desc, hamming asc, start desc, emission order) and walked through the reference's loop, wave-uniformly,
         * the accepted paths living one per lane. */
        float sp_score = -1.0f; int32_t sp_id = 0;                                               /* lane s: score / id of species s (species with paths only) */
        int32_t s_idx = 0;
        const uint32_t ord = __float_as_uint(P.score);                                           /* path scores are positive: the bit pattern orders them */
        const uint32_t tie = ((uint32_t)(1023 - (P.ham < 1023 ? P.ham : 1023)) << 17) | ((uint32_t)(P.start & 0x7FF) << 6) | (uint32_t)(63 - lane);
        for (uint64_t hm = hmask; hm; hm &= hm - 1, s_idx++) {
            const int32_t slo = (int32_t)__builtin_ctzll(hm);
            const int32_t shi = (hm & (hm - 1)) ? (int32_t)__builtin_ctzll(hm & (hm - 1)) : ne;
            const bool in = lane >= slo && lane < shi;
            const uint32_t mh = wave_max_u32(in ? ord : 0u);
            const uint32_t ml = wave_max_u32(in && ord == mh ? tie : 0u);
            const int32_t bl = 63 - (int32_t)(ml & 63u);                                          /* lane of the best path */
            const int32_t f_st = rl_i(P.start, bl), f_en = rl_i(P.end, bl);
            bool surv = in;
            if (in && lane != bl && !((P.end < f_st) || (f_en < P.start))) {
                const int32_t ov = (P.end < f_en ? P.end : f_en) - (P.start > f_st ? P.start : f_st) + 1;
                if (ov == P.end - P.start + 1 || ov >= 24) surv = false;
            }
            const uint64_t smask = __ballot(surv);
            int32_t rank = 0;
            for (uint64_t fm = smask; fm; fm &= fm - 1) {
                const int32_t f = (int32_t)__builtin_ctzll(fm);
                const float fs = rl_f(P.score, f); const int32_t fh = rl_i(P.ham, f), fst = rl_i(P.start, f);
                const bool before = fs != P.score ? fs > P.score : (fh != P.ham ? fh < P.ham : (fst != P.start ? fst > P.start : f < lane));
                rank += (f != lane && before) ? 1 : 0;
            }
            const int32_t ns = (int32_t)__popcll(smask);
            int32_t acc_st = 0, acc_en = 0, na = 0; float sum = 0.0f;
            for (int32_t q = 0; q < ns; q++) {
                const int32_t o = (int32_t)__builtin_ctzll(__ballot(surv && rank == q));
                int32_t cst = rl_i(P.start, o), cen = rl_i(P.end, o), cham = rl_i(P.ham, o);
                float csc = rl_f(P.score, o);
                const uint32_t crehs = (uint32_t)rl_i((int32_t)P.rehs, o);
                bool drop = false;
                uint64_t ovm = __ballot(lane < na && !((cen < acc_st) || (acc_en < cst)));       /* against the untrimmed candidate: a superset */
                while (ovm && !drop) {
                    const int32_t a2 = (int32_t)__builtin_ctzll(ovm); ovm &= ovm - 1;
                    const int32_t ast = rl_i(acc_st, a2), aen = rl_i(acc_en, a2);
                    if (!((cen < ast) || (aen < cst))) {
                        const int32_t ov = (cen < aen ? cen : aen) - (cst > ast ? cst : ast) + 1;
                        if (ov == cen - cst + 1 || ov >= 24) drop = true;
                        else if (cst < ast) {
                            cen = ast - 1;
                            const int32_t h = cham - mtb_part_ham(crehs >> 16, ov / 3, false); cham = h > 0 ? h : 0;
                            csc = csc - mtb_part_score(crehs >> 16, ov / 3, false) - (float)(ov % 3);
                        } else {
                            cst = aen + 1;
                            const int32_t h = cham - mtb_part_ham(crehs & 0xFFFFu, ov / 3, true); cham = h > 0 ? h : 0;
                            csc = csc - mtb_part_score(crehs & 0xFFFFu, ov / 3, true) - (float)(ov % 3);
                        }
                    }
                }
                if (!drop) { if (lane == na) { acc_st = cst; acc_en = cen; } na++; sum += csc; }
            }
            float sc = sum / (float)read_len; sc = sc < 1.0f ? sc : 1.0f;
            const int32_t spid = rl_i(P.species, slo);
            if (lane == s_idx) { sp_score = sc; sp_id = spid; }
        }
        MTB_FAST_MARK(5);       /* combination */
        /* ---- species decision (getBestSpeciesMatches second half, chooseBestTaxon early exits) ---- */
        const bool valid = lane < nsp && !(sp_score < sp.min_score);
        const uint64_t vmask = __ballot(valid);
        const int32_t meaningful = (int32_t)__popcll(__ballot(valid && sp_score > 0.0f));
        if (meaningful == 0) { if (lane == 0) results[r] = R; continue; }
        float best_sp = valid ? sp_score : -1.0f;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) { const float o2 = __shfl_xor(best_sp, d, 64); best_sp = o2 > best_sp ? o2 : best_sp; }
        const float cut = best_sp * sp.tie_ratio;
        uint64_t tied = __ballot(valid && sp_score >= cut);
        const int32_t n_max = (int32_t)__popcll(tied);
        float tsum = 0.0f; int32_t only = 0, lca = -1, first_spc = 0, cnt_t = 0;
        for (uint64_t tm = tied; tm; tm &= tm - 1) {
            const int32_t s = (int32_t)__builtin_ctzll(tm);
            const int32_t spc = rl_i(sp_id, s);
            tsum += rl_f(sp_score, s); only = spc; cnt_t++;
            if (cnt_t == 1) first_spc = spc;
            else {
                if (cnt_t == 2) lca = mtb_tax_exists(&tx, first_spc) ? mtb_tax_canon(&tx, first_spc) : -1;
                if (mtb_tax_exists(&tx, spc)) lca = lca < 0 ? mtb_tax_canon(&tx, spc) : mtb_lca(&tx, lca, spc);
            }
        }
        (void)vmask;
        const float score = n_max > 1 ? tsum / (float)n_max : tsum;
        R.score = score;
        if (score == 0.0f || score < sp.min_score) { if (lane == 0) results[r] = R; continue; }
        if (n_max > 1) { R.is_classified = 1; R.classification = lca < 0 ? 0 : lca; if (lane == 0) results[r] = R; continue; }
        R.is_classified = 1;
        const int32_t species = only;
        const bool sp_ok = species >= 0 && species <= tx.max_taxid;
        mtb_tax_node rs = nodes[sp_ok ? species : 0];                              /* needed by the descent: in flight during the filter */
        if (!sp_ok) { rs.canon = -1; rs.depth = 0; rs.parent = -1; rs.flags = 0; }
        MTB_FAST_MARK(6);       /* decision */
        /* ---- redundancy filter over the best species' matches (filterRedundantMatches, :205-241) ---- */
        for (int32_t q = lane; q < nb; q += 64) { s_hmin[q] = 255u; s_btax[q] = -1; }
        wave_fence();
        int32_t bq[K];
#pragma unroll
        for (int k = 0; k < K; k++) {
            bq[k] = -1;
            if (k < nslot) {
                const int32_t i = lane + 64 * k;
                if (i < n && (int32_t)fk_species(key[k]) == species) {
                    const int32_t q = (int32_t)((fk_pos(key[k]) * div_m) >> 16);
                    if (q < nb) { bq[k] = q; atomicMin(&s_hmin[q], fk_ham(key[k])); }
                }
            }
        }
        wave_fence();
#pragma unroll
        for (int k = 0; k < K; k++) {
            if (k < nslot && bq[k] >= 0 && fk_ham(key[k]) == s_hmin[bq[k]]) {
                const int32_t t = (int32_t)tid[k];
                int32_t old = atomicCAS(&s_btax[bq[k]], -1, t);              /* the first id of a bucket stays raw */
                while (old != -1) {
                    const int32_t merged = old == t ? (tcanon[k] < 0 ? t : tcanon[k]) : mtb_lca(&tx, old, t);      /* LCA(a, a) = canon(a), already here */
                    if (merged == old) break;
                    const int32_t seen = atomicCAS(&s_btax[bq[k]], old, merged);
                    if (seen == old) break;
                    old = seen;
                }
            }
        }
        wave_fence();
        const uint64_t off = tc_off[r], room = tc_off[r + 1] - off;
        MTB_FAST_MARK(7);       /* filter */
        /* Query::taxCnt: distinct bucket taxa in ascending order with their bucket counts (std::map order); the buckets sit two per
         * lane, one wave minimum per distinct taxon (1-2 for most reads).  Taxonomy ids are < 2^22 here. */
        int32_t ntc = 0;
        {
            uint32_t t0 = 0xFFFFFFFFu, t1 = 0xFFFFFFFFu;
            if (lane < nb && s_hmin[lane] != 255u) t0 = (uint32_t)s_btax[lane];
            if (lane + 64 < nb && s_hmin[lane + 64] != 255u) t1 = (uint32_t)s_btax[lane + 64];
            while (ntc < (int32_t)room) {
                const uint32_t mn = wave_min_u32(t0 < t1 ? t0 : t1);
                if (mn == 0xFFFFFFFFu) break;
                const uint32_t cnt = (uint32_t)__popcll(__ballot(t0 == mn)) + (uint32_t)__popcll(__ballot(t1 == mn));
                if (lane == 0) { s_otax[ntc] = (int32_t)mn; s_ocnt[ntc] = cnt; }
                if (t0 == mn) t0 = 0xFFFFFFFFu;
                if (t1 == mn) t1 = 0xFFFFFFFFu;
                ntc++;
            }
        }
        wave_fence();
        MTB_FAST_MARK(8);       /* gather */
        /* ---- sub-species descent (lowerRankClassification, :252-314) ---- */
        int32_t slow_lr = ntc > MTB_LR_MAXE ? 1 : 0;
        if (!slow_lr && lane < ntc) {
            /* mtb_lr_climb on the per-taxon record: canon, depth and parent of the entry arrive with one load */
            const int32_t tax = s_otax[lane];
            const bool t_ok = tax >= 0 && tax <= tx.max_taxid;
            mtb_tax_node rt = nodes[t_ok ? tax : 0];
            if (!t_ok) { rt.canon = -1; rt.depth = 0; rt.parent = -1; rt.flags = 0; }
            int32_t lv;
            const int32_t cs = rs.canon, c = rt.canon;
            if (cs < 0 || c < 0) lv = -1;
            else {
                const int32_t L = rt.depth - rs.depth;
                if (L < 0) lv = -1;
                else if (L > MTB_LR_K) lv = MTB_LR_K + 1;
                else {
                    int32_t a = c;
                    for (int32_t k = L - 1; k >= 0; k--) { s_anc[lane * MTB_LR_K + k] = a; a = (k == L - 1) ? rt.parent : tx.parent[a]; }
                    lv = (a == cs) ? L : -1;
                }
            }
            s_lev[lane] = lv;
            if (lv > MTB_LR_K) slow_lr = 1;
        }
        slow_lr = __any(slow_lr) ? 1 : 0;
        wave_fence();
        if (lane == 0) {
            R.n_taxcnt = (uint16_t)ntc;
            const int32_t cs = rs.canon;
            if (R.score < sp.min_sp_score) R.classification = (species >= 0 && species <= tx.max_taxid) ? tx.sp_parent[species] : 0;
            else if (slow_lr || cs < 0) R.classification = mtb_lower_rank(&tx, s_otax, s_ocnt, ntc, species, read_len, sp.denominator, sp.accession_level);
            else R.classification = mtb_lr_bfs(s_lev, s_anc, s_ocnt, ntc, cs, read_len, sp.denominator, &tx, sp.accession_level);
            R.taxcnt_off = (uint32_t)(off + tc_base);
            for (int32_t k = 0; k < ntc; k++)
                if (off + k < tc_cap) { tc_tax[off + k] = s_otax[k]; tc_cnt[off + k] = s_ocnt[k]; }
            results[r] = R;
        }
        MTB_FAST_MARK(9);       /* descent + output */
    }
#ifdef MTB_FAST_DEBUG
    __syncthreads();
    if (threadIdx.x < 16) atomicAdd(&mtb_fast_reasons[8 + threadIdx.x], s_ph[threadIdx.x]);
#endif
}

#endif
