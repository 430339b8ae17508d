/* kernels_score_long.h -- scorer for long reads (seq_mode 3): one WORKGROUP per read, streaming over the read's sorted segment.
 *
 * Taxonomer::chooseBestTaxon (src/commons/Taxonomer.cpp:130-699) on a read with thousands of matches.  The generic k_score
 * (kernels_score.h) gives such a read to ONE wavefront that keeps ~15 per-match arrays in an HBM slab and sweeps them once or
 * twice per phase (~1.5 KB of traffic per match, 0.012 of the HBM roofline on 10 kb reads, profiles/r02_notes.md).  Here the
 * read's segment -- in compareMatches order (KmerMatcher.cpp:1149-1166: species, frame, position, hamming, dna) -- is read
 * from HBM about three times (block discovery, path walk, redundancy filter) and nothing per match is ever written back:
 *
 *   blocks   all threads: the (species, frame) blocks with at least two matches, in order (a block of one match has a single
 *            position group, for which getMatchPaths' loop never runs, :342 / :528);
 *   walk     every wave takes blocks from that list and walks them front to back, 64 matches per step (getMatchPaths, :487-648).
 *            Window of position groups of one match each: a match can only link to the lane before it, the DP is a segmented
 *            prefix sum over the window (scores are multiples of 0.5, hamming / depth small integers: exact in fp32 in any
 *            association).  Window with a larger group: the groups are taken one after another, lanes =
 *            matches of the group, predecessors read with readlane (strict > on the score, first predecessor wins: :528-560).
 *            The last complete group of a window is carried into the next one with its path state.  Emitted paths (not
 *            connected to the next group, depth >= minConsCnt, :561-572) go to an LDS list with the index of their end match
 *            = their emission order;
 *   combine  all-pairs rank of the paths by (species; score desc, hamming asc, start desc, emission order) = the stable
 *            insertion sort of combineMatchPaths (:410-426); one wave per species walks the greedy loop (:428-468,
 *            trimMatchPath :475-485), the lanes testing a candidate against 64 accepted paths at a time;
 *   decide   thread 0: best species / ties -> LCA (getBestSpeciesMatches second half, :354-407; chooseBestTaxon :130-165);
 *            redundancy filter over the best species' matches with LDS buckets (filterRedundantMatches, :205-241), Query::taxCnt
 *            by repeated workgroup minima, sub-species descent (lowerRankClassification, :252-314).
 *
 * Reads that exceed the LDS budgets below (paths, blocks, species with paths, position buckets, a position group wider than a
 * window) are flagged in `todo` and left to the generic kernel.  Algorithmic HBM bytes: 24 per match + 16 per read.          */
#ifndef MTB_KERNELS_SCORE_LONG_H
#define MTB_KERNELS_SCORE_LONG_H
#include "dev_util.h"
#include "mtb_core.h"
#include "mtb_score_par.h"

#define MTB_LONG_NT 256
#define MTB_LONG_NW (MTB_LONG_NT / 64)
#define MTB_LONG_MAXP 1024          /* emitted paths of a read                        */
#define MTB_LONG_MAXBLK 1024        /* (species, frame) blocks with >= 2 matches       */
#define MTB_LONG_MAXSP 256          /* species with paths                             */
#define MTB_LONG_TINY 8             /* blocks of up to this many matches (single-match position groups) are walked by one lane each */
#define MTB_LONG_MAXBKT 4096        /* position buckets of the redundancy filter (reads up to ~36 kb with syncmers) */

struct mtb_lpath { int32_t start, end; float score; int32_t ham; uint32_t rehs; /* start reh | end reh << 16 */ int32_t species; uint32_t eidx; uint32_t spare; };
static_assert(sizeof(mtb_lpath) == 32, "32-byte path records");
static_assert(MTB_LONG_MAXP * sizeof(mtb_lpath) >= MTB_LONG_MAXBKT * 8, "the filter's buckets live in the (dead) path storage");

/* profiling build only (make libmtb_xlprof.so X=-DMTB_LONG_PHASE_CYCLES): cycles of thread 0 per phase, summed over the workgroups;
 * dev_score_long prints them.  0 setup, 1 block list, 2 walk, 3 rank, 4 species ranges, 5 combination, 6 decision, 7 filter, 8 taxCnt gather, 9 descent + output */
#ifdef MTB_LONG_PHASE_CYCLES
__device__ unsigned long long mtb_long_cycles[16];
#define MTB_LP_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); lp_acc[k] += t_ - lp_t; lp_t = t_; } while (0)
#else
#define MTB_LP_MARK(k) do {} while (0)
#endif
__device__ __forceinline__ int32_t lrl_i(int32_t v, int32_t l) { return __builtin_amdgcn_readlane(v, l); }
__device__ __forceinline__ float lrl_f(float v, int32_t l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); }
__device__ __forceinline__ void lwave_fence() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); }

/* combineMatchPaths' order inside one species: a before b (ties by emission order = index of the end match) */
__device__ __forceinline__ bool lpath_before(const mtb_lpath &a, const mtb_lpath &b) {
    if (a.score != b.score) return a.score > b.score;
    if (a.ham != b.ham) return a.ham < b.ham;
    if (a.start != b.start) return a.start > b.start;
    return a.eidx < b.eidx;
}

/* MAXBLK / MAXSP: LDS budgets for the (species, frame) blocks of two or more matches and for the species with paths.  Long reads run the
 * defaults; the short reads of conserved genes whose organism is NOT in the index (thousands of matches over a thousand species, a few
 * of them with paths: kernels_score_many.h, k_many_sort) run <2048, 256, 256, 256> (below). */
/* MAXP / MAXBKT: emitted paths and position buckets (the filter's buckets live in the dead path storage).  The short-read instantiation
 * <2048, 256, 256, 256> -- a read of <= 4096 records has at most 2048 blocks of two, its paths need four matches in a row (a handful in a read
 * whose species bring two or three matches each), its position buckets are a few dozen -- takes 27 KB of LDS instead of 70: five workgroups per
 * CU instead of two for a kernel whose SIMDs issue a VALU instruction on a quarter of their cycles (profiles/r06_heldout_4M_pmc_sq.tsv). */
template <int MAXBLK = MTB_LONG_MAXBLK, int MAXSP = MTB_LONG_MAXSP, int MAXP = MTB_LONG_MAXP, int MAXBKT = MTB_LONG_MAXBKT>
__global__ __launch_bounds__(MTB_LONG_NT) void k_score_long(const mtb_match *__restrict__ matches, const uint64_t *__restrict__ seg_start, uint64_t n_reads,
                                                             const int32_t *__restrict__ qlen, const int32_t *__restrict__ qlen2, mtb_tax_view tx, mtb_score_params sp,
                                                             const uint64_t *__restrict__ tc_off, mtb_result *__restrict__ results, int32_t *__restrict__ tc_tax,
                                                             uint32_t *__restrict__ tc_cnt, uint64_t tc_cap, uint64_t tc_base, uint8_t *__restrict__ todo,
                                                             unsigned long long *__restrict__ work, const uint32_t *__restrict__ seg_cnt = nullptr,
                                                             const uint32_t *__restrict__ list = nullptr, uint32_t n_list = 0,
                                                             uint32_t only_flag = 0 /* != 0: only the reads whose todo[] equals it (the flag is cleared first): the second launch, with larger budgets */) {
    static_assert(MAXP * sizeof(mtb_lpath) >= MAXBKT * 8, "the filter's buckets live in the (dead) path storage");
    __shared__ __attribute__((aligned(16))) mtb_lpath s_path[MAXP];
    __shared__ uint16_t s_sidx[MAXP], s_acc[MAXP];
    __shared__ uint32_t s_blk[MAXBLK];
    __shared__ int32_t s_carry[MTB_LONG_NW][64][5];
    __shared__ uint16_t s_splo[MAXSP + 1];
    __shared__ int32_t s_spid[MAXSP];
    __shared__ float s_spsc[MAXSP];
    __shared__ int32_t s_otax[MTB_LR_MAXE]; __shared__ uint32_t s_ocnt[MTB_LR_MAXE];
    __shared__ int32_t s_lev[MTB_LR_MAXE], s_anc[MTB_LR_MAXE * MTB_LR_K];
    __shared__ uint32_t s_red[MTB_LONG_NW];
    __shared__ unsigned long long s_r;
    __shared__ uint32_t s_nblk, s_next, s_npath, s_fail, s_nsp, s_nbig;
    __shared__ int32_t s_go, s_species;
    __shared__ mtb_result s_R;
    const int32_t tid = (int32_t)threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const uint64_t lt = lanemask_lt();
#ifdef MTB_LONG_PHASE_CYCLES
    unsigned long long lp_acc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long lp_t = __builtin_readcyclecounter();
#endif

    for (;;) {
        __syncthreads();
        if (tid == 0) { s_r = atomicAdd(work, 1ull); s_nblk = 0; s_next = 0; s_npath = 0; s_fail = 0; s_nsp = 0; s_go = 0; s_nbig = 0; }
        __syncthreads();
        const uint64_t it = s_r;
        if (it >= (list ? (uint64_t)n_list : n_reads)) break;
        const uint64_t r = list ? (uint64_t)list[it] : it;           /* optional: only the listed reads, their segments indexed by the list slot */
        if (only_flag) {
            if (todo[r] != only_flag) continue;                      /* (every thread reads the flag before thread 0 clears it behind the barrier) */
            __syncthreads();
            if (tid == 0) todo[r] = 0;
        }
        const uint64_t s0 = seg_start[it];
        const int32_t n = seg_cnt ? (int32_t)seg_cnt[it] : (int32_t)(seg_start[it + 1] - s0);      /* (ordered slot segments: the read's records fill the front of its slot range) */
        const mtb_match *m = matches + s0;
        MTB_LP_MARK(9);
        const int32_t ql1 = qlen[r], ql2 = qlen2[r], read_len = ql1 + ql2;
        const int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        const uint64_t off = tc_off[r], room = tc_off[r + 1] - off;
        if (tid == 0) {
            mtb_result R;
            R.classification = 0; R.score = 0.0f; R.query_length = ql1; R.query_length2 = ql2; R.is_classified = 0; R.reserved = 0; R.n_taxcnt = 0; R.taxcnt_off = (uint32_t)tc_base;
            s_R = R;
        }
        if (n < 2) { __syncthreads(); if (tid == 0) results[r] = s_R; continue; }          /* no block of two matches: no path, unclassified */
        if (nb > MAXBKT) { if (tid == 0) todo[r] = 1; continue; }

        MTB_LP_MARK(0);
        /* ---- blocks: heads of the (species, frame) blocks that hold at least two matches, in order ---- */
        /* (a wave reserves the places of its step's heads with one LDS atomic: the list's order does not matter -- a path's place in
         * the emission order is its end match's index -- and the two workgroup barriers per step were an eighth of the kernel) */
        for (int32_t c0 = 0; c0 < n; c0 += MTB_LONG_NT) {
            const int32_t i = c0 + tid;
            bool cand = false;
            if (i + 1 < n) {
                const uint64_t k1 = ((uint64_t)(uint32_t)m[i].species_id << 3) | mtb_q_frame(m[i].qinfo);
                const uint64_t k2 = ((uint64_t)(uint32_t)m[i + 1].species_id << 3) | mtb_q_frame(m[i + 1].qinfo);
                bool head = i == 0;
                if (!head) { const uint64_t k0 = ((uint64_t)(uint32_t)m[i - 1].species_id << 3) | mtb_q_frame(m[i - 1].qinfo); head = k0 != k1; }
                cand = head && k1 == k2;
            }
            const uint64_t cm = __ballot(cand);
            if (cm) {
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_nblk, (uint32_t)__popcll(cm));
                at0 = (uint32_t)__shfl((int)at0, 0, 64);
                if (cand) { const uint32_t at = at0 + (uint32_t)__popcll(cm & lt); if (at < (uint32_t)MAXBLK) s_blk[at] = (uint32_t)i; }
            }
        }
        __syncthreads();
        if (tid == 0 && s_nblk > (uint32_t)MAXBLK) s_fail = 1;
        __syncthreads();
        const uint32_t nblk = s_nblk;
        if (s_fail) { if (tid == 0) todo[r] = 1; continue; }
        if (nblk == 0) { if (tid == 0) results[r] = s_R; continue; }

        MTB_LP_MARK(1);
        /* ---- walk, tiny blocks first: ONE LANE per block of at most MTB_LONG_TINY matches whose position groups are single matches.  A
         * short read of a conserved gene of an organism that is not in the index brings a thousand blocks of two or three matches (one
         * per species and frame); a wave step per block -- one dependent HBM round trip each -- was the kernel's time on such reads
         * (32 ms per 163 k of them).  The recurrence is the window's (a match links to the one before it or opens a chain; the block's
         * matches are walked in order, so the sums are the same sums).  Blocks that are longer or hold a wider group are compacted to
         * the front of the list for the wave walk below. ---- */
        for (uint32_t b0 = 0; b0 < nblk; b0 += MTB_LONG_NT) {
            const uint32_t b = b0 + (uint32_t)tid;
            uint32_t bstart = 0;
            if (b < nblk) bstart = s_blk[b];
            __syncthreads();                                 /* every entry of this step is read: the compaction may overwrite them */
            if (b < nblk) {
                const mtb_match x0 = m[bstart];
                const int32_t species = x0.species_id;
                const uint32_t frame = mtb_q_frame(x0.qinfo);
                uint32_t len = 1, ppos = mtb_q_pos(x0.qinfo);
                bool tiny = true;
                for (;;) {
                    const uint32_t idx = bstart + len;
                    if (idx >= (uint32_t)n) break;
                    const int32_t xs = m[idx].species_id; const uint64_t xq = m[idx].qinfo;
                    if (xs != species || mtb_q_frame(xq) != frame) break;
                    const uint32_t xp = mtb_q_pos(xq);
                    if (xp == ppos || len == MTB_LONG_TINY) { tiny = false; break; }
                    ppos = xp; len++;
                }
                if (!tiny) s_blk[atomicAdd(&s_nbig, 1u)] = bstart;
                else {
                    const bool fwd = frame < 3u;
                    const int32_t md = (species >= 0 && species <= tx.max_taxid && tx.under_euk[species]) ? sp.min_cons_cnt_euk : sp.min_cons_cnt;
                    int32_t p_start = 0, p_ham = 0, p_depth = 0; float p_score = 0.0f; uint32_t p_sreh = 0;
                    uint32_t q_pos = 0, q_dna = 0, q_reh = 0;
                    auto out_path = [&](uint32_t idx, uint32_t pos, uint32_t reh) {
                        const uint32_t at = atomicAdd(&s_npath, 1u);
                        if (at < (uint32_t)MAXP) {
                            mtb_lpath P; P.start = p_start; P.end = (int32_t)pos + 23; P.score = p_score; P.ham = p_ham; P.rehs = (p_sreh & 0xFFFFu) | (reh << 16);
                            P.species = species; P.eidx = idx; P.spare = 0;
                            s_path[at] = P;
                        } else s_fail = 1;
                    };
                    for (uint32_t j = 0; j < len; j++) {
                        const mtb_match x = m[bstart + j];
                        const uint32_t pos = mtb_q_pos(x.qinfo), dna = x.dna, reh = x.right_end_hamming;
                        bool linked = false; int32_t sh = 0;
                        if (j) {
                            const int32_t s_ = (int32_t)(pos - q_pos) / 3;
                            if (s_ > 0 && s_ <= sp.max_codon_shift && mtb_consecutive(q_dna, dna, s_, fwd, sp.kmer_format)) { linked = true; sh = s_; }
                            if (!linked && p_depth >= md) out_path(bstart + j - 1, q_pos, q_reh);        /* the previous match is not connected to this one: its path ends */
                        }
                        if (linked) { p_score += mtb_part_score(reh, sh, false); p_ham += mtb_part_ham(reh, sh, false); p_depth += sh; }
                        else { p_start = (int32_t)pos; p_score = mtb_part_score(reh, 8, false); p_ham = (int32_t)x.hamming; p_depth = 1; p_sreh = reh; }
                        q_pos = pos; q_dna = dna; q_reh = reh;
                    }
                    if (p_depth >= md) out_path(bstart + len - 1, q_pos, q_reh);                        /* the block's last match: nothing follows it */
                }
            }
        }
        __syncthreads();
        const uint32_t nbig = s_nbig;
        /* ---- walk: one wave per block ---- */
        for (;;) {
            uint32_t b = 0;
            if (lane == 0) b = atomicAdd(&s_next, 1u);
            b = (uint32_t)__shfl((int)b, 0, 64);
            if (b >= nbig || s_fail) break;
            const uint32_t bstart = s_blk[b];
            const int32_t species = m[bstart].species_id;
            const uint32_t frame = mtb_q_frame(m[bstart].qinfo);
            const bool fwd = frame < 3u;
            const int32_t md = (species >= 0 && species <= tx.max_taxid && tx.under_euk[species]) ? sp.min_cons_cnt_euk : sp.min_cons_cnt;      /* IsAncestor(eukaryota, species), Taxonomer.cpp:497-500 */
            uint32_t base = bstart; int32_t carry_n = 0; bool multi = false, done = false, fail = false;
            int32_t p_start = 0, p_ham = 0, p_depth = 0; float p_score = 0.0f; uint32_t p_sreh = 0;
            /* emitted paths of this wave's step -> LDS list */
            auto emit = [&](bool e, uint32_t idx, uint32_t pos, uint32_t reh) {
                const uint64_t em = __ballot(e);
                if (!em) return;
                uint32_t at0 = 0;
                if (lane == 0) at0 = atomicAdd(&s_npath, (uint32_t)__popcll(em));
                at0 = (uint32_t)__shfl((int)at0, 0, 64);
                const uint32_t at = at0 + (uint32_t)__popcll(em & lt);
                if (e) {
                    if (at < (uint32_t)MAXP) {
                        mtb_lpath P; P.start = p_start; P.end = (int32_t)pos + 23; P.score = p_score; P.ham = p_ham; P.rehs = (p_sreh & 0xFFFFu) | (reh << 16);
                        P.species = species; P.eidx = idx; P.spare = 0;
                        s_path[at] = P;
                    } else s_fail = 1;
                }
            };
            while (!done && !fail) {
                const uint32_t idx = base + (uint32_t)lane;
                const bool valid = idx < (uint32_t)n;
                uint64_t qinfo = 0; uint32_t dna = 0, reh = 0, ham = 0; int32_t spc = -1;
                if (valid) { const mtb_match x = m[idx]; qinfo = x.qinfo; dna = x.dna; reh = x.right_end_hamming; ham = x.hamming; spc = x.species_id; }
                const bool inblk = valid && spc == species && mtb_q_frame(qinfo) == frame;
                const uint64_t inm = __ballot(inblk);
                const int32_t nin = inm == ~0ull ? 64 : (int32_t)__builtin_ctzll(~inm);         /* the block's matches are a prefix of the window */
                const bool ends = nin < 64;
                const uint32_t pos = mtb_q_pos(qinfo);
                const uint32_t ppos = (uint32_t)__shfl_up((int)pos, 1, 64);
                const bool in = lane < nin;
                const bool gh = in && (lane == 0 || pos != ppos);
                const uint64_t gm = __ballot(gh);
                if (carry_n) {
                    if (lane < carry_n) { p_start = s_carry[wv][lane][0]; p_score = __int_as_float(s_carry[wv][lane][1]); p_ham = s_carry[wv][lane][2]; p_depth = s_carry[wv][lane][3]; p_sreh = (uint32_t)s_carry[wv][lane][4]; }
                }
                if (in && lane >= carry_n) { p_start = (int32_t)pos; p_score = mtb_part_score(reh, 8, false); p_ham = (int32_t)ham; p_depth = 1; p_sreh = reh; }
                int32_t ps = 0, pe = 0;            /* the last complete group that was processed: carried if the block goes on */
                const uint64_t all_in = nin == 64 ? ~0ull : ((1ull << nin) - 1ull);
                if (gm == all_in && carry_n <= 1) {
                    /* ---- every position group of the window is one match: lane j can only link to lane j - 1, the DP is a
                     * segmented prefix sum.  Lanes [0, last] are complete groups (the window's last match may be the first of a
                     * wider group that goes on in the next window, unless the block ends here). ---- */
                    const int32_t last = ends ? nin - 1 : nin - 2;
                    const uint32_t pdna = (uint32_t)__shfl_up((int)dna, 1, 64);
                    bool linked = false; int32_t sh = 0;
                    if (lane >= 1 && lane <= last) {
                        const int32_t s = (int32_t)(pos - ppos) / 3;
                        if (s > 0 && s <= sp.max_codon_shift && mtb_consecutive(pdna, dna, s, fwd, sp.kmer_format)) { linked = true; sh = s; }
                    }
                    const uint64_t lm = __ballot(linked);
                    float is = 0.0f; int32_t ihd = 0;
                    if (linked) { is = mtb_part_score(reh, sh, false); ihd = (mtb_part_ham(reh, sh, false) << 16) | sh; }
                    const float psum = wave_inclusive_scan_dpp(is); const int32_t phd = wave_inclusive_scan_dpp(ihd);
                    const uint64_t le = ~lm & (lt | (1ull << lane));             /* chain roots at or below this lane; lane 0 is one */
                    const int32_t root = 63 - (int32_t)__builtin_clzll(le);
                    const float r_psum = __shfl(psum, root, 64); const int32_t r_phd = __shfl(phd, root, 64);
                    const int32_t r_start = __shfl(p_start, root, 64), r_ham = __shfl(p_ham, root, 64), r_depth = __shfl(p_depth, root, 64);
                    const float r_score = __shfl(p_score, root, 64); const uint32_t r_sreh = (uint32_t)__shfl((int)p_sreh, root, 64);
                    if (linked) {
                        const int32_t dhd = phd - r_phd;
                        p_start = r_start; p_score = r_score + (psum - r_psum); p_ham = r_ham + (dhd >> 16); p_depth = r_depth + (dhd & 0xFFFF); p_sreh = r_sreh;
                    }
                    if (last >= 1) multi = true;
                    /* a lane whose next group is complete decides its emission now; the last complete group waits for the next
                     * window unless the block ends with it (then nothing follows it: not connected) */
                    const bool conn = ((lm >> 1) >> lane) & 1ull;                 /* lane + 1 links to this lane */
                    bool e = false;
                    if (lane < last) e = multi && !conn && p_depth >= md;
                    else if (lane == last && ends) e = multi && p_depth >= md;
                    emit(e, idx, pos, reh);
                    if (ends) done = true;
                    else { ps = last; pe = last + 1; }
                } else {
                    /* ---- groups of several matches in the window: one group transition after another ---- */
                    ps = 0;
                    uint64_t rest = gm & (gm - 1);
                    pe = rest ? (int32_t)__builtin_ctzll(rest) : nin;
                    if (!(rest != 0 || ends)) { fail = true; break; }            /* a position group wider than the window */
                    for (;;) {
                        const int32_t gs = pe;
                        if (gs >= nin) break;
                        const uint64_t r2 = rest & (rest - 1);
                        const int32_t ge = r2 ? (int32_t)__builtin_ctzll(r2) : nin;
                        if (!(r2 != 0 || ends)) break;                            /* trailing group may go on in the next window */
                        multi = true;
                        const uint32_t posP = (uint32_t)lrl_i((int32_t)pos, ps), posG = (uint32_t)lrl_i((int32_t)pos, gs);
                        const int32_t s = (int32_t)(posG - posP) / 3;
                        const bool ok = s > 0 && s <= sp.max_codon_shift;
                        const bool inG = lane >= gs && lane < ge;
                        int32_t best = -1; float best_sc = 0.0f; uint64_t conn = 0;
                        if (ok) for (int32_t cu = ps; cu < pe; cu++) {
                            const uint32_t dcu = (uint32_t)lrl_i((int32_t)dna, cu); const float scu = lrl_f(p_score, cu);
                            const bool c = inG && mtb_consecutive(dcu, dna, s, fwd, sp.kmer_format);
                            if (c && scu > best_sc) { best = cu; best_sc = scu; }
                            if (__ballot(c)) conn |= 1ull << cu;
                        }
                        const bool inP = lane >= ps && lane < pe;
                        emit(inP && !((conn >> lane) & 1ull) && p_depth >= md, idx, pos, reh);
                        const int32_t src = best >= 0 ? best : lane;
                        const int32_t b_start = __shfl(p_start, src, 64), b_ham = __shfl(p_ham, src, 64), b_depth = __shfl(p_depth, src, 64);
                        const float b_score = __shfl(p_score, src, 64); const uint32_t b_sreh = (uint32_t)__shfl((int)p_sreh, src, 64);
                        if (inG && best >= 0) {
                            p_start = b_start; p_score = b_score + mtb_part_score(reh, s, false); p_ham = b_ham + mtb_part_ham(reh, s, false);
                            p_depth = b_depth + s; p_sreh = b_sreh;
                        }
                        ps = gs; pe = ge; rest = r2;
                    }
                    if (ends && pe >= nin) {
                        const bool inP = lane >= ps && lane < pe;
                        emit(inP && multi && p_depth >= md, idx, pos, reh);
                        done = true;
                    } else if (ps == 0) { fail = true; break; }                   /* nothing but the carried group fitted: group too wide */
                }
                if (!done) {
                    /* the block goes on: the last processed group opens the next window with its path state */
                    lwave_fence();
                    if (lane >= ps && lane < pe) {
                        s_carry[wv][lane - ps][0] = p_start; s_carry[wv][lane - ps][1] = __float_as_int(p_score); s_carry[wv][lane - ps][2] = p_ham;
                        s_carry[wv][lane - ps][3] = p_depth; s_carry[wv][lane - ps][4] = (int32_t)p_sreh;
                    }
                    lwave_fence();
                    carry_n = pe - ps; base += (uint32_t)ps;
                }
            }
            if (fail) s_fail = 1;
        }
        __syncthreads();
        if (s_fail) { if (tid == 0) todo[r] = 1; continue; }
        const int32_t np = (int32_t)s_npath;
        if (np == 0) { if (tid == 0) results[r] = s_R; continue; }               /* no species produced a path: unclassified, score 0 (:372-375) */

        MTB_LP_MARK(2);
        /* ---- combine: stable order by all-pairs rank ---- */
        for (int32_t e = tid; e < np; e += MTB_LONG_NT) {
            const mtb_lpath pe_ = s_path[e];
            int32_t rank = 0;
            for (int32_t f = 0; f < np; f++) {
                const mtb_lpath pf = s_path[f];
                rank += (pf.species < pe_.species || (pf.species == pe_.species && f != e && lpath_before(pf, pe_))) ? 1 : 0;
            }
            s_sidx[rank] = (uint16_t)e;
        }
        __syncthreads();
        MTB_LP_MARK(3);
        /* species ranges of the sorted list */
        for (int32_t c0 = 0; c0 < np; c0 += MTB_LONG_NT) {
            const int32_t k = c0 + tid;
            bool head = false; int32_t spc = 0;
            if (k < np) { spc = s_path[s_sidx[k]].species; head = k == 0 || s_path[s_sidx[k - 1]].species != spc; }
            const uint64_t hm = __ballot(head);
            if (lane == 0) s_red[wv] = (uint32_t)__popcll(hm);
            __syncthreads();
            uint32_t before = s_nsp, tot = 0;
#pragma unroll
            for (int q = 0; q < MTB_LONG_NW; q++) { const uint32_t c = s_red[q]; if (q < wv) before += c; tot += c; }
            if (head) { const uint32_t at = before + (uint32_t)__popcll(hm & lt); if (at < (uint32_t)MAXSP) { s_splo[at] = (uint16_t)k; s_spid[at] = spc; } }
            __syncthreads();
            if (tid == 0) { s_nsp += tot; if (s_nsp > (uint32_t)MAXSP) s_fail = 1; }
        }
        __syncthreads();
        if (s_fail) { if (tid == 0) todo[r] = 1; continue; }
        const int32_t nsp = (int32_t)s_nsp;
        if (tid == 0) s_splo[nsp] = (uint16_t)np;
        __syncthreads();
        MTB_LP_MARK(4);
        /* greedy combination, one wave per species, 64 candidate paths per step (a lane each).  Every lane runs its candidate against the
         * paths accepted BEFORE the batch, in order (their trimmed ends are broadcast: the first 256 live in registers, path q * 64 + a in
         * lane a of register q; further ones in LDS); then the first surviving lane is accepted -- nothing accepted later can precede it
         * -- and the lanes behind it run against that path, and so on.  A candidate meets exactly the accepted paths it meets in the
         * reference's one-at-a-time loop (combineMatchPaths :428-468, trimMatchPath :475-485), in the same order, and the scores are
         * added in acceptance order: same bits (checked on the host, tests/emu/combine_batched_check.cpp).  The serial form was a
         * quarter of the kernel: one wave tested one candidate per step while the read's true species holds nearly all paths. */
        for (int32_t j = wv; j < nsp; j += MTB_LONG_NW) {
            const int32_t lo = s_splo[j], hi = s_splo[j + 1];
            float score = 0.0f; int32_t na = 0;
            int32_t a_st[4] = {0, 0, 0, 0}, a_en[4] = {0, 0, 0, 0};
            for (int32_t k0 = lo; k0 < hi; k0 += 64) {
                const int32_t nbt = hi - k0 < 64 ? hi - k0 : 64;
                const int32_t pi = lane < nbt ? (int32_t)s_sidx[k0 + lane] : 0;
                mtb_lpath p = s_path[pi];
                bool drop = lane >= nbt, taken = false;
                auto against = [&](int32_t cst, int32_t cen) {        /* the reference's loop body for one accepted path */
                    if (!((p.end < cst) || (cen < p.start))) {
                        const int32_t ov2 = (p.end < cen ? p.end : cen) - (p.start > cst ? p.start : cst) + 1;
                        if (ov2 == p.end - p.start + 1) { drop = true; return; }
                        if (ov2 < 24) {
                            if (p.start < cst) {
                                p.end = cst - 1;
                                const int32_t h = p.ham - mtb_part_ham(p.rehs >> 16, ov2 / 3, false); p.ham = h > 0 ? h : 0;
                                p.score = p.score - mtb_part_score(p.rehs >> 16, ov2 / 3, false) - (float)(ov2 % 3);
                            } else {
                                p.start = cen + 1;
                                const int32_t h = p.ham - mtb_part_ham(p.rehs & 0xFFFFu, ov2 / 3, true); p.ham = h > 0 ? h : 0;
                                p.score = p.score - mtb_part_score(p.rehs & 0xFFFFu, ov2 / 3, true) - (float)(ov2 % 3);
                            }
                        } else drop = true;
                    }
                };
                /* the paths accepted before this batch */
                const int32_t na0 = na;
                for (int32_t a = 0; a < na0; a++) {
                    int32_t cst, cen;
                    if (a < 256) {
                        const int32_t q = a >> 6, l = a & 63;
                        const int32_t vs = q == 0 ? a_st[0] : q == 1 ? a_st[1] : q == 2 ? a_st[2] : a_st[3];
                        const int32_t ve = q == 0 ? a_en[0] : q == 1 ? a_en[1] : q == 2 ? a_en[2] : a_en[3];
                        cst = lrl_i(vs, l); cen = lrl_i(ve, l);
                    } else { const mtb_lpath c = s_path[s_acc[lo + a]]; cst = c.start; cen = c.end; }
                    if (!__any(!drop)) break;
                    if (!drop) against(cst, cen);
                }
                /* the batch's survivors, in order */
                for (;;) {
                    const uint64_t sm = __ballot(!drop && !taken);
                    if (!sm) break;
                    const int32_t f = (int32_t)__builtin_ctzll(sm);
                    const int32_t fst = lrl_i(p.start, f), fen = lrl_i(p.end, f);
                    const float fsc = lrl_f(p.score, f);
                    if (na < 256) {
#pragma unroll
                        for (int q = 0; q < 4; q++) if (q == (na >> 6) && lane == (na & 63)) { a_st[q] = fst; a_en[q] = fen; }
                    } else {
                        if (lane == f) { s_path[pi] = p; s_acc[lo + na] = (uint16_t)pi; }
                        lwave_fence();
                    }
                    na++; score += fsc;
                    if (lane == f) taken = true;
                    if (lane > f && !drop) against(fst, fen);
                }
            }
            float sc = score / (float)read_len; sc = sc < 1.0f ? sc : 1.0f;
            if (lane == 0) s_spsc[j] = sc;
        }
        __syncthreads();
        MTB_LP_MARK(5);
        /* ---- species decision (thread 0): getBestSpeciesMatches second half, chooseBestTaxon's early exits ---- */
        if (tid == 0) {
            mtb_result R = s_R;
            float best_sp = 0.0f; int32_t meaningful = 0;
            for (int32_t j = 0; j < nsp; j++) { const float sc = s_spsc[j]; if (sc < sp.min_score) continue; if (sc > 0.0f) meaningful++; if (sc > best_sp) best_sp = sc; }
            if (meaningful) {
                float sum = 0.0f; int32_t n_max = 0, lca = -1, only = 0, first_spc = 0;
                const float cut = best_sp * sp.tie_ratio;
                for (int32_t j = 0; j < nsp; j++) {
                    const float sc = s_spsc[j];
                    if (sc < sp.min_score) continue;
                    if (sc >= cut) {
                        const int32_t spc = s_spid[j];
                        sum += sc; only = spc; n_max++;
                        if (n_max == 1) first_spc = spc;
                        else {
                            if (n_max == 2) lca = mtb_tax_exists(&tx, first_spc) ? mtb_tax_canon(&tx, first_spc) : -1;
                            if (mtb_tax_exists(&tx, spc)) lca = lca < 0 ? mtb_tax_canon(&tx, spc) : mtb_lca(&tx, lca, spc);
                        }
                    }
                }
                const float score = n_max > 1 ? sum / (float)n_max : sum;
                R.score = score;
                if (!(score == 0.0f || score < sp.min_score)) {
                    R.is_classified = 1;
                    if (n_max > 1) R.classification = lca < 0 ? 0 : lca;
                    else { s_go = 1; s_species = only; }
                }
            }
            s_R = R;
        }
        __syncthreads();
        if (!s_go) { if (tid == 0) results[r] = s_R; continue; }
        const int32_t species = s_species;
        MTB_LP_MARK(6);
        /* ---- redundancy filter over the best species' matches; buckets in the (dead) path storage ---- */
        uint32_t *hmin = (uint32_t *)s_path; int32_t *btax = (int32_t *)(hmin + MAXBKT);
        for (int32_t q = tid; q < nb; q += MTB_LONG_NT) { hmin[q] = 255u; btax[q] = -1; }
        __syncthreads();
        /* (both passes: the records of two steps requested together; the second pass was a quarter of the kernel because nearly every
         * match paid an LCA walk of dependent global loads -- the matches of one species carry a handful of distinct taxa, so a thread
         * keeps its last four (a, b) -> LCA answers in registers) */
        for (int32_t c0 = 0; c0 < n; c0 += 2 * MTB_LONG_NT) {
            const int32_t i0 = c0 + tid, i1 = i0 + MTB_LONG_NT;
            int32_t s0_ = -1, s1_ = -1; uint32_t p0 = 0, p1 = 0, h0 = 0, h1 = 0;
            if (i0 < n) { s0_ = m[i0].species_id; p0 = mtb_q_pos(m[i0].qinfo); h0 = m[i0].hamming; }
            if (i1 < n) { s1_ = m[i1].species_id; p1 = mtb_q_pos(m[i1].qinfo); h1 = m[i1].hamming; }
            if (i0 < n && s0_ == species) { const int32_t q = (int32_t)(p0 / (uint32_t)sp.dna_shift); if (q < nb) atomicMin(&hmin[q], h0); }
            if (i1 < n && s1_ == species) { const int32_t q = (int32_t)(p1 / (uint32_t)sp.dna_shift); if (q < nb) atomicMin(&hmin[q], h1); }
        }
        __syncthreads();
        {
            int32_t ka[4] = {-1, -1, -1, -1}, kb[4] = {-1, -1, -1, -1}, kr[4] = {0, 0, 0, 0}; int kn = 0;
            auto lca_memo = [&](int32_t a, int32_t b) -> int32_t {
#pragma unroll
                for (int u = 0; u < 4; u++) if (ka[u] == a && kb[u] == b) return kr[u];
                const int32_t v = mtb_lca(&tx, a, b);
#pragma unroll
                for (int u = 0; u < 4; u++) if (u == kn) { ka[u] = a; kb[u] = b; kr[u] = v; }
                kn = (kn + 1) & 3;
                return v;
            };
            auto merge = [&](int32_t q, int32_t t) {            /* mtb_ph_filter_merge with the memo */
                int32_t old = atomicCAS(&btax[q], -1, t);       /* first id of the bucket stays raw */
                while (old != -1) {
                    const int32_t merged = lca_memo(old, t);
                    if (merged == old) break;
                    const int32_t seen = atomicCAS(&btax[q], old, merged);
                    if (seen == old) break;
                    old = seen;
                }
            };
            for (int32_t c0 = 0; c0 < n; c0 += 2 * MTB_LONG_NT) {
                const int32_t i0 = c0 + tid, i1 = i0 + MTB_LONG_NT;
                int32_t s0_ = -1, s1_ = -1, t0 = 0, t1 = 0; uint32_t p0 = 0, p1 = 0, h0 = 0, h1 = 0;
                if (i0 < n) { s0_ = m[i0].species_id; p0 = mtb_q_pos(m[i0].qinfo); h0 = m[i0].hamming; t0 = m[i0].target_id; }
                if (i1 < n) { s1_ = m[i1].species_id; p1 = mtb_q_pos(m[i1].qinfo); h1 = m[i1].hamming; t1 = m[i1].target_id; }
                if (i0 < n && s0_ == species) { const int32_t q = (int32_t)(p0 / (uint32_t)sp.dna_shift); if (q < nb && h0 == hmin[q]) merge(q, t0); }
                if (i1 < n && s1_ == species) { const int32_t q = (int32_t)(p1 / (uint32_t)sp.dna_shift); if (q < nb && h1 == hmin[q]) merge(q, t1); }
            }
        }
        __syncthreads();
        MTB_LP_MARK(7);
        /* Query::taxCnt: distinct bucket taxa ascending with their bucket counts (std::map order) */
        int32_t ntc = 0, last = -1;
        while ((uint64_t)ntc < room) {
            int32_t mn = INT32_MAX;
            for (int32_t q = tid; q < nb; q += MTB_LONG_NT) if (hmin[q] != 255u) { const int32_t t = btax[q]; if (t > last && t < mn) mn = t; }
            for (int d = 32; d > 0; d >>= 1) { const int32_t o = __shfl_xor(mn, d, 64); mn = o < mn ? o : mn; }
            __syncthreads();
            if (lane == 0) s_red[wv] = (uint32_t)mn;
            __syncthreads();
#pragma unroll
            for (int q = 0; q < MTB_LONG_NW; q++) { const int32_t o = (int32_t)s_red[q]; mn = o < mn ? o : mn; }
            if (mn == INT32_MAX) break;
            uint32_t cnt = 0;
            for (int32_t q = tid; q < nb; q += MTB_LONG_NT) cnt += (hmin[q] != 255u && btax[q] == mn) ? 1u : 0u;
            for (int d = 32; d > 0; d >>= 1) cnt += (uint32_t)__shfl_xor((int)cnt, d, 64);
            __syncthreads();
            if (lane == 0) s_red[wv] = cnt;
            __syncthreads();
            cnt = 0;
#pragma unroll
            for (int q = 0; q < MTB_LONG_NW; q++) cnt += s_red[q];
            if (tid == 0) {
                if (off + (uint64_t)ntc < tc_cap) { tc_tax[off + ntc] = mn; tc_cnt[off + ntc] = cnt; }
                if (ntc < MTB_LR_MAXE) { s_otax[ntc] = mn; s_ocnt[ntc] = cnt; }
            }
            ntc++; last = mn;
        }
        __syncthreads();
        MTB_LP_MARK(8);
        /* ---- sub-species descent ---- */
        int32_t slow = ntc > MTB_LR_MAXE ? 1 : 0;
        if (!slow && tid < ntc) {
            int32_t lv;
            mtb_lr_climb(&tx, s_otax[tid], species, &lv, s_anc + tid * MTB_LR_K);
            s_lev[tid] = lv;
            if (lv > MTB_LR_K) slow = 1;
        }
        slow = __syncthreads_or(slow);
        if (tid == 0) {
            mtb_result R = s_R;
            R.n_taxcnt = (uint16_t)ntc;
            const int32_t cs = mtb_tax_canon(&tx, species);
            if (R.score < sp.min_sp_score) R.classification = (species >= 0 && species <= tx.max_taxid) ? tx.sp_parent[species] : 0;
            else if (slow || cs < 0) {
                if (ntc <= MTB_LR_MAXE) R.classification = mtb_lower_rank(&tx, s_otax, s_ocnt, ntc, species, read_len, sp.denominator, sp.accession_level);
                else R.classification = mtb_lower_rank(&tx, tc_tax + off, tc_cnt + off, ntc, species, read_len, sp.denominator, sp.accession_level);
            } else R.classification = mtb_lr_bfs(s_lev, s_anc, s_ocnt, ntc, cs, read_len, sp.denominator, &tx, sp.accession_level);
            R.taxcnt_off = (uint32_t)(off + tc_base);
            results[r] = R;
        }
    }
#ifdef MTB_LONG_PHASE_CYCLES
    if (tid == 0) for (int k = 0; k < 10; k++) atomicAdd(&mtb_long_cycles[k], lp_acc[k]);
#endif
}

#endif
