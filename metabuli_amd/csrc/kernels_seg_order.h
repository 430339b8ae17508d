/* kernels_seg_order.h -- long reads: a read's ordinal slots -> its match segment in compareMatches order, without a sort.
 *
 * The reference sorts ALL matches of a batch by (read, species, frame, position, hamming, dna) (KmerMatcher::sortMatches,
 * src/commons/KmerMatcher.cpp:1071-1078, 1149-1166).  Round 2 regrouped the matches of long reads by read (scattered 24-byte
 * stores, 3.3 x write amplification) and sorted every segment (bitonic chunks in LDS + rank merges): 79 + 193 ms per 200 k x 10 kb.
 * With the directory join writing the first match of a read's ord-th metamer to slot ord (k_join_dir<.., LONG>), the direct slots
 * of a read already are in (frame, position) order -- the extractor numbers a read's metamers frame by frame with rising
 * positions -- and what is left of the sort is
 *   (1) a STABLE PARTITION of that stream by species (species ascending, order inside a species kept), and
 *   (2) the few further matches of multi-match metamers (the read's tail slots, unordered): sorted in LDS by (species, key) and
 *       merged by rank -- a direct match is preceded by the tail matches of its species with a smaller key (one binary search), a
 *       tail match by the direct matches of its species with a smaller key (a difference array filled on the way).
 * One workgroup (4 waves) per read: count pass (species hash table in LDS, counts per wave quarter), tail sort, species offsets,
 * scatter pass (every wave its quarter, stable by ballot ranking on the 9-bit hash slot), tail pass.  Output: 24-byte Match
 * records at out[rb[r] ..), live[r] of them.  Reads whose tail overran its slots, or with more distinct species / tail matches
 * than the LDS tables hold, are flagged in `fail` (the caller redoes the batch on the exact-segment path).
 * Algorithmic HBM bytes: 16 per slot read twice + 24 per match written.                                                     */
#ifndef MTB_KERNELS_SEG_ORDER_H
#define MTB_KERNELS_SEG_ORDER_H
#include "dev_util.h"
#include "mtb_core.h"

#define MTB_SO_NT 256
#define MTB_SO_NW 4
#define MTB_SO_BLOOM 65536         /* buckets of the seen-once / seen-twice bit arrays */
#define MTB_SO_HASH 1024           /* exact table of the species that may have two matches (at most 768) */
#define MTB_SO_MAXSP 768
#define MTB_SO_TAIL 2048           /* tail matches of a read held in LDS as (key, index) */

/* profiling build only (make libmtb_xsoprof.so X=-DMTB_SO_PHASE_CYCLES): cycles of thread 0 per phase, summed over the workgroups: 0 claim + clear, 1 pass 1
 * (bits), 2 pass 2 (counts, tail keys), 3 single-match species, 4 tail sort, 5 species sort + offsets, 6 scatter pass, 7 tail pass */
#ifdef MTB_SO_PHASE_CYCLES
__device__ unsigned long long mtb_so_cycles[8];
#define MTB_SO_MARK(k) do { const unsigned long long t_ = __builtin_readcyclecounter(); so_acc[k] += t_ - so_t; so_t = t_; } while (0)
#else
#define MTB_SO_MARK(k) do {} while (0)
#endif
__device__ __forceinline__ uint32_t so_hash(int32_t species) { return ((uint32_t)species * 0x9E3779B1u) >> 22; }      /* 10 bits */
__device__ __forceinline__ uint32_t so_bloom(int32_t species) { return ((uint32_t)species * 0x85EBCA6Bu) >> 16; }     /* 16 bits */

/* a 24-byte Match record at an 8-byte aligned place: ONE 16-byte store + one 8-byte store (which half is the wide one depends on the
 * place's alignment) instead of three 8-byte ones.  The ordering passes write records to unrelated places lane by lane -- their time is
 * the NUMBER of write transactions (the tail pass: 100 M records per 50 k long reads on the heavy-tailed index = 71 % of the kernel). */
typedef unsigned long long so_u64x2 __attribute__((vector_size(16)));
__device__ __forceinline__ void so_store_match(mtb_match *p, const mtb_match &m) {
    const uint64_t *q = (const uint64_t *)&m;
    uint64_t *o = (uint64_t *)p;
    if (((uintptr_t)o & 15u) == 0) { so_u64x2 v; v[0] = q[0]; v[1] = q[1]; *(so_u64x2 *)o = v; o[2] = q[2]; }
    else { o[0] = q[0]; so_u64x2 v; v[0] = q[1]; v[1] = q[2]; *(so_u64x2 *)(o + 1) = v; }
}

/* all-ascending bitonic network over n keys with virtual +inf padding, payload idx[] moves with the key */
__device__ __forceinline__ void so_bitonic(uint64_t *key, uint16_t *idx, uint32_t n, uint32_t tid) {
    uint32_t p2 = 1; while (p2 < n) p2 <<= 1;
    for (uint32_t k = 2; k <= p2; k <<= 1) {
        for (uint32_t j = k >> 1; j > 0; j >>= 1) {
            for (uint32_t q = tid; q < (p2 >> 1); q += MTB_SO_NT) {
                const uint32_t lo = ((q & ~(j - 1)) << 1) | (q & (j - 1));
                const uint32_t hi = (j == (k >> 1)) ? (lo ^ ((j << 1) - 1)) : (lo + j);
                if (hi < n) {
                    const uint64_t a = key[lo], b = key[hi];
                    if (b < a) { key[lo] = b; key[hi] = a; const uint16_t t = idx[lo]; idx[lo] = idx[hi]; idx[hi] = t; }
                }
            }
            __syncthreads();
        }
    }
}

/* A long read of an error-prone technology carries THOUSANDS of stray matches, each of a species of its own (a metamer with a
 * changed amino acid finds some target with the same amino-acid part in a dense index): a match that is alone in its species can
 * never be part of a path ((species, frame) blocks need two position groups, Taxonomer.cpp:342), its species is never scored and
 * the match is never looked at again -- such matches are DROPPED here (live[] counts them, the ordered segment does not hold
 * them).  They are found with two bit arrays (species seen once / twice, by hash bucket); only the species of a bucket seen twice
 * enter the exact table, and an exact count of one drops the match as well. */
__global__ __launch_bounds__(MTB_SO_NT) void k_seg_order(const mtb_slot16 *__restrict__ slots, const uint64_t *__restrict__ rb, const uint32_t *__restrict__ dcnt,
                                                          const uint32_t *__restrict__ cursor, uint32_t tf, uint64_t n_reads, mtb_match *__restrict__ out,
                                                          uint32_t *__restrict__ live, uint32_t *__restrict__ n_fail, uint32_t *__restrict__ fail_list,
                                                          unsigned long long *__restrict__ work, unsigned long long *__restrict__ n_all) {
    /* seen-once / seen-twice bits by species bucket; dead after pass 2, when the same storage holds the quarters' running positions */
    __shared__ uint32_t s_bits[2 * (MTB_SO_BLOOM / 32)];
    static_assert(2 * (MTB_SO_BLOOM / 32) >= MTB_SO_NW * MTB_SO_HASH, "the running positions alias the bit arrays");
    uint32_t *b_once = s_bits, *b_twice = s_bits + MTB_SO_BLOOM / 32;
    uint32_t (*h_run)[MTB_SO_HASH] = (uint32_t (*)[MTB_SO_HASH])s_bits;
    __shared__ int32_t h_key[MTB_SO_HASH];                    /* species of the slot, -1 = empty, <= -2 = dropped */
    __shared__ uint32_t h_cnt[2][MTB_SO_HASH];                /* direct matches per wave quarter, 16 bits each: quarters 0 | 1 << 16 and 2 | 3 << 16 */
    __shared__ uint32_t h_tcnt[MTB_SO_HASH], h_tstart[MTB_SO_HASH];      /* tail matches of the species; their start in the sorted tail */
    __shared__ uint32_t h_S[MTB_SO_HASH];                     /* start of the species' region in the output; ~0 = dropped (one match) */
    __shared__ uint64_t t_key[MTB_SO_TAIL];                   /* hash slot [46..55] | (frame, position, hamming, dna) [0..45] */
    __shared__ uint16_t t_idx[MTB_SO_TAIL];
    __shared__ uint32_t t_diff[MTB_SO_TAIL + 1];
    __shared__ uint64_t sp_sort[MTB_SO_HASH];                 /* species << 16 | hash slot, sorted */
    __shared__ uint16_t sp_dummy[MTB_SO_HASH];
    __shared__ uint32_t s_red[MTB_SO_NW];
    __shared__ unsigned long long s_r;
    __shared__ uint32_t s_nsp, s_bad, s_total, s_live, s_nkept;
    const uint32_t tid = threadIdx.x, lane = tid & 63u, wv = tid >> 6;
    const uint64_t lt = lanemask_lt();
#ifdef MTB_SO_PHASE_CYCLES
    unsigned long long so_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long so_t = __builtin_readcyclecounter();
#endif

    for (;;) {
        __syncthreads();
        MTB_SO_MARK(7);
        if (tid == 0) { s_r = atomicAdd(work, 1ull); s_nsp = 0; s_bad = 0; s_total = 0; s_live = 0; s_nkept = 0; }
        __syncthreads();
        const uint64_t r = s_r;
        if (r >= n_reads) break;
        const uint32_t d = dcnt[r];
        if (d == 0) { if (tid == 0) live[r] = 0; continue; }
        const uint32_t tcap = mtb_lslot_tail(d, tf);
        const uint32_t cur = cursor[r];
        if (cur > tcap || cur > MTB_SO_TAIL) { if (tid == 0) { live[r] = 0; fail_list[atomicAdd(n_fail, 1u)] = (uint32_t)r; } continue; }
        const uint32_t t = cur;
        const mtb_slot16 *seg = slots + rb[r];
        mtb_match *dst = out + rb[r];
        for (uint32_t q = tid; q < MTB_SO_BLOOM / 32; q += MTB_SO_NT) { b_once[q] = 0; b_twice[q] = 0; }
        for (uint32_t q = tid; q < MTB_SO_HASH; q += MTB_SO_NT) { h_key[q] = -1; h_cnt[0][q] = 0; h_cnt[1][q] = 0; h_tcnt[q] = 0; h_tstart[q] = 0; h_S[q] = ~0u; }
        for (uint32_t q = tid; q <= t; q += MTB_SO_NT) t_diff[q] = 0;
        __syncthreads();
        /* wave quarter of the direct slots: whole 64-slot steps */
        const uint32_t q_len = ((d + MTB_SO_NW * 64 - 1) / (MTB_SO_NW * 64)) * 64;
        const uint32_t q_lo = wv * q_len < d ? wv * q_len : d, q_hi = (wv + 1) * q_len < d ? (wv + 1) * q_len : d;
        MTB_SO_MARK(0);
        /* ---- pass 1: which species buckets are seen twice ---- */
        uint32_t my_live = 0;
        auto mark = [&](int32_t species) {
            const uint32_t b = so_bloom(species), w = b >> 5, bit = 1u << (b & 31u);
            const uint32_t old = atomicOr(&b_once[w], bit);
            if (old & bit) atomicOr(&b_twice[w], bit);
        };
        /* (four 64-slot steps of the quarter per iteration, their loads in flight together: with 8 waves per CU a step per round trip
         * was the kernel's time; the counting passes do not care about the order inside a quarter) */
        for (uint32_t i0 = q_lo + lane; i0 < q_hi; i0 += 256) {
            mtb_slot16 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { x[u].a = 0; x[u].b = 0; if (i0 + 64u * u < q_hi) x[u] = seg[i0 + 64u * u]; }
#pragma unroll
            for (int u = 0; u < 4; u++) if (mtb_lslot_live(x[u])) { mark(mtb_lslot_species(x[u])); my_live++; }
        }
        for (uint32_t i = tid; i < t; i += MTB_SO_NT) { const mtb_slot16 x = seg[d + i]; if (mtb_lslot_live(x)) { mark(mtb_lslot_species(x)); my_live++; } else s_bad = 1; }
        if (my_live) atomicAdd(&s_live, my_live);
        __syncthreads();
        /* hash slot of a species that may have two matches (inserting it): open addressing, linear probing */
        auto twice = [&](int32_t species) -> bool { const uint32_t b = so_bloom(species); return (b_twice[b >> 5] >> (b & 31u)) & 1u; };
        auto slot_of = [&](int32_t species) -> uint32_t {
            /* a read with more candidate species than the table holds is handed on (s_bad): nobody keeps probing a table that fills up -- every
             * further species of such a read walked all 1024 entries, one LDS atomic each: a handful of those reads held the kernel for tens of
             * milliseconds after the other workgroups had finished */
            if (*(volatile uint32_t *)&s_bad) return 0;
            uint32_t h = so_hash(species);
            for (uint32_t probe = 0; probe < MTB_SO_HASH; probe++) {
                const int32_t old = atomicCAS(&h_key[h], -1, species);
                if (old == -1) { if (atomicAdd(&s_nsp, 1u) >= MTB_SO_MAXSP) s_bad = 1; return h; }
                if (old == species) return h;
                h = (h + 1) & (MTB_SO_HASH - 1);
            }
            s_bad = 1;
            return 0;
        };
        MTB_SO_MARK(1);
        /* ---- pass 2: exact counts of those species, per wave quarter; the tail's sort keys ---- */
        for (uint32_t i0 = q_lo + lane; i0 < q_hi; i0 += 256) {
            mtb_slot16 x[4];
#pragma unroll
            for (int u = 0; u < 4; u++) { x[u].a = 0; x[u].b = 0; if (i0 + 64u * u < q_hi) x[u] = seg[i0 + 64u * u]; }
#pragma unroll
            for (int u = 0; u < 4; u++)
                if (mtb_lslot_live(x[u])) { const int32_t spc = mtb_lslot_species(x[u]); if (twice(spc)) atomicAdd(&h_cnt[wv >> 1][slot_of(spc)], (wv & 1u) ? 0x10000u : 1u); }
        }
        for (uint32_t i = tid; i < t; i += MTB_SO_NT) {
            const mtb_slot16 x = seg[d + i];
            uint64_t k = ~0ull;                          /* dropped matches sort behind the kept ones */
            if (mtb_lslot_live(x)) { const int32_t spc = mtb_lslot_species(x); if (twice(spc)) { const uint32_t h = slot_of(spc); atomicAdd(&h_tcnt[h], 1u); k = ((uint64_t)h << 46) | mtb_lslot_key(x); } }
            t_key[i] = k; t_idx[i] = (uint16_t)i;
        }
        __syncthreads();
        if (s_bad || s_nsp > MTB_SO_MAXSP) { if (tid == 0) { live[r] = 0; fail_list[atomicAdd(n_fail, 1u)] = (uint32_t)r; } continue; }
        MTB_SO_MARK(2);
        if (tid == 0 && n_all) atomicAdd(n_all, (unsigned long long)s_live);
        /* species with a single match after all (bucket collisions): dropped; their tail keys go behind the kept ones */
        for (uint32_t q = tid; q < MTB_SO_HASH; q += MTB_SO_NT)
            if (h_key[q] >= 0 && (h_cnt[0][q] & 0xFFFFu) + (h_cnt[0][q] >> 16) + (h_cnt[1][q] & 0xFFFFu) + (h_cnt[1][q] >> 16) + h_tcnt[q] < 2u) { h_key[q] = -2 - h_key[q]; }      /* keeps the probe chain intact, marks the species */
        __syncthreads();
        for (uint32_t i = tid; i < t; i += MTB_SO_NT) { const uint64_t k = t_key[i]; if (k != ~0ull && h_key[(uint32_t)(k >> 46)] < -1) t_key[i] = ~0ull; }
        __syncthreads();
        MTB_SO_MARK(3);
        /* ---- tail sorted by (hash slot, key); start of every species' range ---- */
        if (t > 1) so_bitonic(t_key, t_idx, t, tid);
        __syncthreads();
        for (uint32_t i = tid; i < t; i += MTB_SO_NT) {
            const uint64_t k = t_key[i];
            if (k == ~0ull) continue;
            const uint32_t h = (uint32_t)(k >> 46);
            if (i == 0 || (uint32_t)(t_key[i - 1] >> 46) != h) h_tstart[h] = i;
        }
        __syncthreads();
        MTB_SO_MARK(4);
        /* ---- kept species ascending -> start of every species' region in the output ---- */
        /* (only the kept species are sorted -- a few dozen for a typical read, not the table's 1024 slots: 15 - 21 network stages
         * with a barrier each instead of 55) */
        for (uint32_t q = tid; q < MTB_SO_HASH; q += MTB_SO_NT) {
            const int32_t k = h_key[q];
            if (k >= 0) { const uint32_t at = atomicAdd(&s_nkept, 1u); sp_sort[at] = ((uint64_t)(uint32_t)k << 16) | q; sp_dummy[at] = 0; }
        }
        __syncthreads();
        const uint32_t n_kept = s_nkept;
        if (n_kept > 1) so_bitonic(sp_sort, sp_dummy, n_kept, tid);
        __syncthreads();
        {   /* exclusive prefix of the species totals in sorted order (four species per thread); the quarters' running positions
               replace their counts: S[h], S[h] + c0, S[h] + c0 + c1, ... */
            uint32_t c[4], hs[4]; uint32_t sum = 0;
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const uint32_t j = tid * 4 + u;
                c[u] = 0; hs[u] = 0;
                if (j < n_kept) {
                    hs[u] = (uint32_t)(sp_sort[j] & 0xFFFFu);
                    const uint32_t a = h_cnt[0][hs[u]], b = h_cnt[1][hs[u]];
                    c[u] = (a & 0xFFFFu) + (a >> 16) + (b & 0xFFFFu) + (b >> 16) + h_tcnt[hs[u]];
                }
                sum += c[u];
            }
            uint32_t tot;
            uint32_t run = block_exclusive_scan<uint32_t, MTB_SO_NW>(sum, s_red, &tot);
            __syncthreads();                                          /* the bit arrays are dead: their storage takes the running positions */
#pragma unroll
            for (int u = 0; u < 4; u++) {
                if (c[u]) {
                    const uint32_t S = run, h = hs[u];
                    const uint32_t a = h_cnt[0][h], b = h_cnt[1][h];
                    const uint32_t c0 = a & 0xFFFFu, c1 = a >> 16, c2 = b & 0xFFFFu;
                    h_run[0][h] = S; h_run[1][h] = S + c0; h_run[2][h] = S + c0 + c1; h_run[3][h] = S + c0 + c1 + c2;
                    h_S[h] = S;
                }
                run += c[u];
            }
            if (tid == 0) s_total = tot;
        }
        __syncthreads();
        MTB_SO_MARK(5);
        /* ---- scatter pass: every wave its quarter, in order ---- */
        mtb_slot16 x_next; x_next.a = 0; x_next.b = 0;
        if (q_lo + lane < q_hi) x_next = seg[q_lo + lane];
        for (uint32_t c0 = q_lo; c0 < q_hi; c0 += 64) {
            const uint32_t i = c0 + lane;
            const mtb_slot16 x = x_next;                        /* this step's slots were requested a step ago */
            x_next.a = 0; x_next.b = 0;
            if (i + 64 < q_hi) x_next = seg[i + 64];
            bool lv = i < q_hi && mtb_lslot_live(x);
            uint32_t h = 0;
            if (lv) {       /* kept = in the exact table and not marked (a species that never entered the table ends the probe at an empty slot) */
                const int32_t spc = mtb_lslot_species(x);
                h = so_hash(spc);
                for (;;) { const int32_t k = h_key[h]; if (k == spc) break; if (k == -1 || k == -2 - spc) { lv = false; break; } h = (h + 1) & (MTB_SO_HASH - 1); }
            }
            uint64_t peers = __ballot(lv);
#pragma unroll
            for (int b = 0; b < 10; b++) { const bool bit = (h >> b) & 1u; const uint64_t vote = __ballot(bit); peers &= bit ? vote : ~vote; }
            if (lv) {
                const uint32_t rank = (uint32_t)__popcll(peers & lt);
                const uint32_t before = h_run[wv][h];
                uint32_t lb = 0;
                const uint32_t tc = h_tcnt[h];
                if (tc) {           /* tail matches of the species with a smaller key */
                    const uint64_t key = ((uint64_t)h << 46) | mtb_lslot_key(x);
                    uint32_t lo = h_tstart[h], hi = lo + tc;
                    const uint32_t b0 = lo;
                    while (lo < hi) { const uint32_t mid = (lo + hi) >> 1; if (t_key[mid] < key) lo = mid + 1; else hi = mid; }
                    lb = lo - b0;
                    if (lb < tc) atomicAdd(&t_diff[b0 + lb], 1u);           /* this match precedes the tail matches from there on */
                }
                so_store_match(dst + before + rank + lb, mtb_lslot_unpack(x, (uint32_t)r + 1));
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();       /* every peer has read the running position ... */
            if (lv && (peers & lt) == 0) h_run[wv][h] += (uint32_t)__popcll(peers);                         /* ... before the first of them moves it */
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        }
#if defined(MTB_SO_PHASE_CYCLES) && defined(__AMDGCN__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       /* (profiling build: the scatter pass's stores are charged to the scatter pass) */
#endif
        __syncthreads();
        MTB_SO_MARK(6);
        /* ---- tail pass: a tail match sits behind the direct matches of its species that precede it and the tail matches before it ---- */
        if (t) {
            /* inclusive prefix sums of the difference array (t <= 2048: 8 per thread) */
            uint32_t v[8]; uint32_t sum = 0;
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t j = tid * 8 + u; v[u] = j < t ? t_diff[j] : 0u; sum += v[u]; }
            uint32_t tot;
            uint32_t run = block_exclusive_scan<uint32_t, MTB_SO_NW>(sum, s_red, &tot);
            __syncthreads();
#pragma unroll
            for (int u = 0; u < 8; u++) { const uint32_t j = tid * 8 + u; run += v[u]; if (j < t) t_diff[j] = run; }
            __syncthreads();
            for (uint32_t j = tid; j < t; j += MTB_SO_NT) {
                const uint64_t k = t_key[j];
                if (k == ~0ull) continue;
                const uint32_t h = (uint32_t)(k >> 46);
                const uint32_t ts = h_tstart[h];
                const uint32_t direct_before = t_diff[j] - (ts ? t_diff[ts - 1] : 0u);
                const mtb_slot16 x = seg[d + t_idx[j]];
                /* S[h] + all direct matches of the species that precede it + its rank in the species' tail */
                so_store_match(dst + h_S[h] + direct_before + (j - ts), mtb_lslot_unpack(x, (uint32_t)r + 1));
            }
        }
        if (tid == 0) live[r] = s_total;
#if defined(MTB_SO_PHASE_CYCLES) && defined(__AMDGCN__)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        MTB_SO_MARK(7);
#endif
    }
#ifdef MTB_SO_PHASE_CYCLES
    if (tid == 0) for (int k = 0; k < 8; k++) atomicAdd(&mtb_so_cycles[k], so_acc[k]);
#endif
}

/* Reads k_seg_order could not take (more candidate species / tail matches than its LDS tables hold): their live slots are copied
 * into exact segments, which k_segsort_lds then sorts the general way.  One wavefront per listed read. */
__global__ __launch_bounds__(64) void k_lbig_count(const mtb_slot16 *__restrict__ slots, const uint64_t *__restrict__ rb, const uint32_t *__restrict__ dcnt,
                                                    const uint32_t *__restrict__ cursor, uint32_t tf, const uint32_t *__restrict__ list, uint32_t n_list,
                                                    uint32_t *__restrict__ cnt, uint32_t *__restrict__ max_seg) {
    uint32_t mx = 0;
    for (uint32_t b = blockIdx.x; b < n_list; b += gridDim.x) {
        const uint32_t r = list[b];
        const uint32_t d = dcnt[r], tcap = mtb_lslot_tail(d, tf), cur = cursor[r];
        const uint32_t tot = d + (cur < tcap ? cur : tcap);
        const mtb_slot16 *s = slots + rb[r];
        uint32_t n = 0;
        for (uint32_t c0 = 0; c0 < tot; c0 += 64) { const uint32_t i = c0 + threadIdx.x; const bool lv = i < tot && mtb_lslot_live(s[i]); n += (uint32_t)__popcll(__ballot(lv)); }
        if (threadIdx.x == 0) cnt[b] = n;
        mx = n > mx ? n : mx;
    }
    if (threadIdx.x == 0 && mx) atomicMax(max_seg, mx);
}
__global__ __launch_bounds__(64) void k_lbig_copy(const mtb_slot16 *__restrict__ slots, const uint64_t *__restrict__ rb, const uint32_t *__restrict__ dcnt,
                                                   const uint32_t *__restrict__ cursor, uint32_t tf, const uint32_t *__restrict__ list, uint32_t n_list,
                                                   const uint64_t *__restrict__ start, mtb_match *__restrict__ big) {
    for (uint32_t b = blockIdx.x; b < n_list; b += gridDim.x) {
        const uint32_t r = list[b];
        const uint32_t d = dcnt[r], tcap = mtb_lslot_tail(d, tf), cur = cursor[r];
        const uint32_t tot = d + (cur < tcap ? cur : tcap);
        const mtb_slot16 *s = slots + rb[r];
        mtb_match *dst = big + start[b];
        uint32_t n = 0;
        for (uint32_t c0 = 0; c0 < tot; c0 += 64) {
            const uint32_t i = c0 + threadIdx.x;
            mtb_slot16 x; x.a = 0; x.b = 0;
            if (i < tot) x = s[i];
            const bool lv = i < tot && mtb_lslot_live(x);
            const uint64_t mask = __ballot(lv);
            if (lv) {
                const mtb_match m = mtb_lslot_unpack(x, r + 1);
                uint64_t *o = (uint64_t *)(dst + n + (uint32_t)__popcll(mask & lanemask_lt()));
                const uint64_t *q = (const uint64_t *)&m;
                o[0] = q[0]; o[1] = q[1]; o[2] = q[2];
            }
            n += (uint32_t)__popcll(mask);
        }
    }
}

/* slots a read needs: its metamers + the tail */
__global__ __launch_bounds__(256) void k_lslot_sizes(const uint32_t *__restrict__ dcnt, uint64_t n_reads, uint32_t tf, uint32_t *__restrict__ sizes) {
    const uint64_t r = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n_reads) { const uint32_t d = dcnt[r]; sizes[r] = d + mtb_lslot_tail(d, tf); }
}

#endif
