/* kernels_dir.h -- amino-acid prefix directory over the resident target array, and the join that uses it.
 *
 * The reference finds a query's candidates by streaming the delta-coded target list from a `split` checkpoint
 * (KmerMatcher.cpp:157-205, 363-371).  With the list flat in HBM the first version of this engine searched a tile's target
 * window by bisection: ~13 dependent loads per query plus the run scan (k_join, kernels_join.h) -- latency, not bandwidth,
 * bounds that kernel (82 % of its wave cycles parked in s_waitcnt, rocprofv3 SQ counters).  Here the targets are bucketed by
 * the base-21 number of their first L amino-acid letters (L <= 7, chosen so that a bucket holds a handful of targets):
 *     dir[b]      = start of bucket b, as u32 relative to base[b >> 16]           ((21^L + 1) x 4 B: 7.2 GB at L = 7)
 *     base[g]     = absolute start of bucket g * 65536                              (u64)
 * so a query reads two adjacent directory words and finishes with <= 3-4 bisection steps inside one or two cache lines.
 * Sorted queries (top 30 bits) walk the directory and the target array front to back: both streams are L2-friendly.
 * The directory exists only when every 5-bit letter of the index is < 21 (kmer_format 2) and no bucket group spans 2^32
 * targets; otherwise k_join is used.
 *
 * k_join_dir (fused short-read path, slot segments): no LDS, no barrier, no output reservation.  Per query: bucket ->
 * bisection for the amino-acid part -> the run of equal amino-acid parts is scanned once for the minimum hamming sum and
 * once to emit (compareDna, KmerMatcher.cpp:1117-1146): the first selected candidate goes to the query's ordinal slot, the
 * others to the read's tail (one atomic each, rare), beyond that to the overflow list -- same contract as k_join<SEG>.    */
#ifndef MTB_KERNELS_DIR_H
#define MTB_KERNELS_DIR_H
#include <type_traits>
#include "dev_util.h"
#include "kernels_join.h"
#include "mtb_core.h"

struct mtb_dir_view { const uint32_t *dir; const uint64_t *base; int32_t L; int32_t kmer_format; uint32_t n_buckets; };

MTB_HD uint32_t mtb_pow21(int e) { uint32_t p = 1; for (int i = 0; i < e; i++) p *= 21u; return p; }

/* bucket of a metamer value: base-21 number of its first L amino-acid letters (monotone in the value while all letters < 21) */
MTB_HD uint32_t mtb_dir_bucket(uint64_t value, int L, int kmer_format) {
    const uint64_t aa = value >> 24;
    if (kmer_format == 1) {                       /* the amino-acid part already is the base-21 number of the 8 letters */
        switch (L) {
            case 1: return (uint32_t)(aa / 1801088541ull); case 2: return (uint32_t)(aa / 85766121ull); case 3: return (uint32_t)(aa / 4084101ull);
            case 4: return (uint32_t)(aa / 194481ull); case 5: return (uint32_t)(aa / 9261ull); case 6: return (uint32_t)(aa / 441ull);
            default: return (uint32_t)(aa / 21ull);
        }
    }
    uint32_t b = 0;
    for (int j = 0; j < L; j++) b = b * 21u + (uint32_t)((aa >> (35 - 5 * j)) & 31u);
    return b;
}

/* base[g] = first target whose bucket is >= g * 65536 (g = 0 .. n_groups, the last one = T) */
__global__ __launch_bounds__(256) void k_dir_base(const uint64_t *__restrict__ values, uint64_t T, int L, int fmt, uint32_t n_groups, uint64_t *__restrict__ base) {
    const uint32_t g = blockIdx.x * 256 + threadIdx.x;
    if (g > n_groups) return;
    const uint64_t gb = (uint64_t)g << 16;
    uint64_t lo = 0, hi = T;
    while (lo < hi) { const uint64_t mid = lo + ((hi - lo) >> 1); if ((uint64_t)mtb_dir_bucket(values[mid], L, fmt) < gb) lo = mid + 1; else hi = mid; }
    base[g] = lo;
}
/* dir[x] for every bucket that starts at target i (and the empty buckets before it); flags: [0] a letter >= 21, [1] a group of
 * 65536 buckets spans >= 2^32 targets */
__global__ __launch_bounds__(256) void k_dir_fill(const uint64_t *__restrict__ values, uint64_t T, int L, int fmt, uint32_t n_buckets,
                                                   const uint64_t *__restrict__ base, uint32_t *__restrict__ dir, uint32_t *__restrict__ flags) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (uint64_t)gridDim.x * 256) {
        const uint64_t v = values[i];
        if (fmt != 1) {
            bool bad = false;
            for (int j = 0; j < 8; j++) bad |= ((v >> (59 - 5 * j)) & 31u) > 20u;
            if (bad) { flags[0] = 1; continue; }
        }
        const uint32_t b = mtb_dir_bucket(v, L, fmt);
        if (b >= n_buckets) { flags[0] = 1; continue; }
        int64_t pb = -1;
        if (i > 0) { const uint32_t p = mtb_dir_bucket(values[i - 1], L, fmt); if (p == b) continue; pb = (int64_t)p; if (p > b) { flags[0] = 1; continue; } }
        for (int64_t x = pb + 1; x <= (int64_t)b; x++) {
            const uint64_t rel = i - base[(uint64_t)x >> 16];
            if (rel >> 32) flags[1] = 1;
            dir[x] = (uint32_t)rel;
        }
    }
}
/* the empty buckets behind the last target, and the end sentinel dir[n_buckets] */
__global__ __launch_bounds__(256) void k_dir_tail(const uint64_t *__restrict__ values, uint64_t T, int L, int fmt, uint32_t n_buckets,
                                                   const uint64_t *__restrict__ base, uint32_t *__restrict__ dir, uint32_t *__restrict__ flags) {
    const uint64_t first = T ? (uint64_t)mtb_dir_bucket(values[T - 1], L, fmt) + 1 : 0;
    for (uint64_t x = first + (uint64_t)blockIdx.x * 256 + threadIdx.x; x <= n_buckets; x += (uint64_t)gridDim.x * 256) {
        const uint64_t rel = T - base[x >> 16];
        if (rel >> 32) flags[1] = 1;
        dir[x] = (uint32_t)rel;
    }
}

/* The same directory built while the target array arrives in chunks (mtb_index_open: the flat values of chunk [g0, g0 + m) are complete,
 * everything before may already be packed): *prev = the flat value of entry g0 - 1.  First the group bases that start inside the chunk,
 * then the bucket starts (they need the bases); k_dir_finish closes the table behind the last target. */
__device__ __forceinline__ bool mtb_dir_letters_ok(uint64_t v, int fmt) {
    if (fmt == 1) return true;
    bool bad = false;
    for (int j = 0; j < 8; j++) bad |= ((v >> (59 - 5 * j)) & 31u) > 20u;
    return !bad;
}
__global__ __launch_bounds__(256) void k_dir_chunk_base(const uint64_t *__restrict__ values, uint64_t g0, uint64_t m, const uint64_t *__restrict__ prev, int L, int fmt,
                                                         uint32_t n_buckets, uint64_t *__restrict__ base, uint32_t *__restrict__ flags) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) {
        const uint64_t i = g0 + j, v = values[i];
        if (!mtb_dir_letters_ok(v, fmt)) { flags[0] = 1; continue; }
        const uint32_t b = mtb_dir_bucket(v, L, fmt);
        if (b >= n_buckets) { flags[0] = 1; continue; }
        int64_t pb = -1;
        if (i > 0) { pb = (int64_t)mtb_dir_bucket(j ? values[i - 1] : *prev, L, fmt); if (pb > (int64_t)b) { flags[0] = 1; continue; } }
        for (int64_t g = pb < 0 ? 0 : (pb >> 16) + 1; g <= (int64_t)(b >> 16); g++) base[g] = i;       /* groups whose first bucket lies in (pb, b] */
    }
}
__global__ __launch_bounds__(256) void k_dir_chunk_fill(const uint64_t *__restrict__ values, uint64_t g0, uint64_t m, const uint64_t *__restrict__ prev, int L, int fmt,
                                                         uint32_t n_buckets, const uint64_t *__restrict__ base, uint32_t *__restrict__ dir, uint32_t *__restrict__ flags) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) {
        const uint64_t i = g0 + j, v = values[i];
        if (!mtb_dir_letters_ok(v, fmt)) continue;
        const uint32_t b = mtb_dir_bucket(v, L, fmt);
        if (b >= n_buckets) continue;
        int64_t pb = -1;
        if (i > 0) { pb = (int64_t)mtb_dir_bucket(j ? values[i - 1] : *prev, L, fmt); if (pb >= (int64_t)b) continue; }
        for (int64_t x = pb + 1; x <= (int64_t)b; x++) {
            const uint64_t rel = i - base[(uint64_t)x >> 16];
            if (rel >> 32) flags[1] = 1;
            dir[x] = (uint32_t)rel;
        }
    }
}
/* behind the last target (*last = its flat value): the bases of the groups that never started, the empty buckets, the end sentinel */
__global__ __launch_bounds__(256) void k_dir_finish(const uint64_t *__restrict__ last, uint64_t T, int L, int fmt, uint32_t n_buckets, uint32_t n_groups,
                                                     uint64_t *__restrict__ base, uint32_t *__restrict__ dir, uint32_t *__restrict__ flags, int phase) {
    const uint64_t first = T ? (uint64_t)mtb_dir_bucket(*last, L, fmt) + 1 : 0;
    if (phase == 0) {
        for (uint64_t g = (T ? ((first - 1) >> 16) + 1 : 0) + (uint64_t)blockIdx.x * 256 + threadIdx.x; g <= n_groups; g += (uint64_t)gridDim.x * 256) base[g] = T;
        return;
    }
    for (uint64_t x = first + (uint64_t)blockIdx.x * 256 + threadIdx.x; x <= n_buckets; x += (uint64_t)gridDim.x * 256) {
        const uint64_t rel = T - base[x >> 16];
        if (rel >> 32) flags[1] = 1;
        dir[x] = (uint32_t)rel;
    }
}
/* pack entries [g0, g0 + m) with the info entries of that chunk (pack on load: the whole info[] never exists on the device) */
__global__ __launch_bounds__(256) void k_index_pack_chunk(uint64_t *__restrict__ values, const uint32_t *__restrict__ info_chunk, uint64_t g0, uint64_t m, int fmt);

/* ---- packed state of the target array (directory depth 7 only) ------------------------------------------------------------
 * Inside a bucket of the depth-7 directory all targets share their first seven amino-acid letters, so a target is told apart by
 * 29 bits: its eighth letter (5 bits; format 1: the last base-21 digit) and its 24 DNA bits.  The other 35 bits of the 64-bit
 * word carry the `info` entry (taxonomy id, incl. the legacy redundancy bit): one 8-byte load gives the join everything the
 * reference reads from diffIdx AND info (KmerMatcher.cpp:378-381) -- the separate info[] fetch, a random 64-byte HBM sector per
 * match (70 GB per 10 M reads at 16 G targets), disappears.  The conversion is in place (values[] is rewritten, info[] stays
 * allocated as the unpack destination) and reversible: everything that needs the flat arrays unpacks first.            */
#define MTB_PACK_LOW 29
MTB_HD uint64_t mtb_pack_word(uint64_t value, uint32_t info, int fmt) {
    const uint64_t low = fmt == 1 ? ((((value >> 24) % 21ull) << 24) | (value & 0xFFFFFFull)) : (value & ((1ull << MTB_PACK_LOW) - 1));
    return ((uint64_t)info << MTB_PACK_LOW) | low;
}
MTB_HD uint64_t mtb_unpack_value(uint64_t w, uint32_t bucket, int fmt) {
    const uint64_t low = w & ((1ull << MTB_PACK_LOW) - 1);
    if (fmt == 1) return ((((uint64_t)bucket * 21ull) + (low >> 24)) << 24) | (low & 0xFFFFFFull);
    uint64_t pre = 0; uint32_t b = bucket;
    for (int j = 6; j >= 0; j--) { pre |= (uint64_t)(b % 21u) << (5 * (6 - j)); b /= 21u; }       /* letter j at bits 5*(6-j) of the 35-bit prefix */
    return (pre << MTB_PACK_LOW) | low;
}
__global__ __launch_bounds__(256) void k_index_pack(uint64_t *__restrict__ values, const uint32_t *__restrict__ info, uint64_t T, int fmt) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (uint64_t)gridDim.x * 256) values[i] = mtb_pack_word(values[i], info[i], fmt);
}
__global__ __launch_bounds__(256) void k_index_pack_chunk(uint64_t *__restrict__ values, const uint32_t *__restrict__ info_chunk, uint64_t g0, uint64_t m, int fmt) {
    for (uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x; j < m; j += (uint64_t)gridDim.x * 256) values[g0 + j] = mtb_pack_word(values[g0 + j], info_chunk[j], fmt);
}
/* one thread per bucket: its targets get their prefix back, info[] (if given) is rewritten from the upper bits */
__global__ __launch_bounds__(256) void k_index_unpack(uint64_t *__restrict__ values, uint32_t *__restrict__ info, mtb_dir_view dv) {
    for (uint64_t b = (uint64_t)blockIdx.x * 256 + threadIdx.x; b < dv.n_buckets; b += (uint64_t)gridDim.x * 256) {
        const uint64_t lo = dv.base[b >> 16] + dv.dir[b], hi = dv.base[(b + 1) >> 16] + dv.dir[b + 1];
        for (uint64_t t = lo; t < hi; t++) { const uint64_t w = values[t]; if (info) info[t] = (uint32_t)(w >> MTB_PACK_LOW); values[t] = mtb_unpack_value(w, (uint32_t)b, dv.kmer_format); }
    }
}

#if defined(MTB_NO_NT_STORES)    /* experiment builds (timing only; profiles/r06_notes.md section 5): plain stores; */
#define MTB_SLOT_STORE(sl, p) do { *(p) = (sl); } while (0)
#elif defined(MTB_NO_SLOT_STORES)        /* ... no slot stores at all (everything that feeds them still computed: the epoch never has that value); */
#define MTB_SLOT_STORE(sl, p) do { if (sa.epoch == 0xFFFFFFFFu) { __builtin_nontemporal_store((sl).a, &(p)->a); __builtin_nontemporal_store((sl).b, &(p)->b); } } while (0)
#elif defined(MTB_HALF_SLOT_STORES)      /* ... only 8 of a slot's 16 bytes (what an 8-byte slot record would write) */
#define MTB_SLOT_STORE(sl, p) do { __builtin_nontemporal_store((sl).b, &(p)->b); if (sa.epoch == 0xFFFFFFFFu) __builtin_nontemporal_store((sl).a, &(p)->a); } while (0)
#else
#define MTB_SLOT_STORE(sl, p) do { __builtin_nontemporal_store((sl).a, &(p)->a); __builtin_nontemporal_store((sl).b, &(p)->b); } while (0)
#endif
#ifndef MTB_JOIN_DIR_QPT
#define MTB_JOIN_DIR_QPT 2
#endif

/* Candidate runs longer than this are scanned by the whole wave (64 lanes over consecutive 8-byte words, ballot-ranked emission) instead
 * of by their query's lane alone.  Real databases have heavy-tailed runs -- an amino-acid 8-mer of a conserved protein is shared by
 * 10^3 - 10^4 species -- and a lane that walks such a run twice on its own (minimum, then emission: the reference's
 * loop, KmerMatcher.cpp:363-416) stalls the other 63 lanes of its wave for thousands of dependent loads. */
#ifndef MTB_JOIN_WAVES
#define MTB_JOIN_WAVES 6              /* waves per SIMD the short-read instantiation on packed words is compiled for.  Round 5, in-process A/B on the heavy-tailed workload
                                       * (profiles/r05_notes.md): two queries per thread at 5 waves 86.3 ms, at 6 waves (80 registers: 16 spilled) 91.7, ONE query per thread
                                       * at 6 waves (no spill) 80.6, at 5 waves 88.7 -- occupancy beats the second query's instruction-level parallelism */
#endif
#ifndef MTB_JOIN_DIR_QPT0
#define MTB_JOIN_DIR_QPT0 1           /* queries per thread of the short-read instantiation on packed words (the other modes keep MTB_JOIN_DIR_QPT) */
#endif
#ifndef MTB_JOIN_EXACT_MIN
#define MTB_JOIN_EXACT_MIN 8          /* diagnostics (k_join_run_hist): runs beyond this length count as long when a query finds its own DNA part in them */
#endif
/* (Round 5's experiment "queries of a wave that share a long run walk it in lockstep" measured slower -- held-out reads' join 47 -> 71 ms per 2 M reads,
 * headline 73 -> 82 -- and lives in profiles/experiments/join_lockstep_walk.patch, not in this kernel.) */
#ifndef MTB_JOIN_COOP_MIN
#define MTB_JOIN_COOP_MIN 12          /* default of JoinSegArgs::coop_min; MTB_JOIN_COOP_MIN=<n> in the environment of mtb_ctx_create overrides it (A/B runs).  32 until the wave scan took its
                                       * hamming sums from a table and fetched a step ahead; since then (one process each, join ms at 8 / 12 / 16 / 24 / 32): headline 58.1 / 58.1 / 58.2 / 58.4 /
                                       * 58.7, 200 k x 10 kb 118.9 / 118.9 / 119.1 / 119.4 / 119.7, 12.5 M pairs 145.3 / - / 145.6 / - / 146.7 (profiles/r06_notes.md section 10) */
#endif
/* the value of lane `src` (wave-uniform: the callers take it from a ballot) as a scalar: v_readlane, so that what depends on it -- a scanned run's
 * bounds, its loop -- stays in scalar registers and scalar branches */
__device__ __forceinline__ uint32_t wave_bcast32(uint32_t v, int src) { return (uint32_t)__builtin_amdgcn_readlane((int)v, src); }
__device__ __forceinline__ uint64_t wave_bcast64(uint64_t v, int src) {
    return ((uint64_t)wave_bcast32((uint32_t)(v >> 32), src) << 32) | (uint64_t)wave_bcast32((uint32_t)v, src);
}
__device__ __forceinline__ uint32_t wave_min_shfl_u32(uint32_t v) {
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) { const uint32_t o = (uint32_t)__shfl_xor((int)v, d, 64); v = o < v ? o : v; }
    return v;
}
/* Hamming sums of a wave-scanned run out of ONE register per query (round 6).  The query of such a run is the same for all 64 lanes, so its side of
 * getHammingDistanceSum (KmerMatcher.h:348-360) is tabulated across the wave: lane p = (c1 << 3 | c0) holds, in byte j, hammingLookup[q_2j][c0] +
 * hammingLookup[q_2j+1][c1] -- codon pair j of the query against the codon pair (c0, c1).  A target's sum is then four ds_bpermute look-ups (the four 6-bit
 * codon pairs of its DNA part address the lanes), two v_perm that keep byte j of look-up j, and one v_sad_u8 that adds the four bytes: 8 VALU + 4
 * LDS-crossbar instructions instead of ~36 VALU for mtb_ham_sum's eight nibble extractions.  ds_bpermute returns 0 for a source lane that is switched
 * off: every call site runs with all 64 lanes on (dummy targets beyond the run's end, their sums discarded). */
__device__ __forceinline__ uint32_t wave_ham_table(const uint32_t *hamrow, uint32_t qdna) {
    const uint32_t l = threadIdx.x & 63u, s0 = 4u * (l & 7u), s1 = 4u * (l >> 3);
    uint32_t e = 0;
#pragma unroll
    for (int j = 0; j < 4; j++)
        e |= (((hamrow[(qdna >> (6 * j)) & 7u] >> s0) & 15u) + ((hamrow[(qdna >> (6 * j + 3)) & 7u] >> s1) & 15u)) << (8 * j);
    return e;
}
__device__ __forceinline__ uint32_t wave_ham_lookup(uint32_t tab, uint32_t tdna) {
#if defined(__AMDGCN__)
    /* byte address of lane i = 4 i; the instruction divides by four, so the two bits below a codon pair may ride along */
#ifdef MTB_BPERM_EXACT_ADDR
#define MTB_BPERM_ADDR(k) (((k) ? tdna >> (6 * (k) - 2) : tdna << 2) & 0xFCu)
#else
    const uint32_t t4 = tdna << 2;
#define MTB_BPERM_ADDR(k) ((t4 >> (6 * (k))) & 0xFFu)
#endif
    const uint32_t r0 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)MTB_BPERM_ADDR(0), (int)tab), r1 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)MTB_BPERM_ADDR(1), (int)tab);
    const uint32_t r2 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)MTB_BPERM_ADDR(2), (int)tab), r3 = (uint32_t)__builtin_amdgcn_ds_bpermute((int)MTB_BPERM_ADDR(3), (int)tab);
#undef MTB_BPERM_ADDR
    /* v_perm_b32: selector bytes 0-3 take bytes of the second operand, 4-7 of the first, 0x0C is zero: {r0.b0, r1.b1, 0, 0} and {0, 0, r2.b2, r3.b3};
     * the sum of absolute differences of the two words IS the sum of the four bytes */
    return __builtin_amdgcn_sad_u8(__builtin_amdgcn_perm(r1, r0, 0x0C0C0500u), __builtin_amdgcn_perm(r3, r2, 0x07020C0Cu), 0u);
#else
    uint32_t h = 0;
    for (int k = 0; k < 4; k++) h += ((uint32_t)__shfl((int)tab, (int)((tdna >> (6 * k)) & 63u), 64) >> (8 * k)) & 0xFFu;
    return h;
#endif
}

/* MODE 0: slot segments of fixed stride (short reads); 1: per-read slot ranges (long reads); 2: dense list of Match records
 * (owner side of the range-partitioned index: the home rank of the read places them into ITS slot segments, k_slot_place) */
/* QPT = queries per thread, WAVES = waves per SIMD the register allocation aims at: template parameters so that the A/B variants of the
 * short-read instantiation live in ONE library and are compared inside one process, on one index, one allocation (MTB_JOIN_VARIANT=q<Q>w<W>
 * in the environment, read per batch; between processes the placement of a 27 GB slot buffer alone moved the join by 10 %) */
/* WIN (slot modes on packed words, one query per thread): the tile's TARGET WINDOW is staged in LDS.  The workgroup's `qt` sorted queries
 * (qt <= 256, chosen by the host from the batch's density) address buckets that lie next to each other: the span from the first query's
 * bucket to the last one's -- bounded BEFORE the launch by k_join_tile_win, so that the window's loads are the kernel's first instructions
 * and the queries and their directory rows arrive while it is in flight -- is read ONCE, and every later access of the search and the
 * evaluation (bisection steps, run ends, candidates, the wave scan of a long run) is an LDS read.  With 10 M reads against 16 G targets a
 * query owns 12.5 targets of the array on average: the windows of a batch ARE the array.
 * The window holds only the LOW 32 bits of every packed word -- all that the search and the evaluation read (29 bits tell the targets of a
 * bucket apart, 24 of them are the DNA part) -- staged by 4-byte direct-to-LDS loads (a lane per target, no staging registers, no wait
 * between the pieces); the full word is fetched from global memory (L2-warm: the window's load has just brought its sector) for SELECTED
 * candidates only.  15.9 KB per tile and 64 registers: EIGHT waves per SIMD.  That is what the kernel's time depends on -- its waves spend
 * two thirds of their time waiting on dependent accesses (bisection step -> run end -> candidate -> species id -> tail cursor), VALU issue
 * is at 42 %, the array streams at a third of the HBM rate.  Measured in one process on the headline batch (profiles/r06_notes.md): round 5's
 * 8-byte window (31.7 KB, 5 waves) 83.6 ms; low dwords at 5 / 6 / 7 / 8 waves 73.0 / 66.7 / 63.2 / 62.0; sector-random (q1w6) 80.7.
 * A tile whose window exceeds the capacity (sparse tiles, buckets of long candidate runs) keeps reading global memory: same code, rdv()
 * picks the source per tile. */
#ifndef MTB_WIN_AUX
#define MTB_WIN_AUX 0                 /* cache policy bits of the window's direct-to-LDS loads (2 = nt: streamed once; A/B build switch) */
#endif
#if defined(__AMDGCN__)
#define MTB_WAIT_VMEM() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")       /* direct-to-LDS loads count in vmcnt; the barrier that follows publishes them */
#define MTB_DRAIN_VMEM() __builtin_amdgcn_s_waitcnt(0x0F70)                   /* vmcnt(0), expcnt and lgkmcnt left alone (gfx9 encoding: vmcnt [3:0] + [15:14], expcnt [6:4], lgkmcnt [11:8]) */
#else
#define MTB_WAIT_VMEM() do {} while (0)
#define MTB_DRAIN_VMEM() do {} while (0)
#endif
#define MTB_JOIN_WINCAP 3968          /* targets a window holds = 62 pieces of 64 low dwords (one wave-wide 4-byte direct-to-LDS load each): 15.9 KB -> eight workgroups
                                       * (32 waves) per CU */
#define MTB_JOIN_WIN_WAVES 8          /* waves per SIMD the window form is compiled for */
/* The windows of the tiles, BEFORE the join (round 6): the queries are sorted on their top (64 - low_bits) bits -- six amino-acid letters
 * of kmer_format 2, the top 32 bits of format 1 -- so the first and the last record of a tile bound the buckets all its queries can
 * address: [first bucket with the first record's sort key, last bucket with the last record's sort key].  One thread per tile reads two
 * records and four directory words and leaves {first word, words} (words = 0: no window, the tile reads global memory).  The join then
 * STARTS with its window's direct-to-LDS loads and fetches its queries and their directory rows while the window is in flight: one
 * exposed round trip before the search instead of three dependent ones (queries -> directory -> workgroup minimum / maximum -> window).
 * The span is that of the exact minimum / maximum widened to whole sort-key groups at both ends (187 targets a group at 16 G targets);
 * a query whose bucket is not inside it (never, unless the list is not sorted as announced) sends the whole tile to global memory. */
struct mtb_tile_win { uint64_t first, words; };
__global__ __launch_bounds__(256) void k_join_tile_win(const mtb_kmer *__restrict__ q, uint64_t n, uint32_t qt, mtb_dir_view dv, uint64_t limit, int low_bits,
                                                        mtb_tile_win *__restrict__ win, uint32_t n_tiles, unsigned long long *__restrict__ stat) {
    const uint32_t t = blockIdx.x * 256 + threadIdx.x;
    const bool live = t < n_tiles;
    bool windowed = false;
    if (live) {
    const uint64_t j0 = (uint64_t)t * qt, j1 = (j0 + qt < n ? j0 + qt : n) - 1;
    const uint64_t v0 = q[j0].value, v1 = q[j1].value;
    mtb_tile_win w; w.first = 0; w.words = 0;
    uint64_t blo, bhi; bool ok = true;
    if (dv.kmer_format == 1) {
        const int sh = low_bits - 24;                      /* bits of the amino-acid number below the sort key */
        const uint64_t a_lo = ((v0 >> 24) >> sh) << sh, a_hi = (v1 >> 24) | ((1ull << sh) - 1ull);
        blo = a_lo / 21ull; bhi = a_hi / 21ull;
    } else {
        uint32_t b0 = 0, b1 = 0;
        for (int j = 0; j < 6; j++) {
            const uint32_t l0 = (uint32_t)((v0 >> (59 - 5 * j)) & 31u), l1 = (uint32_t)((v1 >> (59 - 5 * j)) & 31u);
            ok &= l0 < 21u && l1 < 21u;
            b0 = b0 * 21u + l0; b1 = b1 * 21u + l1;
        }
        blo = (uint64_t)b0 * 21ull; bhi = (uint64_t)b1 * 21ull + 20ull;
    }
    if (ok && blo < dv.n_buckets && bhi >= blo) {
        if (bhi >= dv.n_buckets) bhi = dv.n_buckets - 1;
        uint64_t a0 = dv.base[blo >> 16] + dv.dir[blo], a1 = dv.base[(bhi + 1) >> 16] + dv.dir[bhi + 1];
        if (a1 > limit) a1 = limit;
        a0 &= ~1ull;                                       /* 16-byte aligned pieces */
        if (a1 > a0 && a1 - a0 <= (uint64_t)MTB_JOIN_WINCAP) { w.first = a0; w.words = a1 - a0; windowed = true; }
    }
    win[t] = w;
    }
    /* statistics (mtb_batch_stats.join_tiles_windowed): one atomic per wave */
    const uint64_t m = __ballot(windowed);
    if ((threadIdx.x & 63u) == 0 && m) atomicAdd(stat, (unsigned long long)__popcll(m));
}

template <bool PACKED, int MODE = 0, int QPT = ((PACKED && MODE == 0) ? MTB_JOIN_DIR_QPT0 : MTB_JOIN_DIR_QPT), int WAVES = MTB_JOIN_WAVES, bool WIN = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu((PACKED && MODE != 2) ? WAVES : 5))) void k_join_dir(const mtb_kmer *__restrict__ q, uint64_t n, mtb_index_view ix, uint64_t limit, mtb_dir_view dv,
                                                   const mtb_tables *__restrict__ tabs, JoinSegArgs sa, uint32_t *__restrict__ overflow, uint32_t qt = 256,
                                                   const mtb_tile_win *__restrict__ tile_win = nullptr, unsigned long long *__restrict__ win_stat = nullptr) {
    constexpr int Q = QPT;
    static_assert(!WIN || (QPT == 1 && PACKED && MODE != 2), "the window form: packed words, one query per thread, slot modes");
    /* (four words of padding at either end: the run-end search reads the four words before and after a landing place, the evaluation four words from a
     * run's first candidate, whatever the run's length) */
    __shared__ __attribute__((aligned(16))) uint32_t s_win_raw[WIN ? MTB_JOIN_WINCAP + 8 : 1];
    uint32_t *const s_win = s_win_raw + (WIN ? 4 : 0);
    uint64_t w0 = 0; bool use_win = false;
    /* rdv: what the search and the evaluation read of a target (inside a window: its low 32 bits); full_of: the whole word of a SELECTED candidate */
    auto rdv = [&](uint64_t t) -> uint64_t { return (WIN && use_win) ? (uint64_t)s_win[t - w0] : ix.values[t]; };
    auto full_of = [&](uint64_t t, uint64_t v) -> uint64_t { return (WIN && use_win) ? ix.values[t] : v; };
    constexpr bool LONG = MODE == 1, LIST = MODE == 2;
    const uint64_t AAM = ~0xFFFFFFull;
    __shared__ uint32_t s_hr[8];                    /* hammingLookup rows as nibble words: the only table the join arithmetic reads (filled below, behind the loads that matter) */
    const uint64_t base_q = WIN ? (uint64_t)blockIdx.x * qt : (uint64_t)blockIdx.x * (256 * Q);
    /* the window [a0, a1) -> s_win: 64 targets a piece, a wave each, every lane the low dword of its own target, straight into LDS
     * (global_load_lds_dword: no staging registers, no wait between the pieces -- a loop of load / ds_write pairs waited for every load: a
     * dozen dependent round trips per tile, measured 96 ms against 80 for the random join).  The last piece may reach beyond the window (never read). */
    auto stage_window = [&](uint64_t a0, uint64_t a1) {
        const uint32_t n_piece = (uint32_t)((a1 - a0 + 63) >> 6), ln_ = threadIdx.x & 63u;
        for (uint32_t pc = threadIdx.x >> 6; pc < n_piece; pc += 4) {
            uint64_t idx = a0 + ((uint64_t)pc << 6) + ln_;
            if (idx >= ix.n_targets) idx = ix.n_targets - 1;                /* (behind the window) */
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ix.values + idx),
                                             (__attribute__((address_space(3))) void *)(s_win + ((uint64_t)pc << 6)), 4, 0, MTB_WIN_AUX);
        }
    };
    uint64_t pre_a0 = 0, pre_len = 0;
    if (WIN) {                                       /* the window was bounded before the launch (k_join_tile_win): its loads go first */
        const mtb_tile_win tw = tile_win[blockIdx.x];          /* (a uniform address of read-only memory: scalar loads) */
        pre_a0 = tw.first; pre_len = tw.words;
        if (pre_len) stage_window(pre_a0, pre_a0 + pre_len);
    }
    MTB_JP_BEGIN();                                  /* profiling build only: cycles of thread 0 per phase (mtb_join_cycles: 0 queries + directory, 1 bisection, 2 run ends, 3 wave-scanned runs, 4 per-lane evaluation + emission) */
    mtb_kmer k[Q]; bool valid[Q]; uint64_t lo[Q], hi[Q];
#pragma unroll
    for (int u = 0; u < Q; u++) {
        const uint64_t j = base_q + (uint64_t)u * 256 + threadIdx.x;
        valid[u] = j < n && (!WIN || threadIdx.x < qt);
        k[u].value = 0; k[u].qinfo = 0;
        if (valid[u]) { k[u] = q[j]; valid[u] = mtb_q_seq(k[u].qinfo) != 0; }        /* blank slots carry sequenceID 0 */
    }
#pragma unroll
    for (int u = 0; u < Q; u++) {
        lo[u] = 0; hi[u] = 0;
        if (valid[u]) {
            const uint32_t b = mtb_dir_bucket(k[u].value, dv.L, dv.kmer_format);
            if (b < dv.n_buckets) {
                lo[u] = dv.base[b >> 16] + dv.dir[b];
                hi[u] = dv.base[(b + 1) >> 16] + dv.dir[b + 1];
                if (hi[u] > limit) hi[u] = limit;           /* the last entry of the (whole) index is never a candidate */
                if (lo[u] > hi[u]) lo[u] = hi[u];
            } else valid[u] = false;                        /* a metamer outside the directory's alphabet (stage API: any 64-bit value may arrive) has no candidate -- and no bucket row to read again below */
        }
    }
    if (threadIdx.x < 8) s_hr[threadIdx.x] = tabs->hamrow[threadIdx.x];        /* (issued behind the window / query / directory loads: as the kernel's first statement it cost wave 0 a round trip of its own) */
    if (WIN) {
        /* every query's bucket inside the announced window?  (one barrier: it also publishes s_hr and -- behind the explicit wait for the
         * direct-to-LDS loads, which the workgroup-scope fence of a barrier is not documented to cover -- the window) */
        const bool outside = pre_len != 0 && valid[0] && lo[0] < hi[0] && (lo[0] < pre_a0 || hi[0] > pre_a0 + pre_len);
        MTB_WAIT_VMEM();
        MTB_JP_MARK(5);
        if (!__syncthreads_or(outside ? 1 : 0)) { if (pre_len) { use_win = true; w0 = pre_a0; } }
        else if (threadIdx.x == 0 && win_stat) atomicAdd(win_stat + 1, 1ull);        /* (mtb_batch_stats.join_tiles_outside: stays 0 while the list is sorted as announced; the tile reads global memory) */
    } else __syncthreads();                          /* s_hr; the query and directory loads above are in flight meanwhile */
    MTB_JP_MARK(6);
    /* what tells targets of one bucket apart: the whole amino-acid part (flat state) or the packed word's eighth letter */
    auto tkey = [&](uint64_t w) -> uint64_t { return PACKED ? (w & 0x1F000000ull) : (w & AAM); };
    auto qkey = [&](uint64_t v) -> uint64_t {
        if (!PACKED) return v & AAM;
        return dv.kmer_format == 1 ? (((v >> 24) % 21ull) << 24) : (v & 0x1F000000ull);
    };
    uint64_t e_hi[Q];
#pragma unroll
    for (int u = 0; u < Q; u++) e_hi[u] = hi[u];
    /* ONE bisection per query, on the whole value: inside a bucket the targets are ordered by (amino-acid part, DNA part), so the lower
     * bound of the query's own (amino-acid part, DNA part) -- the flat value, or the low 29 bits of a packed word -- lands either ON the
     * block of targets equal to the query or inside / next to the run of its amino-acid part.
     *   equal block found: the hamming sum of two DNA parts is 0 exactly when they are equal (every off-diagonal entry of the lookup is
     *     >= 1, KmerMatcher.h:66-70), so the minimum over the run is 0, the threshold min(2 x 0, 7) = 0 (KmerMatcher.cpp:1136) and the
     *     selection IS that block: the rest of the run -- thousands of other species for a conserved protein -- is never read;
     *   no equal target: the run's ends are found by stepping from the landing place (runs are 1-4 entries as a rule; beyond eight steps
     *     a bisection takes over), then minimum and selection as the reference's loop does them (KmerMatcher.cpp:363-416).
     * Afterwards [lo[u], e_hi[u]) is exactly the query's candidate set to evaluate (empty: valid[u] = false). */
    auto tcomp = [&](uint64_t w) -> uint64_t { return PACKED ? (w & 0x1FFFFFFFull) : w; };
    MTB_JP_MARK(0);
    {
        uint64_t qc[Q];
#pragma unroll
        for (int u = 0; u < Q; u++) qc[u] = PACKED ? (qkey(k[u].value) | (k[u].value & 0xFFFFFFull)) : k[u].value;
        bool more = false;
#pragma unroll
        for (int u = 0; u < Q; u++) more |= lo[u] < hi[u];
        while (more) {
            more = false;
#pragma unroll
            for (int u = 0; u < Q; u++) {
                if (lo[u] < hi[u]) {
                    const uint64_t mid = lo[u] + ((hi[u] - lo[u]) >> 1);
                    if (tcomp(rdv(mid)) < qc[u]) lo[u] = mid + 1; else hi[u] = mid;
                    more |= lo[u] < hi[u];
                }
            }
        }
        MTB_JP_MARK(1);
#pragma unroll
        for (int u = 0; u < Q; u++) {
            if (!valid[u]) continue;
            const uint64_t p = lo[u], end = e_hi[u], qk = qkey(k[u].value);
            uint64_t s0 = p, e0 = p;
            if (WIN && use_win) {
                /* a tile with a window: the four words before and the four from the landing place in ONE LDS round trip (one address, immediate offsets), the bucket's
                 * start read again from the directory next to them -- the loops below are a chain of dependent reads, one per step (run ends: 18 % of the kernel's cycles);
                 * runs are 1 - 4 entries as a rule.  A side whose four words all belong to the run is continued by those loops. */
                const uint32_t *pw = s_win + (p - w0);
                const uint32_t m4 = pw[-4], m3 = pw[-3], m2 = pw[-2], m1 = pw[-1], a0 = pw[0], a1 = pw[1], a2 = pw[2], a3 = pw[3];
                const uint32_t room_up = end - p > 4u ? 4u : (uint32_t)(end - p);        /* candidates at p .. p + room_up - 1 are inside the bucket */
                const uint32_t qc32 = (uint32_t)qc[u], qk32 = (uint32_t)qk;
                if (room_up != 0u && (a0 & 0x1FFFFFFFu) == qc32) {
                    uint32_t c = 1;
                    if (c < room_up && (a1 & 0x1FFFFFFFu) == qc32) { c = 2; if (c < room_up && (a2 & 0x1FFFFFFFu) == qc32) { c = 3; if (c < room_up && (a3 & 0x1FFFFFFFu) == qc32) c = 4; } }
                    e0 = p + c;
                    if (c == 4u) {
                        uint32_t n = 0;
                        while (e0 < end && n < 8u && tcomp(rdv(e0)) == qc[u]) { e0++; n++; }
                        if (n == 8u && e0 < end && tcomp(rdv(e0)) == qc[u]) {
                            uint64_t y = end;
                            while (e0 < y) { const uint64_t mid = e0 + ((y - e0) >> 1); if (tcomp(rdv(mid)) <= qc[u]) e0 = mid + 1; else y = mid; }
                        }
                    }
                } else {
                    const uint32_t bk = mtb_dir_bucket(k[u].value, dv.L, dv.kmer_format);
                    const uint64_t blo = dv.base[bk >> 16] + dv.dir[bk];
                    const uint32_t room_dn = p - blo > 4u ? 4u : (uint32_t)(p - blo);
                    uint32_t cu = 0, cd = 0;
                    if (cu < room_up && (a0 & 0x1F000000u) == qk32) { cu = 1; if (cu < room_up && (a1 & 0x1F000000u) == qk32) { cu = 2; if (cu < room_up && (a2 & 0x1F000000u) == qk32) { cu = 3; if (cu < room_up && (a3 & 0x1F000000u) == qk32) cu = 4; } } }
                    if (cd < room_dn && (m1 & 0x1F000000u) == qk32) { cd = 1; if (cd < room_dn && (m2 & 0x1F000000u) == qk32) { cd = 2; if (cd < room_dn && (m3 & 0x1F000000u) == qk32) { cd = 3; if (cd < room_dn && (m4 & 0x1F000000u) == qk32) cd = 4; } } }
                    s0 = p - cd; e0 = p + cu;
                    if (cd == 4u) {
                        uint32_t n = 0;
                        while (s0 > blo && n < 8u && tkey(rdv(s0 - 1)) == qk) { s0--; n++; }
                        if (n == 8u && s0 > blo && tkey(rdv(s0 - 1)) == qk) {
                            uint64_t x = blo, y = s0;
                            while (x < y) { const uint64_t mid = x + ((y - x) >> 1); if (tkey(rdv(mid)) < qk) x = mid + 1; else y = mid; }
                            s0 = x;
                        }
                    }
                    if (cu == 4u) {
                        uint32_t n = 0;
                        while (e0 < end && n < 8u && tkey(rdv(e0)) == qk) { e0++; n++; }
                        if (n == 8u && e0 < end && tkey(rdv(e0)) == qk) {
                            uint64_t y = end;
                            while (e0 < y) { const uint64_t mid = e0 + ((y - e0) >> 1); if (tkey(rdv(mid)) <= qk) e0 = mid + 1; else y = mid; }
                        }
                    }
                }
            } else
            if (p < end && tcomp(rdv(p)) == qc[u]) {
                /* the block of targets equal to the query (several species may file the same metamer) */
                e0 = p + 1;
                uint32_t n = 0;
                while (e0 < end && n < 8u && tcomp(rdv(e0)) == qc[u]) { e0++; n++; }
                if (n == 8u && e0 < end && tcomp(rdv(e0)) == qc[u]) {
                    uint64_t y = end;
                    while (e0 < y) { const uint64_t mid = e0 + ((y - e0) >> 1); if (tcomp(rdv(mid)) <= qc[u]) e0 = mid + 1; else y = mid; }
                }
            } else {
                /* the run of the query's amino-acid part around the landing place (the bucket's start is read again from the directory:
                 * keeping it in registers across the bisection cost the kernel a wave of occupancy) */
                const uint32_t bk = mtb_dir_bucket(k[u].value, dv.L, dv.kmer_format);
                const uint64_t blo = dv.base[bk >> 16] + dv.dir[bk];
                uint32_t n = 0;
                while (s0 > blo && n < 8u && tkey(rdv(s0 - 1)) == qk) { s0--; n++; }
                if (n == 8u && s0 > blo && tkey(rdv(s0 - 1)) == qk) {
                    uint64_t x = blo, y = s0;
                    while (x < y) { const uint64_t mid = x + ((y - x) >> 1); if (tkey(rdv(mid)) < qk) x = mid + 1; else y = mid; }
                    s0 = x;
                }
                n = 0;
                while (e0 < end && n < 8u && tkey(rdv(e0)) == qk) { e0++; n++; }
                if (n == 8u && e0 < end && tkey(rdv(e0)) == qk) {
                    uint64_t y = end;
                    while (e0 < y) { const uint64_t mid = e0 + ((y - e0) >> 1); if (tkey(rdv(mid)) <= qk) e0 = mid + 1; else y = mid; }
                }
            }
            lo[u] = s0; e_hi[u] = e0;
            if (s0 >= e0) valid[u] = false;
        }
    }
    /* runs that are still longer than sa.coop_min (no equal target in a long run: a sequencing error, a variant the index does not hold)
     * are taken away from the lane and scanned by the wave below */
    bool lng[Q];
#pragma unroll
    for (int u = 0; u < Q; u++) {
        lng[u] = valid[u] && e_hi[u] - lo[u] > (uint64_t)sa.coop_min;
        if (lng[u]) valid[u] = false;
    }
    MTB_JP_MARK(2);
    const uint32_t lane = threadIdx.x & 63u;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    /* ONE pass over a wave-scanned run [s, e), four 64-candidate steps in flight per iteration: the minimum hamming sum (-> the
     * selection threshold) and, per lane, the candidates of its stripe with a sum <= 7 -- no other can be selected, the threshold
     * being min(2 x minimum, 7) -- as (offset in the run << 4 | sum): the last four are kept (c0 = newest), n_c counts them all.
     * Runs whose lanes all stay within four are emitted from these registers; the others are walked a second time. */
    auto coop_scan_from = [&](auto from_lds, uint32_t qdna, uint32_t &tab, uint64_t s, uint64_t e, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t &n_c) -> uint32_t {
        uint32_t mn = 255u; n_c = 0; c0 = 0; c1 = 0; c2 = 0; c3 = 0;
        const uint32_t len = (uint32_t)(e - s);                          /* (runs are shorter than 2^28) */
        /* a step's 256 candidates are fetched while the previous step's are evaluated -- and the first ones while the query's table is built: a scanned run
         * costs ONE exposed round trip whatever its length (profiling build, reads of held-out genomes: 48 % of the join's cycles sat in these loops with
         * a dependent fetch per step).  Two register sets taken in turns (no moves: a move would wait for the fetch); one loop per source -- LDS window or
         * global memory, a property of the tile -- so that the compiler's wait counts see straight-line code. */
        uint32_t v[4], w[4];
        auto fetch = [&](uint32_t (&d)[4], uint32_t b0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                /* lanes beyond the run's end read its last candidate again (discarded in eval): an address clamp, not a branch -- with a predicated load the
                 * compiler moved the first use into the load's block and waited for every load on the spot */
                const uint32_t o = b0 + 64 * j + lane, oc = o < len ? o : len - 1u;
                d[j] = decltype(from_lds)::value ? s_win[s + oc - w0] : ((const uint32_t *)ix.values)[2 * (s + oc)];       /* (the low dword: a 4-byte load, no dead high register to wait for) */
            }
        };
        auto eval = [&](const uint32_t (&d)[4], uint32_t b0) {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (b0 + 64 * j >= len) break;                           /* wave-uniform: the table look-ups need all 64 lanes */
                const uint32_t o = b0 + 64 * j + lane;
                const uint32_t h = wave_ham_lookup(tab, d[j]);
                if (o < len) {
                    mn = h < mn ? h : mn;
                    /* (also asking for h <= 2 x the lane's minimum so far -- a necessary condition -- changed nothing: headline join 64.3 - 70.2 vs 62.5 - 65.8 ms,
                     * 10 M held-out reads 202.8 vs 201.6, alternating processes; profiles/r06_notes.md) */
                    if (h <= 7u) { c3 = c2; c2 = c1; c1 = c0; c0 = (o << 4) | h; n_c++; }
                }
            }
        };
        /* the previous run's slot stores may still be in flight; gfx9 counts loads and stores in ONE counter and orders them only among their own kind, so with a
         * store pending the compiler has to wait for vmcnt(0) at every use of a loaded value -- the prefetch would be waited for on the spot.  An explicit
         * s_waitcnt (the builtin: an instruction the compiler's wait-count pass sees, unlike inline assembly) empties the counter here; from then on
         * the loop holds loads only and waits for the older register set with vmcnt(4). */
        MTB_DRAIN_VMEM();
        fetch(v, 0);
        tab = wave_ham_table(s_hr, qdna);
        for (uint32_t b0 = 0; b0 < len; b0 += 512) {
            fetch(w, b0 + 256);
            eval(v, b0);
            if (b0 + 256 >= len) break;
            fetch(v, b0 + 512);
            eval(w, b0 + 256);
        }
        return mtb_ham_threshold(wave_min_shfl_u32(mn));
    };
    auto coop_scan = [&](uint32_t qdna, uint32_t &tab, uint64_t s, uint64_t e, uint32_t &c0, uint32_t &c1, uint32_t &c2, uint32_t &c3, uint32_t &n_c) -> uint32_t {
        if (WIN && use_win) return coop_scan_from(std::true_type(), qdna, tab, s, e, c0, c1, c2, c3, n_c);
        return coop_scan_from(std::false_type(), qdna, tab, s, e, c0, c1, c2, c3, n_c);
    };
    if (LIST) {
        /* dense list (owner side of the partitioned index): count the selected candidates of the thread's queries, one workgroup
         * scan, ONE atomic per workgroup for the output range, then emit (a per-match atomic on the list's single counter cost
         * 65 ms per 217 M matches, measured).  A record keeps its query's qinfo (ordinal tag included); pad = 1 on the query's
         * first match = the one that owns the ordinal slot at the read's home rank. */
        __shared__ uint32_t s_scan[8]; __shared__ unsigned long long s_base;
        uint64_t rs[Q], re[Q]; uint32_t thr_[Q], cnt[Q]; uint32_t tot_c = 0;
#pragma unroll
        for (int u = 0; u < Q; u++) {
            rs[u] = 0; re[u] = 0; thr_[u] = 0; cnt[u] = 0;
            if (!valid[u]) continue;
            const uint64_t s0 = lo[u], e = e_hi[u];
            const uint64_t v0 = rdv(s0);
            mtb_qrows qr; mtb_prepare_query_rows(s_hr, k[u].value, &qr);
            uint32_t mn = mtb_ham_sum(&qr, (uint32_t)v0 & 0xFFFFFFu);
            for (uint64_t t = s0 + 1; t < e; t++) { const uint32_t h = mtb_ham_sum(&qr, (uint32_t)rdv(t) & 0xFFFFFFu); mn = h < mn ? h : mn; }
            const uint32_t thr = mtb_ham_threshold(mn);
            uint32_t c = 0;
            for (uint64_t t = s0; t < e; t++) { const uint64_t v = t == s0 ? v0 : rdv(t); c += mtb_ham_sum(&qr, (uint32_t)v & 0xFFFFFFu) <= thr ? 1u : 0u; }
            rs[u] = s0; re[u] = e; thr_[u] = thr; cnt[u] = c; tot_c += c;
        }
        /* wave-scanned runs: threshold and count now, emission behind the reservation */
#pragma unroll
        for (int u = 0; u < Q; u++) {
            uint64_t todo = __ballot(lng[u]);
            while (todo) {
                const int src = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
                const uint64_t s0 = wave_bcast64(lo[u], src), e = wave_bcast64(e_hi[u], src);
                uint32_t tab, c0, c1, c2, c3, n_c;
                const uint32_t thr = coop_scan(wave_bcast32((uint32_t)k[u].value, src) & 0xFFFFFFu, tab, s0, e, c0, c1, c2, c3, n_c);
                uint32_t c = 0;
                if (!__any(n_c > 4u)) {
                    const uint32_t mine = (n_c > 0 && (c0 & 15u) <= thr ? 1u : 0u) + (n_c > 1 && (c1 & 15u) <= thr ? 1u : 0u) +
                                          (n_c > 2 && (c2 & 15u) <= thr ? 1u : 0u) + (n_c > 3 && (c3 & 15u) <= thr ? 1u : 0u);
                    c = wave_inclusive_scan<uint32_t>(mine);
                    c = (uint32_t)__shfl((int)c, 63, 64);
                } else {
                    for (uint64_t t0 = s0; t0 < e; t0 += 64) {
                        const uint64_t t = t0 + lane;
                        const uint32_t h = wave_ham_lookup(tab, t < e ? (uint32_t)rdv(t) : 0u);
                        c += (uint32_t)__popcll(__ballot(t < e && h <= thr));
                    }
                }
                if ((int)lane == src) { rs[u] = s0; re[u] = e; thr_[u] = thr; cnt[u] = c; tot_c += c; }
            }
        }
        uint32_t tot;
        const uint32_t off = block256_exclusive_scan<uint32_t>(tot_c, s_scan, &tot);
        if (threadIdx.x == 0) s_base = tot ? atomicAdd(sa.ovf_counter, (unsigned long long)tot) : 0ull;
        __syncthreads();
        unsigned long long o_of[Q];
        { unsigned long long o_run = s_base + off;
#pragma unroll
          for (int u = 0; u < Q; u++) { o_of[u] = o_run; o_run += cnt[u]; } }
#pragma unroll
        for (int u = 0; u < Q; u++) {            /* wave-scanned runs: selected candidates in index order from the query's offset, ranks by ballot */
            uint64_t todo = __ballot(lng[u] && cnt[u] != 0);
            while (todo) {
                const int src = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
                const uint64_t s0 = wave_bcast64(rs[u], src), e = wave_bcast64(re[u], src), qi = wave_bcast64(k[u].qinfo, src);
                const uint32_t thr = wave_bcast32(thr_[u], src);
                unsigned long long ob = wave_bcast64(o_of[u], src);
                const uint64_t qv = wave_bcast64(k[u].value, src);
                const uint32_t tab = wave_ham_table(s_hr, (uint32_t)qv & 0xFFFFFFu);
                mtb_qrows qr; mtb_prepare_query_rows(s_hr, qv, &qr);              /* (the per-codon fields of the selected candidates) */
                const bool rev = mtb_hammings_reversed(mtb_q_frame(qi), ix.kmer_format);
                bool first = true;
                for (uint64_t t0 = s0; t0 < e; t0 += 64) {
                    const uint64_t t = t0 + lane;
                    uint64_t v = 0; uint32_t td = 0;
                    if (t < e) { v = rdv(t); td = (uint32_t)v & 0xFFFFFFu; }
                    const uint32_t h = wave_ham_lookup(tab, td);
                    const bool sel = t < e && h <= thr;
                    const uint64_t m = __ballot(sel);
                    if (!m) continue;
                    const uint32_t rk = (uint32_t)__popcll(m & lt_mask);
                    if (sel && ob + rk < sa.ovf_cap) {
                        const int32_t tid = (int32_t)((PACKED ? (uint32_t)(v >> MTB_PACK_LOW) : ix.info[t]) & ix.info_mask);
                        mtb_match mm; mm.qinfo = qi; mm.target_id = tid; mm.species_id = (tid >= 0 && tid <= ix.max_taxid) ? ix.tax2species[tid] : 0;
                        mm.dna = td; mm.right_end_hamming = mtb_hammings(&qr, td, rev); mm.hamming = (uint8_t)h; mm.pad = (first && rk == 0) ? 1 : 0;
                        sa.ovf[ob + rk] = mm;
                    }
                    ob += (uint32_t)__popcll(m); first = false;
                }
            }
        }
#pragma unroll
        for (int u = 0; u < Q; u++) {
            if (cnt[u] == 0 || lng[u]) continue;
            unsigned long long o = o_of[u];
            mtb_qrows qr; mtb_prepare_query_rows(s_hr, k[u].value, &qr);
            const bool rev = mtb_hammings_reversed(mtb_q_frame(k[u].qinfo), ix.kmer_format);
            bool first = true;
            for (uint64_t t = rs[u]; t < re[u]; t++) {
                const uint64_t v = rdv(t);
                const uint32_t td = (uint32_t)v & 0xFFFFFFu;
                const uint32_t h = mtb_ham_sum(&qr, td);
                if (h > thr_[u]) continue;
                if (o < sa.ovf_cap) {
                    const int32_t tid = (int32_t)((PACKED ? (uint32_t)(v >> MTB_PACK_LOW) : ix.info[t]) & ix.info_mask);
                    mtb_match m; m.qinfo = k[u].qinfo; m.target_id = tid; m.species_id = (tid >= 0 && tid <= ix.max_taxid) ? ix.tax2species[tid] : 0;
                    m.dna = td; m.right_end_hamming = mtb_hammings(&qr, td, rev); m.hamming = (uint8_t)h; m.pad = first ? 1 : 0;
                    sa.ovf[o] = m;
                }
                o++; first = false;
            }
        }
        return;
    }
    const uint32_t tail_cap = sa.stride - sa.direct;
    /* a place in the overflow list: the workgroup's stripe (its own counter and region), or the single dense list */
    const uint32_t stripe = sa.ovf_stripes ? blockIdx.x & (sa.ovf_stripes - 1u) : 0u;
    /* `beyond` = the match's place behind the read's tail (its tail cursor value - the tail's capacity: unique and dense per read).  It rides
     * in the record -- bits [16, 32) of qinfo, which the stripped qinfo leaves free, and pad = 2 says so -- so that k_ovf_group places the
     * entry into its read's group WITHOUT a returning atomic per entry (14 of 115 ms on reads of organisms that are not in the index);
     * every reader of the list clears both again.  A read with more than 65535 entries is not grouped (k_ovf_count) and takes the exact path. */
    auto ovf_put = [&](mtb_match mm, uint32_t beyond) {
        const unsigned long long o = atomicAdd(sa.ovf_counter + 8u * stripe, 1ull);
        if (LONG) return;                          /* counted only: the caller retries the join with a larger tail */
        mm.qinfo |= (uint64_t)(beyond < 0xFFFFu ? beyond : 0xFFFFu) << 16; mm.pad = 2;
        const unsigned long long room = sa.ovf_stripes ? sa.ovf_region : sa.ovf_cap;
        if (o < room) sa.ovf[(uint64_t)stripe * sa.ovf_region + o] = mm; else *overflow = 1;
    };
    /* wave-scanned runs: one pass (minimum + the few candidates that can be selected, kept in registers), then emission -- the selected
     * candidate with the lowest index takes the query's ordinal slot, the others the read's tail (ONE returning atomic per step for all
     * of them), beyond that the overflow list: the contract of the per-lane loop below */
#pragma unroll
    for (int u = 0; u < Q; u++) {
        uint64_t todo = __ballot(lng[u]);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
            const uint64_t s0 = wave_bcast64(lo[u], src), e = wave_bcast64(e_hi[u], src), qi_t = wave_bcast64(k[u].qinfo, src);
            const uint64_t qv = wave_bcast64(k[u].value, src);
            uint32_t tab, cb[4], n_c;
            const uint32_t thr = coop_scan((uint32_t)qv & 0xFFFFFFu, tab, s0, e, cb[0], cb[1], cb[2], cb[3], n_c);
            mtb_qrows qr; mtb_prepare_query_rows(s_hr, qv, &qr);                  /* (the per-codon fields of the selected candidates: after the scan, which needs the table only) */
            const uint32_t r = mtb_q_seq(qi_t) - 1, ord = mtb_q_pos(qi_t) >> 16;
            const uint64_t qinfo = qi_t & ~0xFFFF0000ull;
            const bool rev = mtb_hammings_reversed(mtb_q_frame(qinfo), ix.kmer_format);
            uint32_t direct = sa.direct, tcap = tail_cap;
            mtb_slot16 *seg;
            if (LONG) { direct = sa.dcnt[r]; tcap = mtb_lslot_tail(direct, sa.tf); seg = sa.seg + sa.rb[r]; }
            else seg = sa.seg + (uint64_t)r * sa.stride;
            const bool offr = !LONG && sa.off && sa.off[r];            /* a read the slot records cannot hold (positions / metamer count) */
            const uint32_t inc = offr ? tcap + 1u : 1u;
            bool first = ord < direct && !offr;
            /* one selected candidate -> its slot: `at` = place in the read's tail, or ~0u for the query's ordinal slot */
            auto put = [&](uint64_t t, uint64_t v, uint32_t h, uint32_t at) {
                const uint32_t td = (uint32_t)v & 0xFFFFFFu;
                const int32_t tid = (int32_t)((PACKED ? (uint32_t)(v >> MTB_PACK_LOW) : ix.info[t]) & ix.info_mask);
                const int32_t sp = (tid >= 0 && tid <= ix.max_taxid) ? ix.tax2species[tid] : 0;
                const uint16_t reh = mtb_hammings(&qr, td, rev);
                const mtb_slot16 sl = LONG ? mtb_lslot_pack(qinfo, tid, sp, td, reh, h) : mtb_slot_pack(qinfo, tid, sp, td, reh, h, sa.epoch);
                if (at == ~0u) MTB_SLOT_STORE(sl, &seg[ord]);
                else if (at < tcap) MTB_SLOT_STORE(sl, &seg[direct + at]);
                else {
                    mtb_match mm; mm.qinfo = qinfo; mm.target_id = tid; mm.species_id = sp; mm.dna = td; mm.right_end_hamming = reh; mm.hamming = (uint8_t)h; mm.pad = 0;
                    ovf_put(mm, at - tcap);
                }
            };
            if (!__any(n_c > 4u)) {
                /* from the registers: the lowest selected offset of the wave owns the ordinal slot */
                uint32_t low = ~0u;
#pragma unroll
                for (int b = 0; b < 4; b++) if ((uint32_t)b < n_c && (cb[b] & 15u) <= thr) low = (cb[b] >> 4) < low ? (cb[b] >> 4) : low;
                low = first ? wave_min_shfl_u32(low) : ~0u;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const bool sel = (uint32_t)b < n_c && (cb[b] & 15u) <= thr;
                    const uint32_t off = cb[b] >> 4;
                    const bool own = sel && off == low;              /* (~0u never equals an offset: runs are shorter than 2^28) */
                    const uint64_t m = __ballot(sel && !own);
                    uint32_t at0 = 0;
                    if (m) {
                        const int leader = __ffsll((unsigned long long)m) - 1;
                        if ((int)lane == leader) at0 = atomicAdd(&sa.cursor[r], (uint32_t)__popcll(m) * inc);
                        at0 = wave_bcast32(at0, leader);
                    }
                    if (sel) put(s0 + off, full_of(s0 + off, rdv(s0 + off)), cb[b] & 15u, own ? ~0u : (offr ? tcap : at0 + (uint32_t)__popcll(m & lt_mask)));
                }
                continue;
            }
            for (uint64_t t0 = s0; t0 < e; t0 += 64) {      /* a lane met more than four possible candidates: second walk, 64 per step */
                const uint64_t t = t0 + lane;
                uint64_t v = 0;
                if (t < e) v = rdv(t);
                const uint32_t h = wave_ham_lookup(tab, (uint32_t)v);
                const bool sel = t < e && h <= thr;
                const uint64_t m = __ballot(sel);
                if (!m) continue;
                const uint32_t rk = (uint32_t)__popcll(m & lt_mask), n_sel = (uint32_t)__popcll(m);
                const uint32_t skip = first ? 1u : 0u, n_tail = n_sel - skip;
                uint32_t at0 = 0;
                if (n_tail) {
                    const int leader = __ffsll((unsigned long long)m) - 1;
                    if ((int)lane == leader) at0 = atomicAdd(&sa.cursor[r], n_tail * inc);
                    at0 = wave_bcast32(at0, leader);
                }
                if (sel) put(t, full_of(t, v), h, (first && rk == 0) ? ~0u : (offr ? tcap : at0 + rk - skip));
                first = false;
            }
        }
    }
    MTB_JP_MARK(3);
#pragma unroll
    for (int u = 0; u < Q; u++) {
        if (!valid[u]) continue;
        const uint64_t s = lo[u], e = e_hi[u];
        /* a tile with a window reads the run's first four candidates at once (low dwords: all the evaluation needs) with the first one's full word from
         * global memory next to them (it is the one selected more often than not): ONE round trip before the evaluation instead of one per candidate and
         * pass -- runs are 1 - 4 entries as a rule, and the loops below were two chains of dependent reads (per-lane evaluation + emission: 38 % of the
         * kernel's cycles).  Their sums are kept for the emission.  Without a window: the first candidate's word, the others one by one as before. */
        const uint32_t len = (uint32_t)(e - s);                /* (<= coop_min: longer runs went to the wave) */
        const bool win4 = WIN && use_win;                      /* a tile with a window: four LDS words with one address (the array is padded by three words) */
        const uint32_t n_pre = win4 ? 4u : 1u;
        uint32_t d0, d1 = 0, d2 = 0, d3 = 0, hi0 = 0;
        if (win4) { const uint32_t *pw = s_win + (s - w0); d0 = pw[0]; d1 = pw[1]; d2 = pw[2]; d3 = pw[3]; if (PACKED) hi0 = ((const uint32_t *)ix.values)[2 * s + 1]; }
        else { const uint64_t v0 = ix.values[s]; d0 = (uint32_t)v0; hi0 = (uint32_t)(v0 >> 32); }
        const uint32_t info0 = PACKED ? 0u : ix.info[s];
        mtb_qrows qr; mtb_prepare_query_rows(s_hr, k[u].value, &qr);
        /* the sums of the candidates read ahead, a byte each (a sum is <= 32); the others are read one by one */
        uint32_t hp = mtb_ham_sum(&qr, d0 & 0xFFFFFFu);
        uint32_t mn = hp;
        if (win4) {
            { const uint32_t h = mtb_ham_sum(&qr, d1 & 0xFFFFFFu); hp |= h << 8; if (len > 1u) mn = h < mn ? h : mn; }
            { const uint32_t h = mtb_ham_sum(&qr, d2 & 0xFFFFFFu); hp |= h << 16; if (len > 2u) mn = h < mn ? h : mn; }
            { const uint32_t h = mtb_ham_sum(&qr, d3 & 0xFFFFFFu); hp |= h << 24; if (len > 3u) mn = h < mn ? h : mn; }
        }
        for (uint64_t t = s + n_pre; t < e; t++) { const uint32_t h = mtb_ham_sum(&qr, (uint32_t)rdv(t) & 0xFFFFFFu); mn = h < mn ? h : mn; }
        const uint32_t thr = mtb_ham_threshold(mn);
        const uint32_t r = mtb_q_seq(k[u].qinfo) - 1;
        const uint32_t ord = mtb_q_pos(k[u].qinfo) >> 16;
        const uint64_t qinfo = k[u].qinfo & ~0xFFFF0000ull;      /* the record carries the reference's qinfo */
        const bool rev = mtb_hammings_reversed(mtb_q_frame(qinfo), ix.kmer_format);
        uint32_t direct = sa.direct, tcap = tail_cap;
        mtb_slot16 *seg;
        if (LONG) { direct = sa.dcnt[r]; tcap = mtb_lslot_tail(direct, sa.tf); seg = sa.seg + sa.rb[r]; }
        else seg = sa.seg + (uint64_t)r * sa.stride;
        const bool offr = !LONG && sa.off && sa.off[r];
        bool first = ord < direct && !offr;
        for (uint32_t i = 0; i < len; i++) {
            uint32_t low, h;
            if (i < n_pre) { low = d0; h = hp & 255u; d0 = d1; d1 = d2; d2 = d3; hp >>= 8; }       /* (one loop body for all candidates: the registers rotate) */
            else { low = (uint32_t)rdv(s + i); h = mtb_ham_sum(&qr, low & 0xFFFFFFu); }
            if (h > thr) continue;
            const uint64_t t = s + i;
            const uint32_t td = low & 0xFFFFFFu;
            const int32_t tid = (int32_t)((PACKED ? (((i == 0u ? hi0 : ((const uint32_t *)ix.values)[2 * t + 1]) << (32 - MTB_PACK_LOW)) | (low >> MTB_PACK_LOW)) : (i == 0u ? info0 : ix.info[t])) & ix.info_mask);
            const int32_t sp = (tid >= 0 && tid <= ix.max_taxid) ? ix.tax2species[tid] : 0;
            const uint16_t reh = mtb_hammings(&qr, td, rev);
            /* non-temporal stores: a slot line is written ~5 times at unrelated moments of the kernel and never read by it; keeping
             * those lines out of the L2's way measured 47.5 ms against 50-58 ms (and steadier) for the kernel -- which is bound by
             * these 1.1 G scattered 16-byte stores: 20.5 ms without them, 29 ms with dense stores (profiles/r02_notes.md) */
            if (first) { const mtb_slot16 sl = LONG ? mtb_lslot_pack(qinfo, tid, sp, td, reh, h) : mtb_slot_pack(qinfo, tid, sp, td, reh, h, sa.epoch);
                         MTB_SLOT_STORE(sl, &seg[ord]); first = false; continue; }
            const uint32_t at = offr ? (atomicAdd(&sa.cursor[r], tcap + 1u), tcap) : atomicAdd(&sa.cursor[r], 1u);
            if (at < tcap) { const mtb_slot16 sl = LONG ? mtb_lslot_pack(qinfo, tid, sp, td, reh, h) : mtb_slot_pack(qinfo, tid, sp, td, reh, h, sa.epoch);
                             MTB_SLOT_STORE(sl, &seg[direct + at]); }
            else {
                mtb_match m; m.qinfo = qinfo; m.target_id = tid; m.species_id = sp; m.dna = td; m.right_end_hamming = reh; m.hamming = (uint8_t)h; m.pad = 0;
                ovf_put(m, at - tcap);
            }
        }
    }
    MTB_JP_MARK(4);
    MTB_END_RELEASE();
}

/* ---- diagnostic (not on the timed path): what the directory join addresses on the index side --------------------------------
 * One thread per query of the last batch: its bucket, the 64-byte sectors of the two directory words it reads and the 64-byte
 * sectors of the bucket's span in the target array are marked in bitmaps; k_popcount_words counts them.  This is the batch's
 * distinct index-side working set -- the lower bound of what k_join_dir has to fetch whatever the caches do (bench.py reports it
 * next to the PMC traffic: `footprint`). */
__global__ __launch_bounds__(256) void k_join_footprint(const mtb_kmer *__restrict__ q, uint64_t n, mtb_dir_view dv, uint64_t T,
                                                         uint32_t *__restrict__ bm_bucket, uint32_t *__restrict__ bm_dirsec, uint32_t *__restrict__ bm_tgtsec,
                                                         unsigned long long *__restrict__ n_valid) {
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (j >= n) return;
    const mtb_kmer k = q[j];
    if (mtb_q_seq(k.qinfo) == 0) return;
    const uint32_t b = mtb_dir_bucket(k.value, dv.L, dv.kmer_format);
    if (b >= dv.n_buckets) return;
    atomicAdd(n_valid, 1ull);
    atomicOr(&bm_bucket[b >> 5], 1u << (b & 31u));
    const uint64_t ds0 = ((uint64_t)b * 4) >> 6, ds1 = ((uint64_t)(b + 1) * 4) >> 6;
    atomicOr(&bm_dirsec[ds0 >> 5], 1u << (ds0 & 31u));
    if (ds1 != ds0) atomicOr(&bm_dirsec[ds1 >> 5], 1u << (ds1 & 31u));
    uint64_t lo = dv.base[b >> 16] + dv.dir[b], hi = dv.base[(b + 1) >> 16] + dv.dir[b + 1];
    if (hi > T) hi = T;
    for (uint64_t sct = (lo * 8) >> 6; lo < hi && sct <= ((hi * 8 - 1) >> 6); sct++) atomicOr(&bm_tgtsec[sct >> 5], 1u << (sct & 31u));
}
/* ---- diagnostics (not on the timed path): candidate-run lengths ------------------------------------------------------------------
 * k_index_run_hist: runs of equal amino-acid parts of a FLAT target array, counted at their first entry: hist[b] += 1 and
 * hist[32 + b] += length for b = floor(log2(length)) -- the index-side run-length distribution (heavy-tailed in real
 * databases).  k_join_run_hist: for every query of the last batch the length of the run it meets (what k_join_dir scans for it):
 * hist[0] = queries without a candidate, hist[1 + b] = queries with floor(log2(length)) = b, hist[40 + b] = targets in those runs,
 * hist[33] / hist[34] = queries (and the targets of their runs) that find their own DNA part in a run beyond MTB_JOIN_EXACT_MIN: the join
 * selects that block by bisection instead of scanning the run. */
__global__ __launch_bounds__(256) void k_index_run_hist(const uint64_t *__restrict__ values, uint64_t T, unsigned long long *__restrict__ hist) {
    const uint64_t AAM = ~0xFFFFFFull;
    __shared__ unsigned long long s_h[64];           /* per workgroup: 13 G runs on two global counters would serialise on them */
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < T; i += (uint64_t)gridDim.x * 256) {
        const uint64_t aa = values[i] & AAM;
        if (i > 0 && (values[i - 1] & AAM) == aa) continue;
        uint64_t e = i + 1;
        while (e < T && (values[e] & AAM) == aa) e++;
        const uint32_t b = 63u - (uint32_t)__clzll((unsigned long long)(e - i));
        atomicAdd(&s_h[b < 31u ? b : 31u], 1ull); atomicAdd(&s_h[32 + (b < 31u ? b : 31u)], (unsigned long long)(e - i));
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
template <bool PACKED>
__global__ __launch_bounds__(256) void k_join_run_hist(const mtb_kmer *__restrict__ q, uint64_t n, const uint64_t *__restrict__ values, uint64_t limit, mtb_dir_view dv,
                                                        unsigned long long *__restrict__ hist) {
    __shared__ unsigned long long s_h[64];
    if (threadIdx.x < 64) s_h[threadIdx.x] = 0;
    __syncthreads();
    const uint64_t j = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    mtb_kmer k; k.value = 0; k.qinfo = 0;
    if (j < n) k = q[j];
    const bool live = j < n && mtb_q_seq(k.qinfo) != 0;
    const uint64_t AAM = ~0xFFFFFFull;
    auto tkey = [&](uint64_t w) -> uint64_t { return PACKED ? (w & 0x1F000000ull) : (w & AAM); };
    const uint64_t qk = !PACKED ? (k.value & AAM) : (dv.kmer_format == 1 ? (((k.value >> 24) % 21ull) << 24) : (k.value & 0x1F000000ull));
    const uint32_t b = mtb_dir_bucket(k.value, dv.L, dv.kmer_format);
    uint64_t lo = 0, hi = 0;
    if (live && b < dv.n_buckets) { lo = dv.base[b >> 16] + dv.dir[b]; hi = dv.base[(b + 1) >> 16] + dv.dir[b + 1]; if (hi > limit) hi = limit; if (lo > hi) lo = hi; }
    uint64_t a = lo, z = hi;
    while (a < z) { const uint64_t mid = a + ((z - a) >> 1); if (tkey(values[mid]) < qk) a = mid + 1; else z = mid; }
    const uint64_t s = a;
    z = hi;
    while (a < z) { const uint64_t mid = a + ((z - a) >> 1); if (tkey(values[mid]) <= qk) a = mid + 1; else z = mid; }
    const uint64_t len = a - s;
    if (live) {
        if (len == 0) atomicAdd(&s_h[0], 1ull);
        else {
            const uint32_t bin = 63u - (uint32_t)__clzll((unsigned long long)len);
            atomicAdd(&s_h[1 + (bin < 31u ? bin : 31u)], 1ull); atomicAdd(&s_h[40 + (bin < 23u ? bin : 23u)], (unsigned long long)len);
            /* does the run hold the query's own DNA part?  (then the join selects that block without a scan) */
            if (len > (uint64_t)MTB_JOIN_EXACT_MIN) {
                const uint32_t qd = (uint32_t)k.value & 0xFFFFFFu;
                uint64_t x = s, y = a;
                while (x < y) { const uint64_t mid = x + ((y - x) >> 1); if (((uint32_t)values[mid] & 0xFFFFFFu) < qd) x = mid + 1; else y = mid; }
                if (x < a && ((uint32_t)values[x] & 0xFFFFFFu) == qd) { atomicAdd(&s_h[33], 1ull); atomicAdd(&s_h[34], (unsigned long long)len); }
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 64 && s_h[threadIdx.x]) atomicAdd(&hist[threadIdx.x], s_h[threadIdx.x]);
}
__global__ __launch_bounds__(256) void k_popcount_words(const uint32_t *__restrict__ w, uint64_t n_words, unsigned long long *__restrict__ out) {
    unsigned long long acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) acc += (unsigned long long)__popc(w[i]);
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0 && acc) atomicAdd(out, acc);
}

#endif
