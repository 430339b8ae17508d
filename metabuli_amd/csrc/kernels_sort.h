/* kernels_sort.h -- stable LSD radix sort of 16-byte query k-mer records on
 * the 64-bit metamer value (replaces SORT_PARALLEL(..., Kmer::compareQueryKmer),
 * src/commons/KmerExtractor.cpp:79; equal values keep extraction order, which
 * is ascending sequenceID, so the result is a valid (value, seqID) order).
 *
 * Per pass: histogram kernel (counts per tile, digit-major table), device scan,
 * scatter kernel.  Stage API (mtb_sort_kmers): 8-bit digits over all 64 bits,
 * 2048-record tiles.  Fused path: three passes whose digit is a PAIR of
 * amino-acid letters (441 of 512 bins), 4096-record tiles, histograms from
 * 2-byte digit side arrays.  The scatter ranks a tile stably with wave64 ballot
 * matching, reorders it through LDS so that every digit's records leave the CU
 * as one contiguous run, and adds the scanned tile offsets; tiles are handed
 * out in XCD-contiguous order so that the adjacent runs of neighbouring tiles
 * merge in one L2.
 * Algorithmic HBM bytes per pass: 16 (read) + 16 (write) per record, + 2 + 2
 * for the digit side arrays (16 instead of 2 where the histogram reads records). */
#ifndef MTB_KERNELS_SORT_H
#define MTB_KERNELS_SORT_H
#include "dev_util.h"
#include "kernels_scan.h"
#include "mtb_core.h"

#define MTB_SORT_TILE 2048
#ifndef MTB_SORT_ITEMS
#define MTB_SORT_ITEMS 8
#endif

/* Digit of one pass.  MODE 0: 8 binary bits at `shift` (256 bins).  MODE 1 (kmer_format 2 only): the two 5-bit
 * amino-acid letters at `shift` as one base-21 digit, letter codes are 0..20 -> 441 of 512 bins; three such passes
 * (shift 34, 44, 54) order the metamers by their first six amino acids = bits [34,64), where four binary passes
 * order bits [32,64): the join only needs tiles with a narrow amino-acid range, and a six-letter prefix already
 * resolves ~94 targets of an 8 G index.                                                                          */
template <int MODE>
__device__ __forceinline__ uint32_t radix_digit(uint64_t v, int shift) {
    if (MODE == 0) return (uint32_t)(v >> shift) & 255u;
    uint32_t two = (uint32_t)(v >> shift) & 1023u;
    uint32_t d = (two >> 5) * 21u + (two & 31u);
    return d < 511u ? d : 511u;                 /* letters > 20 never leave the extractor; keep the index in range anyway */
}

/* THREADS x MTB_SORT_ITEMS elements per tile: a digit's run inside a tile should be about a 128-byte line
 * (8 records) long, so 256 bins go with 2048-element tiles and 512 bins with 4096-element tiles (measured:
 * 512 bins on 2048-element tiles made a pass 40 % slower -- 4.6-record runs). */
template <int NB, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_radix_hist(const mtb_kmer *__restrict__ in, uint64_t n, int shift,
                                                         uint32_t *__restrict__ hist, uint32_t num_tiles) {
    __shared__ uint32_t s_h[NB];
    for (int b = threadIdx.x; b < NB; b += THREADS) s_h[b] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * (THREADS * MTB_SORT_ITEMS);
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)r * THREADS + threadIdx.x;
        if (i < n) atomicAdd(&s_h[radix_digit<MODE>(in[i].value, shift)], 1u);
    }
    __syncthreads();
    for (int b = threadIdx.x; b < NB; b += THREADS) hist[(uint64_t)b * num_tiles + blockIdx.x] = s_h[b];
}

/* Histogram of a pass from a side array of 16-bit digits (written by the extractor for the first pass and by the
 * previous scatter for the later ones): 2 bytes per record instead of the 16-byte record for a 9-bit digit.
 * A workgroup counts MTB_HIST_GROUP consecutive tiles and writes, per bin, the counts of those tiles as one 64-byte
 * run of the digit-major table (one bin per thread): writing tile by tile put a lone 4-byte word into every sector,
 * 5 GB of HBM writes for a 0.64 GB table (PMC WRITE_SIZE). */
#define MTB_HIST_GROUP 16
template <int NB, int THREADS, int TILE = THREADS * MTB_SORT_ITEMS>
__global__ __launch_bounds__(THREADS) void k_radix_hist_dig(const uint16_t *__restrict__ dig, uint64_t n, uint32_t *__restrict__ hist, uint32_t num_tiles) {
    static_assert(NB == THREADS, "one bin per thread");
    __shared__ uint32_t s_h[MTB_HIST_GROUP][NB];
    const uint32_t tile0 = blockIdx.x * MTB_HIST_GROUP;
#pragma unroll
    for (int g = 0; g < MTB_HIST_GROUP; g++) s_h[g][threadIdx.x] = 0;
    __syncthreads();
    for (int g = 0; g < MTB_HIST_GROUP; g++) {
        const uint64_t base = (uint64_t)(tile0 + g) * TILE;
        if (base >= n) break;
        if (TILE == THREADS * 8 && base + TILE <= n) {
            /* whole tile: eight digits per thread in one 16-byte load (the side arrays are hipMalloc'ed and TILE * 2 bytes is a multiple of 16) */
            const uint4 q = *(const uint4 *)(dig + base + 8u * threadIdx.x);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d0 = w[k] & 0xFFFFu, d1 = w[k] >> 16;
                atomicAdd(&s_h[g][d0 < NB ? d0 : NB - 1], 1u);
                atomicAdd(&s_h[g][d1 < NB ? d1 : NB - 1], 1u);
            }
            continue;
        }
#pragma unroll
        for (int r = 0; r < TILE / THREADS; r++) {
            uint64_t i = base + (uint64_t)r * THREADS + threadIdx.x;
            if (i < n) { uint32_t d = dig[i]; atomicAdd(&s_h[g][d < NB ? d : NB - 1], 1u); }
        }
    }
    __syncthreads();
    uint32_t *row = hist + (uint64_t)threadIdx.x * num_tiles + tile0;
    if (tile0 + MTB_HIST_GROUP <= num_tiles && (((uint64_t)threadIdx.x * num_tiles + tile0) & 3u) == 0) {
#pragma unroll
        for (int g = 0; g < MTB_HIST_GROUP; g += 4) {
            uint4 v; v.x = s_h[g][threadIdx.x]; v.y = s_h[g + 1][threadIdx.x]; v.z = s_h[g + 2][threadIdx.x]; v.w = s_h[g + 3][threadIdx.x];
            *(uint4 *)(row + g) = v;
        }
    } else {
        for (int g = 0; g < MTB_HIST_GROUP && tile0 + g < num_tiles; g++) row[g] = s_h[g][threadIdx.x];
    }
}

#define MTB_SORT_NBKT 512
/* bucket of a tile: last b with first_tile[b] <= t */
__device__ __forceinline__ uint32_t sort_tile_bucket(const uint32_t *__restrict__ first_tile, uint32_t t) {
    uint32_t lo = 0, hi = MTB_SORT_NBKT;
    while (hi - lo > 1) { const uint32_t mid = (lo + hi) >> 1; if (first_tile[mid] <= t) lo = mid; else hi = mid; }
    return lo;
}

/* Scatter of one pass.  Every wavefront owns a contiguous eighth (THREADS/64-th) of the tile and ranks its records
 * on its own: per round the lanes with equal digits find each other with ballots, the first of them bumps the wave's
 * private counter of that digit and all take "counter before + rank among peers" -- LDS operations of one wave
 * execute in order, so no workgroup barrier is needed until all rounds are done (the first version synchronised the
 * workgroup three times per round).  Wave-major ranks = index order inside the tile, so the pass is stable.     */
template <int NB, int MODE, int THREADS>
__global__ __launch_bounds__(THREADS) void k_radix_scatter(const mtb_kmer *__restrict__ in, mtb_kmer *__restrict__ out,
                                                            uint64_t n, int shift, const uint32_t *__restrict__ tile_off,
                                                            uint32_t num_tiles, uint16_t *__restrict__ dig_out = nullptr, int next_shift = 0, int xcd_map = 1,
                                                            const uint32_t *__restrict__ plan = nullptr /* bucket-local pass: tiles and table are bucket-major */) {
    constexpr int BITS = NB == 256 ? 8 : 9;
    constexpr int NW = THREADS / 64;
    constexpr int TILE = THREADS * MTB_SORT_ITEMS;
    static_assert(NB <= THREADS, "at least one thread per bin");
    __shared__ uint16_t s_cnt[NW][NB];             /* per wave: records of the digit so far; later: start of the wave's run */
    __shared__ uint16_t s_start[NB];
    __shared__ uint32_t s_tmp[NW];
    __shared__ mtb_kmer s_buf[TILE];
    __shared__ uint32_t s_goff[NB];
    const uint32_t t = threadIdx.x, w = t >> 6, lane = t & 63u;
    /* workgroups go to the 8 XCDs round-robin: XCD x takes the x-th eighth of the tiles in order, so that the runs of one bin
     * written by neighbouring tiles (adjacent in memory) meet in ONE L2 and leave it as whole lines, and the tile_off
     * sectors are fetched once per 16 tiles instead of once per tile */
    if (plan) num_tiles = plan[513 + 512];
    const uint32_t per_xcd = (num_tiles + 7u) >> 3;
    const uint32_t tile = xcd_map ? (blockIdx.x & 7u) * per_xcd + (blockIdx.x >> 3) : blockIdx.x;
    if (tile >= num_tiles) return;
    uint64_t base = (uint64_t)tile * TILE;
    uint32_t my_off;
    if (plan) {
        const uint32_t *tfirst = plan + 513;
        const uint32_t b = sort_tile_bucket(tfirst, tile);
        const uint32_t j = tile - tfirst[b], ntb = tfirst[b + 1] - tfirst[b];
        base = (uint64_t)plan[b] + (uint64_t)j * TILE;
        n = plan[b + 1];                                                     /* the bucket's end bounds the (possibly partial) tile */
        my_off = t < NB ? tile_off[(uint64_t)512 * tfirst[b] + (uint64_t)t * ntb + j] : 0u;
    } else
    my_off = t < NB ? tile_off[(uint64_t)t * num_tiles + tile] : 0u;          /* bin t of this tile: global start */
    mtb_kmer e[MTB_SORT_ITEMS];
    uint32_t lrank[MTB_SORT_ITEMS];
    if (t < NB) {
#pragma unroll
        for (int k = 0; k < NW; k++) s_cnt[k][t] = 0;
    }
    /* all loads of the thread in flight: wave w owns records [w*64*ITEMS, (w+1)*64*ITEMS) of the tile */
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)w * (64 * MTB_SORT_ITEMS) + (uint64_t)r * 64 + lane;
        if (i < n) e[r] = in[i]; else { e[r].value = 0; e[r].qinfo = 0; }
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)w * (64 * MTB_SORT_ITEMS) + (uint64_t)r * 64 + lane;
        bool valid = i < n;
        uint32_t d = radix_digit<MODE>(e[r].value, shift);
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < BITS; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
        uint32_t rank_in_wave = (uint32_t)__popcll(peers & lanemask_lt());
        uint32_t before = s_cnt[w][d];                                  /* every peer reads the same value ... */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        if (valid && rank_in_wave == 0) s_cnt[w][d] = (uint16_t)(before + (uint32_t)__popcll(peers));   /* ... before the first one bumps it */
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier();
        lrank[r] = before + rank_in_wave;
    }
    __syncthreads();
    /* bin t: exclusive offsets of the waves' runs inside the bin, bin total -> exclusive scan over bins */
    uint32_t run = 0;
    if (t < NB) {
#pragma unroll
        for (int k = 0; k < NW; k++) { uint32_t c = s_cnt[k][t]; s_cnt[k][t] = (uint16_t)run; run += c; }
    }
    uint32_t tot;
    uint32_t ex = block_exclusive_scan<uint32_t, NW>(run, s_tmp, &tot);
    if (t < NB) {
        s_start[t] = (uint16_t)ex;
        s_goff[t] = my_off - ex;                                        /* destination of record i of bin t = s_goff[t] + i */
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)w * (64 * MTB_SORT_ITEMS) + (uint64_t)r * 64 + lane;
        if (i < n) {
            uint32_t d = radix_digit<MODE>(e[r].value, shift);
            s_buf[(uint32_t)s_start[d] + s_cnt[w][d] + lrank[r]] = e[r];
        }
    }
    __syncthreads();
    uint32_t cnt = (uint32_t)((n - base) < TILE ? (n - base) : TILE);
    for (uint32_t i = t; i < cnt; i += THREADS) {
        mtb_kmer x = s_buf[i];
        uint32_t d = radix_digit<MODE>(x.value, shift);
        const uint64_t dst = (uint64_t)(s_goff[d] + i);
        out[dst] = x;
        if (dig_out) dig_out[dst] = (uint16_t)radix_digit<MODE>(x.value, next_shift);      /* next pass's histogram input */
    }
}

/* ---- bucket-local passes (fused path, kmer_format 2) -------------------------------------------------------------------------
 * The three letter-pair passes used to be LSD: every pass scatters every tile's 441 runs over the whole 20 GB output, 441 open pages
 * per tile, and the scatter is bound by address translation (UTCL2 busy 96 %, profiles/r02_notes.md).  Now the FIRST pass takes the
 * TOP letter pair (the extractor knows it) and leaves 441 buckets of ~46 MB; the two lower pairs are then sorted INSIDE every bucket
 * (low pair first, stable): same records moved, same final order (top, middle, low), but a tile's 441 runs of those two passes fall
 * into one bucket = 23 pages.  A bucket is cut into tiles of its own (the last one partial); the tile list is bucket-major, and so is
 * the histogram table (bucket, bin, tile-in-bucket) -- one global exclusive scan of it yields global destinations, because all
 * records of bucket b precede those of bucket b + 1.
 * plan[0 .. 512]   = first record of every bucket (plan[512] = n)        plan[513 .. 1025] = first tile of every bucket (last = tiles) */
__global__ __launch_bounds__(512) void k_sort_plan(const uint32_t *__restrict__ scanned, uint32_t num_tiles, uint64_t n, uint32_t tile, uint32_t *__restrict__ plan) {
    __shared__ uint32_t s_tmp[8];
    const uint32_t b = threadIdx.x;
    const uint32_t start = scanned[(uint64_t)b * num_tiles];               /* bin b of the first pass, tile 0: where the bucket begins */
    const uint32_t next = b + 1 < MTB_SORT_NBKT ? scanned[(uint64_t)(b + 1) * num_tiles] : (uint32_t)n;
    const uint32_t nt = (next - start + tile - 1) / tile;
    uint32_t tot;
    const uint32_t first = block_exclusive_scan<uint32_t, 8>(nt, s_tmp, &tot);
    plan[b] = start; plan[513 + b] = first;
    if (b == 0) { plan[512] = (uint32_t)n; plan[513 + 512] = tot; }
}
/* histogram of one bucket-local pass from the 2-byte digit side array; table entry of (bucket b, bin x, tile j of the bucket) =
 * 512 * first_tile[b] + x * tiles_of(b) + j.  A workgroup takes MTB_HIST_GROUP consecutive tiles; inside one bucket it writes, per
 * bin, their counts as one run (the common case), across a bucket boundary tile by tile. */
template <int NB, int THREADS, int TILE = THREADS * MTB_SORT_ITEMS>
__global__ __launch_bounds__(THREADS) void k_radix_hist_seg(const uint16_t *__restrict__ dig, const uint32_t *__restrict__ plan, uint32_t *__restrict__ hist) {
    static_assert(NB == THREADS, "one bin per thread");
    __shared__ uint32_t s_h[MTB_HIST_GROUP][NB];
    const uint32_t *bstart = plan, *tfirst = plan + 513;
    const uint32_t tiles = tfirst[512];
    const uint32_t tile0 = blockIdx.x * MTB_HIST_GROUP;
    if (tile0 >= tiles) return;
#pragma unroll
    for (int g = 0; g < MTB_HIST_GROUP; g++) s_h[g][threadIdx.x] = 0;
    __syncthreads();
    const uint32_t b0 = sort_tile_bucket(tfirst, tile0);
    uint32_t b = b0;
    for (int g = 0; g < MTB_HIST_GROUP; g++) {
        const uint32_t t = tile0 + g;
        if (t >= tiles) break;
        while (tfirst[b + 1] <= t) b++;                                     /* (empty buckets own no tile) */
        const uint32_t base = bstart[b] + (t - tfirst[b]) * TILE;
        const uint32_t end = bstart[b + 1] < base + TILE ? bstart[b + 1] : base + TILE;
        if (TILE == THREADS * 8 && end == base + TILE && (base & 7u) == 0) {
            const uint4 q = *(const uint4 *)(dig + base + 8u * threadIdx.x);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t d0 = w[k] & 0xFFFFu, d1 = w[k] >> 16;
                atomicAdd(&s_h[g][d0 < NB ? d0 : NB - 1], 1u);
                atomicAdd(&s_h[g][d1 < NB ? d1 : NB - 1], 1u);
            }
            continue;
        }
        /* a bucket starts anywhere, so most tiles are not 16-byte aligned: aligned 16-byte chunks over [base, end) with the digits
         * outside the tile masked (eight 2-byte loads per thread made this kernel 1.8 ms against the aligned kernel's 1.0; the side
         * arrays end with slack, dev_sort) */
        const uint32_t a0 = base & ~7u;
        for (uint32_t ch = threadIdx.x; a0 + 8u * ch < end; ch += THREADS) {
            const uint32_t first = a0 + 8u * ch;
            const uint4 q = *(const uint4 *)(dig + first);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t i0 = first + 2u * k, d0 = w[k] & 0xFFFFu, d1 = w[k] >> 16;
                if (i0 >= base && i0 < end) atomicAdd(&s_h[g][d0 < NB ? d0 : NB - 1], 1u);
                if (i0 + 1 >= base && i0 + 1 < end) atomicAdd(&s_h[g][d1 < NB ? d1 : NB - 1], 1u);
            }
        }
    }
    __syncthreads();
    const uint32_t last = tile0 + MTB_HIST_GROUP <= tiles ? tile0 + MTB_HIST_GROUP - 1 : tiles - 1;
    if (tfirst[b0 + 1] > last) {                                             /* all tiles of the group in bucket b0: one run per bin */
        const uint32_t ntb = tfirst[b0 + 1] - tfirst[b0];
        uint32_t *row = hist + (uint64_t)512 * tfirst[b0] + (uint64_t)threadIdx.x * ntb + (tile0 - tfirst[b0]);
        for (uint32_t g = 0; g <= last - tile0; g++) row[g] = s_h[g][threadIdx.x];
    } else {
        uint32_t bb = b0;
        for (uint32_t g = 0; g <= last - tile0; g++) {
            const uint32_t t = tile0 + g;
            while (tfirst[bb + 1] <= t) bb++;
            const uint32_t ntb = tfirst[bb + 1] - tfirst[bb];
            hist[(uint64_t)512 * tfirst[bb] + (uint64_t)threadIdx.x * ntb + (t - tfirst[bb])] = s_h[g][threadIdx.x];
        }
    }
}

static inline uint64_t radix_hist_elems(uint64_t n, uint32_t bins = 256) { uint64_t tile = (uint64_t)bins * MTB_SORT_ITEMS; return (uint64_t)bins * ((n + tile - 1) / tile); }

/* Sort on bits [first_bit, 64).  a = input, b = scratch (same size); returns
 * the buffer that holds the result.  hist: radix_hist_elems(n) u32; ws:
 * scan_ws_elems(radix_hist_elems(n)) u32.  n < 2^32.                         */
static mtb_kmer *radix_sort_kmers(hipStream_t st, mtb_kmer *a, mtb_kmer *b, uint64_t n, int first_bit,
                                  uint32_t *hist, uint32_t *ws) {
    if (n == 0) return a;
    uint32_t tiles = (uint32_t)((n + MTB_SORT_TILE - 1) / MTB_SORT_TILE);
    mtb_kmer *src = a, *dst = b;
    for (int shift = first_bit; shift < 64; shift += 8) {
        hipLaunchKernelGGL((k_radix_hist<256, 0, 256>), dim3(tiles), dim3(256), 0, st, src, n, shift, hist, tiles);
        scan_launch<uint32_t, uint32_t, false>(st, hist, 256ull * tiles, false, hist, ws);
        hipLaunchKernelGGL((k_radix_scatter<256, 0, 256>), dim3((tiles + 7u) / 8u * 8u), dim3(256), 0, st, src, dst, n, shift, hist, tiles);
        mtb_kmer *tmp = src; src = dst; dst = tmp;
    }
    return src;
}

#endif
