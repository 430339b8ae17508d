/* kernels_sort.h -- stable LSD radix sort of 16-byte query k-mer records on
 * the 64-bit metamer value (replaces SORT_PARALLEL(..., Kmer::compareQueryKmer),
 * src/commons/KmerExtractor.cpp:79; equal values keep extraction order, which
 * is ascending sequenceID, so the result is a valid (value, seqID) order).
 *
 * 8-bit digits; per pass: histogram kernel (tile counts, digit-major), device
 * scan, scatter kernel.  The scatter ranks a 2048-element tile stably with
 * wave64 ballot matching, reorders it through LDS so that every digit's
 * elements leave the CU as one contiguous run (full 128-byte lines instead of
 * scattered 16-byte writes), and adds the scanned tile offsets.
 * Algorithmic HBM bytes per pass: 16 (hist read) + 16 (read) + 16 (write).   */
#ifndef MTB_KERNELS_SORT_H
#define MTB_KERNELS_SORT_H
#include "dev_util.h"
#include "kernels_scan.h"
#include "mtb_core.h"

#define MTB_SORT_TILE 2048
#define MTB_SORT_ITEMS 8

__global__ __launch_bounds__(256) void k_radix_hist(const mtb_kmer *__restrict__ in, uint64_t n, int shift,
                                                     uint32_t *__restrict__ hist, uint32_t num_tiles) {
    __shared__ uint32_t s_h[256];
    s_h[threadIdx.x] = 0;
    __syncthreads();
    uint64_t base = (uint64_t)blockIdx.x * MTB_SORT_TILE;
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)r * 256 + threadIdx.x;
        if (i < n) atomicAdd(&s_h[(uint32_t)(in[i].value >> shift) & 255u], 1u);
    }
    __syncthreads();
    hist[(uint64_t)threadIdx.x * num_tiles + blockIdx.x] = s_h[threadIdx.x];
}

__global__ __launch_bounds__(256) void k_radix_scatter(const mtb_kmer *__restrict__ in, mtb_kmer *__restrict__ out,
                                                        uint64_t n, int shift, const uint32_t *__restrict__ tile_off,
                                                        uint32_t num_tiles) {
    __shared__ uint32_t s_cnt[4][256];
    __shared__ uint32_t s_run[256];
    __shared__ uint32_t s_start[256];
    __shared__ uint32_t s_tmp[8];
    __shared__ mtb_kmer s_buf[MTB_SORT_TILE];
    const uint32_t t = threadIdx.x, w = t >> 6;
    const uint64_t base = (uint64_t)blockIdx.x * MTB_SORT_TILE;
    mtb_kmer e[MTB_SORT_ITEMS];
    uint32_t lrank[MTB_SORT_ITEMS];
    s_run[t] = 0;
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)r * 256 + t;
        bool valid = i < n;
        if (valid) e[r] = in[i]; else { e[r].value = 0; e[r].qinfo = 0; }
        uint32_t d = (uint32_t)(e[r].value >> shift) & 255u;
        s_cnt[0][t] = 0; s_cnt[1][t] = 0; s_cnt[2][t] = 0; s_cnt[3][t] = 0;
        __syncthreads();
        /* lanes of this wave holding the same digit */
        uint64_t peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; b++) {
            bool bit = (d >> b) & 1u;
            uint64_t vote = __ballot(bit);
            peers &= bit ? vote : ~vote;
        }
        uint32_t rank_in_wave = (uint32_t)__popcll(peers & lanemask_lt());
        if (valid && rank_in_wave == 0) s_cnt[w][d] = (uint32_t)__popcll(peers);
        __syncthreads();
        uint32_t pre = s_run[d];
        if (w > 0) pre += s_cnt[0][d];
        if (w > 1) pre += s_cnt[1][d];
        if (w > 2) pre += s_cnt[2][d];
        lrank[r] = pre + rank_in_wave;
        __syncthreads();
        s_run[t] += s_cnt[0][t] + s_cnt[1][t] + s_cnt[2][t] + s_cnt[3][t];
    }
    __syncthreads();
    uint32_t tot;
    uint32_t ex = block256_exclusive_scan<uint32_t>(s_run[t], s_tmp, &tot);
    s_start[t] = ex;
    __syncthreads();
#pragma unroll
    for (int r = 0; r < MTB_SORT_ITEMS; r++) {
        uint64_t i = base + (uint64_t)r * 256 + t;
        if (i < n) {
            uint32_t d = (uint32_t)(e[r].value >> shift) & 255u;
            s_buf[s_start[d] + lrank[r]] = e[r];
        }
    }
    __syncthreads();
    uint32_t cnt = (uint32_t)((n - base) < MTB_SORT_TILE ? (n - base) : MTB_SORT_TILE);
    for (uint32_t i = t; i < cnt; i += 256) {
        mtb_kmer x = s_buf[i];
        uint32_t d = (uint32_t)(x.value >> shift) & 255u;
        out[(uint64_t)tile_off[(uint64_t)d * num_tiles + blockIdx.x] + (i - s_start[d])] = x;
    }
}

static inline uint64_t radix_hist_elems(uint64_t n) { return 256ull * ((n + MTB_SORT_TILE - 1) / MTB_SORT_TILE); }

/* Sort on bits [first_bit, 64).  a = input, b = scratch (same size); returns
 * the buffer that holds the result.  hist: radix_hist_elems(n) u32; ws:
 * scan_ws_elems(radix_hist_elems(n)) u32.  n < 2^32.                         */
static mtb_kmer *radix_sort_kmers(hipStream_t st, mtb_kmer *a, mtb_kmer *b, uint64_t n, int first_bit,
                                  uint32_t *hist, uint32_t *ws) {
    if (n == 0) return a;
    uint32_t tiles = (uint32_t)((n + MTB_SORT_TILE - 1) / MTB_SORT_TILE);
    mtb_kmer *src = a, *dst = b;
    for (int shift = first_bit; shift < 64; shift += 8) {
        hipLaunchKernelGGL(k_radix_hist, dim3(tiles), dim3(256), 0, st, src, n, shift, hist, tiles);
        scan_launch<uint32_t, uint32_t, false>(st, hist, 256ull * tiles, false, hist, ws);
        hipLaunchKernelGGL(k_radix_scatter, dim3(tiles), dim3(256), 0, st, src, dst, n, shift, hist, tiles);
        mtb_kmer *tmp = src; src = dst; dst = tmp;
    }
    return src;
}

#endif
