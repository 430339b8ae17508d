/* kernels_scan.h -- device-wide prefix sums (reduce / recurse / apply).
 * Tile = 256 threads x 8 thread-contiguous items (32-bit arrays: two 16-byte
 * accesses per thread where the chunk lies inside the array); HBM traffic =
 * 2 reads + 1 write per element.  Used for k-mer offsets, radix histograms, per-read
 * match segments and the diffIdx decode.                                    */
#ifndef MTB_KERNELS_SCAN_H
#define MTB_KERNELS_SCAN_H
#include "dev_util.h"

#define MTB_SCAN_TILE 2048

template <typename TIn, typename TOut>
__global__ __launch_bounds__(256) void k_scan_reduce(const TIn *__restrict__ in, uint64_t n_in, TOut *__restrict__ sums) {
    __shared__ TOut s_tmp[8];
    uint64_t base = (uint64_t)blockIdx.x * MTB_SCAN_TILE + (uint64_t)threadIdx.x * 8;
    TOut v = 0;
    if (sizeof(TIn) == 4 && base + 8 <= n_in && (((uintptr_t)(in + base)) & 15u) == 0) {
        /* whole chunk inside the array: two 16-byte loads instead of eight guarded 4-byte ones (the radix tables are 640 MB per pass) */
        const uint4 a = ((const uint4 *)(in + base))[0], b = ((const uint4 *)(in + base))[1];
        v = (TOut)a.x + (TOut)a.y + (TOut)a.z + (TOut)a.w + (TOut)b.x + (TOut)b.y + (TOut)b.z + (TOut)b.w;
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { uint64_t i = base + k; if (i < n_in) v += (TOut)in[i]; }
    }
    TOut tot;
    block256_exclusive_scan<TOut>(v, s_tmp, &tot);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

template <typename TIn, typename TOut, bool INCL>
__global__ __launch_bounds__(256) void k_scan_final(const TIn *in, uint64_t n_in, uint64_t n_out, TOut *out,
                                                     const TOut *__restrict__ prefix) {
    __shared__ TOut s_tmp[8];
    uint64_t base = (uint64_t)blockIdx.x * MTB_SCAN_TILE + (uint64_t)threadIdx.x * 8;
    TOut x[8];
    TOut v = 0;
    const bool whole_in = sizeof(TIn) == 4 && base + 8 <= n_in && (((uintptr_t)(in + base)) & 15u) == 0;
    if (whole_in) {
        const uint4 a = ((const uint4 *)(in + base))[0], b = ((const uint4 *)(in + base))[1];
        x[0] = (TOut)a.x; x[1] = (TOut)a.y; x[2] = (TOut)a.z; x[3] = (TOut)a.w; x[4] = (TOut)b.x; x[5] = (TOut)b.y; x[6] = (TOut)b.z; x[7] = (TOut)b.w;
#pragma unroll
        for (int k = 0; k < 8; k++) v += x[k];
    } else {
#pragma unroll
        for (int k = 0; k < 8; k++) { uint64_t i = base + k; x[k] = (i < n_in) ? (TOut)in[i] : (TOut)0; v += x[k]; }
    }
    TOut tot;
    TOut run = block256_exclusive_scan<TOut>(v, s_tmp, &tot) + (prefix ? prefix[blockIdx.x] : (TOut)0);
    if (sizeof(TOut) == 4 && base + 8 <= n_out && (((uintptr_t)(out + base)) & 15u) == 0) {
        uint32_t y[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { y[k] = (uint32_t)(INCL ? run + x[k] : run); run += x[k]; }
        ((uint4 *)(out + base))[0] = make_uint4(y[0], y[1], y[2], y[3]); ((uint4 *)(out + base))[1] = make_uint4(y[4], y[5], y[6], y[7]);
        return;
    }
#pragma unroll
    for (int k = 0; k < 8; k++) {
        uint64_t i = base + k;
        if (i < n_out) out[i] = INCL ? run + x[k] : run;
        run += x[k];
    }
}

static inline uint64_t scan_ws_elems(uint64_t n) {
    uint64_t tot = 0;
    while (n > MTB_SCAN_TILE) { n = (n + MTB_SCAN_TILE - 1) / MTB_SCAN_TILE; tot += n; }
    return tot + 8;
}

/* out[i] = sum(in[0..i)) (or inclusive); with want_total, out has n_in+1
 * entries and out[n_in] = grand total.  ws: scan_ws_elems(n_in+1) TOut's.   */
template <typename TIn, typename TOut, bool INCL>
static void scan_launch(hipStream_t st, const TIn *in, uint64_t n_in, bool want_total, TOut *out, TOut *ws) {
    uint64_t n_out = n_in + (want_total ? 1 : 0);
    if (n_out == 0) return;
    uint64_t tiles = (n_out + MTB_SCAN_TILE - 1) / MTB_SCAN_TILE;
    if (tiles == 1) {
        hipLaunchKernelGGL((k_scan_final<TIn, TOut, INCL>), dim3(1), dim3(256), 0, st, in, n_in, n_out, out, (const TOut *)nullptr);
        return;
    }
    TOut *sums = ws;
    hipLaunchKernelGGL((k_scan_reduce<TIn, TOut>), dim3((uint32_t)tiles), dim3(256), 0, st, in, n_in, sums);
    scan_launch<TOut, TOut, false>(st, sums, tiles, false, sums, ws + tiles);
    hipLaunchKernelGGL((k_scan_final<TIn, TOut, INCL>), dim3((uint32_t)tiles), dim3(256), 0, st, in, n_in, n_out, out, (const TOut *)sums);
}

#endif
