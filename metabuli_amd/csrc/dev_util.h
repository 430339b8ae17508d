/* dev_util.h -- wave64 / workgroup primitives for gfx950 (CDNA4). */
#ifndef MTB_DEV_UTIL_H
#define MTB_DEV_UTIL_H
#include <hip/hip_runtime.h>
#include <stdint.h>

#define MTB_WAVE 64

__device__ __forceinline__ uint32_t lane_id() { return threadIdx.x & 63u; }
__device__ __forceinline__ uint64_t lanemask_lt() { return (1ull << lane_id()) - 1ull; }

/* inclusive scan across the 64 lanes of a wavefront */
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan(T v) {
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        T o = __shfl_up(v, d, 64);
        if ((int)lane_id() >= d) v += o;
    }
    return v;
}

/* the same with DPP row shifts / row broadcasts (gfx9: row_shr:n = 0x110+n, row_bcast:15 = 0x142, row_bcast:31 = 0x143):
 * six VALU adds with a DPP operand instead of six ds_bpermute round trips.  32-bit int / float only. */
__device__ __forceinline__ int32_t dpp_add_step(int32_t v, int32_t moved) { return v + moved; }
__device__ __forceinline__ float dpp_add_step(float v, int32_t moved) { return v + __int_as_float(moved); }
template <typename T>
__device__ __forceinline__ T wave_inclusive_scan_dpp(T v) {
    static_assert(sizeof(T) == 4, "32-bit values");
#define MTB_DPP_STEP(ctrl, rmask, bctl) do { int32_t b_; __builtin_memcpy(&b_, &v, 4); v = dpp_add_step(v, __builtin_amdgcn_update_dpp(0, b_, ctrl, rmask, 0xF, bctl)); } while (0)
    MTB_DPP_STEP(0x111, 0xF, true);
    MTB_DPP_STEP(0x112, 0xF, true);
    MTB_DPP_STEP(0x114, 0xF, true);
    MTB_DPP_STEP(0x118, 0xF, true);
    MTB_DPP_STEP(0x142, 0xA, false);
    MTB_DPP_STEP(0x143, 0xC, false);
#undef MTB_DPP_STEP
    return v;
}

/* Exclusive scan over a 256-thread workgroup; every thread gets its exclusive
 * prefix and *total (sum over the workgroup).  s_tmp: >= 5 elements of LDS.   */
template <typename T>
__device__ __forceinline__ T block256_exclusive_scan(T v, T *s_tmp, T *total) {
    T inc = wave_inclusive_scan(v);
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();                       /* protect s_tmp from a previous use */
    if (lane_id() == 63) s_tmp[w] = inc;
    __syncthreads();
    T pre = 0;
    T t0 = s_tmp[0], t1 = s_tmp[1], t2 = s_tmp[2], t3 = s_tmp[3];
    if (w > 0) pre += t0;
    if (w > 1) pre += t1;
    if (w > 2) pre += t2;
    *total = t0 + t1 + t2 + t3;
    return pre + inc - v;
}

/* same for a workgroup of NW wavefronts; s_tmp: >= NW elements of LDS */
template <typename T, int NW>
__device__ __forceinline__ T block_exclusive_scan(T v, T *s_tmp, T *total) {
    T inc = wave_inclusive_scan(v);
    uint32_t w = threadIdx.x >> 6;
    __syncthreads();
    if (lane_id() == 63) s_tmp[w] = inc;
    __syncthreads();
    T pre = 0, tot = 0;
#pragma unroll
    for (int k = 0; k < NW; k++) { T t = s_tmp[k]; if ((int)w > k) pre += t; tot += t; }
    *total = tot;
    return pre + inc - v;
}

/* experiment build (-DMTB_SYS_FENCES, profiles/scripts/contig_diag.py): system-scope release at the end of the kernels that write
 * the slot buffer and system-scope acquire at the start of those that read it -- on gfx950 these emit the L2 write-back /
 * invalidate that a kernel boundary normally implies for ordinary device memory */
#ifdef MTB_SYS_FENCES          /* 1: both, 2: release only, 3: acquire only */
#define MTB_END_RELEASE() do { if (MTB_SYS_FENCES != 3) __builtin_amdgcn_fence(__ATOMIC_RELEASE, ""); } while (0)
#define MTB_BEGIN_ACQUIRE() do { if (MTB_SYS_FENCES != 2) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, ""); } while (0)
#else
#define MTB_END_RELEASE() do {} while (0)
#define MTB_BEGIN_ACQUIRE() do {} while (0)
#endif

#endif
