/* gather_probe.hip -- round-5 go / no-go experiment: does the directory join need its queries SORTED?
 * The step spends 31 ms sorting 1.28 G query metamers and 13 ms writing them, only so that k_join_dir meets the directory and the
 * target array in ascending order.  A read-major fused kernel (one wavefront per read: extract -> lookup -> slots in LDS -> score)
 * would drop the sort, the 20 GB of metamer records and the 1.26 G scattered slot stores -- if the lookups of UNSORTED queries are
 * cheap enough.  This probe times nothing but those lookups, with the sizes of the timed index:
 *   T = 16 G target words (128 GB), directory of 21^7 buckets (7.2 GB, uniform: ~8.9 targets per bucket), N = 1.28 G queries;
 *   per query: two adjacent directory words -> [lo, hi) -> lower-bound bisection of a pseudo-random key inside the bucket (what
 *   k_join_dir does: 3 - 4 dependent 8-byte loads) -> one more load of the word next to the landing place; NO stores.
 * Query orders: sorted by bucket (what the radix sort buys), random (read-major: a wave's 64 lanes hold the metamers of one read),
 * and "sorted inside blocks of B queries" for B = 2^16 .. 2^26 (a partial sort: one MSD pass, or reads binned by nothing at all).
 * Build: hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip ; run: ./gather_probe [G targets] [M queries] */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef unsigned long long u64;
__device__ __forceinline__ u64 mix(u64 x) { x ^= x >> 30; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 27; x *= 0x94D049BB133111EBull; return x ^ (x >> 31); }

__global__ __launch_bounds__(256) void k_fill_values(u64 *v, u64 T) {
    for (u64 i = (u64)blockIdx.x * 256 + threadIdx.x; i < T; i += (u64)gridDim.x * 256) v[i] = i << 8;      /* ascending: a bisection converges as on real data */
}
__global__ __launch_bounds__(256) void k_fill_dir(unsigned *dir, u64 *base, u64 nb, u64 T) {
    for (u64 b = (u64)blockIdx.x * 256 + threadIdx.x; b <= nb; b += (u64)gridDim.x * 256) {
        const u64 start = (u64)((__uint128_t)b * T / nb), gb = (u64)((__uint128_t)(b & ~0xFFFFull) * T / nb);
        dir[b] = (unsigned)(start - gb);
        if ((b & 0xFFFF) == 0) base[b >> 16] = gb;
    }
}
/* query j -> its bucket.  mode 0: ascending in j (sorted); 1: random; 2: ascending inside blocks of `blk` queries, the blocks' ranges random
 * (every block sweeps the WHOLE directory, as an unsorted tile of reads does) */
__device__ __forceinline__ u64 bucket_of(u64 j, u64 n, u64 nb, int mode, u64 blk) {
    if (mode == 0) return (u64)((__uint128_t)j * nb / n);
    if (mode == 1) return mix(j * 0x9E3779B97F4A7C15ull + 12345) % nb;
    const u64 in = j % blk;
    return ((u64)((__uint128_t)in * nb / blk) + mix(j / blk) % (nb / blk + 1)) % nb;
}
template <int Q>
__global__ __launch_bounds__(256) void k_gather(const u64 *__restrict__ v, const unsigned *__restrict__ dir, const u64 *__restrict__ base, u64 nb, u64 n, int mode, u64 blk,
                                                u64 *__restrict__ sink) {
    u64 acc = 0;
    u64 lo[Q], hi[Q], key[Q];
#pragma unroll
    for (int u = 0; u < Q; u++) {
        const u64 j = ((u64)blockIdx.x * Q + u) * 256 + threadIdx.x;
        lo[u] = hi[u] = 0; key[u] = 0;
        if (j < n) {
            const u64 b = bucket_of(j, n, nb, mode, blk);
            lo[u] = base[b >> 16] + dir[b]; hi[u] = base[(b + 1) >> 16] + dir[b + 1];
            key[u] = ((lo[u] + mix(j) % (hi[u] - lo[u] + 1)) << 8) | 1;
        }
    }
    bool more = true;
    while (more) {
        more = false;
#pragma unroll
        for (int u = 0; u < Q; u++)
            if (lo[u] < hi[u]) { const u64 mid = lo[u] + ((hi[u] - lo[u]) >> 1); if (v[mid] < key[u]) lo[u] = mid + 1; else hi[u] = mid; more |= lo[u] < hi[u]; }
    }
#pragma unroll
    for (int u = 0; u < Q; u++) acc += v[lo[u]];
    if (acc == 0x123456789ull) sink[0] = acc;          /* never true: keeps the loads */
}
int main(int argc, char **argv) {
    const u64 T = (u64)((argc > 1 ? atof(argv[1]) : 16.0) * 1e9), N = (u64)((argc > 2 ? atof(argv[2]) : 1280.0) * 1e6);
    const u64 nb = 1801088541ull;                    /* 21^7 */
    CK(hipSetDevice(0));
    u64 *v, *base, *sink; unsigned *dir;
    CK(hipMalloc((void **)&v, (T + 2) * 8)); CK(hipMalloc((void **)&dir, (nb + 2) * 4)); CK(hipMalloc((void **)&base, ((nb >> 16) + 3) * 8)); CK(hipMalloc((void **)&sink, 64));
    void *ballast = nullptr; CK(hipMalloc(&ballast, 60ull << 30));         /* what the batch workspace occupies next to the index */
    hipLaunchKernelGGL(k_fill_values, dim3(1 << 16), dim3(256), 0, 0, v, T + 2);
    hipLaunchKernelGGL(k_fill_dir, dim3(1 << 16), dim3(256), 0, 0, dir, base, nb, T);
    CK(hipDeviceSynchronize());
    printf("%.2f G target words (%.0f GB), directory 21^7 (%.1f GB), %.0f M queries, 2 queries per thread; ms (best of 2 after a warm-up)\n", T / 1e9, T * 8 / 1e9, nb * 4 / 1e9, N / 1e6);
    struct { int mode; u64 blk; const char *name; } runs[] = {
        {0, 0, "sorted by bucket"}, {1, 0, "random (read-major)"}, {2, 1ull << 16, "sorted inside blocks of 2^16"}, {2, 1ull << 20, "sorted inside blocks of 2^20"},
        {2, 1ull << 23, "sorted inside blocks of 2^23"}, {2, 1ull << 26, "sorted inside blocks of 2^26"} };
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (auto &r : runs) {
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {
            CK(hipEventRecord(e0, 0));
            hipLaunchKernelGGL((k_gather<2>), dim3((unsigned)((N + 511) / 512)), dim3(256), 0, 0, v, dir, base, nb, N, r.mode, r.blk, sink);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        printf("%-32s %9.2f ms  = %6.1f ns per 1000 queries\n", r.name, best, best * 1e6 / (N / 1000.0));
        fflush(stdout);
    }
    (void)hipFree(v); (void)hipFree(dir); (void)hipFree(base); (void)hipFree(sink); (void)hipFree(ballast);
    return 0;
}
