/* store_window_probe.hip -- round-3 experiment: the directory join emits ~1.1 G matches per 10 M-read batch, each a 16-byte
 * non-temporal store at a random place of the 28 GB slot buffer (the queries arrive in metamer order, their reads are random).
 * alloc_probe.hip: 134 M such stores take 5.07 ms -> 1.1 G take ~42 ms = the whole join kernel.  What would the same stores cost
 * if they arrived grouped by destination window (matches first binned by read range, sequentially, then each bin scattered into
 * its own window of the slot buffer), for windows from L2 size to the whole buffer?
 *   store j (read sequentially from a 16-byte source list) goes to   window (j / per_window) , random 16-byte slot inside it
 * Build: hipcc --offload-arch=gfx950 -O3 -o store_window_probe store_window_probe.hip ; run: ./store_window_probe [GiB buffer] [M stores] */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct slot16 { unsigned long long a, b; };
template <bool NT, bool SRC>
__global__ __launch_bounds__(256) void k_probe(slot16 *buf, const slot16 *src, unsigned long long n_stores, unsigned long long per_window,
                                               unsigned long long window_slots, unsigned seed) {
    const unsigned long long j = (unsigned long long)blockIdx.x * 256 + threadIdx.x;
    if (j >= n_stores) return;
    unsigned long long x = j * 0x9E3779B97F4A7C15ull + seed;
    x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
    slot16 v; v.a = x; v.b = j;
    if (SRC) v = src[j];
    slot16 *p = buf + (j / per_window) * window_slots + (x % window_slots);
    if (NT) { __builtin_nontemporal_store(v.a, &p->a); __builtin_nontemporal_store(v.b, &p->b); }
    else *p = v;
}
template <bool NT, bool SRC>
static float run(slot16 *buf, const slot16 *src, size_t n_stores, size_t n_slots, size_t window_bytes) {
    size_t window_slots = window_bytes / 16; if (window_slots > n_slots) window_slots = n_slots;
    const size_t n_windows = n_slots / window_slots;
    const size_t per_window = (n_stores + n_windows - 1) / n_windows;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 3; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((k_probe<NT, SRC>), dim3((unsigned)((n_stores + 255) / 256)), dim3(256), 0, 0, buf, src, (unsigned long long)n_stores,
                           (unsigned long long)per_window, (unsigned long long)window_slots, 777u + rep);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;
}
int main(int argc, char **argv) {
    const size_t buf_gib = argc > 1 ? atol(argv[1]) : 28, n_stores = (size_t)(argc > 2 ? atol(argv[2]) : 1100) * 1000000;
    CK(hipSetDevice(0));
    void *ballast = nullptr; CK(hipMalloc(&ballast, 135ull << 30));         /* the index next to it */
    slot16 *buf = nullptr, *src = nullptr;
    CK(hipMalloc((void **)&buf, buf_gib << 30)); CK(hipMalloc((void **)&src, n_stores * 16));
    CK(hipMemset(src, 1, n_stores * 16)); CK(hipMemset(buf, 0, buf_gib << 30));
    const size_t n_slots = (buf_gib << 30) / 16;
    printf("%zu M 16-byte stores into a %zu GiB buffer, grouped by destination window; ms (best of 2 after a warm-up)\n", n_stores / 1000000, buf_gib);
    printf("%-12s %10s %10s %10s %10s\n", "window", "NT", "NT+src", "plain", "plain+src");
    for (size_t wb : {(size_t)1 << 20, (size_t)4 << 20, (size_t)16 << 20, (size_t)32 << 20, (size_t)64 << 20, (size_t)128 << 20, (size_t)256 << 20,
                      (size_t)512 << 20, (size_t)2 << 30, (size_t)8 << 30, buf_gib << 30}) {
        printf("%8zu MiB %10.2f %10.2f %10.2f %10.2f\n", wb >> 20, run<true, false>(buf, src, n_stores, n_slots, wb), run<true, true>(buf, src, n_stores, n_slots, wb),
               run<false, false>(buf, src, n_stores, n_slots, wb), run<false, true>(buf, src, n_stores, n_slots, wb));
        fflush(stdout);
    }
    (void)hipFree(buf); (void)hipFree(src); (void)hipFree(ballast);
    return 0;
}
