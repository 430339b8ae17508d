/* alloc_probe.hip -- round-3 experiment: what decides the cost of 16-byte scattered stores into a 28 GB buffer (the join's slot
 * stores: one UTCL1 miss each, the UTCL2 busy 99 % of the kernel, join 42.5 - 57 ms depending on nothing but where the buffer
 * landed, profiles/r02_notes.md)?  With ~135 GB of other allocations in place (the resident index), a 28 GB buffer is obtained
 *   A  hipMalloc
 *   B  hipExtMallocWithFlags(hipDeviceMallocContiguous)
 *   C  hipMemCreate + hipMemAddressReserve(alignment) + hipMemMap, alignment 2 MiB / 1 GiB / 32 GiB
 * several times each, and probed with the join's access pattern (random non-temporal 16-byte stores over the whole buffer).
 * Build: hipcc --offload-arch=gfx950 -O3 -o alloc_probe alloc_probe.hip ; run: ./alloc_probe [GiB of ballast] [GiB of buffer] */
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
struct slot16 { unsigned long long a, b; };
__global__ __launch_bounds__(256) void k_probe(slot16 *buf, unsigned long long n_slots, unsigned per_thread, unsigned seed) {
    unsigned long long x = ((unsigned long long)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
    for (unsigned j = 0; j < per_thread; j++) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        slot16 *p = buf + (x % n_slots);
        __builtin_nontemporal_store(0ull, &p->a); __builtin_nontemporal_store(0ull, &p->b);
    }
}
static float probe(void *p, size_t bytes) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int rep = 0; rep < 4; rep++) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k_probe, dim3(16384), dim3(256), 0, 0, (slot16 *)p, (unsigned long long)(bytes / 16), 32u, 777u + rep);
        hipEventRecord(e1, 0); hipEventSynchronize(e1);
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        if (rep && ms < best) best = ms;
    }
    hipEventDestroy(e0); hipEventDestroy(e1);
    return best;          /* 134 M stores */
}
int main(int argc, char **argv) {
    const size_t ballast_gib = argc > 1 ? atol(argv[1]) : 135, buf_gib = argc > 2 ? atol(argv[2]) : 28;
    const size_t bytes = buf_gib << 30;
    CK(hipSetDevice(0));
    std::vector<void *> ballast;
    {   /* the index: one 128 GiB array, a 64 GiB one that is freed again (info[] before the seal), a 7 GiB directory */
        void *a = nullptr, *b = nullptr, *c = nullptr;
        if (ballast_gib >= 135) { CK(hipMalloc(&a, 128ull << 30)); CK(hipMalloc(&b, 64ull << 30)); CK(hipMalloc(&c, 7ull << 30)); CK(hipFree(b)); ballast.push_back(a); ballast.push_back(c); }
        else if (ballast_gib) { CK(hipMalloc(&a, ballast_gib << 30)); ballast.push_back(a); }
        /* the workspace buffers the library allocates before the slot buffer: two 22 GiB metamer buffers, two 3 GiB digit arrays */
        for (size_t g : {22, 22, 3, 3}) { void *w = nullptr; CK(hipMalloc(&w, g << 30)); ballast.push_back(w); }
    }
    size_t fr = 0, tot = 0; CK(hipMemGetInfo(&fr, &tot));
    printf("free %.1f GiB of %.1f GiB; buffer %zu GiB; probe = 134 M random 16-byte non-temporal stores\n", fr / 1073741824.0, tot / 1073741824.0, buf_gib);
    for (int round = 0; round < 3; round++) {
        { void *p = nullptr; CK(hipMalloc(&p, bytes)); printf("A hipMalloc                     %p  %.3f ms\n", p, probe(p, bytes)); CK(hipFree(p)); }
        { void *p = nullptr; hipError_t e = hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous);
          if (e == hipSuccess) { printf("B contiguous                    %p  %.3f ms\n", p, probe(p, bytes)); CK(hipFree(p)); } else { (void)hipGetLastError(); printf("B contiguous: %s\n", hipGetErrorString(e)); } }
        for (size_t align : {(size_t)2 << 20, (size_t)1 << 30, (size_t)32 << 30}) {
            hipMemAllocationProp prop = {}; prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
            size_t gran = 0; CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
            hipMemGenericAllocationHandle_t h; hipError_t e = hipMemCreate(&h, bytes, &prop, 0);
            if (e != hipSuccess) { (void)hipGetLastError(); printf("C hipMemCreate: %s\n", hipGetErrorString(e)); continue; }
            void *va = nullptr; e = hipMemAddressReserve(&va, bytes, align, nullptr, 0);
            if (e != hipSuccess) { (void)hipGetLastError(); printf("C reserve(align %zu MiB): %s\n", align >> 20, hipGetErrorString(e)); (void)hipMemRelease(h); continue; }
            CK(hipMemMap(va, bytes, 0, h, 0));
            hipMemAccessDesc ad = {}; ad.location = prop.location; ad.flags = hipMemAccessFlagsProtReadWrite;
            CK(hipMemSetAccess(va, bytes, &ad, 1));
            printf("C vmm align %5zu MiB (gran %zu KiB) %p  %.3f ms\n", align >> 20, gran >> 10, va, probe(va, bytes));
            CK(hipMemUnmap(va, bytes)); CK(hipMemAddressFree(va, bytes)); CK(hipMemRelease(h));
        }
        /* the library's way of making the next hipMalloc land elsewhere: park a few GiB first */
        void *pad = nullptr; CK(hipMalloc(&pad, (size_t)(round + 1) * (3ull << 30))); ballast.push_back(pad);
    }
    for (void *p : ballast) (void)hipFree(p);
    return 0;
}
