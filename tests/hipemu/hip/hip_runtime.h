/*
 * TEST INFRASTRUCTURE ONLY -- a host-side stand-in for <hip/hip_runtime.h> that lets the library's sources (metabuli_amd/csrc/mtb_api.hip and
 * its kernel headers, UNCHANGED) be compiled with g++ and run on a machine without a GPU:
 *
 *     g++ -x c++ -std=c++17 -I tests/hipemu -include tests/hipemu/hipemu_dyn_shared.h ... metabuli_amd/csrc/mtb_api.hip -o libmtb_hipemu.so
 *
 * Why: kernel LOGIC can be checked (against the oracle, under AddressSanitizer) before GPU minutes are spent on it, and the parity tests of
 * tests/test_gpu_parity.py get a second executor.  What it is NOT: a product path (nothing under metabuli_amd/ refers to it; libmtb.so is
 * built by hipcc for gfx950 and fails without a device), a performance model, or a memory-model checker (lanes are cooperative fibers:
 * no data race of the real machine shows here).
 *
 * Execution model: a workgroup = blockDim.x fibers on one OS thread (own stacks, a 30-instruction context switch); workgroups of a launch are
 * taken in blockIdx order by a few OS threads (so a look-back on lower-numbered workgroups makes progress).  A fiber runs until it reaches
 * __syncthreads() or a wavefront operation (ballot / shuffle / DPP / readlane / wave barrier); when every live lane of a wavefront is blocked,
 * the lanes blocked at the same call site form the active set of that operation (= the exec mask of the real machine for code whose
 * wavefront operations sit in wave-uniform control flow; if several call sites wait at once the textually first one goes first and the event
 * is counted -- HIPEMU_VERBOSE=1 reports the sites).  Streams are synchronous: every runtime call completes before it returns.
 */
#ifndef HIPEMU_HIP_RUNTIME_H
#define HIPEMU_HIP_RUNTIME_H
#define __HIPCC__ 1
#define __HIP_DEVICE_COMPILE__ 1
#define HIPEMU 1

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include <sys/mman.h>
#include <signal.h>
#include <unistd.h>

/* ---- qualifiers ------------------------------------------------------------------------------------------------------------------ */
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ thread_local            /* block scope: implies static; `extern __shared__ T x[]` finds the definition in hipemu_dyn_shared.h */
#define __constant__
#define HIP_SYMBOL(x) x
#define HIP_DYNAMIC_SHARED(type, var) extern thread_local type var[];

struct dim3 { uint32_t x, y, z; constexpr dim3(uint32_t x_ = 1, uint32_t y_ = 1, uint32_t z_ = 1) : x(x_), y(y_), z(z_) {} };
struct uint4 { uint32_t x, y, z, w; };
static inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { uint4 v; v.x = x; v.y = y; v.z = z; v.w = w; return v; }

/* ---- runtime API (synchronous; "device" memory is host memory) ---------------------------------------------------------------------- */
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorOutOfMemory = 2, hipErrorInvalidValue = 1 };
typedef struct hipemu_stream *hipStream_t;
struct hipemu_event { std::chrono::steady_clock::time_point t; };
typedef hipemu_event *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
enum { hipHostMallocDefault = 0, hipStreamNonBlocking = 1, hipEventDisableTiming = 2, hipDeviceMallocContiguous = 0x4 };
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };

namespace hipemu {
inline std::atomic<long long> g_allocated{0};
/* the "device" reports the real one's 288 GB by default: the library sizes some pools by what is free (two streams of very long reads ask
 * for two 48 GB slab pools), and pages nobody touches cost no host memory -- buffers of 256 MB and more are mapped MAP_NORESERVE */
inline long long mem_total() { const char *v = getenv("HIPEMU_MEM_MB"); return (v && *v ? atoll(v) : 294912ll) << 20; }
inline int env_int(const char *k, int d) { const char *v = getenv(k); return v && *v ? atoi(v) : d; }
struct AllocHdr { size_t bytes; size_t mapped; };
inline void *dev_alloc(size_t bytes) {
    if ((long long)bytes + g_allocated.load() > mem_total()) return nullptr;
    void *p = nullptr; size_t mapped = 0;
    if (bytes >= (256u << 20)) {
        mapped = (bytes + 256 + 4095) & ~(size_t)4095;
        p = mmap(nullptr, mapped, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) return nullptr;
    } else if (posix_memalign(&p, 256, bytes + 256) != 0) return nullptr;
    /* header in the first 256 bytes: the user pointer stays 256-byte aligned like hipMalloc's */
    ((AllocHdr *)p)->bytes = bytes; ((AllocHdr *)p)->mapped = mapped;
    g_allocated += (long long)bytes;
    if (env_int("HIPEMU_POISON", 1) && !mapped) memset((char *)p + 256, 0xA5, bytes);      /* fresh device memory is not zero (the big mapped buffers stay untouched zero pages) */
    return (char *)p + 256;
}
inline void dev_free(void *u) {
    if (!u) return;
    char *p = (char *)u - 256; const AllocHdr h = *(AllocHdr *)p;
    g_allocated -= (long long)h.bytes;
    if (h.mapped) munmap(p, h.mapped); else free(p);
}
}  // namespace hipemu

static inline hipError_t hipGetLastError() { return hipSuccess; }
enum { hipDeviceAttributeMultiprocessorCount = 63 };
static inline hipError_t hipDeviceGetAttribute(int *v, int, int) { *v = 3; return hipSuccess; }      /* (a small "chip": persistent kernels loop over several tiles per workgroup) */
static inline const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess (emulated)" : e == hipErrorOutOfMemory ? "out of memory (emulated device)" : "error (emulated device)"; }
static inline hipError_t hipGetDeviceCount(int *n) { *n = hipemu::env_int("HIPEMU_DEVICES", 2); return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
static inline hipError_t hipDeviceCanAccessPeer(int *can, int, int) { *can = 1; return hipSuccess; }
static inline hipError_t hipDeviceEnablePeerAccess(int, unsigned) { return hipSuccess; }
template <class T> static inline hipError_t hipMalloc(T **p, size_t bytes) { *p = (T *)hipemu::dev_alloc(bytes ? bytes : 1); return *p ? hipSuccess : hipErrorOutOfMemory; }
template <class T> static inline hipError_t hipExtMallocWithFlags(T **p, size_t bytes, unsigned) { return hipMalloc(p, bytes); }
static inline hipError_t hipFree(void *p) { hipemu::dev_free(p); return hipSuccess; }
template <class T> static inline hipError_t hipHostMalloc(T **p, size_t bytes, unsigned = 0) { void *q = nullptr; if (posix_memalign(&q, 4096, bytes ? bytes : 1) != 0) return hipErrorOutOfMemory; *p = (T *)q; return hipSuccess; }
static inline hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
static inline hipError_t hipMemGetInfo(size_t *fr, size_t *tot) { *tot = (size_t)hipemu::mem_total(); long long f = hipemu::mem_total() - hipemu::g_allocated.load(); *fr = f > 0 ? (size_t)f : 0; return hipSuccess; }
static inline hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
/* inter-process handles: the stand-in has one address space per process -- a handle is the pointer itself (an import inside the exporting
 * process never opens it; across processes the real runtime is needed and the stand-in reports an error) */
typedef void *hipDeviceptr_t;
struct hipIpcMemHandle_t { char reserved[64]; };
enum { hipIpcMemLazyEnablePeerAccess = 1 };
static inline hipError_t hipMemGetAddressRange(hipDeviceptr_t *base, size_t *size, hipDeviceptr_t p) { *base = p; *size = 0; return hipSuccess; }
static inline hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t *h, void *p) { memset(h, 0, sizeof(*h)); memcpy(h->reserved, &p, sizeof(p)); return hipSuccess; }
static inline hipError_t hipIpcOpenMemHandle(void **, hipIpcMemHandle_t, unsigned) { return hipErrorInvalidValue; }
static inline hipError_t hipIpcCloseMemHandle(void *) { return hipSuccess; }
static inline hipError_t hipMemcpyPeerAsync(void *d, int, const void *s, int, size_t n, hipStream_t = nullptr) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t = nullptr) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void *d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
template <class S> static inline hipError_t hipMemcpyToSymbol(S &sym, const void *s, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyHostToDevice) { memcpy((char *)&sym + off, s, n); return hipSuccess; }
template <class S> static inline hipError_t hipMemcpyFromSymbol(void *d, const S &sym, size_t n, size_t off = 0, hipMemcpyKind = hipMemcpyDeviceToHost) { memcpy(d, (const char *)&sym + off, n); return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned) { *s = (hipStream_t)malloc(8); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned = 0) { return hipSuccess; }
static inline hipError_t hipEventCreate(hipEvent_t *e) { *e = new hipemu_event(); (*e)->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned) { return hipEventCreate(e); }
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t = nullptr) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b) { *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count(); return hipSuccess; }
template <class F> static inline hipError_t hipFuncSetAttribute(F, hipFuncAttribute, int) { return hipSuccess; }

/* the few runtime calls tests make from Python (tests/test_gpu_parity.py: _Hip), exported by the emulated library under their own names */
extern "C" {
inline __attribute__((used)) int hipemu_c_malloc(void **p, size_t n) { return hipMalloc(p, n); }
inline __attribute__((used)) int hipemu_c_memcpy(void *d, const void *s, size_t n, int) { if (n) memmove(d, s, n); return 0; }
inline __attribute__((used)) int hipemu_c_free(void *p) { return hipFree(p); }
inline __attribute__((used)) int hipemu_c_device_synchronize() { return 0; }
}

/* ---- the fiber machine ------------------------------------------------------------------------------------------------------------ */
extern "C" void hipemu_switch(void **save_sp, void *load_sp);
asm(".text\n.globl hipemu_switch\n.type hipemu_switch,@function\nhipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  movq %rsp, (%rdi)\n  movq %rsi, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n  ret\n"
    ".size hipemu_switch, .-hipemu_switch\n");

namespace hipemu {
enum Op : int { OP_NONE = 0, OP_BALLOT, OP_SHFL, OP_SHFL_UP, OP_SHFL_DOWN, OP_SHFL_XOR, OP_READLANE, OP_DPP, OP_WAVE_BARRIER };
enum State : int { RUNNABLE = 0, WAIT_WAVE, WAIT_BLOCK, DONE };
struct Idx3 { uint32_t x, y, z; };
struct Lane {
    void *sp = nullptr; char *stack = nullptr;
    State state = DONE; Op op = OP_NONE; uint32_t site = 0;
    uint64_t a = 0, r = 0; int32_t b = 0, c = 0; uint32_t d = 0;       /* payload: value, source / delta / mask, width, DPP control words */
    Idx3 tid{0, 0, 0};
};
struct Block {
    dim3 grid, block; Idx3 bid{0, 0, 0};
    std::vector<Lane> lanes; void *sched_sp = nullptr; const std::function<void()> *body = nullptr;
    int bar_or = 0, bar_and = 1, bar_count = 0;      /* __syncthreads_or / _and / _count of the barrier just released */
};
inline thread_local Block *t_blk = nullptr;
inline thread_local Lane *t_lane = nullptr;
inline const char *g_kernel = "";
inline std::atomic<unsigned long> g_divergent{0}, g_inactive_reads{0};
struct ExitReport { ~ExitReport() { if (env_int("HIPEMU_VERBOSE", 0)) fprintf(stderr, "hipemu: %lu wavefront operations fired with several call sites waiting, %lu shuffle / readlane / DPP reads of a lane outside the active set (0 returned)\n",
                                                                              g_divergent.load(), g_inactive_reads.load()); } };
inline ExitReport g_exit_report;
constexpr size_t STACK = 192 << 10;

#if defined(__SANITIZE_ADDRESS__)
extern "C" void __sanitizer_start_switch_fiber(void **fake_stack_save, const void *bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void *fake_stack_save, const void **bottom_old, size_t *size_old);
inline thread_local const void *t_sched_bottom = nullptr; inline thread_local size_t t_sched_size = 0;
#define HIPEMU_ASAN_ENTER_FIBER(fake, lane) __sanitizer_start_switch_fiber(&(fake), (lane).stack, hipemu::STACK)
#define HIPEMU_ASAN_BACK_IN_SCHEDULER(fake) __sanitizer_finish_switch_fiber((fake), nullptr, nullptr)
#define HIPEMU_ASAN_FIBER_STARTED() __sanitizer_finish_switch_fiber(nullptr, &hipemu::t_sched_bottom, &hipemu::t_sched_size)
#define HIPEMU_ASAN_LEAVE_FIBER(fake_ptr) __sanitizer_start_switch_fiber((fake_ptr), hipemu::t_sched_bottom, hipemu::t_sched_size)
#define HIPEMU_ASAN_BACK_IN_FIBER(fake) __sanitizer_finish_switch_fiber((fake), &hipemu::t_sched_bottom, &hipemu::t_sched_size)
#else
#define HIPEMU_ASAN_ENTER_FIBER(fake, lane) ((void)(fake))
#define HIPEMU_ASAN_BACK_IN_SCHEDULER(fake) ((void)0)
#define HIPEMU_ASAN_FIBER_STARTED() ((void)0)
#define HIPEMU_ASAN_LEAVE_FIBER(fake_ptr) ((void)0)
#define HIPEMU_ASAN_BACK_IN_FIBER(fake) ((void)(fake))
#endif
inline void fiber_main() {
    HIPEMU_ASAN_FIBER_STARTED();
    (*t_blk->body)();
    t_lane->state = DONE;
    HIPEMU_ASAN_LEAVE_FIBER(nullptr);          /* (no fake stack to keep: this fiber is over) */
    hipemu_switch(&t_lane->sp, t_blk->sched_sp);
    abort();            /* a finished fiber is never resumed */
}
inline void yield_to_scheduler() {
    Lane *me = t_lane; void *fake = nullptr;
    HIPEMU_ASAN_LEAVE_FIBER(&fake);
    hipemu_switch(&me->sp, t_blk->sched_sp);
    HIPEMU_ASAN_BACK_IN_FIBER(fake);
}

inline uint64_t wave_op(Op op, uint32_t site, uint64_t a, int32_t b = 0, int32_t c = 0, uint32_t d = 0) {
    Lane *me = t_lane;
    me->op = op; me->site = site; me->a = a; me->b = b; me->c = c; me->d = d; me->state = WAIT_WAVE;
    yield_to_scheduler();
    return me->r;
}
inline void block_barrier(int pred = 0) { t_lane->a = pred ? 1 : 0; t_lane->state = WAIT_BLOCK; yield_to_scheduler(); }

inline void report_inactive(uint32_t site, int lane, int src) {
    if (env_int("HIPEMU_VERBOSE", 0) >= 3) fprintf(stderr, "hipemu: call site %u in %s: lane %d reads lane %d, which is not in the active set\n", site, g_kernel, lane, src);
}
/* the operation of one active set: `idx` = lanes of the wavefront (0..63) that take part, L = their Lane records by lane number */
inline void exec_wave_op(Lane **L, const int *idx, int n) {
    bool in[64] = {false};
    for (int k = 0; k < n; k++) in[idx[k]] = true;
    const Op op = L[idx[0]]->op;
    uint64_t ballot = 0;
    if (op == OP_BALLOT) for (int k = 0; k < n; k++) if (L[idx[k]]->a) ballot |= 1ull << idx[k];
    uint64_t res[64];
    for (int k = 0; k < n; k++) {
        const int l = idx[k]; Lane *x = L[l];
        uint64_t r = 0;
        switch (op) {
        case OP_BALLOT: r = ballot; break;
        case OP_WAVE_BARRIER: break;
        case OP_SHFL: { const int w = x->c, s = (l & ~(w - 1)) | (x->b & (w - 1)); r = in[s] ? L[s]->a : (++g_inactive_reads, report_inactive(x->site, l, s), 0); break; }
        case OP_SHFL_UP: { const int w = x->c, s = l - x->b; r = (s < (l & ~(w - 1))) ? x->a : (in[s] ? L[s]->a : 0); break; }
        case OP_SHFL_DOWN: { const int w = x->c, s = l + x->b; r = (s > ((l & ~(w - 1)) | (w - 1))) ? x->a : (in[s] ? L[s]->a : 0); break; }
        case OP_SHFL_XOR: { const int w = x->c, s = l ^ x->b; r = (s > ((l & ~(w - 1)) | (w - 1)) || s < 0 || s > 63) ? x->a : (in[s] ? L[s]->a : 0); break; }
        case OP_READLANE: { const int s = x->b & 63; r = in[s] ? L[s]->a : (++g_inactive_reads, report_inactive(x->site, l, s), 0); break; }
        case OP_DPP: {
            /* a = src (low 32) | old (high 32); b = dpp_ctrl; c = row_mask | bank_mask << 4; d = bound_ctrl */
            const uint32_t src = (uint32_t)x->a, old = (uint32_t)(x->a >> 32);
            const int ctrl = x->b, row = l >> 4, pos = l & 15, bank = pos >> 2;
            const bool enabled = ((x->c >> row) & 1) && (((x->c >> 4) >> bank) & 1);
            int s = -1;
            if (ctrl >= 0x101 && ctrl <= 0x10F) { const int k2 = ctrl - 0x100; if (pos + k2 <= 15) s = l + k2; }                 /* row_shl */
            else if (ctrl >= 0x111 && ctrl <= 0x11F) { const int k2 = ctrl - 0x110; if (pos >= k2) s = l - k2; }                  /* row_shr */
            else if (ctrl >= 0x121 && ctrl <= 0x12F) { const int k2 = ctrl - 0x120; s = (l & ~15) | ((pos - k2) & 15); }          /* row_ror */
            else if (ctrl == 0x142) { if (row > 0) s = row * 16 - 1; }                                                            /* row_bcast:15 */
            else if (ctrl == 0x143) { if (row >= 2) s = 31; }                                                                     /* row_bcast:31 */
            else if (ctrl >= 0 && ctrl <= 0xFF) { s = (l & ~3) | ((ctrl >> (2 * (l & 3))) & 3); }                                 /* quad_perm */
            else { fprintf(stderr, "hipemu: DPP control 0x%x is not modelled\n", ctrl); abort(); }
            (void)src;
            if (!enabled) r = old;
            else if (s >= 0 && in[s]) r = (uint32_t)L[s]->a;
            else r = x->d ? 0u : old;
            break; }
        default: abort();
        }
        res[l] = r;
    }
    for (int k = 0; k < n; k++) { Lane *x = L[idx[k]]; x->r = res[idx[k]]; x->state = RUNNABLE; x->op = OP_NONE; }
}

struct StackPool { std::vector<char *> st; ~StackPool() { for (char *p : st) munmap(p, STACK); } };
inline thread_local StackPool t_stacks;

inline void run_block(Block &B) {
    const uint32_t n = B.block.x;
    if (B.lanes.size() < n) B.lanes.resize(n);
    while (t_stacks.st.size() < n) {
        void *p = mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) { fprintf(stderr, "hipemu: cannot map a fiber stack\n"); abort(); }
        mprotect(p, 4096, PROT_NONE);            /* a frame that runs off the stack faults here instead of scribbling over the neighbour's */
        t_stacks.st.push_back((char *)p);
    }
    t_blk = &B;
    for (uint32_t t = 0; t < n; t++) {
        Lane &x = B.lanes[t];
        x.stack = t_stacks.st[t]; x.state = RUNNABLE; x.op = OP_NONE; x.tid = Idx3{t, 0, 0};
        uintptr_t top = ((uintptr_t)x.stack + STACK) & ~(uintptr_t)15;
        void **sp = (void **)top;
        *--sp = nullptr;                         /* the frame of a caller that does not exist */
        *--sp = (void *)&fiber_main;             /* `ret` of the first switch lands here with rsp = top - 8 */
        for (int k = 0; k < 6; k++) *--sp = nullptr;
        x.sp = sp;
    }
    const uint32_t nw = (n + 63) / 64;
    auto run_lane = [&](Lane &x) { void *fake = nullptr; t_lane = &x; HIPEMU_ASAN_ENTER_FIBER(fake, x); hipemu_switch(&B.sched_sp, x.sp); HIPEMU_ASAN_BACK_IN_SCHEDULER(fake); t_lane = nullptr; };
    for (;;) {
        bool any_live = false;
        for (uint32_t w = 0; w < nw; w++) {
            Lane *L[64]; int cnt = 0;
            for (uint32_t l = 0; l < 64; l++) { const uint32_t t = w * 64 + l; L[l] = t < n ? &B.lanes[t] : nullptr; if (L[l]) cnt++; }
            for (;;) {
                for (int l = 0; l < 64; l++) if (L[l] && L[l]->state == RUNNABLE) run_lane(*L[l]);
                /* every live lane of the wavefront is blocked now: the lanes waiting at one call site are an operation's active set */
                int idx[64], m = 0; uint32_t best = ~0u; int sites = 0; uint32_t seen = ~0u;
                for (int l = 0; l < 64; l++) if (L[l] && L[l]->state == WAIT_WAVE) { if (L[l]->site != seen) { sites++; seen = L[l]->site; } best = std::min(best, L[l]->site); }
                if (best == ~0u) break;
                for (int l = 0; l < 64; l++) if (L[l] && L[l]->state == WAIT_WAVE && L[l]->site == best) idx[m++] = l;
                if (sites > 1) {
                    ++g_divergent;
                    if (env_int("HIPEMU_VERBOSE", 0)) {
                        static std::mutex mu; std::lock_guard<std::mutex> g(mu);
                        fprintf(stderr, "hipemu: wavefront operations wait at %d call sites at once (site %u goes first) in block %u\n", sites, best, B.bid.x);
                    }
                }
                exec_wave_op(L, idx, m);
            }
            for (int l = 0; l < 64; l++) if (L[l] && L[l]->state != DONE) any_live = true;
        }
        if (!any_live) break;
        /* every live lane of the workgroup waits at the barrier (wavefronts that have ended do not take part) */
        B.bar_or = 0; B.bar_and = 1; B.bar_count = 0;
        for (uint32_t t = 0; t < n; t++) if (B.lanes[t].state == WAIT_BLOCK) { const int p2 = B.lanes[t].a != 0; B.bar_or |= p2; B.bar_and &= p2; B.bar_count += p2; B.lanes[t].state = RUNNABLE; }
    }
    t_blk = nullptr;
}

inline thread_local const char *t_kernel = "";
inline void segv_report(int sig, siginfo_t *si, void *) {
    char buf[512];
    const Block *b = t_blk; const Lane *l = t_lane;
    int n = snprintf(buf, sizeof buf, "hipemu: signal %d at address %p in kernel %s, workgroup %u, thread %u (stack %p .. %p)\n", sig, si->si_addr, g_kernel,
                     b ? b->bid.x : 0u, l ? l->tid.x : 0u, l ? (void *)l->stack : nullptr, l ? (void *)(l->stack + STACK) : nullptr);
    if (n > 0) { ssize_t w = write(2, buf, (size_t)n); (void)w; }
    _exit(139);
}
inline void install_segv_report() {
    static std::once_flag once;
    std::call_once(once, [] {
        if (!env_int("HIPEMU_SEGV_REPORT", 1)) return;
        static char alt[1 << 16];
        stack_t ss; ss.ss_sp = alt; ss.ss_size = sizeof alt; ss.ss_flags = 0; sigaltstack(&ss, nullptr);
        struct sigaction sa; memset(&sa, 0, sizeof sa); sa.sa_sigaction = segv_report; sa.sa_flags = SA_SIGINFO | SA_ONSTACK; sigemptyset(&sa.sa_mask);
        sigaction(SIGSEGV, &sa, nullptr); sigaction(SIGBUS, &sa, nullptr);
    });
}
inline void trace_launch(const char *name, dim3 grid, dim3 block, size_t shmem) {
    install_segv_report(); g_kernel = name;
    if (shmem > (160u << 10)) { fprintf(stderr, "hipemu: %s asks for %zu bytes of dynamic LDS (160 KB per workgroup)\n", name, shmem); abort(); }
    if (env_int("HIPEMU_VERBOSE", 0) >= 2) fprintf(stderr, "hipemu: launch %s grid (%u, %u, %u) x %u threads, %zu bytes of dynamic LDS\n", name, grid.x, grid.y, grid.z, block.x, shmem);
}
/* OS threads that take workgroups: created once and kept (their fiber stacks with them); a launch that finds the pool busy -- launches from
 * several host threads at once -- runs its workgroups on the launching thread alone */
struct Pool {
    std::mutex use, m; std::condition_variable cv, cv_done;
    const std::function<void()> *job = nullptr; uint64_t gen = 0; int want = 0, running = 0, n_threads = 0;
    void grow(int n) {
        while (n_threads < n) {
            const int idx = n_threads++;
            std::thread([this, idx] {
                uint64_t seen = 0;
                for (;;) {
                    const std::function<void()> *f = nullptr;
                    { std::unique_lock<std::mutex> l(m); cv.wait(l, [&] { return gen != seen; }); seen = gen; if (idx < want) f = job; }
                    if (!f) continue;
                    (*f)();
                    { std::lock_guard<std::mutex> l(m); if (--running == 0) cv_done.notify_all(); }
                }
            }).detach();                 /* (they wait for work until the process ends) */
        }
    }
    void run(const std::function<void()> &f, int helpers) {
        grow(helpers);
        { std::lock_guard<std::mutex> l(m); job = &f; want = helpers; running = helpers; gen++; }
        cv.notify_all();
        f();
        { std::unique_lock<std::mutex> l(m); cv_done.wait(l, [&] { return running == 0; }); job = nullptr; want = 0; }
    }
};
inline Pool &pool() { static Pool *p = new Pool(); return *p; }      /* never destroyed: its threads wait on it until the process ends */
template <class F> inline void launch(dim3 grid, dim3 block, F &&body_fn) {
    if (block.y != 1 || block.z != 1) { fprintf(stderr, "hipemu: only one-dimensional workgroups are modelled\n"); abort(); }
    const uint64_t total = (uint64_t)grid.x * grid.y * grid.z;
    if (total == 0 || block.x == 0) return;
    const std::function<void()> body(body_fn);
    std::atomic<uint64_t> next{0};
    auto worker = [&] {
        Block B; B.grid = grid; B.block = block; B.body = &body;
        for (;;) {
            const uint64_t i = next.fetch_add(1);
            if (i >= total) break;
            B.bid = Idx3{(uint32_t)(i % grid.x), (uint32_t)((i / grid.x) % grid.y), (uint32_t)(i / ((uint64_t)grid.x * grid.y))};
            run_block(B);
        }
    };
    const int hw = std::max(1, std::min<int>(env_int("HIPEMU_THREADS", (int)std::thread::hardware_concurrency()), 64));
    const int nt = (int)std::min<uint64_t>((uint64_t)hw, total);
    if (nt <= 1 || total * block.x < 4096 || !pool().use.try_lock()) { worker(); return; }
    const std::function<void()> w(worker);
    pool().run(w, nt - 1);
    pool().use.unlock();
}
}  // namespace hipemu

#define threadIdx (hipemu::t_lane->tid)
#define blockIdx (hipemu::t_blk->bid)
#define blockDim (hipemu::t_blk->block)
#define gridDim (hipemu::t_blk->grid)
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
    do { (void)(stream); hipemu::trace_launch(#kernel, dim3(grid), dim3(block), (size_t)(shmem)); hipemu::launch(dim3(grid), dim3(block), [=]() { kernel(__VA_ARGS__); }); } while (0)

/* ---- device intrinsics ------------------------------------------------------------------------------------------------------------ */
#define HIPEMU_SITE ((uint32_t)__COUNTER__)
static inline void hipemu_syncthreads() { hipemu::block_barrier(); }
#define __syncthreads() hipemu_syncthreads()
static inline int __syncthreads_or(int p) { hipemu::block_barrier(p); return hipemu::t_blk->bar_or; }
static inline int __syncthreads_and(int p) { hipemu::block_barrier(p); return hipemu::t_blk->bar_and; }
static inline int __syncthreads_count(int p) { hipemu::block_barrier(p); return hipemu::t_blk->bar_count; }
#define __builtin_assume(x) ((void)0)
#define __ballot(p) hipemu::wave_op(hipemu::OP_BALLOT, HIPEMU_SITE, (uint64_t)((p) ? 1 : 0))
#define __any(p) (hipemu::wave_op(hipemu::OP_BALLOT, HIPEMU_SITE, (uint64_t)((p) ? 1 : 0)) != 0)
#define __all(p) (hipemu::wave_op(hipemu::OP_BALLOT, HIPEMU_SITE, (uint64_t)((p) ? 0 : 1)) == 0)
namespace hipemu {
template <class T> inline uint64_t to_bits(T v) { static_assert(sizeof(T) <= 8, "shuffle of at most 64 bits"); uint64_t b = 0; memcpy(&b, &v, sizeof(T)); return b; }
template <class T> inline T from_bits(uint64_t b) { T v; memcpy(&v, &b, sizeof(T)); return v; }
template <class T> inline T shfl(Op op, uint32_t site, T v, int arg, int width) { return from_bits<T>(wave_op(op, site, to_bits(v), arg, width)); }
}
#define HIPEMU_SHFL3(op, v, arg, width, ...) hipemu::shfl(op, HIPEMU_SITE, (v), (int)(arg), (int)(width))
#define __shfl(...) HIPEMU_SHFL3(hipemu::OP_SHFL, __VA_ARGS__, 64, 64)
#define __shfl_up(...) HIPEMU_SHFL3(hipemu::OP_SHFL_UP, __VA_ARGS__, 64, 64)
#define __shfl_down(...) HIPEMU_SHFL3(hipemu::OP_SHFL_DOWN, __VA_ARGS__, 64, 64)
#define __shfl_xor(...) HIPEMU_SHFL3(hipemu::OP_SHFL_XOR, __VA_ARGS__, 64, 64)
#define __builtin_amdgcn_readlane(v, l) ((int)(uint32_t)hipemu::wave_op(hipemu::OP_READLANE, HIPEMU_SITE, (uint64_t)(uint32_t)(v), (int)(l)))
#define __builtin_amdgcn_update_dpp(old, src, ctrl, row_mask, bank_mask, bound_ctrl) \
    ((int)(uint32_t)hipemu::wave_op(hipemu::OP_DPP, HIPEMU_SITE, (uint64_t)(uint32_t)(src) | ((uint64_t)(uint32_t)(old) << 32), (int)(ctrl), (int)((row_mask) | ((bank_mask) << 4)), (bound_ctrl) ? 1u : 0u))
#define __builtin_amdgcn_wave_barrier() ((void)hipemu::wave_op(hipemu::OP_WAVE_BARRIER, HIPEMU_SITE, 0))
#define __builtin_amdgcn_fence(order, scope) __atomic_thread_fence(__ATOMIC_SEQ_CST)
#define __builtin_nontemporal_store(v, p) (*(p) = (v))
#define __builtin_readcyclecounter() ((unsigned long long)__builtin_ia32_rdtsc())
static inline uint32_t hipemu_alignbyte(uint32_t hi, uint32_t lo, uint32_t sh) { return (uint32_t)(((((uint64_t)hi) << 32) | lo) >> (8 * (sh & 3))); }
#define __builtin_amdgcn_alignbyte(hi, lo, sh) hipemu_alignbyte((hi), (lo), (sh))
/* global_load_lds, 4 or 16 bytes per lane: the wavefront's LDS base + size x lane <- every lane's own global address */
static inline void hipemu_global_load_lds(const void *g, void *lds, int size, int, int) {
    if (size != 4 && size != 16) { fprintf(stderr, "hipemu: global_load_lds of %d bytes is not modelled\n", size); abort(); }
    memcpy((char *)lds + (size_t)size * (hipemu::t_lane->tid.x & 63u), g, (size_t)size);
}
#define __builtin_amdgcn_global_load_lds(g, lds, size, off, aux) hipemu_global_load_lds((const void *)(g), (void *)(lds), (size), (off), (aux))

static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_block() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline void __threadfence_system() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }
static inline int __popc(uint32_t v) { return __builtin_popcount(v); }
static inline int __popcll(uint64_t v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline uint32_t __float_as_uint(float f) { uint32_t i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(uint32_t i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline uint32_t __umulhi(uint32_t a, uint32_t b) { return (uint32_t)(((uint64_t)a * b) >> 32); }

template <class T> static inline T atomicAdd(T *p, T v) { return __atomic_fetch_add(p, v, __ATOMIC_SEQ_CST); }
static inline float atomicAdd(float *p, float v) { uint32_t *q = (uint32_t *)p, o = __atomic_load_n(q, __ATOMIC_SEQ_CST); for (;;) { float f; memcpy(&f, &o, 4); f += v; uint32_t n; memcpy(&n, &f, 4); if (__atomic_compare_exchange_n(q, &o, n, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) { float r; memcpy(&r, &o, 4); return r; } } }
template <class T> static inline T atomicSub(T *p, T v) { return __atomic_fetch_sub(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicOr(T *p, T v) { return __atomic_fetch_or(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicAnd(T *p, T v) { return __atomic_fetch_and(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicExch(T *p, T v) { return __atomic_exchange_n(p, v, __ATOMIC_SEQ_CST); }
template <class T> static inline T atomicCAS(T *p, T expect, T v) { __atomic_compare_exchange_n(p, &expect, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST); return expect; }
template <class T> static inline T atomicMax(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o < v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
template <class T> static inline T atomicMin(T *p, T v) { T o = __atomic_load_n(p, __ATOMIC_SEQ_CST); while (o > v && !__atomic_compare_exchange_n(p, &o, v, false, __ATOMIC_SEQ_CST, __ATOMIC_SEQ_CST)) {} return o; }
/* mixed argument types as the HIP overloads take them (unsigned int * with an int literal, unsigned long long * with uint64_t) */
template <class T, class U> static inline T atomicAdd(T *p, U v) { return atomicAdd<T>(p, (T)v); }
template <class T, class U> static inline T atomicOr(T *p, U v) { return atomicOr<T>(p, (T)v); }
template <class T, class U> static inline T atomicMax(T *p, U v) { return atomicMax<T>(p, (T)v); }
template <class T, class U> static inline T atomicMin(T *p, U v) { return atomicMin<T>(p, (T)v); }
template <class T, class U, class V> static inline T atomicCAS(T *p, U e, V v) { return atomicCAS<T>(p, (T)e, (T)v); }

#endif
