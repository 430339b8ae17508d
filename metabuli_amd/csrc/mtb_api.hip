/* mtb_api.hip -- C ABI (include/mtb.h) over the HIP kernels.  gfx950 only.
 * The library never falls back to a CPU path: every entry point that computes
 * launches kernels on the context's stream and reports HIP errors.           */
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <functional>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mtb.h"
#include "host_db.h"
#include "kernels_extract.h"
#include "kernels_index.h"
#include "kernels_join.h"
#include "kernels_dir.h"
#include "kernels_scan.h"
#include "kernels_score.h"
#include "kernels_score_fast.h"
#include "kernels_score_long.h"
#include "kernels_score_many.h"
#include "kernels_seg_order.h"
#include "kernels_sort.h"
#include "mtb_core.h"
#include "mtb_options.h"

#if defined(__HIP_DEVICE_COMPILE__) && defined(__AMDGCN__) && !defined(__gfx950__)
#error "libmtb is written for gfx950 (MI355X) only: 160 KB of LDS per CU (k_score_long holds 70 KB per workgroup), global_load_lds_dwordx4, wave64"
#endif

static thread_local std::string g_err;
static mtb_status fail(mtb_status s, const std::string &m) { g_err = m; return s; }

#define HIPCHK(x)                                                                                  \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess)                                                                      \
            return fail(MTB_ERR_DEVICE, std::string(#x) + ": " + hipGetErrorString(e_));           \
    } while (0)
#define STCHK(x)                                                                                   \
    do { mtb_status s_ = (x); if (s_ != MTB_OK) return s_; } while (0)

/* Query::taxCnt lists packed back to back before they cross PCIe: the scorer leaves every read's entries inside a slot sized by
 * its bound (one per position bucket, 18 for a 150 bp read) while a read has one or two entries -- copying the slots moved 10 x
 * the payload over pageable D2H copies (the driver's GPU stage was 1.05 s per 8 M reads, the kernels 0.1 s of it). */
__global__ __launch_bounds__(256) void k_taxcnt_n(const mtb_result *__restrict__ res, uint64_t n, uint32_t *__restrict__ cnt) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) cnt[i] = res[i].n_taxcnt;
}
__global__ __launch_bounds__(256) void k_taxcnt_pack(mtb_result *__restrict__ res, uint64_t n, const uint64_t *__restrict__ new_off, const int32_t *__restrict__ tt,
                                                      const uint32_t *__restrict__ tc, int32_t *__restrict__ tt2, uint32_t *__restrict__ tc2) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint32_t k = res[i].n_taxcnt, src = res[i].taxcnt_off;
    const uint64_t dst = new_off[i];
    for (uint32_t j = 0; j < k; j++) { tt2[dst + j] = tt[src + j]; tc2[dst + j] = tc[src + j]; }
    res[i].taxcnt_off = (uint32_t)dst;
}

/* 2-bit reads (mtb_classify_batch_packed): slots of 8 bases a read needs; bases of read r written as text behind offs[r] */
__global__ __launch_bounds__(256) void k_pack_slots(const uint32_t *__restrict__ lens, uint64_t n, uint32_t *__restrict__ slots) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) slots[i] = (lens[i] + 7u) >> 3;
}
__global__ __launch_bounds__(64) void k_unpack_reads(const uint8_t *__restrict__ packed2, const uint8_t *__restrict__ nmask, const uint64_t *__restrict__ offs,
                                                     const uint64_t *__restrict__ slot_off, uint64_t n, char *__restrict__ bases) {
    for (uint64_t r = blockIdx.x; r < n; r += gridDim.x) {
        const uint64_t o = offs[r], s0 = slot_off[r];
        const uint32_t L = (uint32_t)(offs[r + 1] - o);
        for (uint32_t k = threadIdx.x; k < L; k += 64) {            /* one base per lane: a wave writes 64 consecutive bytes */
            const uint64_t g = s0 + (k >> 3); const uint32_t j = k & 7u;
            const uint32_t w = (uint32_t)packed2[2 * g] | ((uint32_t)packed2[2 * g + 1] << 8);
            const bool bad = (nmask[g] >> j) & 1u;
            const uint32_t c = (w >> (2 * j)) & 3u;
            bases[o + k] = bad ? 'N' : (char)((0x47544341u >> (8 * c)) & 0xFFu);       /* "ACTG": GeneticCode's nuc2int order */
        }
    }
}

struct DevBuf { void *p = nullptr; size_t cap = 0; };
/* the slab pool of the generic scorer's large-segment launches (reads beyond every LDS budget: 2 of 200 k long reads on the bench's index).  Round 6: 16 GiB
 * instead of 48 -- the 32 GiB went to the long-read sub-batches (200 k x 10 kb: two of 100 k reads instead of four of 50 k; denser sorted queries) */
#define MTB_SLAB_POOL_MAX (16ull << 30)
#define MTB_LONG_WIN_MAX_PER_Q 24.0   /* long reads: targets per query metamer up to which the window form is taken without being asked for */
#define MTB_OVF_STRIPES 256u

struct mtb_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    mtb_tables *d_tabs = nullptr;
    mtb_tables h_tabs;
    std::map<std::string, DevBuf> bufs;
    std::mutex bufs_mu;              /* the buffer table may be grown from a helper thread (mtb_ctx_reserve) while the context's thread opens an index */
    std::mutex reserve_mu;           /* held by mtb_ctx_reserve for its whole run */
    std::atomic<bool> reserve_cancel{false};     /* an index open on this context found device memory short: a running reservation stops after its current buffer, a new one does nothing, and what was reserved is handed back (open_make_room) */
    uint64_t *d_scal = nullptr;      /* [0] match counter, [1] overflow, [2] n_large, [3] max_seg, [4] max_len, [5] n_big */
    uint64_t *d_xscal = nullptr;     /* = d_scal + 8: single-pass extraction counters */
    unsigned long long *d_ovfctr = nullptr;      /* MTB_OVF_STRIPES counters of the striped overflow list, 64 bytes apart (JoinSegArgs::ovf_stripes) */
    uint64_t ovf_region = 0, ovf_max_region = 0; /* entries per region in the last slot-mode join; fullest region */
    hipEvent_t ev[8];
    mtb_batch_stats stats;
    int profiling = 0;
    struct KEv { int id; hipEvent_t a, b; };
    std::vector<KEv> kev;            /* events of the current batch      */
    std::vector<hipEvent_t> ev_pool; /* recycled events                  */
    std::vector<mtb_ctx *> lanes;    /* extra stream contexts (mtb_ctx_set_streams) */
    bool is_lane = false;            /* lanes share the parent's tables  */
    uint32_t seg_epoch = 0;          /* tag of the live slots in the "segm" buffer (1..MTB_SLOT_EPOCHS) */
    const void *seg_clean_p = nullptr; size_t seg_clean_cap = 0;     /* the allocation the epoch count belongs to: the one prepare_slots last cleared (another pointer or size, e.g. after mtb_ctx_reserve grew or created "segm", holds stale bytes) */
    double extract_yield = 0.0;      /* metamers per base of the previous batch (single-pass extraction buffer sizing) */
    uint64_t part_n_reads = 0; uint32_t part_max_len = 0;   /* batch state between mtb_part_extract and mtb_part_score */
    int part_mode = 0; uint32_t part_max_q = 0; uint64_t part_nk_real = 0;      /* part_mode 1: the batch's metamers carry ordinals, the matches come home into slot segments */
    bool placement_probe = false;    /* mtb_ctx_set_placement_probe: a new big slot buffer is chosen among candidate allocations */
    uint64_t ws_limit = 0;           /* workspace budget of a batch in bytes; 0 = what hipMemGetInfo reports free (+ what the context holds) */
    double ws_per_base = 0.0;        /* workspace bytes per base: what the buffers hold / the largest sub-batch they were grown for (HBM-budgeted batching) */
    uint64_t ws_max_sub_bases = 0;   /* bases of the largest sub-batch since the workspace was last released */
    int ws_seq_mode = 0;             /* the buffer sets of short and long reads differ: a change of mode releases the workspace */
    uint32_t last_sub_batches = 0;
    bool fast_used = false;          /* the last dev_score call launched k_score_fast (its slow-list count sits in d_scal[6]) */
    bool no_lslot = false;           /* classify_one is redoing a long-read range on the exact-segment path */
    const mtb_kmer *last_sorted = nullptr; uint64_t last_sorted_n = 0;       /* the last fused slot-path batch's sorted metamers (mtb_ctx_join_footprint) */
    /* what the join leaves dead until the next batch's extraction -- the metamer buffer that does NOT hold the sorted list, both digit arrays: the scorer's large
     * per-batch temporaries (the grouped overflow list, the deferred reads' segments) are carved out of them (scratch()) instead of being allocations of their own:
     * 10 M reads of held-out genomes need 133 instead of 154 GiB of workspace -- one sub-batch instead of two on a 288 GB part next to a 126 GiB index */
    struct DeadSeg { char *p; size_t cap, used; };
    DeadSeg dead[3]; int n_dead = 0;
    uint64_t last_scratch_bytes = 0;     /* mtb_ctx_last_scratch_bytes */
    /* upload of the NEXT batch's packed reads while the current batch computes (mtb_prefetch_batch_packed): a copy stream, two sets of
     * input buffers, and what the set that is being filled holds */
    hipStream_t copy_stream = nullptr; hipEvent_t copy_done[2] = {nullptr, nullptr};       /* per input buffer set: the prefetch into it is complete */
    hipEvent_t unpacked[2] = {nullptr, nullptr};     /* per set: the last classify call that read it has unpacked it (recorded on the compute stream; a prefetch into the set waits for it) */
    bool unpacked_rec[2] = {false, false};
    uint64_t prefetch_issued = 0, prefetch_used = 0; /* mtb_ctx_prefetch_stats */
    /* results on their way back while the next batch computes (mtb_classify_batch_packed_async): a download stream, the event that closes the
     * queued copies, and which of the two device-side result buffer sets the next call writes */
    hipStream_t down_stream = nullptr; hipEvent_t down_ready = nullptr, down_done = nullptr; bool down_pending = false; int res_set = 0;
    int pk_set = 0;                                  /* the set the last classify call read */
    /* one record per set.  The protocol of mtb.h is prefetch(k+1), classify(k): when batch k+1 is prefetched, batch k (prefetched one call
     * earlier) is still waiting for its classify call, so TWO prefetches are outstanding for a moment -- with a single record classify(k)
     * found batch k+1's key, discarded it and uploaded k again */
    struct Prefetched { const void *key = nullptr, *key2 = nullptr; uint64_t n_reads = 0; bool valid = false; } pre[2];
    struct JoinTune { uint64_t key = 0; int calls = 0, pending = -1, best = -1; float ms[3] = {0.0f, 0.0f, 0.0f}; hipEvent_t e0 = nullptr, e1 = nullptr; } join_tunes[4]; uint32_t join_tune_next = 0;      /* dev_join: the short-read instantiation that is fastest for this index and batch size */
    uint64_t many_stats[4] = {0, 0, 0, 0};           /* last slot-path batch: reads deferred by the first scoring launches, of those scored by k_score_many, their matches, the survivors of the dead-species drop */
    uint32_t lslot_tf_start = 1;                     /* long-read slot ranges: tail factor the next batch starts with (1 = a quarter of the metamers, 4 = all) */
    MtbOptions opt;                                  /* the experiment / diagnosis switches: the MTB_* environment at mtb_ctx_create, mtb_ctx_set_option afterwards (mtb_options.h) */
};
/* buffers that carry a call's inputs / outputs (host-buffer entry points) are not workspace */
static bool is_io_buf(const std::string &n) { return n == "bases" || n == "offs" || n == "bases2" || n == "offs2" || n == "results" || n == "resultsb" || n == "tctax" || n == "tccnt" || n.compare(0, 2, "pk") == 0; }
static size_t held_bytes(mtb_ctx *c) { std::lock_guard<std::mutex> lk(c->bufs_mu); size_t b = 0; for (auto &kv : c->bufs) if (!is_io_buf(kv.first)) b += kv.second.cap; return b; }
/* the part of it that grows with the sub-batch (the slab pool of the large-segment scorer is bounded on its own) */
static size_t scaling_bytes(mtb_ctx *c) { std::lock_guard<std::mutex> lk(c->bufs_mu); size_t b = 0; for (auto &kv : c->bufs) if (!is_io_buf(kv.first) && kv.first != "slabs") b += kv.second.cap; return b; }
static void release_workspace(mtb_ctx *c) {
    std::lock_guard<std::mutex> lk(c->bufs_mu);
    for (auto &kv : c->bufs) if (kv.second.p && !is_io_buf(kv.first)) { hipError_t e = hipFree(kv.second.p); (void)e; kv.second.p = nullptr; kv.second.cap = 0; }
    c->ws_per_base = 0.0; c->ws_max_sub_bases = 0;
    c->seg_clean_p = nullptr; c->seg_clean_cap = 0;       /* a later "segm" at the same address is not the buffer that was cleared */
}

/* RAII bracket around one kernel launch (only when profiling is on) */
struct KTimer {
    mtb_ctx *c; int id; hipEvent_t a = nullptr, b = nullptr;
    KTimer(mtb_ctx *c_, int id_) : c(c_), id(id_) {
        if (!c->profiling) return;
        a = take(); b = take();
        hipError_t e = hipEventRecord(a, c->stream); (void)e;
    }
    ~KTimer() {
        if (!c->profiling) return;
        hipError_t e = hipEventRecord(b, c->stream); (void)e;
        c->kev.push_back({id, a, b});
    }
    hipEvent_t take() {
        hipEvent_t ev;
        if (!c->ev_pool.empty()) { ev = c->ev_pool.back(); c->ev_pool.pop_back(); return ev; }
        hipError_t e = hipEventCreate(&ev); (void)e;
        return ev;
    }
};
static void collect_kernel_times(mtb_ctx *c) {
    for (auto &k : c->kev) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, k.a, k.b) == hipSuccess) { c->stats.ms_kernel[k.id] += ms; c->stats.n_launch[k.id]++; }
        c->ev_pool.push_back(k.a); c->ev_pool.push_back(k.b);
    }
    c->kev.clear();
}

struct mtb_index {
    mtb_ctx *ctx = nullptr;
    uint64_t T = 0;
    uint64_t *d_values = nullptr; uint32_t *d_info = nullptr; bool own = false;
    bool own_tax = true;             /* slices share the parent's device taxonomy */
    bool match_last = false;         /* partitions other than the last one match their final entry too */
    mtbhost::Taxonomy tax;
    int32_t *d_canon = nullptr, *d_parent = nullptr, *d_depth = nullptr, *d_spparent = nullptr, *d_tax2species = nullptr;
    uint8_t *d_under = nullptr, *d_accleaf = nullptr;
    mtb_tax_node *d_node = nullptr;
    mtb_params params;
    uint32_t info_mask = 0xFFFFFFFFu;
    /* amino-acid prefix directory (kernels_dir.h); absent for views and for indices it cannot describe */
    uint32_t *d_dir = nullptr; uint64_t *d_dirbase = nullptr; int32_t dir_L = 0; uint32_t dir_buckets = 0;
    bool packed = false;             /* d_values holds packed words (kernels_dir.h): the fused join's state; everything else unpacks first */
    bool info_owned = false;         /* d_info was (re)allocated by the library although the value array is borrowed (after mtb_index_seal) */
    /* The flat <-> packed conversion rewrites the target array in place and is not idempotent: it happens under `state_mu`, only
     * while no kernel that reads the array is in flight (`users` = join launches between acquire and the end of their stream
     * sync, whatever context or stream they run on), and is complete (stream-synchronised) before the lock is released. */
    std::mutex state_mu; std::condition_variable state_cv; int users = 0;
    int views = 0;                   /* live mtb_index_slice views: they read the parent's flat arrays, so the parent stays flat */
    mtb_index *parent = nullptr;     /* of a view */
    uint64_t open_chunks = 0, open_chunk_words = 0, open_peak_bytes = 0, open_free0 = 0;      /* mtb_index_open_stats */
    void *ipc_stage[4] = {nullptr, nullptr, nullptr, nullptr};       /* mtb_index_export: copies of the arrays too small for an inter-process handle of their own */
};

template <typename T>
static mtb_status ensure(mtb_ctx *c, const char *name, size_t elems, T **out) {
    if (c->down_pending) {
        /* asynchronous results (mtb_classify_batch_packed_async): whoever asks for a buffer that queued copies still read waits for them first
         * (the asynchronous path itself asks for the other set) */
        static const char *const sets[2][3] = {{"results", "tctax2", "tccnt2"}, {"resultsb", "tctax2b", "tccnt2b"}};
        const int fl = c->res_set ^ 1;
        if (!strcmp(name, sets[fl][0]) || !strcmp(name, sets[fl][1]) || !strcmp(name, sets[fl][2])) { HIPCHK(hipEventSynchronize(c->down_done)); c->down_pending = false; }
    }
    std::lock_guard<std::mutex> lk(c->bufs_mu);
    DevBuf &b = c->bufs[name];
    size_t bytes = elems * sizeof(T);
    if (bytes == 0) bytes = 64;
    if (b.cap < bytes) {
        if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
        size_t want = bytes + bytes / 16 + 256;
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        if (want > fr) { want = bytes; if (want > fr) return fail(MTB_ERR_OOM, std::string("not enough HBM for buffer ") + name); }
        hipError_t e = hipMalloc(&b.p, want);
        if (e != hipSuccess) {
            b.p = nullptr;
            (void)hipGetLastError();          /* the failure is reported here: do not leave it in the runtime's sticky slot for the next hipGetLastError() check */
            return fail(MTB_ERR_OOM, std::string("hipMalloc failed for ") + name + ": " + hipGetErrorString(e));
        }
        b.cap = want;
#ifdef MTB_POISON_ALLOC     /* robustness build (make libmtb_xpoison.so X=-DMTB_POISON_ALLOC): no kernel may rely on fresh device memory being zero */
        HIPCHK(hipMemsetAsync(b.p, 0xA5, want, c->stream));         /* on the library's stream: it must not overtake or trail the kernels that use the buffer */
        HIPCHK(hipStreamSynchronize(c->stream));                    /* ... nor a copy that another stream (mtb_prefetch_batch_packed) is about to make into it */
#endif
    }
    *out = (T *)b.p;
    return MTB_OK;
}
/* a per-batch temporary that is written before it is read: out of the buffers the join left dead (first fit, 256-byte aligned), else a buffer of its own */
template <typename T>
static mtb_status scratch(mtb_ctx *c, const char *name, size_t elems, T **out) {
    const size_t bytes = (elems * sizeof(T) + 255) & ~(size_t)255;
    for (int i = 0; i < c->n_dead; i++) {
        mtb_ctx::DeadSeg &d = c->dead[i];
        if (d.cap - d.used >= bytes) { *out = (T *)(d.p + d.used); d.used += bytes; c->last_scratch_bytes += bytes; return MTB_OK; }
    }
    return ensure(c, name, elems, out);
}
/* Placement of the slot buffer.  The join's 1.1 G scattered 16-byte slot stores miss the L1 TLB once each, and what a miss costs
 * depends on where the 27 GB buffer landed: the same process runs the join at 47.5 or at 56.5 ms depending on nothing but a
 * re-allocation of this buffer (profiles/scripts/join_vs_placement.py; the other buffers do not matter).  So a new slot buffer is
 * chosen among a few candidate allocations by a probe that does what the join does to it -- random 16-byte non-temporal stores
 * over the whole buffer -- and the losers are handed back.  One-time cost per (re)allocation of a big buffer: 0.3 - 1 s per candidate
 * (hipMalloc of tens of GB), at most ~2.5 s; buffers below 8 GB (host batches of the stand-alone driver) are allocated as they come.
 * Measured over processes on one box: join 45 / 57 / 57 / 57 ms without, 42.5 - 47.6 with (profiles/r02_notes.md). */
__global__ __launch_bounds__(256) void k_probe_scatter(mtb_slot16 *buf, uint64_t n_slots, uint32_t per_thread, uint32_t seed) {
    uint64_t x = ((uint64_t)blockIdx.x * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull + seed;
    for (uint32_t j = 0; j < per_thread; j++) {
        x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        mtb_slot16 *p = buf + (x % n_slots);
        __builtin_nontemporal_store((uint64_t)0, &p->a); __builtin_nontemporal_store((uint64_t)0, &p->b);      /* epoch 0 = not live */
    }
}
__global__ __launch_bounds__(256) void k_clear_words(uint64_t *p, uint64_t n_words) {
    for (uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x; i < n_words; i += (uint64_t)gridDim.x * 256) p[i] = 0;
    MTB_END_RELEASE();
}
static mtb_status ensure_placed(mtb_ctx *c, const char *name, size_t elems, mtb_slot16 **out) {
    DevBuf &b = c->bufs[name];
    const size_t bytes = std::max<size_t>(elems * sizeof(mtb_slot16), 64);
    /* experiment switch (tests/test_gpu_contig.py, profiles/r03_notes.md): the slot buffer -- of ANY size -- from physically
     * contiguous VRAM, the configuration in which round 2 saw pair scores differ from the oracle's */
    if (c->opt.segm_contig) {
        if (b.cap >= bytes) { *out = (mtb_slot16 *)b.p; return MTB_OK; }
        if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
        const size_t want = bytes + bytes / 16 + 256 + (c->opt.segm_pad ? (1u << 20) : 0);
        hipError_t e = hipExtMallocWithFlags(&b.p, want, hipDeviceMallocContiguous);
        if (e != hipSuccess) { b.p = nullptr; (void)hipGetLastError(); return fail(MTB_ERR_OOM, std::string("contiguous allocation failed for ") + name + ": " + hipGetErrorString(e)); }
        b.cap = want;
        *out = (mtb_slot16 *)b.p;
        return MTB_OK;
    }
    /* Off unless the caller asks for it (mtb_ctx_set_placement_probe, or MTB_PLACEMENT_PROBE=1): in a process whose only device
     * allocations are the library's own (the stand-alone driver) a plain hipMalloc lands as well as any candidate
     * (profiles/scripts/alloc_probe.hip, profiles/r03_notes.md section 3), and the search costs the first big batch 2.5 - 3.8 s
     * (profiles/r03_e2e_driver_30Mreads.txt).  A process with a long allocation history of its own (bench.py under torch's caching
     * allocator) turns it on. */
    const bool no_probe = c->opt.no_placement_probe != 0, env_probe = c->opt.placement_probe != 0;
    if (b.cap >= bytes || bytes < (8ull << 30) || no_probe || !(c->placement_probe || env_probe)) return ensure(c, name, elems, out);
    if (b.p) { hipError_t e = hipFree(b.p); (void)e; b.p = nullptr; b.cap = 0; }
    const size_t want = bytes + bytes / 16 + 256;
    const auto t_begin = std::chrono::steady_clock::now();
    struct Cand { void *p; float ms; size_t bytes; };
    std::vector<Cand> held;                 /* the best candidate so far + rejected ones kept allocated so that the next one lands elsewhere */
    std::vector<float> seen; std::vector<void *> seen_p, pads; std::vector<size_t> sizes;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    struct Guard {                          /* every early return (a failing HIP call below) hands the candidates, the pads and the events back */
        std::vector<Cand> &held; std::vector<void *> &pads; hipEvent_t &e0, &e1; bool armed = true;
        ~Guard() { if (!armed) return; for (auto &h : held) { hipError_t e = hipFree(h.p); (void)e; } for (void *q : pads) { hipError_t e = hipFree(q); (void)e; }
                   if (e0) { hipError_t e = hipEventDestroy(e0); (void)e; } if (e1) { hipError_t e = hipEventDestroy(e1); (void)e; } }
    } guard{held, pads, e0, e1};
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int attempt = 0; attempt < 8; attempt++) {
        if (attempt >= 2 && std::chrono::duration<double>(std::chrono::steady_clock::now() - t_begin).count() > 2.0) break;      /* one-time cost, bounded */
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        if (want + (attempt ? (8ull << 30) : 0) > fr) {                /* no room for another candidate next to the ones held: drop the worst rejected one */
            if (held.size() < 2) break;
            size_t worst = 0; for (size_t k = 1; k < held.size(); k++) if (held[k].ms > held[worst].ms) worst = k;
            { hipError_t e = hipFree(held[worst].p); (void)e; } held.erase(held.begin() + (long)worst);
            /* the hole just freed would be handed out again as it is: park a few GB in it first so that the next candidate is
             * composed differently (part of the hole, part of what else is free) */
            void *pad = nullptr;
            if (hipMalloc(&pad, ((size_t)(attempt % 3) + 1) * (3ull << 30)) == hipSuccess) pads.push_back(pad); else (void)hipGetLastError();
            HIPCHK(hipMemGetInfo(&fr, &tot));
            if (want + (4ull << 30) > fr) break;
        }
        void *p = nullptr;
        const size_t ask = want;        /* (a power-of-two request -- one buddy block, if the device still has one -- probed 0.60 - 0.79 ms: no better) */
        if (hipMalloc(&p, ask) != hipSuccess) { (void)hipGetLastError(); break; }
        sizes.push_back(ask);
        float best = 1e30f;
        for (int rep = 0; rep < 3; rep++) {                             /* first repetition = warm-up of the translations */
            HIPCHK(hipEventRecord(e0, c->stream));
            hipLaunchKernelGGL(k_probe_scatter, dim3(8192), dim3(256), 0, c->stream, (mtb_slot16 *)p, (uint64_t)(bytes / sizeof(mtb_slot16)), 8u, 12345u + (uint32_t)rep);
            HIPCHK(hipEventRecord(e1, c->stream));
            HIPCHK(hipEventSynchronize(e1));
            float ms = 0; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
            if (rep && ms < best) best = ms;
        }
        held.push_back({p, best, ask}); seen.push_back(best); seen_p.push_back(p);
        /* after four candidates: two placements within 3 % of the fastest one seen = that is as good as it gets here, stop looking
         * (two alone can agree on a mediocre kind: 0.596 / 0.582 ms when 0.545 existed) */
        float lo = 1e30f; for (float v : seen) lo = std::min(lo, v);
        int near = 0; for (float v : seen) if (v <= lo * 1.03f) near++;
        float hi = 0.0f; for (float v : seen) hi = std::max(hi, v);
        if (seen.size() >= 4 && near >= 2 && lo <= 0.9f * hi) break;   /* ... and clearly better than the worst one seen */
    }
    guard.armed = false;
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    for (void *q : pads) { hipError_t e = hipFree(q); (void)e; }
    if (held.empty()) return ensure(c, name, elems, out);              /* not even one candidate fitted: the plain path reports the error */
    size_t pick = 0;
    for (size_t k = 1; k < held.size(); k++) if (held[k].ms < held[pick].ms) pick = k;
    for (size_t k = 0; k < held.size(); k++) if (k != pick) { hipError_t e = hipFree(held[k].p); (void)e; }
    std::vector<Cand> &cands = held;
    if (c->opt.placement_verbose) { fprintf(stderr, "mtb: slot buffer placement probe:"); for (size_t k = 0; k < seen.size(); k++) fprintf(stderr, " %.3f@%p", seen[k], seen_p[k]); fprintf(stderr, " ms -> %.3f (%.0f ms spent)\n", cands[pick].ms, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count()); }
    b.p = cands[pick].p; b.cap = cands[pick].bytes;
    *out = (mtb_slot16 *)b.p;
    return MTB_OK;
}
static void release(mtb_ctx *c, const char *name) {
    std::lock_guard<std::mutex> lk(c->bufs_mu);
    auto it = c->bufs.find(name);
    if (it != c->bufs.end()) { if (it->second.p) { hipError_t e = hipFree(it->second.p); (void)e; } c->bufs.erase(it); }
}

extern "C" {

const char *mtb_version(void) { return "metabuli_amd 0.1 (gfx950)"; }
const char *mtb_last_error(void) { return g_err.c_str(); }

void mtb_default_params(mtb_params *p) {   /* setClassifyDefaults, classify.cpp:10-37: kmerFormat 1 unless db.parameters says otherwise */
    p->seq_mode = 2; p->syncmer = 0; p->smer_len = 5; p->kmer_format = 1; p->min_cons_cnt = 4; p->min_cons_cnt_euk = 9;
    p->min_score = 0.0f; p->min_sp_score = 0.0f; p->tie_ratio = 0.95f; p->accession_level = 0; p->skip_redundancy = 0;
}

mtb_status mtb_ctx_create(int device, void *stream, mtb_ctx **out) {
    if (!out) return fail(MTB_ERR_ARG, "out is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(MTB_ERR_DEVICE, "no HIP device available (this library has no CPU path)");
    if (device < 0 || device >= n) return fail(MTB_ERR_ARG, "bad device ordinal");
    HIPCHK(hipSetDevice(device));
    mtb_ctx *c = new mtb_ctx();
    c->device = device; c->stream = (hipStream_t)stream;
    mtb_build_tables(&c->h_tabs);
    HIPCHK(hipMalloc((void **)&c->d_tabs, sizeof(mtb_tables)));
    HIPCHK(hipMemcpy(c->d_tabs, &c->h_tabs, sizeof(mtb_tables), hipMemcpyHostToDevice));
    HIPCHK(hipMalloc((void **)&c->d_scal, 24 * sizeof(uint64_t)));      /* [16..23]: the join's tile statistics */
    c->d_xscal = c->d_scal + 8;
    HIPCHK(hipMalloc((void **)&c->d_ovfctr, MTB_OVF_STRIPES * 64));
    for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&c->ev[i]));
    memset(&c->stats, 0, sizeof(c->stats));
    mtbopt::from_environment(&c->opt);                 /* the ONLY place the library reads its MTB_* switches from the environment */
    *out = c;
    return MTB_OK;
}
void mtb_ctx_destroy(mtb_ctx *c) {
    if (!c) return;
    for (mtb_ctx *l : c->lanes) mtb_ctx_destroy(l);
    c->lanes.clear();
    hipError_t e = hipSetDevice(c->device); (void)e;
    e = hipStreamSynchronize(c->stream);
    for (auto &kv : c->bufs) if (kv.second.p) e = hipFree(kv.second.p);
    if (c->d_tabs && !c->is_lane) e = hipFree(c->d_tabs);
    if (c->d_scal) e = hipFree(c->d_scal);
    if (c->d_ovfctr) e = hipFree(c->d_ovfctr);
    if (c->is_lane && c->stream) e = hipStreamDestroy(c->stream);
    if (c->down_stream) { e = hipStreamSynchronize(c->down_stream); e = hipStreamDestroy(c->down_stream); }
    if (c->down_ready) e = hipEventDestroy(c->down_ready);
    if (c->down_done) e = hipEventDestroy(c->down_done);
    if (c->copy_stream) { e = hipStreamSynchronize(c->copy_stream); e = hipStreamDestroy(c->copy_stream); }
    for (int k = 0; k < 2; k++) { if (c->copy_done[k]) e = hipEventDestroy(c->copy_done[k]); if (c->unpacked[k]) e = hipEventDestroy(c->unpacked[k]); }
    for (int q = 0; q < 4; q++) if (c->join_tunes[q].e0) { e = hipEventDestroy(c->join_tunes[q].e0); e = hipEventDestroy(c->join_tunes[q].e1); }
    for (int i = 0; i < 8; i++) e = hipEventDestroy(c->ev[i]);
    for (auto &k : c->kev) { e = hipEventDestroy(k.a); e = hipEventDestroy(k.b); }
    for (auto &x : c->ev_pool) e = hipEventDestroy(x);
    delete c;
}
mtb_status mtb_ctx_sync(mtb_ctx *c) { HIPCHK(hipStreamSynchronize(c->stream)); return MTB_OK; }
mtb_status mtb_ctx_set_profiling(mtb_ctx *c, int on) {
    if (!c) return fail(MTB_ERR_ARG, "NULL ctx");
    c->profiling = on;
    for (mtb_ctx *l : c->lanes) l->profiling = on;
    return MTB_OK;
}
mtb_status mtb_ctx_set_placement_probe(mtb_ctx *c, int on) {
    if (!c) return fail(MTB_ERR_ARG, "NULL ctx");
    c->placement_probe = on != 0;
    for (mtb_ctx *l : c->lanes) l->placement_probe = on != 0;
    return MTB_OK;
}
mtb_status mtb_ctx_set_workspace_limit(mtb_ctx *c, uint64_t bytes) {
    if (!c) return fail(MTB_ERR_ARG, "NULL ctx");
    c->ws_limit = bytes;
    for (mtb_ctx *l : c->lanes) l->ws_limit = bytes;
    return MTB_OK;
}
mtb_status mtb_ctx_set_option(mtb_ctx *c, const char *name, const char *value) {
    if (!c || !name) return fail(MTB_ERR_ARG, "NULL ctx / name");
    if (!mtbopt::set(&c->opt, name, value)) return fail(MTB_ERR_ARG, std::string("unknown switch or bad value: ") + name + "=" + (value ? value : "(unset)"));
    for (mtb_ctx *l : c->lanes) l->opt = c->opt;
    return MTB_OK;
}
mtb_status mtb_ctx_set_join_variant(mtb_ctx *c, int variant) {
    static const char *const names[] = {"auto", "q1w6", "q2w5", "window"};
    if (!c || variant < 0 || variant > 3) return fail(MTB_ERR_ARG, "variant must be MTB_JOIN_AUTO .. MTB_JOIN_WINDOW");
    return mtb_ctx_set_option(c, "MTB_JOIN_VARIANT", names[variant]);
}
uint32_t mtb_ctx_last_sub_batches(const mtb_ctx *c) { return c ? c->last_sub_batches : 0; }
uint64_t mtb_ctx_last_scratch_bytes(const mtb_ctx *c) { return c ? c->last_scratch_bytes : 0; }
mtb_status mtb_ctx_set_streams(mtb_ctx *c, int n) {
    if (!c || n < 1 || n > 8) return fail(MTB_ERR_ARG, "streams must be 1..8");
    HIPCHK(hipSetDevice(c->device));
    while ((int)c->lanes.size() > (n == 1 ? 0 : n)) { mtb_ctx_destroy(c->lanes.back()); c->lanes.pop_back(); }
    while (n > 1 && (int)c->lanes.size() < n) {
        mtb_ctx *l = new mtb_ctx();
        l->device = c->device; l->is_lane = true; l->d_tabs = c->d_tabs; l->h_tabs = c->h_tabs; l->profiling = c->profiling; l->placement_probe = c->placement_probe; l->opt = c->opt;
        HIPCHK(hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking));
        HIPCHK(hipMalloc((void **)&l->d_scal, 24 * sizeof(uint64_t)));
        l->d_xscal = l->d_scal + 8;
        HIPCHK(hipMalloc((void **)&l->d_ovfctr, MTB_OVF_STRIPES * 64));
        for (int i = 0; i < 8; i++) HIPCHK(hipEventCreate(&l->ev[i]));
        memset(&l->stats, 0, sizeof(l->stats));
        c->lanes.push_back(l);
    }
    return MTB_OK;
}

} // extern "C"

/* ------------------------------------------------------------------ */
/* internal device-side stages                                         */
/* ------------------------------------------------------------------ */
static mtb_status d2h(mtb_ctx *c, void *dst, const void *src, size_t bytes) {
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MTB_OK;
}
static mtb_status h2d(mtb_ctx *c, void *dst, const void *src, size_t bytes) {
    if (bytes == 0) return MTB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MTB_OK;
}

/* fused-path sort: first pass on the top letter pair, the two lower pairs bucket-local (kernels_sort.h); MTB_SORT_LSD=1: the three LSD passes of round 2 (A/B) */
static bool sort_msd_first(const mtb_ctx *c) { return !c->opt.sort_lsd; }

/* records the single-pass extractor's output buffer is sized for: six frames x L/3 windows bound the output by 2 metamers per base;
 * syncmer selection keeps about half of them, so the buffer follows the previous batch's yield (first batch of a context: < 1 metamer
 * per base with s = 5 -- 0.86 measured --, 2 in dense mode) and the extractor reports an overflow if that was too optimistic */
static uint64_t extract_bound(uint64_t n_bases, uint32_t grid) { return 2 * n_bases + (uint64_t)grid * MTB_EXTRACT_CHUNK; }      /* + at most one partly used chunk per wave */
static uint64_t extract_cap_guess(const mtb_ctx *c, const mtb_params *p, uint64_t n_bases, uint32_t grid) {
    const double guess = c->extract_yield > 0.0 ? c->extract_yield : (p->syncmer ? 1.0 : 2.0);
    return std::min<uint64_t>(extract_bound(n_bases, grid), (uint64_t)((double)n_bases * guess * 1.15) + 4096 + (uint64_t)grid * 64);
}

/* extract.  Result in buffer "kmersA".  Two passes (counts -> offsets -> emit) give the reference's emission order
 * (stage API); `single_pass` (fused path, n_bases = bases of the batch) runs the arithmetic once and lets every
 * wave reserve its output with an atomic: run order arbitrary, which the radix sort does not mind. */
static mtb_status dev_extract(mtb_ctx *c, const mtb_params *p, const char *d_bases, const uint64_t *d_offs, const char *d_bases2,
                              const uint64_t *d_offs2, uint64_t n_reads, mtb_kmer **out, uint64_t *count, int32_t *d_qlen,
                              int32_t *d_qlen2, uint32_t *max_len, bool single_pass = false, uint64_t n_bases = 0,
                              uint64_t *real_count = nullptr, bool tag_ord = false, uint32_t *max_q = nullptr, uint16_t **dig = nullptr,
                              uint32_t *d_counts = nullptr /* single pass: metamers per read (long-read slot path) */,
                              uint64_t *n_bases_exact = nullptr /* single pass: bases of the read range, from its offsets */,
                              uint8_t *d_off = nullptr /* tagged single pass: reads the slot records cannot hold are marked here ... */,
                              uint64_t *off_stats = nullptr /* ... [0] most metamers of an unmarked read, [1] marked reads, [2] longest unmarked read */) {
    if (n_reads >= (1ull << 29)) return fail(MTB_ERR_ARG, "more than 2^29-1 reads per batch (sequenceID is 29 bits, Kmer.h:13)");
    if (p->kmer_format != 1 && p->kmer_format != 2) return fail(MTB_ERR_UNSUPPORTED, "only kmer_format 1 and 2 are implemented");
    if (p->syncmer && (p->smer_len < 1 || p->smer_len > 8)) return fail(MTB_ERR_ARG, "smer_len out of range");
    *count = 0; *out = nullptr;
    if (real_count) *real_count = 0;
    if (n_reads == 0) return MTB_OK;
    ExtractArgs a{d_bases, d_offs, d_bases2, d_offs2, n_reads, p->seq_mode, p->syncmer, p->smer_len, p->kmer_format, (single_pass && tag_ord) ? 1 : 0, sort_msd_first(c) ? 54 : 34};
    /* grid sweep, 10 M reads: 7424 workgroups 23.5 ms, 16384 18.7, 65536 17.3, 262144 17.7 (and more blank tail records) */
    uint32_t grid = (uint32_t)std::min<uint64_t>(n_reads, 256ull * 256);
    HIPCHK(hipMemsetAsync(c->d_scal + 4, 0, 8, c->stream));
    if (single_pass) {
        /* exact number of bases of this read range (the caller's figure may be an estimate) */
        uint64_t o[2];
        STCHK(d2h(c, &o[0], d_offs, 8)); STCHK(d2h(c, &o[1], d_offs + n_reads, 8));
        n_bases = o[1] - o[0];
        if (p->seq_mode == 2 && d_offs2) { STCHK(d2h(c, &o[0], d_offs2, 8)); STCHK(d2h(c, &o[1], d_offs2 + n_reads, 8)); n_bases += o[1] - o[0]; }
        if (n_bases_exact) *n_bases_exact = n_bases;
    }
    if (single_pass && n_bases) {
        /* six frames x L/3 windows bound the output by 2 metamers per base; syncmer selection keeps about half of them:
         * size the buffer from the previous batch's yield and fall back to the bound if that was too optimistic */
        const uint64_t bound = extract_bound(n_bases, grid);
        uint64_t cap = extract_cap_guess(c, p, n_bases, grid);
        cap = std::max<uint64_t>(cap, std::min<uint64_t>(bound, c->bufs["kmersA"].cap / sizeof(mtb_kmer)));
        for (int attempt = 0; attempt < 2; attempt++) {
            mtb_kmer *d_k; uint16_t *d_dig = nullptr;
            STCHK(ensure(c, "kmersA", cap, &d_k));
            if (dig) { STCHK(ensure(c, "digA", cap + 8, &d_dig)); *dig = d_dig; }      /* (+8: the bucket-local histograms read whole 16-byte chunks) */
            HIPCHK(hipMemsetAsync(c->d_xscal, 0, 64, c->stream));
            if (d_off) HIPCHK(hipMemsetAsync(d_off, 0, n_reads, c->stream));
            { KTimer kt(c, MTB_K_EXTRACT_EMIT);
            hipLaunchKernelGGL((k_extract<2>), dim3(grid), dim3(64), 0, c->stream, a, c->d_tabs, d_counts, (const uint64_t *)nullptr,
                               d_k, d_qlen, d_qlen2, (uint32_t *)(c->d_scal + 4), (unsigned long long *)c->d_xscal, cap, d_dig, d_off); }
            HIPCHK(hipGetLastError());
            uint64_t sc[8];
            STCHK(d2h(c, sc, c->d_xscal, 64));             /* records allocated (incl. blank tails), overflow, real metamers, most metamers of a read; marked-read statistics */
            if (off_stats) { off_stats[0] = sc[4]; off_stats[1] = sc[5]; off_stats[2] = sc[6]; }
            if (max_len) { uint64_t ml = 0; STCHK(d2h(c, &ml, c->d_scal + 4, 8)); *max_len = (uint32_t)ml; }
            if (sc[1] == 0) {
                if (sc[0] >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 query metamers in one batch; split the batch");
                c->extract_yield = (double)sc[0] / (double)n_bases;
                *out = d_k; *count = sc[0];
                if (real_count) *real_count = sc[2];
                if (max_q) *max_q = (uint32_t)sc[3];
                return MTB_OK;
            }
            if (cap >= bound) return fail(MTB_ERR_DEVICE, "extract: output bound exceeded");
            cap = bound;
        }
    }
    uint32_t *d_cnt; uint64_t *d_koff; uint64_t *d_ws;
    STCHK(ensure(c, "counts", n_reads, &d_cnt));
    STCHK(ensure(c, "koff", n_reads + 1, &d_koff));
    STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    { KTimer kt(c, MTB_K_EXTRACT_COUNT);
    hipLaunchKernelGGL((k_extract<0>), dim3(grid), dim3(64), 0, c->stream, a, c->d_tabs, d_cnt, (const uint64_t *)nullptr,
                       (mtb_kmer *)nullptr, d_qlen, d_qlen2, (uint32_t *)(c->d_scal + 4), (unsigned long long *)nullptr, (uint64_t)0); }
    { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(c->stream, d_cnt, n_reads, true, d_koff, d_ws); }
    uint64_t total = 0;
    STCHK(d2h(c, &total, d_koff + n_reads, 8));
    if (max_len) { uint64_t ml = 0; STCHK(d2h(c, &ml, c->d_scal + 4, 8)); *max_len = (uint32_t)ml; }
    if (total >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 query metamers in one batch; split the batch");
    mtb_kmer *d_k;
    STCHK(ensure(c, "kmersA", total, &d_k));
    if (total) { KTimer kt(c, MTB_K_EXTRACT_EMIT);
        hipLaunchKernelGGL((k_extract<1>), dim3(grid), dim3(64), 0, c->stream, a, c->d_tabs, (uint32_t *)nullptr,
                           (const uint64_t *)d_koff, d_k, (int32_t *)nullptr, (int32_t *)nullptr, (uint32_t *)nullptr,
                           (unsigned long long *)nullptr, (uint64_t)0); }
    HIPCHK(hipGetLastError());
    *out = d_k; *count = total;
    if (real_count) *real_count = total;
    return MTB_OK;
}

/* first_bit >= 0: binary LSD passes over bits [first_bit, 64).  first_bit == MTB_SORT_AA6 (kmer_format 2): three
 * passes on amino-acid letter pairs = order by bits [34, 64) (kernels_sort.h). */
#define MTB_SORT_AA6 (-6)
static mtb_status dev_sort(mtb_ctx *c, mtb_kmer *d_a, uint64_t n, int first_bit, mtb_kmer **sorted, uint16_t *d_dig = nullptr, int aa_first_shift = 34) {
    *sorted = d_a;
    if (n == 0) return MTB_OK;
    const bool aa = first_bit == MTB_SORT_AA6;
    const uint32_t bins = aa ? 512u : 256u;
    mtb_kmer *d_b; uint32_t *d_hist; uint64_t *d_ws;
    STCHK(ensure(c, "kmersB", n, &d_b));
    STCHK(ensure(c, "hist", radix_hist_elems(n, bins), &d_hist));
    STCHK(ensure(c, "scanws", scan_ws_elems(radix_hist_elems(n, bins)), &d_ws));
    {
        const int xcd_map = c->opt.sort_no_xcd ? 0 : 1;
        const uint32_t sc_threads = aa ? 512u : 256u;
        const uint64_t tile = (uint64_t)sc_threads * MTB_SORT_ITEMS;  /* MTB_SORT_ITEMS records per scatter thread */
        uint32_t tiles = (uint32_t)((n + tile - 1) / tile);
        mtb_kmer *src = d_a, *dst = d_b;
        /* AA6 with a digit side array: every histogram reads 2-byte digits (the extractor wrote the first pass's, each
         * scatter writes the next pass's in output order) instead of the 16-byte records */
        uint16_t *dig_src = aa ? d_dig : nullptr, *dig_dst = nullptr;
        if (dig_src) STCHK(ensure(c, "digB", n + 8, &dig_dst));
        /* the directory join (kernels_dir.h) looks every query up on its own: the sort only buys locality of the directory /
         * target accesses, so the fused path may stop after fewer letter pairs (aa_first_shift: 34 = six letters, 44 = four, 54 = two) */
        if (aa && dig_src && aa_first_shift == 34 && sort_msd_first(c)) {
            /* pass A: the top letter pair, over the whole list (its digits came with the records); passes B, C: the low and the middle
             * pair inside every bucket of pass A */
            uint32_t *d_plan;
            STCHK(ensure(c, "sortplan", 1026, &d_plan));
            const uint32_t seg_tiles_max = tiles + MTB_SORT_NBKT;
            STCHK(ensure(c, "hist", (uint64_t)bins * seg_tiles_max, &d_hist));
            STCHK(ensure(c, "scanws", scan_ws_elems((uint64_t)bins * seg_tiles_max), &d_ws));
            { KTimer kt(c, MTB_K_RADIX_HIST);
              hipLaunchKernelGGL((k_radix_hist_dig<512, 512>), dim3((tiles + MTB_HIST_GROUP - 1) / MTB_HIST_GROUP), dim3(512), 0, c->stream, (const uint16_t *)dig_src, n, d_hist, tiles); }
            { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint32_t, false>(c->stream, d_hist, (uint64_t)bins * tiles, false, d_hist, (uint32_t *)d_ws); }
            hipLaunchKernelGGL(k_sort_plan, dim3(1), dim3(512), 0, c->stream, (const uint32_t *)d_hist, tiles, n, (uint32_t)tile, d_plan);
            { KTimer kt(c, MTB_K_RADIX_SCATTER);
              hipLaunchKernelGGL((k_radix_scatter<512, 1, 512>), dim3((tiles + 7u) / 8u * 8u), dim3(512), 0, c->stream, (const mtb_kmer *)src, dst, n, 54, (const uint32_t *)d_hist, tiles, dig_dst, 34, xcd_map); }
            { mtb_kmer *tmp = src; src = dst; dst = tmp; uint16_t *t2 = dig_src; dig_src = dig_dst; dig_dst = t2; }
            for (int shift = 34; shift <= 44; shift += 10) {
                const dim3 hg((seg_tiles_max + MTB_HIST_GROUP - 1) / MTB_HIST_GROUP), sg((seg_tiles_max + 7u) / 8u * 8u);
                { KTimer kt(c, MTB_K_RADIX_HIST);
                  hipLaunchKernelGGL((k_radix_hist_seg<512, 512>), hg, dim3(512), 0, c->stream, (const uint16_t *)dig_src, (const uint32_t *)d_plan, d_hist); }
                /* (the table's length is on the device: the scan covers the most it can be; entries behind the last tile are never read) */
                { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint32_t, false>(c->stream, d_hist, (uint64_t)bins * seg_tiles_max, false, d_hist, (uint32_t *)d_ws); }
                { KTimer kt(c, MTB_K_RADIX_SCATTER);
                  hipLaunchKernelGGL((k_radix_scatter<512, 1, 512>), sg, dim3(512), 0, c->stream, (const mtb_kmer *)src, dst, n, shift, (const uint32_t *)d_hist, tiles,
                                     shift == 34 ? dig_dst : (uint16_t *)nullptr, 44, xcd_map, (const uint32_t *)d_plan); }
                { mtb_kmer *tmp = src; src = dst; dst = tmp; uint16_t *t2 = dig_src; dig_src = dig_dst; dig_dst = t2; }
            }
            *sorted = src;
            HIPCHK(hipGetLastError());
            return MTB_OK;
        }
        for (int shift = aa ? aa_first_shift : first_bit; shift < 64; shift += aa ? 10 : 8) {
            { KTimer kt(c, MTB_K_RADIX_HIST);
              const dim3 hg((tiles + MTB_HIST_GROUP - 1) / MTB_HIST_GROUP);
              if (aa && dig_src) hipLaunchKernelGGL((k_radix_hist_dig<512, 512>), hg, dim3(512), 0, c->stream, (const uint16_t *)dig_src, n, d_hist, tiles);
              else if (aa) hipLaunchKernelGGL((k_radix_hist<512, 1, 512>), dim3(tiles), dim3(512), 0, c->stream, (const mtb_kmer *)src, n, shift, d_hist, tiles);
              else hipLaunchKernelGGL((k_radix_hist<256, 0, 256>), dim3(tiles), dim3(256), 0, c->stream, (const mtb_kmer *)src, n, shift, d_hist, tiles); }
            { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint32_t, false>(c->stream, d_hist, (uint64_t)bins * tiles, false, d_hist, (uint32_t *)d_ws); }
            { KTimer kt(c, MTB_K_RADIX_SCATTER);
              const dim3 sg((tiles + 7u) / 8u * 8u);
              uint16_t *dnext = (dig_src && shift + 10 < 64) ? dig_dst : (uint16_t *)nullptr;
              if (aa) hipLaunchKernelGGL((k_radix_scatter<512, 1, 512>), sg, dim3(512), 0, c->stream, (const mtb_kmer *)src, dst, n, shift, (const uint32_t *)d_hist, tiles, dnext, shift + 10, xcd_map);
              else hipLaunchKernelGGL((k_radix_scatter<256, 0, 256>), sg, dim3(256), 0, c->stream, (const mtb_kmer *)src, dst, n, shift, (const uint32_t *)d_hist, tiles); }
            mtb_kmer *tmp = src; src = dst; dst = tmp;
            if (dig_src) { uint16_t *t2 = dig_src; dig_src = dig_dst; dig_dst = t2; }
        }
        *sorted = src;
    }
    HIPCHK(hipGetLastError());
    return MTB_OK;
}

struct IndexUse;
static mtb_dir_view dir_view(const mtb_index *ix);
static mtb_index_view index_view(const mtb_index *ix) {
    mtb_index_view v;
    v.values = ix->d_values; v.info = ix->d_info; v.n_targets = ix->T; v.tax2species = ix->d_tax2species;
    v.max_taxid = ix->tax.max_id; v.info_mask = ix->info_mask; v.kmer_format = ix->params.kmer_format;
    return v;
}
static mtb_tax_view tax_view(const mtb_index *ix) {
    mtb_tax_view v;
    v.acc_leaf = ix->d_accleaf; v.canon = ix->d_canon; v.parent = ix->d_parent; v.depth = ix->d_depth; v.under_euk = ix->d_under; v.sp_parent = ix->d_spparent;
    v.max_taxid = ix->tax.max_id;
    v.node = ix->d_node;
    return v;
}

static mtb_status ensure_flat_locked(mtb_index *ix) {       /* caller holds state_mu and has seen users == 0 */
    if (!ix || !ix->packed) return MTB_OK;
    mtb_ctx *c = ix->ctx;
    if (!ix->d_info) {                                  /* sealed: info[] was let go of; the flat state needs it back */
        hipError_t e = hipMalloc((void **)&ix->d_info, std::max<uint64_t>(ix->T, 1) * 4);
        if (e != hipSuccess) { ix->d_info = nullptr; (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no HBM to restore info[] of a sealed index"); }
        ix->info_owned = true;
    }
    hipLaunchKernelGGL(k_index_unpack, dim3((uint32_t)std::min<uint64_t>(((uint64_t)ix->dir_buckets + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, ix->d_values, ix->d_info, dir_view(ix));
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    ix->packed = false;
    return MTB_OK;
}
static bool can_pack(const mtb_index *ix) {
    return ix->d_dir && ix->dir_L == 7 && ix->own_tax && ix->views == 0 && !(ix->ctx && ix->ctx->opt.no_pack);
}
static mtb_status ensure_packed_locked(mtb_index *ix) {
    if (ix->packed || !can_pack(ix)) return MTB_OK;
    mtb_ctx *c = ix->ctx;
    hipLaunchKernelGGL(k_index_pack, dim3((uint32_t)std::min<uint64_t>((ix->T + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, ix->d_values, (const uint32_t *)ix->d_info, ix->T, ix->params.kmer_format);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));            /* complete before any other stream's join may read the array */
    ix->packed = true;
    return MTB_OK;
}
/* A view's arrays are its parent's: the state (and the lock) live there. */
static mtb_index *state_owner(mtb_index *ix) { return ix && ix->parent ? ix->parent : ix; }
/* Exclusive change of state for the synchronous entry points (download, write, seal, slice): waits for the joins in flight. */
static mtb_status ensure_flat(mtb_index *ix) {
    mtb_index *o = state_owner(ix);
    if (!o) return MTB_OK;
    std::unique_lock<std::mutex> lk(o->state_mu);
    o->state_cv.wait(lk, [&] { return o->users == 0; });
    return ensure_flat_locked(o);
}
static mtb_status ensure_packed(mtb_index *ix) {
    mtb_index *o = state_owner(ix);
    std::unique_lock<std::mutex> lk(o->state_mu);
    o->state_cv.wait(lk, [&] { return o->users == 0; });
    return ensure_packed_locked(o);
}
/* A join's hold on the state it was launched for: taken before the launch, released after the launching stream has been
 * synchronised.  Several lanes (mtb_ctx_set_streams) or contexts may hold the same state at once; a lane that needs the other
 * state waits until the holders are gone, converts, and holds in turn. */
struct IndexUse {
    mtb_index *o = nullptr;
    mtb_status acquire(mtb_index *ix, bool want_packed) {
        mtb_index *own = state_owner(ix);
        std::unique_lock<std::mutex> lk(own->state_mu);
        const bool target = want_packed && can_pack(own);
        own->state_cv.wait(lk, [&] { return own->users == 0 || own->packed == target; });
        if (own->packed != target) { mtb_status st = target ? ensure_packed_locked(own) : ensure_flat_locked(own); if (st != MTB_OK) return st; }
        own->users++; o = own;
        return MTB_OK;
    }
    void release() { if (!o) return; { std::lock_guard<std::mutex> lk(o->state_mu); o->users--; } o->state_cv.notify_all(); o = nullptr; }
    ~IndexUse() { release(); }
};

/* join into d_out (cap entries); *count = matches found (may exceed cap -> MTB_ERR_CAPACITY).
 * With `seg` (per-read slot segments, k_join<SEG>) matches go to seg->seg and the overflow list instead; *count is then
 * the number of overflow entries needed and MTB_ERR_CAPACITY refers to the overflow list.                     */
static mtb_status dev_join(mtb_ctx *c, mtb_index *ix, const mtb_kmer *d_q, uint64_t n, mtb_match *d_out, uint64_t cap,
                           uint32_t *d_read_cnt, uint64_t *count, const JoinSegArgs *seg = nullptr, int sort_low_bits = 32) {
    *count = 0;
    if (n == 0) return MTB_OK;
    HIPCHK(hipMemsetAsync(c->d_scal, 0, 16, c->stream));
    uint32_t grid = (uint32_t)((n + MTB_JOIN_QPB - 1) / MTB_JOIN_QPB);
    uint64_t *d_bounds;
    STCHK(ensure(c, "jbounds", 2ull * grid, &d_bounds));
    uint64_t limit = ix->T ? ix->T - (ix->match_last ? 0 : 1) : 0;          /* the last entry of the (whole) index is never a candidate */
    IndexUse use;                     /* released after the stream sync below (the d2h of the counters) */
    uint32_t win_tiles = 0;           /* the window variant ran: tiles launched (their statistics sit in d_scal[16..17]) */
    const bool striped = seg && ix->d_dir && !seg->list && !seg->dense_ovf;       /* the slot modes of the directory join: striped overflow list */
    if (seg && ix->d_dir) {
        STCHK(use.acquire(ix, true));
        KTimer kt(c, MTB_K_JOIN);
        JoinSegArgs sa = *seg; sa.ovf_counter = (unsigned long long *)c->d_scal; sa.coop_min = c->opt.join_coop_min > 0 ? (uint32_t)c->opt.join_coop_min : (uint32_t)MTB_JOIN_COOP_MIN;
        if (striped) {
            HIPCHK(hipMemsetAsync(c->d_ovfctr, 0, MTB_OVF_STRIPES * 64, c->stream));
            sa.ovf_counter = c->d_ovfctr; sa.ovf_stripes = MTB_OVF_STRIPES; sa.ovf_region = sa.ovf_cap / MTB_OVF_STRIPES;
            c->ovf_region = sa.ovf_region;
        }
        const uint32_t g2 = (uint32_t)((n + 256 * MTB_JOIN_DIR_QPT - 1) / (256 * MTB_JOIN_DIR_QPT));
        if (sa.list) {      /* owner side of the partitioned index: a dense list of Match records */
            if (state_owner(ix)->packed) hipLaunchKernelGGL((k_join_dir<true, 2>), dim3(g2), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
            else hipLaunchKernelGGL((k_join_dir<false, 2>), dim3(g2), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
        } else
        if (sa.rb) {        /* long reads: per-read slot ranges */
            if (state_owner(ix)->packed) {
                /* one query per thread at 6 waves per SIMD here too (43.2 -> 41.5 ms per 50 k x 10 kb; MTB_JOIN_VARIANT=q2w5: round 4's instantiation, A/B) */
                /* the window form for long reads (round 6): the same tiles and windows, where the (sub-)batch is dense enough (100 k x 10 kb = 0.86 G metamers
                 * against 16 G targets: 18.6 targets per query) */
                const double per_q = (double)ix->T / (double)std::max<uint64_t>(n, 1);
                uint32_t qt = 256;                   /* full tiles: measured at 18.6 targets per query (200 k x 10 kb in two sub-batches): 256 queries per tile 124.3 ms, 175 (the density rule
                                                      * of the short reads) 127.0, 128: 143.5, sector-random q1w6 146.3 -- tiles beyond the capacity read global memory at eight waves per SIMD */
                if (c->opt.join_win_qt > 0) qt = (uint32_t)std::min(256, c->opt.join_win_qt);
                const bool sorted_ok = ix->params.kmer_format == 2 ? sort_low_bits == 34 : (sort_low_bits >= 24 && sort_low_bits <= 32);
                const bool lwin = sorted_ok && (c->opt.join_variant == 0x100 || c->opt.join_win == 1 || (c->opt.join_variant == 0 && c->opt.join_win < 0 && per_q <= MTB_LONG_WIN_MAX_PER_Q));
                if (lwin) {
                    const uint32_t n_tiles = (uint32_t)((n + qt - 1) / qt);
                    mtb_tile_win *d_tw;
                    STCHK(ensure(c, "jtilewin", (size_t)n_tiles, &d_tw));
                    HIPCHK(hipMemsetAsync(c->d_scal + 16, 0, 16, c->stream));
                    hipLaunchKernelGGL(k_join_tile_win, dim3((n_tiles + 255) / 256), dim3(256), 0, c->stream, d_q, n, qt, dir_view(ix), limit, sort_low_bits, d_tw, n_tiles, (unsigned long long *)(c->d_scal + 16));
                    hipLaunchKernelGGL((k_join_dir<true, 1, 1, MTB_JOIN_WIN_WAVES, true>), dim3(n_tiles), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa,
                                       (uint32_t *)(c->d_scal + 1), qt, (const mtb_tile_win *)d_tw, (unsigned long long *)(c->d_scal + 16));
                    win_tiles = n_tiles; c->stats.join_variant = MTB_JOIN_WINDOW;
                } else
                if (c->opt.join_variant != 0x25) hipLaunchKernelGGL((k_join_dir<true, 1, 1, 6>), dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
                else hipLaunchKernelGGL((k_join_dir<true, 1, MTB_JOIN_DIR_QPT, 5>), dim3(g2), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
            }
            else hipLaunchKernelGGL((k_join_dir<false, 1>), dim3(g2), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
        } else
        if (state_owner(ix)->packed) {
            /* the product instantiation, or one of its A/B variants (queries per thread x waves per SIMD) */
#define MTB_LAUNCH_JV(QV, WV) hipLaunchKernelGGL((k_join_dir<true, 0, QV, WV>), dim3((uint32_t)((n + 256 * QV - 1) / (256 * QV))), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1))
            /* Three exact instantiations of the short-read join on packed words (kernels_dir.h): sector-random lookups with one query per thread at 6 waves
             * per SIMD (q1w6) or two at 5 (q2w5), and the LDS window.  The window needs a batch dense enough that a tile of sorted queries addresses a
             * span of the target array that fits LDS -- a query owns T / n targets on average; qt queries per workgroup so that the expected window is
             * ~0.82 of the capacity; below 240 queries per workgroup (T / n > 13.5: smaller batches against a big index) idle lanes cost more than the
             * window saves (r05_notes.md).  Which one is fastest depends on the batch's locality, which the host cannot see: reads of a few genomes at
             * high coverage run best with two queries per thread, a metagenome of thousands of genomes with one, a dense one with the window.  So a
             * context TRIES them on its first batches of a shape (second to fourth join: one instantiation each, timed with a pair of events) and
             * keeps the fastest; a few (index, batch size) shapes are remembered.  The caller can pin the choice (mtb_ctx_set_join_variant, or
             * MTB_JOIN_VARIANT / MTB_JOIN_WIN in the environment of mtb_ctx_create); the choice and the tuner's timings are reported in
             * mtb_batch_stats (join_variant, join_tuned, join_tune_ms). */
            const double per_q = (double)ix->T / (double)std::max<uint64_t>(n, 1);
            /* FULL tiles (256 queries) whatever the density: a tile whose span exceeds the window reads global memory at eight waves per SIMD, which beat smaller
             * tiles with idle lanes everywhere it was measured (2 M pairs, 31 targets per query: 256 per tile 32.4 ms, 105 per tile 38.5, q1w6 37.0; long reads:
             * 124.3 vs 127.0 at 175).  Beyond ~48 targets per query hardly a tile has a window (2 M held-out reads, 62 per query: 48.6 vs 47.1 for q1w6): the
             * window form is not a candidate there */
            uint32_t qt = 256;
            const bool win_ok = per_q <= 48.0;
            if (c->opt.join_win_qt > 0) qt = (uint32_t)std::min(256, c->opt.join_win_qt);
            qt = std::max<uint32_t>(qt, 1);
            int choice = -1;                                 /* 0: q1w6, 1: q2w5, 2: window, 3 / 4: the A/B-only instantiations q1w5 / q2w6 */
            int win_waves = MTB_JOIN_WIN_WAVES;              /* the window form's instantiation: waves per SIMD (5..7: A/B only) */
            switch (c->opt.join_variant) { case 0x16: choice = 0; break; case 0x25: choice = 1; break; case 0x100: choice = 2; break; case 0x15: choice = 3; break; case 0x26: choice = 4; break;
                                           case 0x205: case 0x206: case 0x207: choice = 2; win_waves = c->opt.join_variant & 15; break; default: break; }
            if (choice < 0 && c->opt.join_win >= 0) choice = c->opt.join_win ? 2 : 0;
            const bool forced = choice >= 0;
            uint32_t lg_n = 0; for (uint64_t x = n; x > 1; x >>= 1) lg_n++;
            const uint64_t tune_key = (uint64_t)(uintptr_t)ix ^ ((uint64_t)lg_n << 56) ^ (ix->T << 8);
            int slot = -1;
            for (int q = 0; q < 4; q++) if (c->join_tunes[q].key == tune_key) slot = q;
            mtb_ctx::JoinTune *jt = slot >= 0 ? &c->join_tunes[slot] : nullptr;
            bool timing_this = false; int tuned_flag = 0;
            if (!forced && !c->is_lane && !sa.retry) {
                if (!jt) { slot = (int)(c->join_tune_next++ & 3u); jt = &c->join_tunes[slot]; const hipEvent_t k0 = jt->e0, k1 = jt->e1; *jt = mtb_ctx::JoinTune(); jt->key = tune_key; jt->e0 = k0; jt->e1 = k1; }
                if (jt->pending >= 0) {                      /* the previous join's time (that batch is long finished) */
                    float ms = 0.0f;
                    if (hipEventElapsedTime(&ms, jt->e0, jt->e1) == hipSuccess && ms > 0.0f) jt->ms[jt->pending] = ms; else { (void)hipGetLastError(); jt->ms[jt->pending] = 1e30f; }
                    jt->pending = -1;
                }
                jt->calls++;
                if (jt->best < 0 && jt->calls >= 2) {
                    int next = -1;
                    for (int v = 0; v < 3; v++) if (jt->ms[v] == 0.0f && (v != 2 || win_ok)) { next = v; break; }
                    if (next >= 0) {
                        if (!jt->e0) { HIPCHK(hipEventCreate(&jt->e0)); HIPCHK(hipEventCreate(&jt->e1)); }
                        choice = next; jt->pending = next; timing_this = true;
                        HIPCHK(hipEventRecord(jt->e0, c->stream));
                    } else {
                        jt->best = 0;
                        for (int v = 1; v < 3; v++) if (jt->ms[v] > 0.0f && jt->ms[v] < jt->ms[jt->best]) jt->best = v;
                        if (c->opt.join_verbose) fprintf(stderr, "mtb: join tuned for this index and batch size: q1w6 %.2f ms, q2w5 %.2f ms, window %.2f ms -> %s\n", jt->ms[0], jt->ms[1], jt->ms[2],
                                                         jt->best == 0 ? "q1w6" : jt->best == 1 ? "q2w5" : "window");
                    }
                }
            }
            /* a remembered choice serves every batch of its log2-size bucket: the window only where THIS batch is dense enough for it;
             * retries of a batch (overflow list too small) and lanes never time anything, they follow what is known */
            if (choice < 0 && jt && jt->best >= 0) { choice = (jt->best == 2 && !win_ok) ? 0 : jt->best; tuned_flag = 1; }
            if (choice < 0) choice = win_ok ? 2 : 0;         /* not tuned (yet): by density */
            c->stats.join_variant = choice == 0 ? MTB_JOIN_Q1W6 : choice == 1 ? MTB_JOIN_Q2W5 : choice == 2 ? (win_waves != MTB_JOIN_WIN_WAVES ? -(10 + win_waves) : MTB_JOIN_WINDOW) : -choice;
            c->stats.join_tuned = tuned_flag;
            if (jt) for (int v = 0; v < 3; v++) c->stats.join_tune_ms[v] = jt->ms[v] < 1e29f ? jt->ms[v] : 0.0f;
            if (choice == 2) {
                /* the tiles' windows first (k_join_tile_win): needs the list sorted on the announced bits (else every tile reads global memory) */
                const uint32_t n_tiles = (uint32_t)((n + qt - 1) / qt);
                mtb_tile_win *d_tw;
                const bool sorted_ok = ix->params.kmer_format == 2 ? sort_low_bits == 34 : (sort_low_bits >= 24 && sort_low_bits <= 32);
                STCHK(ensure(c, "jtilewin", (size_t)n_tiles, &d_tw));
                HIPCHK(hipMemsetAsync(c->d_scal + 16, 0, 16, c->stream));
                if (sorted_ok) hipLaunchKernelGGL(k_join_tile_win, dim3((n_tiles + 255) / 256), dim3(256), 0, c->stream, d_q, n, qt, dir_view(ix), limit, sort_low_bits, d_tw, n_tiles,
                                                  (unsigned long long *)(c->d_scal + 16));
                else HIPCHK(hipMemsetAsync(d_tw, 0, (size_t)n_tiles * sizeof(mtb_tile_win), c->stream));
#define MTB_LAUNCH_JW(WV) hipLaunchKernelGGL((k_join_dir<true, 0, 1, WV, true>), dim3(n_tiles), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), \
                                   (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1), qt, (const mtb_tile_win *)d_tw, (unsigned long long *)(c->d_scal + 16))
                switch (win_waves) {
                case 5: MTB_LAUNCH_JW(5); break;
                case 6: MTB_LAUNCH_JW(6); break;
                case 7: MTB_LAUNCH_JW(7); break;
                default: MTB_LAUNCH_JW(MTB_JOIN_WIN_WAVES); break;
                }
#undef MTB_LAUNCH_JW
                win_tiles = n_tiles;
            } else
            switch (choice) {
            case 1: MTB_LAUNCH_JV(2, 5); break;
            case 3: MTB_LAUNCH_JV(1, 5); break;
            case 4: MTB_LAUNCH_JV(2, 6); break;
            default: MTB_LAUNCH_JV(1, 6); break;
            }
#undef MTB_LAUNCH_JV
            if (timing_this) HIPCHK(hipEventRecord(jt->e1, c->stream));
        }
        else hipLaunchKernelGGL((k_join_dir<false>), dim3(g2), dim3(256), 0, c->stream, d_q, n, index_view(ix), limit, dir_view(ix), (const mtb_tables *)c->d_tabs, sa, (uint32_t *)(c->d_scal + 1));
    } else
    { STCHK(use.acquire(ix, false));
    KTimer kt(c, MTB_K_JOIN);
    hipLaunchKernelGGL(k_join_bounds, dim3((grid + 255) / 256), dim3(256), 0, c->stream, d_q, n, (const uint64_t *)ix->d_values, limit,
                       (uint64_t)grid, d_bounds, sort_low_bits);
    if (seg) {
        JoinSegArgs sa = *seg; sa.ovf_counter = (unsigned long long *)c->d_scal;
        hipLaunchKernelGGL((k_join<true>), dim3(grid), dim3(256), 0, c->stream, d_q, n, index_view(ix), (const mtb_tables *)c->d_tabs,
                           (const uint64_t *)d_bounds, (mtb_match *)nullptr, (uint64_t)0, (unsigned long long *)nullptr, (uint32_t *)nullptr,
                           (uint32_t *)(c->d_scal + 1), sa);
    } else {
        JoinSegArgs sa; memset(&sa, 0, sizeof(sa));
        hipLaunchKernelGGL((k_join<false>), dim3(grid), dim3(256), 0, c->stream, d_q, n, index_view(ix), (const mtb_tables *)c->d_tabs,
                           (const uint64_t *)d_bounds, d_out, cap, (unsigned long long *)c->d_scal, d_read_cnt, (uint32_t *)(c->d_scal + 1), sa);
    } }
    HIPCHK(hipGetLastError());
    uint64_t sc[2];
    STCHK(d2h(c, sc, c->d_scal, 16));
    if (win_tiles) {
        uint64_t ws[2];
        STCHK(d2h(c, ws, c->d_scal + 16, 16));
        c->stats.join_tiles = win_tiles; c->stats.join_tiles_windowed = (uint32_t)ws[0]; c->stats.join_tiles_outside = (uint32_t)ws[1];
    }
    if (striped) {
        /* entries per stripe: the list fits when the fullest region does; else the caller comes again with room for 256 x that */
        unsigned long long h[MTB_OVF_STRIPES * 8];
        STCHK(d2h(c, h, c->d_ovfctr, sizeof(h)));
        uint64_t tot = 0, mx = 0;
        for (uint32_t k = 0; k < MTB_OVF_STRIPES; k++) { tot += h[8 * k]; mx = std::max<uint64_t>(mx, h[8 * k]); }
        c->ovf_max_region = mx;
        if (seg->rb) {                                 /* long reads: counted only */
            *count = tot;
            if (tot > seg->ovf_cap) return fail(MTB_ERR_CAPACITY, "match buffer too small");
            return MTB_OK;
        }
        *count = mx > c->ovf_region ? (mx + mx / 8 + 64) * MTB_OVF_STRIPES : tot;          /* (>= tot: also enough for a dense list) */
        if (tot >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 matches in one batch; split the batch");
        if (mx > c->ovf_region) return fail(MTB_ERR_CAPACITY, "match buffer too small");
        return MTB_OK;
    }
    c->ovf_region = 0;
    *count = sc[0];
    if (sc[0] >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 matches in one batch; split the batch");
    if (sc[0] > (seg ? seg->ovf_cap : cap)) return fail(MTB_ERR_CAPACITY, "match buffer too small");
    return MTB_OK;
}

/* per-read counters -> seg_start (n_reads+1) -> regroup into d_out */
template <typename REC>
static mtb_status dev_regroup(mtb_ctx *c, const mtb_match *d_in, uint64_t m, uint64_t n_reads, uint32_t *d_read_cnt,
                              uint64_t **seg_start, REC *d_out) {
    uint64_t *d_seg; uint64_t *d_ws; uint32_t *d_cur;
    STCHK(ensure(c, "segstart", n_reads + 1, &d_seg));
    STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    STCHK(ensure(c, "cursor", n_reads, &d_cur));
    { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(c->stream, d_read_cnt, n_reads, true, d_seg, d_ws); }
    HIPCHK(hipMemsetAsync(d_cur, 0, n_reads * 4, c->stream));
    if (m) { KTimer kt(c, MTB_K_REGROUP); hipLaunchKernelGGL((k_regroup<REC>), dim3((uint32_t)((m + 255) / 256)), dim3(256), 0, c->stream, d_in, m, (const uint64_t *)d_seg, d_cur, d_out); }
    HIPCHK(hipGetLastError());
    *seg_start = d_seg;
    return MTB_OK;
}

static mtb_status dev_segsort(mtb_ctx *c, mtb_match *d_m, const uint64_t *d_seg, uint64_t n_reads, uint32_t *max_seg) {
    uint32_t *d_large;
    STCHK(ensure(c, "large", n_reads, &d_large));
    HIPCHK(hipMemsetAsync(c->d_scal + 2, 0, 16, c->stream));
    uint32_t grid = (uint32_t)std::min<uint64_t>(n_reads, 256ull * 40);
    { KTimer kt(c, MTB_K_SEGSORT);
    hipLaunchKernelGGL(k_segsort_small, dim3(grid), dim3(64), 0, c->stream, d_m, d_seg, n_reads, d_large, (uint32_t *)(c->d_scal + 2),
                       (uint32_t *)(c->d_scal + 3)); }
    hipLaunchKernelGGL((k_segsort_large<mtb_match>), dim3(1024), dim3(256), 0, c->stream, d_m, d_seg, (const uint32_t *)d_large,
                       (const uint32_t *)(c->d_scal + 2));
    HIPCHK(hipGetLastError());
    if (max_seg) { uint64_t sc[2]; STCHK(d2h(c, sc, c->d_scal + 2, 16)); *max_seg = (uint32_t)sc[1]; }
    return MTB_OK;
}

/* where a scoring launch finds its segments */
struct ScoreSrc {
    const mtb_match *m = nullptr;
    const uint64_t *seg = nullptr;        /* seg_start (by read, or by list slot if seg_by_list)            */
    const uint32_t *cursor = nullptr;     /* slot mode (k_join<SEG>): read r owns m[r*stride .. +stride), cursor[r] = entries in its tail */
    uint32_t stride = 0, direct = 0, epoch = 0;
    uint32_t *big_list = nullptr, *n_big = nullptr;      /* slot mode: reads deferred to the large-segment path */
    uint32_t *cnt_out = nullptr;                         /* slot mode: live records per read */
    uint32_t cap = MTB_SCORE_LDS;                        /* matches staged in LDS per read: 160, or 320 (read pairs on the slot path) */
    const uint32_t *list = nullptr, *n_list = nullptr;   /* only these reads */
    int seg_by_list = 0;
    bool sort = false;                    /* segments arrive unordered: rank sort in the kernel */
    uint32_t max_seg = 0;                 /* largest segment this launch can meet */
    uint32_t grid = 0;                    /* 0 = default */
    const uint8_t *only_flagged = nullptr; /* slot mode after k_score_fast: score only the reads it flagged */
    const uint32_t *seg_cnt = nullptr;     /* slab launches: matches per read when the segment does not fill [seg[r], seg[r + 1]) */
    const void *bound_by_buckets = nullptr; /* non-NULL: every read gets one taxcnt slot per position bucket (its match count is not in seg[]) */
};

/* d_results/d_tc_* are device outputs; *n_tc = sum of per-read bounds.  `second` (optional) runs after the launch
 * over `first`, may build a source for the reads that launch deferred (large-segment path) and returns true to
 * have it launched with the same per-read taxcnt slots. */
struct ScoreLaunch {                      /* what dev_score has set up by the time `second` runs: enough to launch a scorer of its own */
    const uint64_t *d_tcoff; bool key64; mtb_score_params sp;
    const int32_t *d_qlen, *d_qlen2; mtb_result *d_res; int32_t *d_tc_tax; uint32_t *d_tc_cnt; uint64_t tc_cap, tc_base;
};
typedef std::function<mtb_status(ScoreSrc *, bool *, const ScoreLaunch &)> ScoreSecond;
static mtb_status dev_score(mtb_ctx *c, mtb_index *ix, const mtb_params *p, uint64_t n_reads, const int32_t *d_qlen, const int32_t *d_qlen2,
                            uint32_t max_len, mtb_result *d_res, int32_t *d_tc_tax, uint32_t *d_tc_cnt, uint64_t tc_cap, uint64_t *n_tc,
                            uint64_t tc_base, const ScoreSrc &first, const ScoreSecond *second,
                            uint32_t max_len_second = 0 /* the deferred reads' launch may meet longer reads than the first one (reads routed around the slot segments) */) {
    mtb_score_params sp; mtb_make_score_params(p, &sp);
    uint32_t *d_bound; uint64_t *d_tcoff; uint64_t *d_ws;
    STCHK(ensure(c, "bound", n_reads, &d_bound));
    STCHK(ensure(c, "tcoff", n_reads + 1, &d_tcoff));
    STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    hipLaunchKernelGGL(k_taxcnt_bound, dim3((uint32_t)((n_reads + 63) / 64)), dim3(64), 0, c->stream, first.seg, first.cursor ? first.cursor : (const uint32_t *)first.bound_by_buckets,
                       d_qlen, d_qlen2, n_reads, sp.dna_shift, d_bound);
    { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(c->stream, d_bound, n_reads, true, d_tcoff, d_ws); }
    uint64_t tot = 0;
    STCHK(d2h(c, &tot, d_tcoff + n_reads, 8));
    *n_tc = tot;
    if (tot > tc_cap) return fail(MTB_ERR_CAPACITY, "taxcnt buffers too small");
    if (tc_base + tot >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 taxcnt slots in one batch (mtb_result.taxcnt_off is 32 bits); split the batch");
    /* single-word sort key when taxids < 2^22 and positions < 2^11 (hamming of a match is <= 7) */
    const bool key64 = ix->tax.max_id < (1 << 22) && max_len + 3 < (1u << 11);
    const uint32_t max_nb_first = (uint32_t)mtb_num_buckets((int32_t)max_len, sp.dna_shift);
    const uint32_t max_nb_second = (uint32_t)mtb_num_buckets((int32_t)std::max(max_len, max_len_second), sp.dna_shift);
    if (max_nb_second > 65535u) return fail(MTB_ERR_ARG, "read too long for mtb_result.n_taxcnt (16 bits): more than 65535 position buckets");
    ScoreSrc second_src;
    for (int pass = 0; pass < 2; pass++) {
        const uint32_t max_nb = pass ? max_nb_second : max_nb_first;
        const ScoreSrc *S = &first;
        if (pass == 1) {
            if (!second) break;
            bool go = false;
            ScoreLaunch SL; SL.d_tcoff = d_tcoff; SL.key64 = key64; SL.sp = sp; SL.d_qlen = d_qlen; SL.d_qlen2 = d_qlen2; SL.d_res = d_res;
            SL.d_tc_tax = d_tc_tax; SL.d_tc_cnt = d_tc_cnt; SL.tc_cap = tc_cap; SL.tc_base = tc_base;
            STCHK((*second)(&second_src, &go, SL));
            if (!go) break;
            S = &second_src;
        }
        /* many more workgroups than resident slots (14 single-wave workgroups per CU): reads differ a lot in cost, and ~8 reads
         * per workgroup balanced best (10 M reads: 3584 workgroups 66.6 ms, 57 k 60.2, 1 M 59.1, 2.5 M 60.4, 10 M 66.8) */
        uint32_t grid = S->grid ? S->grid : (uint32_t)std::min<uint64_t>(n_reads, std::max<uint64_t>(256ull * 14, n_reads / 8));
        /* reads with a big segment OR many position buckets are scored entirely out of a slab */
        bool need_slab = S->max_seg > S->cap || max_nb > MTB_SCORE_BKT;
        uint32_t slab_n = need_slab ? std::max<uint32_t>(S->max_seg, 1) : 0;
        uint32_t slab_nb = need_slab ? max_nb : 0;
        uint64_t slab_bytes = need_slab ? score_slab_bytes(slab_n, slab_nb) : 0;
        uint8_t *d_slabs = nullptr;
        const bool dynamic = need_slab && !S->cursor;       /* slab launches claim reads from a counter: the grid is what is resident */
        if (dynamic) grid = std::min<uint32_t>(grid, 256u * 14u);
        if (slab_bytes) {
            while ((uint64_t)grid * slab_bytes > (MTB_SLAB_POOL_MAX) && grid > 64) grid /= 2;     /* keep the slab pool bounded (a halved grid halves the waves that hide the slab's HBM latency) */
            STCHK(ensure(c, "slabs", (size_t)grid * slab_bytes, &d_slabs));
        }
        unsigned long long *d_work = nullptr;
        if (dynamic) { d_work = (unsigned long long *)(c->d_xscal + 4 + pass); HIPCHK(hipMemsetAsync(d_work, 0, 8, c->stream)); }
#define MTB_LAUNCH_SCORE(SRT, K, CAPV, DYNV, SLOTV) hipLaunchKernelGGL((k_score<SRT, K, mtb_match, CAPV, DYNV, SLOTV>), dim3(grid), dim3(64), 0, c->stream, S->m, S->seg, n_reads, d_qlen, \
        d_qlen2, tax_view(ix), sp, (const uint64_t *)d_tcoff, d_res, d_tc_tax, d_tc_cnt, tc_cap, d_slabs, slab_bytes, slab_n, slab_nb, (mtb_match *)nullptr,  \
        tc_base, S->list, S->n_list, S->cursor, S->stride, S->seg_by_list, S->direct, S->epoch, S->big_list, S->n_big, S->cnt_out, d_work, S->only_flagged, S->seg_cnt)
        ScoreSrc S_rest;
        const bool pairs = p->seq_mode == 2;
        if (S->cursor && pass == 0 && key64 && S->stride <= 384u && !c->opt.no_fast_scorer && !(pairs && c->opt.no_fast_pairs)) {
            /* slot mode: the register-resident scorer takes every read with the common structure (slots in compareMatches order
             * once the species -- for pairs the (species, frame) runs -- are laid one after another, one match per position
             * group) and lists the others for the generic kernel below */
            uint8_t *d_slow;
            STCHK(ensure(c, "slowflag", n_reads, &d_slow));
            HIPCHK(hipMemsetAsync(d_slow, 0, n_reads, c->stream));
            HIPCHK(hipMemsetAsync(c->d_scal + 6, 0, 8, c->stream));
#define MTB_LAUNCH_FAST(...) hipLaunchKernelGGL((k_score_fast<__VA_ARGS__>), dim3(grid), dim3(64), 0, c->stream, (const mtb_slot16 *)S->m, n_reads, d_qlen, d_qlen2, tax_view(ix), sp, \
            (const uint64_t *)d_tcoff, d_res, d_tc_tax, d_tc_cnt, tc_cap, tc_base, S->cursor, S->stride, S->direct, S->epoch, d_slow, S->cnt_out)
            { KTimer ktf(c, MTB_K_SCORE_FAST);
              if (pairs) { if (S->stride <= 192) MTB_LAUNCH_FAST(3, 3, true); else if (S->stride <= 256) MTB_LAUNCH_FAST(4, 4, true); else MTB_LAUNCH_FAST(5, 6, true); }
              else if (S->stride <= 128) MTB_LAUNCH_FAST(2); else if (S->stride <= 192) MTB_LAUNCH_FAST(3); else if (S->stride <= 256) MTB_LAUNCH_FAST(4); else MTB_LAUNCH_FAST(5, 6, false); }
#undef MTB_LAUNCH_FAST
            hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, c->stream, (const uint8_t *)d_slow, n_reads, (unsigned long long *)(c->d_scal + 6));
            if (S->big_list && S->n_big) hipLaunchKernelGGL(k_list_flag2, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, c->stream, (const uint8_t *)d_slow, n_reads, S->big_list, S->n_big);
            c->fast_used = true;
            S_rest = *S; S_rest.only_flagged = d_slow;
            S = &S_rest;
        }
        KTimer kt(c, pass == 0 ? MTB_K_SCORE : MTB_K_SEGSORT);      /* the deferred reads' launch is booked with the large-segment path */
        if (S->cursor) {              /* slot mode (always sorts in the kernel); LDS staging capacity chosen by the caller */
#define MTB_LAUNCH_SLOT(CAPV) do { if (key64) MTB_LAUNCH_SCORE(true, true, CAPV, false, true); else MTB_LAUNCH_SCORE(true, false, CAPV, false, true); } while (0)
            if (S->cap <= 144) MTB_LAUNCH_SLOT(144);
            else if (S->cap <= 160) MTB_LAUNCH_SLOT(160);
            else if (S->cap <= 224) MTB_LAUNCH_SLOT(224);
            else if (S->cap <= 288) MTB_LAUNCH_SLOT(288);
            else MTB_LAUNCH_SLOT(320);
#undef MTB_LAUNCH_SLOT
        } else if (dynamic) {
            if (!S->sort) MTB_LAUNCH_SCORE(false, false, MTB_SCORE_LDS, true, false);
            else if (key64) MTB_LAUNCH_SCORE(true, true, MTB_SCORE_LDS, true, false); else MTB_LAUNCH_SCORE(true, false, MTB_SCORE_LDS, true, false);
        }
        else if (S->sort) { if (key64) MTB_LAUNCH_SCORE(true, true, MTB_SCORE_LDS, false, false); else MTB_LAUNCH_SCORE(true, false, MTB_SCORE_LDS, false, false); }
        else if (S->cap >= 320) MTB_LAUNCH_SCORE(false, false, 320, false, false);      /* the deferred reads of a slot batch: a few hundred matches each stay in LDS, only the rest works out of slabs */
        else MTB_LAUNCH_SCORE(false, false, MTB_SCORE_LDS, false, false);
#undef MTB_LAUNCH_SCORE
    }
    HIPCHK(hipGetLastError());
    return MTB_OK;
}

/* Long reads (seq_mode 3): exact segments in compareMatches order -> k_score_long, one workgroup per read (kernels_score_long.h).
 * Reads it cannot take (LDS budgets) are flagged and scored by the generic slab launch of dev_score afterwards. */
static mtb_status dev_score_long(mtb_ctx *c, mtb_index *ix, const mtb_params *p, uint64_t n_reads, const int32_t *d_qlen, const int32_t *d_qlen2,
                                 uint32_t max_len, mtb_result *d_res, int32_t *d_tc_tax, uint32_t *d_tc_cnt, uint64_t tc_cap, uint64_t *n_tc,
                                 uint64_t tc_base, const mtb_match *d_m, const uint64_t *d_seg, uint32_t max_seg, const uint32_t *d_segcnt = nullptr,
                                 const uint32_t *d_list = nullptr, uint32_t n_list = 0 /* only these reads; segments indexed by the list slot; taxcnt slots as laid out by a previous call */) {
    mtb_score_params sp; mtb_make_score_params(p, &sp);
    uint32_t *d_bound; uint64_t *d_tcoff; uint64_t *d_ws; uint8_t *d_todo;
    STCHK(ensure(c, "bound", n_reads, &d_bound));
    STCHK(ensure(c, "tcoff", n_reads + 1, &d_tcoff));
    STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    STCHK(ensure(c, "slowflag", n_reads, &d_todo));
    if (!d_list) {
        /* per-read taxcnt slots: one per position bucket (slot ranges: the match count is not known yet), else min(matches, buckets) */
        hipLaunchKernelGGL(k_taxcnt_bound, dim3((uint32_t)((n_reads + 63) / 64)), dim3(64), 0, c->stream, d_seg, d_segcnt, d_qlen, d_qlen2,
                           n_reads, sp.dna_shift, d_bound);
        { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(c->stream, d_bound, n_reads, true, d_tcoff, d_ws); }
        uint64_t tot = 0;
        STCHK(d2h(c, &tot, d_tcoff + n_reads, 8));
        *n_tc = tot;
        if (tot > tc_cap) return fail(MTB_ERR_CAPACITY, "taxcnt buffers too small");
        if (tc_base + tot >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 taxcnt slots in one batch (mtb_result.taxcnt_off is 32 bits); split the batch");
        const uint32_t max_nb = (uint32_t)mtb_num_buckets((int32_t)max_len, sp.dna_shift);
        if (max_nb > 65535u) return fail(MTB_ERR_ARG, "read too long for mtb_result.n_taxcnt (16 bits): more than 65535 position buckets");
    }
    HIPCHK(hipMemsetAsync(d_todo, 0, n_reads, c->stream));
    unsigned long long *d_work = (unsigned long long *)(c->d_xscal + 6);
    HIPCHK(hipMemsetAsync(d_work, 0, 8, c->stream));
    HIPCHK(hipMemsetAsync(c->d_scal + 6, 0, 8, c->stream));
    {   KTimer kt(c, MTB_K_SCORE_FAST);       /* booked with the register-resident scorer's id: the workgroup-per-read kernel of long reads */
        /* two launches (round 6): <.., 512 paths, 2048 position buckets> first -- 33 KB of LDS, four workgroups per CU; reads up to ~18 kb with up to 512 emitted
         * paths: nearly all -- then the full budgets (70 KB, two per CU) for the reads the first one flagged */
        const uint64_t n_it = d_list ? n_list : n_reads;
        hipLaunchKernelGGL((k_score_long<MTB_LONG_MAXBLK, MTB_LONG_MAXSP, 512, 2048>), dim3((uint32_t)std::min<uint64_t>(n_it, 256ull * 4)), dim3(MTB_LONG_NT), 0, c->stream, d_m, d_seg, n_reads, d_qlen, d_qlen2, tax_view(ix), sp, (const uint64_t *)d_tcoff,
                           d_res, d_tc_tax, d_tc_cnt, tc_cap, tc_base, d_todo, d_work, d_segcnt, d_list, n_list, 0u);
        hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, c->stream, (const uint8_t *)d_todo, n_reads, (unsigned long long *)(c->d_scal + 6));
        uint64_t n_second = 0;
        STCHK(d2h(c, &n_second, c->d_scal + 6, 8));
        HIPCHK(hipMemsetAsync(c->d_scal + 6, 0, 8, c->stream));
        if (n_second) {
            HIPCHK(hipMemsetAsync(d_work, 0, 8, c->stream));
            hipLaunchKernelGGL((k_score_long<MTB_LONG_MAXBLK, MTB_LONG_MAXSP>), dim3((uint32_t)std::min<uint64_t>(n_it, 256ull * 2)), dim3(MTB_LONG_NT), 0, c->stream, d_m, d_seg, n_reads, d_qlen, d_qlen2, tax_view(ix), sp, (const uint64_t *)d_tcoff,
                               d_res, d_tc_tax, d_tc_cnt, tc_cap, tc_base, d_todo, d_work, d_segcnt, d_list, n_list, 1u);
        }
#ifdef MTB_LONG_PHASE_CYCLES
    {   /* profiling build: cycles of thread 0 per phase of k_score_long, summed over the workgroups (and reset) */
        HIPCHK(hipStreamSynchronize(c->stream));
        unsigned long long h[16], z[16] = {0};
        HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(mtb_long_cycles), sizeof(h)));
        HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mtb_long_cycles), z, sizeof(z)));
        static const char *nm[10] = {"setup", "block list", "walk", "rank", "species ranges", "combination", "decision", "filter", "taxCnt gather", "descent + output + next read"};
        unsigned long long tot = 0; for (int k = 0; k < 10; k++) tot += h[k];
        fprintf(stderr, "k_score_long phases (%llu reads):", (unsigned long long)(d_list ? n_list : n_reads));
        for (int k = 0; k < 10; k++) fprintf(stderr, " %s %.1f %%;", nm[k], tot ? 100.0 * (double)h[k] / (double)tot : 0.0);
        fprintf(stderr, "\n");
    }
#endif
    }
    hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, c->stream, (const uint8_t *)d_todo, n_reads, (unsigned long long *)(c->d_scal + 6));
    HIPCHK(hipGetLastError());
    uint64_t n_left = 0;
    STCHK(d2h(c, &n_left, c->d_scal + 6, 8));
    c->fast_used = true;                      /* statistics: d_scal[6] = reads the generic kernel scores */
    if (n_left == 0) return MTB_OK;
    ScoreSrc a; a.m = d_m; a.seg = d_seg; a.sort = false; a.max_seg = std::max<uint32_t>(max_seg, MTB_SCORE_LDS + 1); a.only_flagged = d_todo; a.seg_cnt = d_segcnt;
    a.bound_by_buckets = d_segcnt ? (const void *)d_segcnt : (const void *)d_list;      /* the same taxcnt layout as above */
    uint32_t *d_nl = nullptr;
    if (d_list) { STCHK(ensure(c, "lnlist", 2, &d_nl)); STCHK(h2d(c, d_nl, &n_list, 4)); a.list = d_list; a.n_list = d_nl; a.seg_by_list = 1; }
    uint64_t n_tc2 = 0;
    STCHK(dev_score(c, ix, p, n_reads, d_qlen, d_qlen2, max_len, d_res, d_tc_tax, d_tc_cnt, tc_cap, &n_tc2, tc_base, a, nullptr));
    HIPCHK(hipMemsetAsync(c->d_scal + 6, 0, 8, c->stream));         /* dev_score's statistics slot: put the count back */
    hipLaunchKernelGGL(k_count_flags, dim3(256), dim3(256), 0, c->stream, (const uint8_t *)d_todo, n_reads, (unsigned long long *)(c->d_scal + 6));
    return MTB_OK;
}

/* ------------------------------------------------------------------ */
/* index                                                               */
/* ------------------------------------------------------------------ */
/* per-read match counts of caller-supplied records; rejects sequenceIDs outside 1..n_reads */
static mtb_status count_reads(mtb_ctx *c, const mtb_match *d_in, uint64_t n, uint64_t n_reads, uint32_t *d_rc) {
    HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, c->stream));
    if (n == 0) return MTB_OK;
    HIPCHK(hipMemsetAsync(c->d_scal + 1, 0, 8, c->stream));
    hipLaunchKernelGGL(k_count_reads, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, c->stream, d_in, n, d_rc, n_reads, (uint32_t *)(c->d_scal + 1));
    uint64_t bad = 0;
    STCHK(d2h(c, &bad, c->d_scal + 1, 8));
    if (bad) return fail(MTB_ERR_ARG, "match records with a sequenceID outside 1..n_reads");
    return MTB_OK;
}

static mtb_status upload_taxonomy(mtb_index *ix) {
    const mtbhost::Taxonomy &t = ix->tax;
    size_t n = (size_t)t.max_id + 1;
    HIPCHK(hipMalloc((void **)&ix->d_canon, n * 4)); HIPCHK(hipMalloc((void **)&ix->d_parent, n * 4));
    HIPCHK(hipMalloc((void **)&ix->d_depth, n * 4)); HIPCHK(hipMalloc((void **)&ix->d_spparent, n * 4));
    HIPCHK(hipMalloc((void **)&ix->d_tax2species, n * 4)); HIPCHK(hipMalloc((void **)&ix->d_under, n)); HIPCHK(hipMalloc((void **)&ix->d_accleaf, n));
    /* parent must be indexable for every canonical id; absent ids keep -1 (never dereferenced) */
    HIPCHK(hipMemcpy(ix->d_canon, t.canon.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_parent, t.parent.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_depth, t.depth.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_spparent, t.sp_parent.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_tax2species, t.tax2species.data(), n * 4, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_under, t.under_euk.data(), n, hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(ix->d_accleaf, t.acc_leaf.data(), n, hipMemcpyHostToDevice));
    {   /* the same per taxid in one record (k_score_fast) */
        std::vector<mtb_tax_node> nodes(n);
        for (size_t i = 0; i < n; i++) {
            const int32_t c = t.canon[i];
            nodes[i].canon = c;
            nodes[i].depth = c >= 0 ? t.depth[(size_t)c] : 0;
            nodes[i].parent = c >= 0 ? t.parent[(size_t)c] : -1;
            nodes[i].flags = c >= 0 ? ((t.under_euk[(size_t)c] ? 1u : 0u) | (t.acc_leaf[(size_t)c] ? 2u : 0u)) : 0u;
        }
        HIPCHK(hipMalloc((void **)&ix->d_node, n * sizeof(mtb_tax_node)));
        HIPCHK(hipMemcpy(ix->d_node, nodes.data(), n * sizeof(mtb_tax_node), hipMemcpyHostToDevice));
    }
    return MTB_OK;
}

/* The amino-acid prefix directory of an index that owns (or borrows) a complete flat array: L letters with 21^L >= T / 8, at most
 * 7 (7.2 GB).  Not built (k_join is used) when the index holds a 5-bit letter >= 21 or a directory group spans 2^32 targets.
 * MTB_NO_DIR=1 disables it (A/B measurements). */
static mtb_status build_directory(mtb_ctx *c, mtb_index *ix) {
    if (ix->T < 2 || c->opt.no_dir) return MTB_OK;
    int L = 1;
    while (L < 7 && (uint64_t)mtb_pow21(L) < ix->T / 8) L++;
    if (c->opt.dir_depth > 0) L = std::max(1, std::min(7, c->opt.dir_depth));       /* tests force depth 7 (packed state) on toy indices */
    const uint32_t nbk = mtb_pow21(L);
    const uint32_t n_groups = (nbk >> 16) + 1;
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    if (((size_t)nbk + 1) * 4 + ((size_t)n_groups + 2) * 8 + (64u << 20) > fr) return MTB_OK;       /* no room: the bisection join still works */
    uint32_t *d_flags = (uint32_t *)(c->d_scal + 6);
    HIPCHK(hipMalloc((void **)&ix->d_dir, ((size_t)nbk + 1) * 4));
    HIPCHK(hipMalloc((void **)&ix->d_dirbase, ((size_t)n_groups + 2) * 8));
    HIPCHK(hipMemsetAsync(d_flags, 0, 8, c->stream));
    const int fmt = ix->params.kmer_format;
    hipLaunchKernelGGL(k_dir_base, dim3((n_groups + 1 + 255) / 256), dim3(256), 0, c->stream, (const uint64_t *)ix->d_values, ix->T, L, fmt, n_groups, ix->d_dirbase);
    hipLaunchKernelGGL(k_dir_fill, dim3((uint32_t)std::min<uint64_t>((ix->T + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, (const uint64_t *)ix->d_values, ix->T, L, fmt,
                       nbk, (const uint64_t *)ix->d_dirbase, ix->d_dir, d_flags);
    hipLaunchKernelGGL(k_dir_tail, dim3(4096), dim3(256), 0, c->stream, (const uint64_t *)ix->d_values, ix->T, L, fmt, nbk, (const uint64_t *)ix->d_dirbase, ix->d_dir, d_flags);
    HIPCHK(hipGetLastError());
    uint32_t fl[2] = {0, 0};
    STCHK(d2h(c, fl, d_flags, 8));
    if (fl[0] || fl[1]) { hipError_t e = hipFree(ix->d_dir); e = hipFree(ix->d_dirbase); (void)e; ix->d_dir = nullptr; ix->d_dirbase = nullptr; return MTB_OK; }
    ix->dir_L = L; ix->dir_buckets = nbk;
    return MTB_OK;
}

static mtb_dir_view dir_view(const mtb_index *ix) {
    mtb_dir_view dv; dv.dir = ix->d_dir; dv.base = ix->d_dirbase; dv.L = ix->dir_L; dv.kmer_format = ix->params.kmer_format; dv.n_buckets = ix->dir_buckets;
    return dv;
}
/* the two states of the target array: flat {value[T], info[T]} (every stage-level entry point, download / write / slices) and packed
 * (the fused join; depth-7 directory only).  Conversions are in place, ~25 ms each at 16 G targets, and happen only on a change of use. */
/* How a database directory is cut into n_parts value ranges at `split` checkpoints (IndexCreator.cpp:848-857: a
 * checkpoint {value, diffIdx offset after it, info index + 1} sits on the first metamer of an amino-acid group).  */
struct PartPlan {
    struct P { bool empty = true, explicit_first = false, drop_last = false; uint64_t ad = 0, diff_lo = 0, diff_hi = 0, info_lo = 0, info_hi = 0; };
    std::vector<P> parts;
    std::vector<uint64_t> bounds;       /* lower amino-acid-part bound of every partition */
};
static mtb_status plan_parts(const std::string &d, uint32_t n_parts, PartPlan *plan) {
    struct Split { uint64_t ad, diff_off, info_off; };
    std::vector<Split> sp;
    FILE *f = fopen((d + "/diffIdx").c_str(), "rb"); if (!f) return fail(MTB_ERR_IO, "cannot open " + d + "/diffIdx");
    fseek(f, 0, SEEK_END); const uint64_t n16 = (uint64_t)ftell(f) / 2; fclose(f);
    f = fopen((d + "/info").c_str(), "rb"); if (!f) return fail(MTB_ERR_IO, "cannot open " + d + "/info");
    fseek(f, 0, SEEK_END); const uint64_t T = (uint64_t)ftell(f) / 4; fclose(f);
    std::vector<Split> use; use.push_back(Split{0, 0, 0});
    if (n_parts > 1) {
        if (!mtbhost::read_whole(d + "/split", &sp)) return fail(MTB_ERR_IO, "cannot read " + d + "/split");
        for (size_t i = 1; i < sp.size(); i++) if (sp[i].ad != 0 && sp[i].ad != UINT64_MAX && sp[i].info_off > use.back().info_off) use.push_back(sp[i]);
    }
    plan->parts.assign(n_parts, PartPlan::P()); plan->bounds.assign(n_parts, UINT64_MAX);
    const size_t U = use.size();
    std::vector<size_t> k(n_parts + 1);
    for (uint32_t p = 0; p <= n_parts; p++) k[p] = (size_t)((uint64_t)p * U / n_parts);
    for (uint32_t p = 0; p < n_parts; p++) {
        PartPlan::P &P = plan->parts[p];
        if (k[p] == k[p + 1] || T == 0) continue;
        P.empty = false;
        const Split &s0 = use[k[p]];
        if (k[p] == 0) { P.explicit_first = false; P.ad = 0; P.diff_lo = 0; P.info_lo = 0; }
        else { P.explicit_first = true; P.ad = s0.ad; P.diff_lo = s0.diff_off; P.info_lo = s0.info_off - 1; }
        if (k[p + 1] >= U) { P.drop_last = false; P.diff_hi = n16; P.info_hi = T; }
        else { P.drop_last = true; P.diff_hi = use[k[p + 1]].diff_off; P.info_hi = use[k[p + 1]].info_off - 1; }
        plan->bounds[p] = k[p] == 0 ? 0 : (s0.ad & ~0xFFFFFFull);
    }
    for (int64_t p = (int64_t)n_parts - 2; p >= 0; p--) if (plan->parts[(size_t)p].empty) plan->bounds[(size_t)p] = plan->bounds[(size_t)p + 1];
    if (T == 0) plan->bounds[0] = 0;
    return MTB_OK;
}

/* bytes [off, off + len) of a file into a (pinned) host buffer, with a few threads: one pread stream tops out near 3 GB/s from the page
 * cache, the files of a GTDB-scale database are ~150 GB */
static bool pread_parallel(int fd, void *dst, uint64_t off, size_t len, int n_threads) {
    if (len == 0) return true;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, len >> 22));
    std::vector<char> ok((size_t)nt, 1);
    auto part = [&](int t) {
        size_t lo = len * (size_t)t / (size_t)nt, hi = len * ((size_t)t + 1) / (size_t)nt;
        while (lo < hi) {
            const ssize_t r = pread(fd, (char *)dst + lo, hi - lo, (off_t)(off + lo));
            if (r <= 0) { ok[(size_t)t] = 0; return; }
            lo += (size_t)r;
        }
    };
    if (nt == 1) part(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(part, t); for (auto &x : th) x.join(); }
    for (char k : ok) if (!k) return false;
    return true;
}

static bool pwrite_parallel(int fd, const void *src, uint64_t off, size_t len, int n_threads) {
    if (len == 0) return true;
    const int nt = (int)std::max<size_t>(1, std::min<size_t>((size_t)n_threads, len >> 22));
    std::vector<char> ok((size_t)nt, 1);
    auto part = [&](int t) {
        size_t lo = len * (size_t)t / (size_t)nt, hi = len * ((size_t)t + 1) / (size_t)nt;
        while (lo < hi) {
            const ssize_t r = pwrite(fd, (const char *)src + lo, hi - lo, (off_t)(off + lo));
            if (r <= 0) { ok[(size_t)t] = 0; return; }
            lo += (size_t)r;
        }
    };
    if (nt == 1) part(0);
    else { std::vector<std::thread> th; for (int t = 0; t < nt; t++) th.emplace_back(part, t); for (auto &x : th) x.join(); }
    for (char k : ok) if (!k) return false;
    return true;
}

/* bytes [off, off + len) of a file -> device memory, double-buffered through pinned host memory */
static mtb_status stream_file_to_device(mtb_ctx *c, const std::string &path, uint64_t off, uint64_t len, void *d_dst) {
    if (len == 0) return MTB_OK;
    const int fd = open(path.c_str(), O_RDONLY);
    if (fd < 0) return fail(MTB_ERR_IO, "cannot open " + path);
    const size_t CH = 64u << 20;
    void *buf[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; bool used[2] = {false, false};
    mtb_status st = MTB_OK;
    if (hipHostMalloc(&buf[0], CH, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&buf[1], CH, hipHostMallocDefault) != hipSuccess ||
        hipEventCreate(&ev[0]) != hipSuccess || hipEventCreate(&ev[1]) != hipSuccess) { (void)hipGetLastError(); st = fail(MTB_ERR_OOM, "no pinned host memory for the index upload"); }
    uint64_t done = 0; int k = 0;
    while (st == MTB_OK && done < len) {
        const size_t n = (size_t)std::min<uint64_t>(CH, len - done);
        if (used[k] && hipEventSynchronize(ev[k]) != hipSuccess) { st = fail(MTB_ERR_DEVICE, "hipEventSynchronize failed during the index upload"); break; }
        if (!pread_parallel(fd, buf[k], off + done, n, 8)) { st = fail(MTB_ERR_IO, "short read from " + path); break; }
        if (hipMemcpyAsync((char *)d_dst + done, buf[k], n, hipMemcpyHostToDevice, c->stream) != hipSuccess || hipEventRecord(ev[k], c->stream) != hipSuccess) {
            st = fail(MTB_ERR_DEVICE, "H2D copy failed during the index upload"); break; }
        used[k] = true; done += n; k ^= 1;
    }
    hipError_t e = hipStreamSynchronize(c->stream); (void)e;
    close(fd);
    for (int i = 0; i < 2; i++) { if (buf[i]) e = hipHostFree(buf[i]); if (ev[i]) e = hipEventDestroy(ev[i]); }
    (void)e;
    return st;
}

/* The target list of a database (or of one value range of it) into HBM, decoded in CHUNKS of the diffIdx stream (the reference streams
 * the files too: KmerMatcher.cpp:212-217, 256-271, getNextTargetKmer KmerMatcher.h:282-297):
 *   chunk of 16-bit words (+ the words of a metamer cut by the previous chunk's end) -> terminators per tile -> offsets -> deltas
 *   written to their final places in value[] -> first delta += the previous chunk's last value -> 64-bit inclusive scan of the chunk
 *   -> the amino-acid directory rows of the chunk (k_dir_chunk_*) -> with `pack`, the chunk's info entries (streamed into a chunk
 *   buffer) are folded into packed 8-byte words at once (kernels_dir.h) and info[] is never resident.
 * Peak HBM: 8 bytes per target (+ 4 for info[] without `pack`) + the directory + one chunk (2 B per word, 4 B per metamer of it):
 * a 16 G-target database opens on one 288 GB GPU; the whole-file decode of round 3 held ~18 bytes per target at once.
 * On return *dir_ok says whether the directory is usable (else it has been freed); with `pack` a false *dir_ok is an error to the
 * caller (the array is partly packed): it opens again without. */
struct OpenPlan { uint64_t n16 = 0, T = 0, expect = 0, lead = 0, diff_off = 0, info_off = 0, first_value = 0; };
static mtb_status decode_chunked(mtb_ctx *c, mtb_index *ix, const std::string &d, const OpenPlan &P, bool want_dir, int L, bool pack, bool *dir_ok) {
    *dir_ok = false;
    hipStream_t st = c->stream;
    /* 16-bit words per chunk: 128 M (256 MB), less when the context's workspace limit asks for it (a chunk costs ~16 bytes per word:
     * the words twice, tile tables, the info entries of its metamers, scan workspace); MTB_OPEN_CHUNK: tests (a few dozen words) */
    uint64_t CH = 128ull << 20;
    if (c->ws_limit) CH = std::max<uint64_t>(1u << 16, std::min<uint64_t>(CH, c->ws_limit / 16));
    if (c->opt.open_chunk > 0) CH = std::max<uint64_t>(16, (uint64_t)c->opt.open_chunk);
    const int fmt = ix->params.kmer_format;
    const uint32_t nbk = want_dir ? mtb_pow21(L) : 0, n_groups = want_dir ? (nbk >> 16) + 1 : 0;
    uint32_t *d_flags = (uint32_t *)(c->d_scal + 6);
    uint64_t *d_carry = c->d_scal + 5;                      /* last flat value of the previous chunk */
    if (want_dir) {
        HIPCHK(hipMalloc((void **)&ix->d_dir, ((size_t)nbk + 1) * 4));
        HIPCHK(hipMalloc((void **)&ix->d_dirbase, ((size_t)n_groups + 2) * 8));
        HIPCHK(hipMemsetAsync(d_flags, 0, 8, st));
    }
    const int fd_d = open((d + "/diffIdx").c_str(), O_RDONLY), fd_i = open((d + "/info").c_str(), O_RDONLY);
    struct Files { int a, b; ~Files() { if (a >= 0) close(a); if (b >= 0) close(b); } } files{fd_d, fd_i};
    if (fd_d < 0 || fd_i < 0) return fail(MTB_ERR_IO, "cannot open diffIdx / info in " + d);
    const uint64_t chunk_words = std::min<uint64_t>(CH, std::max<uint64_t>(P.n16, 1));
    const uint64_t info_cap = std::min<uint64_t>(chunk_words + 8, P.T + 1);      /* metamers of a chunk <= its words */
    struct Pinned { void *p[3] = {nullptr, nullptr, nullptr}; ~Pinned() { for (void *q : p) if (q) { hipError_t e = hipHostFree(q); (void)e; } } } pin;
    if (hipHostMalloc(&pin.p[0], (chunk_words + 8) * 2, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&pin.p[1], (chunk_words + 8) * 2, hipHostMallocDefault) != hipSuccess ||
        hipHostMalloc(&pin.p[2], info_cap * 4, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no pinned host memory for the index upload"); }
    uint16_t *d_chunk; uint32_t *d_tc; uint64_t *d_toff, *d_ws; uint32_t *d_ichunk = nullptr;
    const uint64_t max_tiles = (chunk_words + 8 + 2047) / 2048 + 1;
    STCHK(ensure(c, "diffraw", chunk_words + 8, &d_chunk)); STCHK(ensure(c, "difftc", max_tiles, &d_tc)); STCHK(ensure(c, "difftoff", max_tiles + 1, &d_toff));
    STCHK(ensure(c, "scanws", scan_ws_elems(std::max<uint64_t>(max_tiles + 1, info_cap + 1)), &d_ws));
    if (pack) STCHK(ensure(c, "infochunk", info_cap, &d_ichunk));
    { const uint64_t c0 = P.lead ? P.first_value : 0; STCHK(h2d(c, d_carry, &c0, 8)); }
    if (P.lead) {
        /* a range that starts at a split checkpoint: its first metamer is given by the checkpoint, not coded in the byte range -- a chunk
         * of one entry (value, info entry, directory rows, packing) */
        STCHK(h2d(c, ix->d_values, &P.first_value, 8));
        if (!pread_parallel(fd_i, pin.p[2], P.info_off * 4, 4, 1)) return fail(MTB_ERR_IO, "short read from " + d + "/info");
        STCHK(h2d(c, pack ? d_ichunk : ix->d_info, pin.p[2], 4));
        if (want_dir) {
            hipLaunchKernelGGL(k_dir_chunk_base, dim3(1), dim3(256), 0, st, (const uint64_t *)ix->d_values, (uint64_t)0, (uint64_t)1, (const uint64_t *)d_carry, L, fmt, nbk, ix->d_dirbase, d_flags);
            hipLaunchKernelGGL(k_dir_chunk_fill, dim3(1), dim3(256), 0, st, (const uint64_t *)ix->d_values, (uint64_t)0, (uint64_t)1, (const uint64_t *)d_carry, L, fmt, nbk,
                               (const uint64_t *)ix->d_dirbase, ix->d_dir, d_flags);
        }
        if (pack) hipLaunchKernelGGL(k_index_pack_chunk, dim3(1), dim3(256), 0, st, ix->d_values, (const uint32_t *)d_ichunk, (uint64_t)0, (uint64_t)1, fmt);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
    }
    ix->open_chunk_words = chunk_words;
    uint64_t done_words = 0, g = P.lead, found = 0;       /* g: next free place in value[] */
    uint16_t carry_w[8]; uint32_t n_carry = 0;
    /* the first chunk is read here, every further one by a helper thread while the device works on its predecessor */
    uint64_t cur_new = std::min<uint64_t>(chunk_words, P.n16);
    if (cur_new && !pread_parallel(fd_d, pin.p[0], (P.diff_off + done_words) * 2, cur_new * 2, 16)) return fail(MTB_ERR_IO, "short read from " + d + "/diffIdx");
    int cur = 0;
    while (done_words < P.n16) {
        const uint64_t next_off = done_words + cur_new, next_new = std::min<uint64_t>(chunk_words, P.n16 - next_off);
        bool next_ok = true;
        std::thread reader;
        if (next_new) reader = std::thread([&, next_off, next_new] { next_ok = pread_parallel(fd_d, pin.p[cur ^ 1], (P.diff_off + next_off) * 2, next_new * 2, 16); });
        struct Join { std::thread &t; ~Join() { if (t.joinable()) t.join(); } } join_reader{reader};
        const uint16_t *hw = (const uint16_t *)pin.p[cur];
        /* words of a metamer cut by this chunk's end stay for the next one */
        uint32_t tail = 0;
        while (tail < 5 && tail < cur_new && !(hw[cur_new - 1 - tail] & 0x8000u)) tail++;
        if (tail == cur_new && cur_new < 5) { /* (a chunk of fewer than five words without a terminator: all of it is carried) */ }
        if (tail >= 5 || (next_new == 0 && tail != 0)) return fail(MTB_ERR_IO, "diffIdx is corrupt: a metamer without its terminator word");
        const uint64_t use_new = cur_new - tail, n_use = n_carry + use_new;
        if (n_carry) HIPCHK(hipMemcpyAsync(d_chunk, carry_w, n_carry * 2, hipMemcpyHostToDevice, st));
        if (use_new) HIPCHK(hipMemcpyAsync(d_chunk + n_carry, hw, use_new * 2, hipMemcpyHostToDevice, st));
        uint64_t n_k = 0;
        if (n_use) {
            const uint64_t tiles = (n_use + 2047) / 2048;
            hipLaunchKernelGGL(k_diff_tile_count, dim3((uint32_t)tiles), dim3(256), 0, st, (const uint16_t *)d_chunk, n_use, d_tc);
            scan_launch<uint32_t, uint64_t, false>(st, d_tc, tiles, true, d_toff, d_ws);
            STCHK(d2h(c, &n_k, d_toff + tiles, 8));
            if (found + n_k > P.expect) return fail(MTB_ERR_IO, "diffIdx holds more metamers than info and split announce (" + std::to_string(P.expect) + ")");
            if (n_k) {
                const uint64_t m = std::min<uint64_t>(g + n_k, P.T) - std::min<uint64_t>(g, P.T);       /* entries of the chunk that belong to the index (a dropped last one does not) */
                /* info entries of the chunk: into the chunk buffer (pack) or straight to their places */
                if (m) {
                    if (!pread_parallel(fd_i, pin.p[2], (P.info_off + g) * 4, m * 4, 16)) return fail(MTB_ERR_IO, "short read from " + d + "/info");
                    HIPCHK(hipMemcpyAsync(pack ? d_ichunk : ix->d_info + g, pin.p[2], m * 4, hipMemcpyHostToDevice, st));
                }
                hipLaunchKernelGGL(k_diff_assemble, dim3((uint32_t)tiles), dim3(256), 0, st, (const uint16_t *)d_chunk, n_use, (const uint64_t *)d_toff, ix->d_values + g);
                hipLaunchKernelGGL(k_diff_add_carry, dim3(1), dim3(1), 0, st, ix->d_values + g, (const uint64_t *)d_carry);
                scan_launch<uint64_t, uint64_t, true>(st, ix->d_values + g, n_k, false, ix->d_values + g, d_ws);
                if (want_dir && m) {
                    const dim3 grid((uint32_t)std::min<uint64_t>((m + 255) / 256, 1u << 16));
                    hipLaunchKernelGGL(k_dir_chunk_base, grid, dim3(256), 0, st, (const uint64_t *)ix->d_values, g, m, (const uint64_t *)d_carry, L, fmt, nbk, ix->d_dirbase, d_flags);
                    hipLaunchKernelGGL(k_dir_chunk_fill, grid, dim3(256), 0, st, (const uint64_t *)ix->d_values, g, m, (const uint64_t *)d_carry, L, fmt, nbk,
                                       (const uint64_t *)ix->d_dirbase, ix->d_dir, d_flags);
                }
                hipLaunchKernelGGL(k_save_last, dim3(1), dim3(1), 0, st, (const uint64_t *)(ix->d_values + g + n_k - 1), d_carry);
                if (pack && m) hipLaunchKernelGGL(k_index_pack_chunk, dim3((uint32_t)std::min<uint64_t>((m + 255) / 256, 1u << 16)), dim3(256), 0, st, ix->d_values, (const uint32_t *)d_ichunk, g, m, fmt);
                HIPCHK(hipGetLastError());
                HIPCHK(hipStreamSynchronize(st));          /* the pinned buffers are refilled next */
                g += n_k; found += n_k;
                if (ix->open_chunks < 4 || (ix->open_chunks & 63u) == 0) {        /* (a driver query: not for every one of thousands of tiny test chunks) */
                    size_t fr = 0, tot = 0; if (hipMemGetInfo(&fr, &tot) == hipSuccess && ix->open_free0 > fr) ix->open_peak_bytes = std::max<uint64_t>(ix->open_peak_bytes, ix->open_free0 - fr); }
                ix->open_chunks++;
            }
        }
        for (uint32_t k = 0; k < tail; k++) carry_w[k] = hw[cur_new - tail + k];
        if (n_use == 0 && n_carry) { /* nothing decodable yet: keep the old carry in front (cannot happen: a carry is < 5 words and a metamer ends within 5) */
            return fail(MTB_ERR_IO, "diffIdx is corrupt: a metamer of more than five words"); }
        n_carry = tail;
        if (reader.joinable()) reader.join();
        if (!next_ok) return fail(MTB_ERR_IO, "short read from " + d + "/diffIdx");
        done_words = next_off; cur_new = next_new; cur ^= 1;
    }
    if (found != P.expect)    /* validateDatabase.cpp:17-142: #terminators must equal #info entries */
        return fail(MTB_ERR_IO, "diffIdx holds " + std::to_string(found) + " metamers where info and split announce " + std::to_string(P.expect));
    if (want_dir) {
        /* (*d_carry = the last decoded value; with a dropped last entry the index's own last flat value is one before it: re-read it) */
        if (P.T && g > P.T) {
            if (pack) return fail(MTB_ERR_UNSUPPORTED, "pack on load of a range whose last coded metamer is dropped");       /* (never chosen: see open_impl) */
            hipLaunchKernelGGL(k_save_last, dim3(1), dim3(1), 0, st, (const uint64_t *)(ix->d_values + P.T - 1), d_carry);
        }
        hipLaunchKernelGGL(k_dir_finish, dim3(64), dim3(256), 0, st, (const uint64_t *)d_carry, P.T, L, fmt, nbk, n_groups, ix->d_dirbase, ix->d_dir, d_flags, 0);
        hipLaunchKernelGGL(k_dir_finish, dim3(4096), dim3(256), 0, st, (const uint64_t *)d_carry, P.T, L, fmt, nbk, n_groups, ix->d_dirbase, ix->d_dir, d_flags, 1);
        HIPCHK(hipGetLastError());
        uint32_t fl[2] = {0, 0};
        STCHK(d2h(c, fl, d_flags, 8));
        if (fl[0] || fl[1]) { hipError_t e = hipFree(ix->d_dir); e = hipFree(ix->d_dirbase); (void)e; ix->d_dir = nullptr; ix->d_dirbase = nullptr; }
        else { ix->dir_L = L; ix->dir_buckets = nbk; *dir_ok = true; }
    }
    HIPCHK(hipStreamSynchronize(st));
    return MTB_OK;
}

/* An open that finds device memory short of what the index needs: the workspace a concurrent (or earlier) mtb_ctx_reserve took is worth
 * less than the directory or the database itself -- stop the reservation, give the workspace back (batches grow it again as they did
 * before there was a reservation).  Returns the free bytes afterwards. */
static size_t open_make_room(mtb_ctx *c) {
    c->reserve_cancel = true;
    /* nothing may still be reading the workspace that is handed back: batches queued on the compute stream (another index of this context),
     * results on their way down (asynchronous entry point), a prefetch into the input sets (those are I/O buffers and stay, but their
     * records must not outlive a context whose streams were drained) -- ADVICE r5 */
    { hipError_t e = hipStreamSynchronize(c->stream); (void)e;
      if (c->down_stream) { e = hipStreamSynchronize(c->down_stream); (void)e; c->down_pending = false; }
      if (c->copy_stream) { e = hipStreamSynchronize(c->copy_stream); (void)e; } }
    c->last_sorted = nullptr; c->last_sorted_n = 0;       /* (pointed into the workspace) */
    { std::lock_guard<std::mutex> lk(c->reserve_mu); release_workspace(c); }
    size_t fr = 0, tot = 0;
    hipError_t e = hipMemGetInfo(&fr, &tot); (void)e;
    return fr;
}

static mtb_status open_impl(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, uint32_t part, uint32_t n_parts, mtb_index **out) {
    if (!c || !dbdir || !params || !out) return fail(MTB_ERR_ARG, "NULL argument");
    if (n_parts == 0 || part >= n_parts) return fail(MTB_ERR_ARG, "partition index out of range");
    HIPCHK(hipSetDevice(c->device));
    std::string d(dbdir);
    int reduced_aa = 0;
    mtbhost::load_db_parameters(d, params, &reduced_aa);
    if (reduced_aa) return fail(MTB_ERR_UNSUPPORTED, "database was built with the reduced amino-acid alphabet (Reduced_alphabet 1 in db.parameters); not implemented");
    if (params->kmer_format != 1 && params->kmer_format != 2) return fail(MTB_ERR_UNSUPPORTED, "database uses a k-mer format other than 1 or 2");
    /* loadTaxonomy (common.cpp:50-86): DBDIR/taxonomyDB wins when it is there and readable; else --taxonomy-path; else
     * DBDIR/taxonomy/.  (A taxonomyDB with an outdated serialization version makes the reference fall back to DBDIR/taxonomy.) */
    std::string taxdir = taxonomy_dir && *taxonomy_dir ? std::string(taxonomy_dir) : d + "/taxonomy";
    const bool have_bin = mtbhost::file_exists(d + "/taxonomyDB");
    if (!have_bin && !mtbhost::file_exists(taxdir + "/nodes.dmp"))
        return fail(MTB_ERR_IO, "no taxonomy: neither " + d + "/taxonomyDB nor dump files in " + taxdir);
    PartPlan plan;
    STCHK(plan_parts(d, n_parts, &plan));
    const PartPlan::P &P = plan.parts[part];
    mtb_index *ix = new mtb_index();
    /* every early return below hands the index (and what it holds on the device) back */
    struct Guard { mtb_index *ix; ~Guard() { if (ix) mtb_index_close(ix); } } guard{ix};
    ix->ctx = c; ix->params = *params; ix->own = true;
    for (uint32_t q = part + 1; q < n_parts; q++) if (!plan.parts[q].empty) ix->match_last = true;
    std::string err;
    bool tax_ok = false;
    if (have_bin) {
        tax_ok = mtbhost::load_taxonomy_db(d + "/taxonomyDB", &ix->tax, &err);
        if (!tax_ok) {
            const std::string fb = d + "/taxonomy";
            if (!mtbhost::file_exists(fb + "/nodes.dmp")) return fail(MTB_ERR_IO, err);
            taxdir = fb; err.clear();
        }
    }
    if (!tax_ok && !mtbhost::load_taxonomy(taxdir, &ix->tax, &err)) return fail(MTB_ERR_IO, err);
    std::vector<int32_t> ids;
    if (!mtbhost::read_taxid_list(d + "/taxID_list", &ids)) return fail(MTB_ERR_IO, "cannot open " + d + "/taxID_list");
    mtbhost::build_tax2species(&ix->tax, ids.data(), ids.size());
    ix->info_mask = ~((uint32_t)(params->skip_redundancy == 0) << 31);   /* KmerMatcher.cpp:204-205 */
    STCHK(upload_taxonomy(ix));
    if (P.empty) { ix->T = 0; guard.ix = nullptr; *out = ix; return MTB_OK; }
    /* only this partition's byte ranges are read from the files, chunk by chunk through pinned buffers straight into HBM (the next
     * chunk is read while the device decodes the current one): no whole-file copy in host OR device memory, whatever the database size */
    OpenPlan O;
    O.n16 = P.diff_hi - P.diff_lo; O.T = P.info_hi - P.info_lo; O.lead = P.explicit_first ? 1 : 0;
    O.expect = O.T - O.lead + (P.drop_last ? 1 : 0);     /* metamers coded in the byte range */
    O.diff_off = P.diff_lo; O.info_off = P.info_lo; O.first_value = P.ad;
    const uint64_t T = O.T;
    /* the directory this index will have (build_directory's rule), decided before the load so that it is built chunk by chunk */
    int L = 1;
    while (L < 7 && (uint64_t)mtb_pow21(L) < T / 8) L++;
    if (c->opt.dir_depth > 0) L = std::max(1, std::min(7, c->opt.dir_depth));
    bool want_dir = T >= 2 && !c->opt.no_dir;
    struct CancelReset { mtb_ctx *c; ~CancelReset() { c->reserve_cancel = false; } } cancel_reset{c};
    if (want_dir) {
        size_t fr = 0, tot = 0;
        HIPCHK(hipMemGetInfo(&fr, &tot));
        const uint32_t nbk = mtb_pow21(L);
        /* directory + target words + (flat state) info[] + the chunk buffers of the decode */
        const bool pack_guess = L == 7 && !P.drop_last && !c->opt.no_pack && (c->opt.open_packed >= 0 ? c->opt.open_packed != 0 : T >= (1ull << 28));
        const size_t need_dir = ((size_t)nbk + 1) * 4 + ((size_t)(nbk >> 16) + 3) * 8 + (T + 1) * 8 + (64u << 20);
        const size_t need_all = need_dir + (pack_guess ? 0 : (size_t)T * 4) + (std::min<uint64_t>(O.n16, 1ull << 27) * 12 + (64u << 20));
        if (need_all > fr && held_bytes(c) > 0) fr = open_make_room(c);      /* a reservation made for the first batches must not cost the index its directory */
        if (need_dir > fr) want_dir = false;        /* no room next to the values: the bisection join still works */
    }
    /* pack on load: big databases (>= 2^28 targets; MTB_OPEN_PACKED=1 / 0 forces it on toy databases / off) whose directory has depth 7
     * open in the SEALED state -- packed 8-byte words, info[] never resident (mtb_index_seal's state; everything that needs the flat
     * arrays unpacks on demand as for any sealed index) */
    bool pack = want_dir && L == 7 && !P.drop_last && !c->opt.no_pack && (c->opt.open_packed >= 0 ? c->opt.open_packed != 0 : T >= (1ull << 28));
    { size_t fr = 0, tot = 0; HIPCHK(hipMemGetInfo(&fr, &tot)); ix->open_free0 = fr; }
    for (int attempt = 0; attempt < 2; attempt++) {
        ix->open_chunks = 0; ix->open_peak_bytes = 0;
        {   /* short of memory with a workspace reservation in the way: hand the reservation back and try once more */
            hipError_t e = hipMalloc((void **)&ix->d_values, (T + 1) * 8);
            if (e == hipErrorOutOfMemory && held_bytes(c) > 0) { (void)hipGetLastError(); open_make_room(c); e = hipMalloc((void **)&ix->d_values, (T + 1) * 8); }
            HIPCHK(e);
            if (!pack) {
                e = hipMalloc((void **)&ix->d_info, std::max<uint64_t>(T, 1) * 4);
                if (e == hipErrorOutOfMemory && held_bytes(c) > 0) { (void)hipGetLastError(); open_make_room(c); e = hipMalloc((void **)&ix->d_info, std::max<uint64_t>(T, 1) * 4); }
                HIPCHK(e);
            }
        }
        bool dir_ok = false;
        {   /* the chunk buffers go back whatever way the decode ends */
            struct Scratch { mtb_ctx *c; ~Scratch() { release(c, "diffraw"); release(c, "difftc"); release(c, "difftoff"); release(c, "infochunk"); } } scratch{c};
            STCHK(decode_chunked(c, ix, d, O, want_dir, L, pack, &dir_ok));
        }
        if (pack && !dir_ok) {      /* a letter >= 21 or a bucket group of 2^32 targets: no directory, hence no packed state -- once more, flat */
            hipError_t e = hipFree(ix->d_values); (void)e; ix->d_values = nullptr;
            pack = false;
            continue;
        }
        if (pack) ix->packed = true;
        break;
    }
    ix->T = T;
    guard.ix = nullptr;
    *out = ix;
    return MTB_OK;
}

extern "C" {

mtb_status mtb_index_open(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, mtb_index **out) {
    return open_impl(c, dbdir, taxonomy_dir, params, 0, 1, out);
}
mtb_status mtb_index_open_part(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, uint32_t part, uint32_t n_parts,
                               mtb_index **out) {
    return open_impl(c, dbdir, taxonomy_dir, params, part, n_parts, out);
}
mtb_status mtb_index_part_bounds(const char *dbdir, uint32_t n_parts, uint64_t *bounds) {
    if (!dbdir || !bounds || n_parts == 0) return fail(MTB_ERR_ARG, "NULL argument");
    PartPlan plan;
    STCHK(plan_parts(dbdir, n_parts, &plan));
    for (uint32_t p = 0; p < n_parts; p++) bounds[p] = plan.bounds[p];
    return MTB_OK;
}

mtb_status mtb_index_from_device(mtb_ctx *c, uint64_t *d_values, uint32_t *d_info, uint64_t n_targets, const char *taxonomy_dir,
                                 const int32_t *taxid_list, size_t n_taxids, const mtb_params *params, mtb_index **out) {
    if (!c || !taxonomy_dir || !params || !out) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    mtb_index *ix = new mtb_index();
    ix->ctx = c; ix->params = *params; ix->own = false;
    std::string err;
    if (!mtbhost::load_taxonomy(taxonomy_dir, &ix->tax, &err)) { delete ix; return fail(MTB_ERR_IO, err); }
    mtbhost::build_tax2species(&ix->tax, taxid_list, n_taxids);
    ix->info_mask = ~((uint32_t)(params->skip_redundancy == 0) << 31);
    ix->d_values = d_values; ix->d_info = d_info; ix->T = n_targets;
    mtb_status st = upload_taxonomy(ix);
    if (st != MTB_OK) { mtb_index_close(ix); return st; }
    if ((st = build_directory(c, ix)) != MTB_OK) { mtb_index_close(ix); return st; }
    *out = ix;
    return MTB_OK;
}

/* "load once, broadcast over xGMI".  The resident index of one GPU copied to another context's GPU with peer
 * copies (hipMemcpyPeerAsync: device to device over xGMI where the devices can reach each other; same-device contexts get a
 * device-to-device copy) -- the target words in the state they are in (packed / flat), info[] if it is resident, the directory --
 * and the taxonomy tables uploaded from the host copy.  A node that classifies on 8 GPUs reads and decodes the database files once
 * instead of 8 times.  The reference has one address space and no analogue (KmerMatcher.cpp:212-217 opens the files per thread). */
mtb_status mtb_index_clone(mtb_index *src, mtb_ctx *dst, mtb_index **out) {
    if (!src || !dst || !out) return fail(MTB_ERR_ARG, "NULL argument");
    if (src->parent) return fail(MTB_ERR_ARG, "a view cannot be cloned: clone its parent");
    mtb_ctx *sc = src->ctx;
    if (sc->device != dst->device) {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, dst->device, sc->device) == hipSuccess && can) {
            if (hipSetDevice(dst->device) == hipSuccess) { const hipError_t e = hipDeviceEnablePeerAccess(sc->device, 0); if (e != hipSuccess) (void)hipGetLastError(); }      /* (twice = "already on") */
        } else (void)hipGetLastError();
    }
    HIPCHK(hipSetDevice(dst->device));
    mtb_index *ix = new mtb_index();
    struct Guard { mtb_index *ix; ~Guard() { if (ix) mtb_index_close(ix); } } guard{ix};
    ix->ctx = dst; ix->params = src->params; ix->own = true; ix->tax = src->tax; ix->info_mask = src->info_mask; ix->match_last = src->match_last; ix->T = src->T;
    STCHK(upload_taxonomy(ix));
    IndexUse use;                                   /* the source keeps its state (and is not converted) while it is read */
    {
        std::unique_lock<std::mutex> lk(src->state_mu);
        src->users++; use.o = src;
    }
    const bool packed = src->packed;
    auto peer = [&](void *d, const void *s_, size_t bytes) -> mtb_status {
        const size_t CH = 1ull << 30;               /* 1 GiB pieces: progress is visible to the stream, a failure is local */
        for (size_t o = 0; o < bytes; o += CH) {
            const size_t nb = std::min(CH, bytes - o);
            if (sc->device == dst->device) HIPCHK(hipMemcpyAsync((char *)d + o, (const char *)s_ + o, nb, hipMemcpyDeviceToDevice, dst->stream));
            else HIPCHK(hipMemcpyPeerAsync((char *)d + o, dst->device, (const char *)s_ + o, sc->device, nb, dst->stream));
        }
        return MTB_OK;
    };
    HIPCHK(hipStreamSynchronize(sc->stream));       /* whatever the source context still had in flight on the arrays */
    HIPCHK(hipMalloc((void **)&ix->d_values, (src->T + 1) * 8));
    STCHK(peer(ix->d_values, src->d_values, src->T * 8));
    if (src->d_info) { HIPCHK(hipMalloc((void **)&ix->d_info, std::max<uint64_t>(src->T, 1) * 4)); STCHK(peer(ix->d_info, src->d_info, src->T * 4)); }
    if (src->d_dir) {
        const uint32_t n_groups = (src->dir_buckets >> 16) + 1;
        HIPCHK(hipMalloc((void **)&ix->d_dir, ((size_t)src->dir_buckets + 1) * 4));
        HIPCHK(hipMalloc((void **)&ix->d_dirbase, ((size_t)n_groups + 2) * 8));
        STCHK(peer(ix->d_dir, src->d_dir, ((size_t)src->dir_buckets + 1) * 4));
        STCHK(peer(ix->d_dirbase, src->d_dirbase, ((size_t)n_groups + 2) * 8));
        ix->dir_L = src->dir_L; ix->dir_buckets = src->dir_buckets;
    }
    HIPCHK(hipStreamSynchronize(dst->stream));
    ix->packed = packed;
    guard.ix = nullptr;
    *out = ix;
    return MTB_OK;
}

/* ---- the resident index handed to other PROCESSES of the node (include/mtb.h: mtb_index_share) ---- */
/* An allocation below a few MB is carved out of a larger block by the runtime and cannot be opened by another process (dmabuf handles name whole
 * blocks: hipIpcOpenMemHandle answered "invalid argument" for the 160 KB target array of a toy database): such an array is copied to a 4 MiB
 * allocation of its own first (`stage`, kept by the index until it is closed or exported again) and that one is handed over. */
static mtb_status ipc_describe(mtb_ctx *c, const void *p, size_t bytes, void **stage, uint8_t handle[64], uint64_t *off) {
    static_assert(sizeof(hipIpcMemHandle_t) <= 64, "handle record");
    memset(handle, 0, 64); *off = 0;
    if (*stage) { hipError_t e = hipFree(*stage); (void)e; *stage = nullptr; }
    if (!p) return MTB_OK;
    if (bytes < (4ull << 20)) {
        HIPCHK(hipMalloc(stage, 4ull << 20));
        HIPCHK(hipMemcpyAsync(*stage, p, bytes, hipMemcpyDeviceToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        p = *stage;
    }
    hipDeviceptr_t base = nullptr; size_t size = 0;
    HIPCHK(hipMemGetAddressRange(&base, &size, (hipDeviceptr_t)p));            /* the handle names the ALLOCATION (a torch tensor may sit inside a larger block) */
    hipIpcMemHandle_t h;
    HIPCHK(hipIpcGetMemHandle(&h, (void *)base));
    memcpy(handle, &h, sizeof(h));
    *off = (uint64_t)((const char *)p - (const char *)base);
    return MTB_OK;
}
mtb_status mtb_index_export(mtb_index *src, mtb_index_share *out) {
    if (!src || !out) return fail(MTB_ERR_ARG, "NULL argument");
    if (src->parent) return fail(MTB_ERR_ARG, "a view cannot be exported: export its parent");
    memset(out, 0, sizeof(*out));
    HIPCHK(hipSetDevice(src->ctx->device));
    HIPCHK(hipStreamSynchronize(src->ctx->stream));
    std::unique_lock<std::mutex> lk(src->state_mu);
    src->state_cv.wait(lk, [&] { return src->users == 0; });
    const uint32_t n_groups = (src->dir_buckets >> 16) + 1;
    STCHK(ipc_describe(src->ctx, src->d_values, src->T * 8, &src->ipc_stage[0], out->values_handle, &out->values_off));
    STCHK(ipc_describe(src->ctx, src->d_info, src->T * 4, &src->ipc_stage[1], out->info_handle, &out->info_off));
    STCHK(ipc_describe(src->ctx, src->d_dir, ((size_t)src->dir_buckets + 1) * 4, &src->ipc_stage[2], out->dir_handle, &out->dir_off));
    STCHK(ipc_describe(src->ctx, src->d_dirbase, ((size_t)n_groups + 2) * 8, &src->ipc_stage[3], out->dirbase_handle, &out->dirbase_off));
    out->n_targets = src->T; out->dir_buckets = src->dir_buckets; out->info_mask = src->info_mask; out->dir_depth = src->dir_L;
    out->packed = src->packed ? 1 : 0; out->has_info = src->d_info ? 1 : 0; out->match_last = src->match_last ? 1 : 0; out->device = src->ctx->device;
    out->exporter_pid = (int64_t)getpid();
    out->values_ptr = (uint64_t)(uintptr_t)src->d_values; out->info_ptr = (uint64_t)(uintptr_t)src->d_info;
    out->dir_ptr = (uint64_t)(uintptr_t)src->d_dir; out->dirbase_ptr = (uint64_t)(uintptr_t)src->d_dirbase;
    return MTB_OK;
}
mtb_status mtb_index_import(mtb_ctx *c, const mtb_index_share *sh, const char *taxonomy_dir, const int32_t *taxid_list, size_t n_taxids,
                            const mtb_params *params, mtb_index **out) {
    if (!c || !sh || !taxonomy_dir || !params || !out) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    mtb_index *ix = new mtb_index();
    struct Guard { mtb_index *ix; ~Guard() { if (ix) mtb_index_close(ix); } } guard{ix};
    ix->ctx = c; ix->params = *params; ix->own = true; ix->T = sh->n_targets; ix->info_mask = sh->info_mask; ix->match_last = sh->match_last != 0;
    std::string err;
    if (!mtbhost::load_taxonomy(taxonomy_dir, &ix->tax, &err)) return fail(MTB_ERR_IO, err);
    mtbhost::build_tax2species(&ix->tax, taxid_list, n_taxids);
    STCHK(upload_taxonomy(ix));
    const bool same_process = sh->exporter_pid == (int64_t)getpid();
    if (!same_process && sh->device != c->device) {                      /* the exporter's GPU is another one: copies go over the fabric */
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, c->device, sh->device) == hipSuccess && can) { const hipError_t e = hipDeviceEnablePeerAccess(sh->device, 0); if (e != hipSuccess) (void)hipGetLastError(); }
        else (void)hipGetLastError();
    }
    void *opened[4] = {nullptr, nullptr, nullptr, nullptr};
    struct Closer { void **o; ~Closer() { for (int k = 0; k < 4; k++) if (o[k]) { hipError_t e = hipIpcCloseMemHandle(o[k]); (void)e; } } } closer{opened};
    /* array k of the exporter as a pointer this process may copy from */
    auto source = [&](int k, const uint8_t handle[64], uint64_t off, uint64_t direct, const void **p) -> mtb_status {
        *p = nullptr;
        if (same_process) { *p = (const void *)(uintptr_t)direct; return MTB_OK; }
        hipIpcMemHandle_t h; memcpy(&h, handle, sizeof(h));
        HIPCHK(hipIpcOpenMemHandle(&opened[k], h, hipIpcMemLazyEnablePeerAccess));
        *p = (const char *)opened[k] + off;
        return MTB_OK;
    };
    auto copy = [&](void *d, const void *s_, size_t bytes) -> mtb_status {
        const size_t CH = 1ull << 30;
        for (size_t o = 0; o < bytes; o += CH) HIPCHK(hipMemcpyAsync((char *)d + o, (const char *)s_ + o, std::min(CH, bytes - o), hipMemcpyDefault, c->stream));
        return MTB_OK;
    };
    const void *sp = nullptr;
    STCHK(source(0, sh->values_handle, sh->values_off, sh->values_ptr, &sp));
    HIPCHK(hipMalloc((void **)&ix->d_values, (ix->T + 1) * 8));
    STCHK(copy(ix->d_values, sp, ix->T * 8));
    if (sh->has_info) {
        STCHK(source(1, sh->info_handle, sh->info_off, sh->info_ptr, &sp));
        HIPCHK(hipMalloc((void **)&ix->d_info, std::max<uint64_t>(ix->T, 1) * 4));
        STCHK(copy(ix->d_info, sp, ix->T * 4));
    }
    if (sh->dir_depth > 0) {
        const uint32_t n_groups = (sh->dir_buckets >> 16) + 1;
        HIPCHK(hipMalloc((void **)&ix->d_dir, ((size_t)sh->dir_buckets + 1) * 4));
        HIPCHK(hipMalloc((void **)&ix->d_dirbase, ((size_t)n_groups + 2) * 8));
        STCHK(source(2, sh->dir_handle, sh->dir_off, sh->dir_ptr, &sp));
        STCHK(copy(ix->d_dir, sp, ((size_t)sh->dir_buckets + 1) * 4));
        STCHK(source(3, sh->dirbase_handle, sh->dirbase_off, sh->dirbase_ptr, &sp));
        STCHK(copy(ix->d_dirbase, sp, ((size_t)n_groups + 2) * 8));
        ix->dir_L = sh->dir_depth; ix->dir_buckets = sh->dir_buckets;
    }
    HIPCHK(hipStreamSynchronize(c->stream));
    ix->packed = sh->packed != 0;
    guard.ix = nullptr;
    *out = ix;
    return MTB_OK;
}

void mtb_index_close(mtb_index *ix) {
    if (!ix) return;
    hipError_t e = hipSuccess;
    for (int k = 0; k < 4; k++) if (ix->ipc_stage[k]) { e = hipFree(ix->ipc_stage[k]); ix->ipc_stage[k] = nullptr; }
    if (ix->parent) {                /* a view: the parent may be packed again once the last view is gone */
        { std::lock_guard<std::mutex> lk(ix->parent->state_mu); ix->parent->views--; }
        if (ix->d_dir) e = hipFree(ix->d_dir);
        if (ix->d_dirbase) e = hipFree(ix->d_dirbase);
        (void)e;
        delete ix; return;
    }
    if (!ix->own && ix->packed && ix->d_values) {
        /* a borrowed target array (mtb_index_from_device) is handed back the way it was lent: flat.  A sealed index has let go of
         * info[] (the lender was told so and may have freed it): only the values are restored then. */
        std::unique_lock<std::mutex> lk(ix->state_mu);
        ix->state_cv.wait(lk, [&] { return ix->users == 0; });
        if (hipSetDevice(ix->ctx->device) == hipSuccess) {
            hipLaunchKernelGGL(k_index_unpack, dim3((uint32_t)std::min<uint64_t>(((uint64_t)ix->dir_buckets + 255) / 256, 1u << 20)), dim3(256), 0, ix->ctx->stream,
                               ix->d_values, ix->d_info, dir_view(ix));
            e = hipStreamSynchronize(ix->ctx->stream);
        }
        ix->packed = false;
    }
    if (ix->own) { if (ix->d_values) e = hipFree(ix->d_values); if (ix->d_info) e = hipFree(ix->d_info); }
    else if (ix->info_owned && ix->d_info) e = hipFree(ix->d_info);
    if (ix->d_dir) e = hipFree(ix->d_dir);
    if (ix->d_dirbase) e = hipFree(ix->d_dirbase);
    if (ix->d_canon) e = hipFree(ix->d_canon);
    if (ix->d_parent) e = hipFree(ix->d_parent);
    if (ix->d_depth) e = hipFree(ix->d_depth);
    if (ix->d_spparent) e = hipFree(ix->d_spparent);
    if (ix->d_tax2species) e = hipFree(ix->d_tax2species);
    if (ix->d_under) e = hipFree(ix->d_under);
    if (ix->d_accleaf) e = hipFree(ix->d_accleaf);
    if (ix->d_node) e = hipFree(ix->d_node);
    (void)e;
    delete ix;
}
uint64_t mtb_index_num_targets(const mtb_index *ix) { return ix ? ix->T : 0; }
mtb_status mtb_index_open_stats(const mtb_index *ix, uint64_t *out4) {
    if (!ix || !out4) return fail(MTB_ERR_ARG, "NULL argument");
    out4[0] = ix->open_chunks; out4[1] = ix->open_chunk_words; out4[2] = ix->open_peak_bytes; out4[3] = (ix->packed && !ix->d_info) ? 1 : 0;
    return MTB_OK;
}
mtb_status mtb_index_state(const mtb_index *ix, int32_t *dir_depth, int32_t *packed, int32_t *sealed) {
    if (!ix) return fail(MTB_ERR_ARG, "NULL index");
    const mtb_index *o = ix->parent ? ix->parent : ix;
    if (dir_depth) *dir_depth = ix->d_dir ? ix->dir_L : 0;
    if (packed) *packed = o->packed ? 1 : 0;
    if (sealed) *sealed = (o->packed && !o->d_info) ? 1 : 0;
    return MTB_OK;
}

mtb_status mtb_index_seal(mtb_index *ix) {
    if (!ix) return fail(MTB_ERR_ARG, "NULL index");
    mtb_ctx *c = ix->ctx;
    HIPCHK(hipSetDevice(c->device));
    if (!ix->d_dir || ix->dir_L != 7 || !ix->own_tax) return fail(MTB_ERR_UNSUPPORTED, "index has no depth-7 directory (too small, or a view): nothing to seal");
    std::unique_lock<std::mutex> lk(ix->state_mu);
    ix->state_cv.wait(lk, [&] { return ix->users == 0; });
    if (ix->views) return fail(MTB_ERR_UNSUPPORTED, "index has live views (mtb_index_slice): close them before sealing");
    STCHK(ensure_packed_locked(ix));
    if (!ix->packed) return fail(MTB_ERR_UNSUPPORTED, "packing is disabled");
    if (ix->d_info && (ix->own || ix->info_owned)) { hipError_t e = hipFree(ix->d_info); (void)e; }
    ix->d_info = nullptr; ix->info_owned = false;       /* a borrowed info[] now belongs to the caller alone */
    return MTB_OK;
}

mtb_status mtb_index_download(mtb_index *ix, uint64_t *values, uint32_t *info, uint64_t cap) {
    if (!ix) return fail(MTB_ERR_ARG, "NULL index");
    if (cap < ix->T) return fail(MTB_ERR_CAPACITY, "output too small");
    if (ix->T == 0) return MTB_OK;
    STCHK(ensure_flat(ix));
    if (values) HIPCHK(hipMemcpy(values, ix->d_values, ix->T * 8, hipMemcpyDeviceToHost));
    if (info) HIPCHK(hipMemcpy(info, ix->d_info, ix->T * 4, hipMemcpyDeviceToHost));
    return MTB_OK;
}
/* IndexCreator::writeTargetFilesAndSplits + writeDbParameters (IndexCreator.cpp:817-892, 1251-1272) and the
 * taxID_list dump (:329-333) for an index that is resident on the device -- e.g. a synthetic one.  The delta coder runs on the
 * device, a slice of targets at a time (words per entry -> exclusive scan -> every entry writes its 15-bit groups at its offset), the
 * coded slice and its info entries cross PCIe into pinned buffers and a writer thread appends them to the files while the next slice
 * is coded; the split checkpoints -- armed at every size_of_split-th entry, recorded at the first later entry of another amino-acid
 * part -- are located by one small kernel up front and get their word offsets from the slice that holds them. */
mtb_status mtb_index_write(const mtb_index *cix, const char *dbdir, int split_num) {
    if (!cix || !dbdir || split_num < 2) return fail(MTB_ERR_ARG, "NULL argument / split_num < 2");
    mtb_index *ix = const_cast<mtb_index *>(cix);
    mtb_ctx *c = ix->ctx;
    HIPCHK(hipSetDevice(c->device));
    STCHK(ensure_flat(ix));
    hipStream_t st = c->stream;
    const std::string d(dbdir);
    const int fd = open((d + "/diffIdx").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644), fi = open((d + "/info").c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644);
    struct Files { int a, b; ~Files() { if (a >= 0) close(a); if (b >= 0) close(b); } } files{fd, fi};
    if (fd < 0 || fi < 0) return fail(MTB_ERR_IO, "cannot create diffIdx/info in " + d);
    struct Split { uint64_t ad, diff_off, info_off; };
    std::vector<Split> splits((size_t)split_num, Split{0, 0, 0});
    const uint64_t n = ix->T;
    const uint64_t size_of_split = n / (uint64_t)(split_num - 1);
    /* where the checkpoints fall: j[k] for every arming position k * size_of_split; equal neighbours are one checkpoint (an arming
     * inside a run that is still armed changes nothing); at most split_num - 1 are kept */
    std::vector<uint64_t> cps;
    if (size_of_split && n) {
        const uint64_t n_k64 = n / size_of_split;
        const uint32_t n_k = (uint32_t)std::min<uint64_t>(n_k64, 1u << 24);
        uint64_t *d_j;
        STCHK(ensure(c, "wsplitj", n_k, &d_j));
        hipLaunchKernelGGL(k_split_find, dim3((n_k + 255) / 256), dim3(256), 0, st, (const uint64_t *)ix->d_values, n, size_of_split, n_k, d_j);
        HIPCHK(hipGetLastError());
        std::vector<uint64_t> hj(n_k);
        STCHK(d2h(c, hj.data(), d_j, (size_t)n_k * 8));
        for (uint32_t k = 0; k < n_k && cps.size() + 1 < (size_t)split_num; k++) if (hj[k] < n && (cps.empty() || cps.back() != hj[k])) cps.push_back(hj[k]);
        release(c, "wsplitj");
    }
    const uint64_t SLICE = std::min<uint64_t>(1ull << 25, std::max<uint64_t>(n, 1));
    uint32_t *d_nw; uint64_t *d_off, *d_ws; uint16_t *d_enc; uint8_t *d_seen; int32_t *d_extra; uint32_t *d_nextra; uint64_t *d_cpj, *d_cpv, *d_cpo;
    const uint32_t EXTRA_CAP = 1u << 20;
    STCHK(ensure(c, "wnw", SLICE, &d_nw)); STCHK(ensure(c, "woff", SLICE + 1, &d_off)); STCHK(ensure(c, "scanws", scan_ws_elems(SLICE + 1), &d_ws));
    STCHK(ensure(c, "wenc", SLICE * 5, &d_enc)); STCHK(ensure(c, "wseen", (size_t)ix->tax.max_id + 2, &d_seen));
    STCHK(ensure(c, "wextra", EXTRA_CAP + 1, &d_extra)); d_nextra = (uint32_t *)(d_extra + EXTRA_CAP);
    STCHK(ensure(c, "wcp", 3 * (cps.size() + 1), &d_cpj)); d_cpv = d_cpj + cps.size() + 1; d_cpo = d_cpv + cps.size() + 1;
    HIPCHK(hipMemsetAsync(d_seen, 0, (size_t)ix->tax.max_id + 2, st)); HIPCHK(hipMemsetAsync(d_nextra, 0, 4, st));
    if (!cps.empty()) STCHK(h2d(c, d_cpj, cps.data(), cps.size() * 8));
    struct Pinned { void *e[2] = {nullptr, nullptr}, *i[2] = {nullptr, nullptr}; ~Pinned() { for (int k = 0; k < 2; k++) { if (e[k]) { hipError_t x = hipHostFree(e[k]); (void)x; } if (i[k]) { hipError_t x = hipHostFree(i[k]); (void)x; } } } } pin;
    for (int k = 0; k < 2; k++)
        if (hipHostMalloc(&pin.e[k], SLICE * 10, hipHostMallocDefault) != hipSuccess || hipHostMalloc(&pin.i[k], SLICE * 4, hipHostMallocDefault) != hipSuccess) {
            (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no pinned host memory for the index download"); }
    std::thread writers[2]; bool write_ok = true;         /* writers[k]: appends the slice that sits in buffer pair k; one at a time, in slice order */
    struct Join { std::thread *t; ~Join() { for (int k = 0; k < 2; k++) if (t[k].joinable()) t[k].join(); } } join_writers{writers};
    uint64_t diff_count = 0; size_t cp_at = 0; int cur = 0;
    std::vector<uint64_t> cpv, cpo;
    for (uint64_t s0 = 0; s0 < n; s0 += SLICE, cur ^= 1) {
        const uint64_t m = std::min<uint64_t>(SLICE, n - s0);
        const dim3 grid((uint32_t)((m + 255) / 256));
        hipLaunchKernelGGL(k_diff_nwords, grid, dim3(256), 0, st, (const uint64_t *)ix->d_values, s0, m, d_nw);
        scan_launch<uint32_t, uint64_t, false>(st, d_nw, m, true, d_off, d_ws);
        hipLaunchKernelGGL(k_diff_encode, grid, dim3(256), 0, st, (const uint64_t *)ix->d_values, s0, m, (const uint64_t *)d_off, d_enc);
        hipLaunchKernelGGL(k_mark_taxids, grid, dim3(256), 0, st, (const uint32_t *)ix->d_info, s0, m, d_seen, ix->tax.max_id, d_extra, EXTRA_CAP, d_nextra);
        HIPCHK(hipGetLastError());
        uint64_t words = 0;
        STCHK(d2h(c, &words, d_off + m, 8));
        /* checkpoints inside this slice: value and word offset behind the entry */
        size_t cp_hi = cp_at;
        while (cp_hi < cps.size() && cps[cp_hi] < s0 + m) cp_hi++;
        if (cp_hi > cp_at) {
            const uint32_t q = (uint32_t)(cp_hi - cp_at);
            hipLaunchKernelGGL(k_split_gather, dim3((q + 63) / 64), dim3(64), 0, st, (const uint64_t *)ix->d_values, (const uint64_t *)d_off, s0, (const uint64_t *)(d_cpj + cp_at), q, d_cpv, d_cpo);
            HIPCHK(hipGetLastError());
            cpv.resize(q); cpo.resize(q);
            STCHK(d2h(c, cpv.data(), d_cpv, (size_t)q * 8)); STCHK(d2h(c, cpo.data(), d_cpo, (size_t)q * 8));
            for (uint32_t t = 0; t < q; t++) splits[cp_at + t + 1] = Split{cpv[t], diff_count + cpo[t], cps[cp_at + t] + 1};
            cp_at = cp_hi;
        }
        if (writers[cur].joinable()) writers[cur].join();  /* (two slices back: its buffers are `cur`'s) */
        if (!write_ok) return fail(MTB_ERR_IO, "short write while writing " + d);
        HIPCHK(hipMemcpyAsync(pin.e[cur], d_enc, words * 2, hipMemcpyDeviceToHost, st));          /* while the previous slice is being written from the other pair */
        HIPCHK(hipMemcpyAsync(pin.i[cur], ix->d_info + s0, m * 4, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        if (writers[cur ^ 1].joinable()) writers[cur ^ 1].join();          /* file order */
        if (!write_ok) return fail(MTB_ERR_IO, "short write while writing " + d);
        const void *pe = pin.e[cur], *pi = pin.i[cur];
        const uint64_t at_d = diff_count * 2, at_i = s0 * 4;
        /* several pwrite streams per file: one stream into the page cache tops out near 2-3 GB/s, a 16 G-target database is 150 GB */
        writers[cur] = std::thread([&write_ok, fd, fi, pe, pi, words, m, at_d, at_i] {
            bool ok_i = true;
            std::thread ti([&] { ok_i = pwrite_parallel(fi, pi, at_i, m * 4, 6); });
            const bool ok_d = pwrite_parallel(fd, pe, at_d, words * 2, 10);
            ti.join();
            if (!ok_d || !ok_i) write_ok = false;
        });
        diff_count += words;
    }
    for (int k = 0; k < 2; k++) if (writers[k].joinable()) writers[k].join();
    if (!write_ok) return fail(MTB_ERR_IO, "short write while writing " + d);
    std::vector<uint8_t> seen((size_t)ix->tax.max_id + 2, 0);
    STCHK(d2h(c, seen.data(), d_seen, seen.size()));
    uint32_t n_extra = 0;
    STCHK(d2h(c, &n_extra, d_nextra, 4));
    if (n_extra > EXTRA_CAP) return fail(MTB_ERR_UNSUPPORTED, "more than 2^20 info entries outside the taxonomy's id range");
    std::vector<int32_t> extra_ids(n_extra);           /* ids outside the taxonomy's range (kept for taxID_list) */
    if (n_extra) STCHK(d2h(c, extra_ids.data(), d_extra, (size_t)n_extra * 4));
    release(c, "wnw"); release(c, "woff"); release(c, "wenc"); release(c, "wseen"); release(c, "wextra"); release(c, "wcp");
    FILE *f = fopen((d + "/split").c_str(), "wb"); if (!f) return fail(MTB_ERR_IO, "cannot create " + d + "/split");
    fwrite(splits.data(), sizeof(Split), splits.size(), f); fclose(f);
    std::sort(extra_ids.begin(), extra_ids.end()); extra_ids.erase(std::unique(extra_ids.begin(), extra_ids.end()), extra_ids.end());
    f = fopen((d + "/taxID_list").c_str(), "w"); if (!f) return fail(MTB_ERR_IO, "cannot create " + d + "/taxID_list");
    {   size_t e = 0;
        for (size_t t = 0; t < seen.size(); t++) {
            while (e < extra_ids.size() && extra_ids[e] < (int32_t)t) fprintf(f, "%d\n", extra_ids[e++]);
            if (seen[t]) fprintf(f, "%zu\n", t);
        }
        while (e < extra_ids.size()) fprintf(f, "%d\n", extra_ids[e++]);
    }
    fclose(f);
    f = fopen((d + "/db.parameters").c_str(), "w"); if (!f) return fail(MTB_ERR_IO, "cannot create " + d + "/db.parameters");
    const mtb_params &p = ix->params;
    fprintf(f, "DB_name\t%s\nCreation_date\t-\nReduced_alphabet\t0\nAccession_level\t%d\n", "mtb", p.accession_level == 2 ? 1 : 0);
    fprintf(f, "Mask_mode\t0\nMask_prob\t0.900000\nSkip_redundancy\t%d\nSyncmer\t%d\n", p.skip_redundancy ? 1 : 0, p.syncmer);
    if (p.syncmer == 1) fprintf(f, "Syncmer_len\t%d\n", p.smer_len);
    fprintf(f, "Kmer_format\t%d\n", p.kmer_format);
    fclose(f);
    return MTB_OK;
}

int32_t mtb_tax_lca(const mtb_index *ix, int32_t a, int32_t b) { return ix->tax.lca(a, b); }
int32_t mtb_tax_species(const mtb_index *ix, int32_t t) { return (t >= 0 && t <= ix->tax.max_id) ? ix->tax.tax2species[(size_t)t] : 0; }
int32_t mtb_tax_parent(const mtb_index *ix, int32_t t) { int32_t c = ix->tax.cn(t); return c < 0 ? -1 : ix->tax.parent[(size_t)c]; }
int32_t mtb_tax_max_id(const mtb_index *ix) { return ix->tax.max_id; }
int32_t mtb_tax_original_id(const mtb_index *ix, int32_t t) { return (t >= 0 && t <= ix->tax.max_id) ? ix->tax.orig[(size_t)t] : t; }
int32_t mtb_tax_num_children(const mtb_index *ix, int32_t t) { int32_t c = ix->tax.cn(t); return c < 0 ? 0 : (int32_t)ix->tax.children_of(c).size(); }
int32_t mtb_tax_child(const mtb_index *ix, int32_t t, int32_t k) {
    int32_t c = ix->tax.cn(t); if (c < 0) return -1;
    const std::vector<int32_t> &v = ix->tax.children_of(c);
    return (k >= 0 && (size_t)k < v.size()) ? v[(size_t)k] : -1;
}
const char *mtb_tax_rank(const mtb_index *ix, int32_t t) { int32_t c = ix->tax.cn(t); return c < 0 ? "" : ix->tax.rank[(size_t)c].c_str(); }
const char *mtb_tax_name(const mtb_index *ix, int32_t t) { int32_t c = ix->tax.cn(t); return c < 0 ? "" : ix->tax.name[(size_t)c].c_str(); }

/* ------------------------------------------------------------------ */
/* stage-level entry points (host buffers)                             */
/* ------------------------------------------------------------------ */
static mtb_status upload_reads(mtb_ctx *c, const mtb_params *p, const char *bases, const uint64_t *offs, const char *bases2,
                               const uint64_t *offs2, uint64_t n_reads, char **d_b, uint64_t **d_o, char **d_b2, uint64_t **d_o2,
                               uint64_t *n_bases) {
    *d_b = nullptr; *d_o = nullptr; *d_b2 = nullptr; *d_o2 = nullptr; *n_bases = 0;
    if (n_reads == 0) return MTB_OK;
    if (!bases || !offs) return fail(MTB_ERR_ARG, "bases/offs NULL");
    uint64_t nb = offs[n_reads];
    STCHK(ensure(c, "bases", nb + 8, d_b)); STCHK(ensure(c, "offs", n_reads + 1, d_o));
    STCHK(h2d(c, *d_b, bases, nb)); STCHK(h2d(c, *d_o, offs, (n_reads + 1) * 8));
    *n_bases = nb;
    if (p->seq_mode == 2) {
        if (!bases2 || !offs2) return fail(MTB_ERR_ARG, "seq_mode 2 needs bases2/offs2");
        uint64_t nb2 = offs2[n_reads];
        STCHK(ensure(c, "bases2", nb2 + 8, d_b2)); STCHK(ensure(c, "offs2", n_reads + 1, d_o2));
        STCHK(h2d(c, *d_b2, bases2, nb2)); STCHK(h2d(c, *d_o2, offs2, (n_reads + 1) * 8));
        *n_bases += nb2;
    }
    return MTB_OK;
}

mtb_status mtb_extract(mtb_ctx *c, const mtb_params *p, const char *bases, const uint64_t *offs, const char *bases2, const uint64_t *offs2,
                       uint64_t n_reads, mtb_kmer *out, uint64_t cap, uint64_t *count, int32_t *qlen, int32_t *qlen2) {
    if (!c || !p || !count) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *count = 0;
    if (n_reads == 0) return MTB_OK;
    char *d_b, *d_b2; uint64_t *d_o, *d_o2; uint64_t nb;
    STCHK(upload_reads(c, p, bases, offs, bases2, offs2, n_reads, &d_b, &d_o, &d_b2, &d_o2, &nb));
    int32_t *d_ql, *d_ql2;
    STCHK(ensure(c, "qlen", n_reads, &d_ql)); STCHK(ensure(c, "qlen2", n_reads, &d_ql2));
    mtb_kmer *d_k; uint64_t n;
    STCHK(dev_extract(c, p, d_b, d_o, d_b2, d_o2, n_reads, &d_k, &n, d_ql, d_ql2, nullptr));
    *count = n;
    if (qlen) STCHK(d2h(c, qlen, d_ql, n_reads * 4));
    if (qlen2) STCHK(d2h(c, qlen2, d_ql2, n_reads * 4));
    if (n > cap) return fail(MTB_ERR_CAPACITY, "k-mer buffer too small");
    if (n) STCHK(d2h(c, out, d_k, n * sizeof(mtb_kmer)));
    return MTB_OK;
}

mtb_status mtb_sort_kmers(mtb_ctx *c, mtb_kmer *kmers, uint64_t n) {
    if (!c) return fail(MTB_ERR_ARG, "NULL ctx");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0) return MTB_OK;
    if (n >= (1ull << 32)) return fail(MTB_ERR_ARG, "n must be < 2^32");
    mtb_kmer *d_a, *d_s;
    STCHK(ensure(c, "kmersA", n, &d_a));
    STCHK(h2d(c, d_a, kmers, n * sizeof(mtb_kmer)));
    STCHK(dev_sort(c, d_a, n, 0, &d_s));
    STCHK(d2h(c, kmers, d_s, n * sizeof(mtb_kmer)));
    return MTB_OK;
}

mtb_status mtb_match_kmers(mtb_ctx *c, mtb_index *ix, const mtb_kmer *sorted, uint64_t n, mtb_match *out, uint64_t cap, uint64_t *count) {
    if (!c || !ix || !count) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *count = 0;
    if (n == 0) return MTB_OK;
    mtb_kmer *d_q; mtb_match *d_m;
    STCHK(ensure(c, "kmersA", n, &d_q));
    STCHK(h2d(c, d_q, sorted, n * sizeof(mtb_kmer)));
    STCHK(ensure(c, "jtemp", cap, &d_m));
    mtb_status st = dev_join(c, ix, d_q, n, d_m, cap, nullptr, count);
    if (st != MTB_OK) return st;
    if (*count) STCHK(d2h(c, out, d_m, *count * sizeof(mtb_match)));
    return MTB_OK;
}

mtb_status mtb_sort_matches(mtb_ctx *c, mtb_match *matches, uint64_t n, uint64_t n_reads) {
    if (!c) return fail(MTB_ERR_ARG, "NULL ctx");
    HIPCHK(hipSetDevice(c->device));
    if (n == 0 || n_reads == 0) return MTB_OK;
    mtb_match *d_in, *d_out; uint32_t *d_rc; uint64_t *d_seg;
    STCHK(ensure(c, "jtemp", n, &d_in)); STCHK(ensure(c, "matches", n, &d_out)); STCHK(ensure(c, "readcnt", n_reads, &d_rc));
    STCHK(h2d(c, d_in, matches, n * sizeof(mtb_match)));
    STCHK(count_reads(c, d_in, n, n_reads, d_rc));
    STCHK(dev_regroup(c, d_in, n, n_reads, d_rc, &d_seg, d_out));
    STCHK(dev_segsort(c, d_out, d_seg, n_reads, nullptr));
    STCHK(d2h(c, matches, d_out, n * sizeof(mtb_match)));
    return MTB_OK;
}

mtb_status mtb_score(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const mtb_match *sorted, uint64_t n_matches, uint64_t n_reads,
                     const int32_t *qlen, const int32_t *qlen2, mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt,
                     uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    if (!c || !ix || !p || !qlen || !results || !n_taxcnt) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    if (n_reads == 0) return MTB_OK;
    mtb_match *d_m; uint32_t *d_rc; uint64_t *d_seg; uint64_t *d_ws; int32_t *d_ql, *d_ql2;
    STCHK(ensure(c, "matches", n_matches, &d_m)); STCHK(ensure(c, "readcnt", n_reads, &d_rc));
    STCHK(ensure(c, "segstart", n_reads + 1, &d_seg)); STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    STCHK(ensure(c, "qlen", n_reads, &d_ql)); STCHK(ensure(c, "qlen2", n_reads, &d_ql2));
    STCHK(h2d(c, d_m, sorted, n_matches * sizeof(mtb_match)));
    STCHK(h2d(c, d_ql, qlen, n_reads * 4));
    if (qlen2) STCHK(h2d(c, d_ql2, qlen2, n_reads * 4)); else HIPCHK(hipMemsetAsync(d_ql2, 0, n_reads * 4, c->stream));
    STCHK(count_reads(c, d_m, n_matches, n_reads, d_rc));
    scan_launch<uint32_t, uint64_t, false>(c->stream, d_rc, n_reads, true, d_seg, d_ws);
    /* maxima for slab sizing */
    std::vector<uint32_t> rc(n_reads);
    STCHK(d2h(c, rc.data(), d_rc, n_reads * 4));
    uint32_t max_seg = 0, max_len = 0;
    for (uint64_t i = 0; i < n_reads; i++) { max_seg = std::max(max_seg, rc[i]); max_len = std::max<uint32_t>(max_len, (uint32_t)(qlen[i] + (qlen2 ? qlen2[i] : 0))); }
    mtb_result *d_res; int32_t *d_tt; uint32_t *d_tc;
    STCHK(ensure(c, "results", n_reads, &d_res)); STCHK(ensure(c, "tctax", taxcnt_cap, &d_tt)); STCHK(ensure(c, "tccnt", taxcnt_cap, &d_tc));
    ScoreSrc src; src.m = d_m; src.seg = d_seg; src.sort = false; src.max_seg = max_seg;
    mtb_status st = (p->seq_mode == 3 && !c->opt.no_long_scorer)
        ? dev_score_long(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_res, d_tt, d_tc, taxcnt_cap, n_taxcnt, 0, d_m, d_seg, max_seg)
        : dev_score(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_res, d_tt, d_tc, taxcnt_cap, n_taxcnt, 0, src, nullptr);
    if (st != MTB_OK) return st;
    STCHK(d2h(c, results, d_res, n_reads * sizeof(mtb_result)));
    if (*n_taxcnt) { STCHK(d2h(c, taxcnt_tax, d_tt, *n_taxcnt * 4)); STCHK(d2h(c, taxcnt_cnt, d_tc, *n_taxcnt * 4)); }
    return MTB_OK;
}

/* ------------------------------------------------------------------ */
/* fused batch                                                         */
/* ------------------------------------------------------------------ */
} // extern "C"

/* Exact-segment tail of the pipeline: matches in join order (d_tmp, clobbered: it becomes the sort scratch) +
 * per-read counts -> scan, regroup, big segments sorted (chunk sort in LDS + rank merges), scoring.
 * Records ev[4] (regrouped) and ev[5] (sorted). */
static mtb_status score_join_order(mtb_ctx *c, mtb_index *ix, const mtb_params *p, mtb_match *d_tmp, uint64_t nm, uint64_t n_reads, uint32_t *d_rc,
                                   const int32_t *d_ql, const int32_t *d_ql2, uint32_t max_len, mtb_result *d_results, int32_t *d_taxcnt_tax,
                                   uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt, uint64_t tc_base) {
    hipStream_t st = c->stream;
    mtb_match *d_m; uint64_t *d_seg;
    STCHK(ensure(c, "matches", nm, &d_m));
    STCHK(dev_regroup(c, d_tmp, nm, n_reads, d_rc, &d_seg, d_m));
    HIPCHK(hipEventRecord(c->ev[4], st));
    /* segments that fit LDS are sorted inside k_score; only the big ones are sorted here.  Long reads: k_score_long streams sorted
     * segments, so every segment of two or more matches is sorted here */
    const bool long_scorer = p->seq_mode == 3 && !c->opt.no_long_scorer;
    uint32_t max_seg = 0;
    {
        uint32_t *d_large;
        STCHK(ensure(c, "large", n_reads, &d_large));
        HIPCHK(hipMemsetAsync(c->d_scal + 2, 0, 16, st));
        { KTimer kt(c, MTB_K_SEGSORT);
        hipLaunchKernelGGL(k_list_large, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, st, (const uint64_t *)d_seg, n_reads,
                           long_scorer ? 1u : (uint32_t)MTB_SCORE_LDS, d_large, (uint32_t *)(c->d_scal + 2), (uint32_t *)(c->d_scal + 3)); }
        uint64_t sc[2];
        STCHK(d2h(c, sc, c->d_scal + 2, 16));
        max_seg = (uint32_t)sc[1];
        if (sc[0]) {
            /* big segments: chunk sort in LDS + rank merges; the join-order buffer is the scratch */
            const size_t lds = (size_t)MTB_SEGLDS_CHUNK * 14;
            HIPCHK(hipFuncSetAttribute((const void *)k_segsort_lds<mtb_match>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            KTimer kt(c, MTB_K_SEGSORT);
            hipLaunchKernelGGL((k_segsort_lds<mtb_match>), dim3(std::min<uint32_t>((uint32_t)sc[0], 2048)), dim3(MTB_SEGLDS_THREADS), lds, st, d_m,
                               (const uint64_t *)d_seg, (const uint32_t *)d_large, (const uint32_t *)(c->d_scal + 2), d_tmp);
        }
    }
    HIPCHK(hipEventRecord(c->ev[5], st));
    if (long_scorer) return dev_score_long(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base, d_m, d_seg, max_seg);
    ScoreSrc a; a.m = d_m; a.seg = d_seg; a.sort = true; a.max_seg = max_seg;
    return dev_score(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base, a, nullptr);
}

/* The slot segments of a short-read batch: n_reads x stride 16-byte slots in buffer "segm".  Live slots carry the batch's epoch tag;
 * the buffer is cleared only when it is new or the tag wraps. */
static mtb_status prepare_slots(mtb_ctx *c, uint64_t n_reads, uint32_t stride, mtb_slot16 **out, uint32_t *epoch_out) {
    hipStream_t st = c->stream;
    mtb_slot16 *d_segm;
    DevBuf &sb = c->bufs["segm"];
    STCHK(ensure_placed(c, "segm", n_reads * (uint64_t)stride, &d_segm));
    const char *clear_mode = c->opt.segm_clear[0] ? c->opt.segm_clear : nullptr;      /* experiment switch: kernel | sync | always (default: hipMemsetAsync when new / on epoch wrap) */
    const bool always = clear_mode && !strcmp(clear_mode, "always");
    /* cleared when the allocation is not the one this function cleared last (new, grown, or created by mtb_ctx_reserve: never
     * written by a clear), or when the tag wraps */
    if (sb.p != c->seg_clean_p || sb.cap != c->seg_clean_cap || c->seg_epoch == 0 || c->seg_epoch >= MTB_SLOT_EPOCHS || always) {
        if (clear_mode && !strcmp(clear_mode, "kernel")) hipLaunchKernelGGL(k_clear_words, dim3(2048), dim3(256), 0, st, (uint64_t *)sb.p, (uint64_t)(sb.cap / 8));
        else HIPCHK(hipMemsetAsync(sb.p, 0, sb.cap, st));
        if (clear_mode && !strcmp(clear_mode, "sync")) HIPCHK(hipDeviceSynchronize());
        c->seg_epoch = 0; c->seg_clean_p = sb.p; c->seg_clean_cap = sb.cap;
    }
    c->seg_epoch++;
    *out = d_segm; *epoch_out = c->seg_epoch;
    return MTB_OK;
}
static void slot_geometry(const mtb_ctx *c, uint32_t max_q, uint32_t *direct, uint32_t *stride) {
    const uint32_t tail_min = c->opt.tail_min > 0 ? (uint32_t)std::max(8, c->opt.tail_min) & ~7u : 16u;      /* experiment switch: longer tails */
    *direct = std::max<uint32_t>(8, (max_q + 7) & ~7u);
    *stride = *direct + std::max<uint32_t>(tail_min, (*direct / 8 + 7) & ~7u);
}

/* Scoring out of filled slot segments (fused path after the join; partitioned path after the matches came home): the
 * register-resident scorer, the generic one for the reads it flags, exact segments for the reads neither can take from their slots.
 * d_rc = per-read tail cursors, d_ovf / n_ovf = the overflow list.  *nm = matches seen (statistics). */
static mtb_status score_fixed_slots(mtb_ctx *c, mtb_index *ix, const mtb_params *p, uint64_t n_reads, const int32_t *d_ql, const int32_t *d_ql2, uint32_t max_len,
                                    uint64_t nk_real, mtb_slot16 *d_segm, uint32_t *d_rc, uint32_t stride, uint32_t direct, uint32_t epoch, mtb_match *d_ovf, uint64_t n_ovf,
                                    mtb_result *d_results, int32_t *d_taxcnt_tax, uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt, uint64_t tc_base, uint64_t *nm_out,
                                    uint32_t max_len_deferred = 0, const uint8_t *d_off_reads = nullptr, uint64_t ovf_region = 0 /* != 0: the overflow list is striped (dev_join) */) {
    hipStream_t st = c->stream;
    uint64_t nm = 0;
    memset(c->many_stats, 0, sizeof(c->many_stats));
    uint32_t *d_biglist, *d_bigcnt, *d_bigidx, *d_bigcur, *d_cnt; uint64_t *d_bigstart = nullptr, *d_ws2, *d_tot; mtb_match *d_big = nullptr;
    STCHK(ensure(c, "biglist", n_reads, &d_biglist)); STCHK(ensure(c, "bigidx", n_reads, &d_bigidx)); STCHK(ensure(c, "livecnt", n_reads, &d_cnt));
    STCHK(ensure(c, "segstart", n_reads + 1, &d_tot)); STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws2));
    HIPCHK(hipMemsetAsync(d_cnt, 0, n_reads * 4, st));
    HIPCHK(hipMemsetAsync(c->d_scal + 2, 0, 32, st));            /* [2] unused, [3] max big segment, [5] reads deferred by the first launch */
    uint64_t big_total = 0;
    ScoreSrc a; a.m = (const mtb_match *)d_segm; a.cursor = d_rc;       /* slot mode: 16-byte slot records behind the pointer */ a.stride = stride; a.direct = direct; a.epoch = epoch; a.sort = true;
    /* LDS staging of the scorer = smallest instantiation that holds one match for every metamer of the longest read
     * (144 records: 16 waves per CU, 160: 14, 224: 10, 288: 8, 320: 7); reads with more live records are deferred */
    {   /* sized for the typical read (mean metamer count + 12 %), not the longest: the few reads beyond it take the deferred
           path, and 144 instead of 160 records is worth 10 % of the kernel on 150 bp reads (47.1 vs 52.3 ms) */
        const uint32_t want = std::min<uint32_t>(direct, (uint32_t)((double)nk_real / (double)n_reads * 1.12) + 1);
        a.cap = want <= 144 ? 144 : want <= 160 ? 160 : want <= 224 ? 224 : want <= 288 ? 288 : 320;
    }
    a.max_seg = a.cap;
    a.big_list = d_biglist; a.n_big = (uint32_t *)(c->d_scal + 5); a.cnt_out = d_cnt;
    /* reads the first launch could not take from their slots: exact segments (live slots + overflow list), sorted in HBM */
    ScoreSecond second = [&](ScoreSrc *b, bool *go, const ScoreLaunch &SL) -> mtb_status {
        uint64_t sc = 0;
        STCHK(d2h(c, &sc, c->d_scal + 5, 8));
        uint32_t n_big = (uint32_t)sc;
        *go = n_big != 0;
        if (!n_big) return MTB_OK;
        c->many_stats[0] = n_big; c->many_stats[1] = 0;
        const bool no_many = c->opt.no_score_many != 0;        /* A/B switch: every deferred read through exact segments (round 4's path) */
        uint32_t *d_novf = nullptr; uint64_t *d_ostart = nullptr; mtb_match *d_ovfg = nullptr;      /* the overflow list grouped by read, once the many-species path has built it */
        if (!no_many && stride <= 384u) {
            /* ---- the reads of conserved genes (kernels_score_many.h): scored straight from their slots + their overflow entries, dead
             * species dropped before anything is ordered.  The overflow list is grouped by read first (counts from the tail cursors). ---- */
            KTimer ktm(c, MTB_K_SCORE_MANY);
            uint32_t *d_ocur = nullptr, *d_rest;
            STCHK(ensure(c, "restlist", n_reads, &d_rest));
            const bool have_ovf = ovf_region ? c->ovf_max_region != 0 : n_ovf != 0;
            if (have_ovf) {
                STCHK(ensure(c, "novf", n_reads, &d_novf)); STCHK(ensure(c, "ocur", n_reads, &d_ocur)); STCHK(ensure(c, "ovfstart", n_reads + 1, &d_ostart));
                STCHK(scratch(c, "ovfg", n_ovf + 1, &d_ovfg));
                HIPCHK(hipMemsetAsync(d_ocur, 0, n_reads * 4, st));
                hipLaunchKernelGGL(k_ovf_count, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, st, (const uint32_t *)d_rc, d_off_reads, n_reads, stride - direct, d_novf);
                scan_launch<uint32_t, uint64_t, false>(st, d_novf, n_reads, true, d_ostart, d_ws2);
                if (ovf_region) hipLaunchKernelGGL(k_ovf_group, dim3((uint32_t)((c->ovf_max_region + 255) / 256), MTB_OVF_STRIPES), dim3(256), 0, st, (const mtb_match *)d_ovf, n_ovf,
                                                   (const uint64_t *)d_ostart, (const uint32_t *)d_novf, d_ocur, d_ovfg, ovf_region, (const unsigned long long *)c->d_ovfctr);
                else hipLaunchKernelGGL(k_ovf_group, dim3((uint32_t)((n_ovf + 255) / 256)), dim3(256), 0, st, (const mtb_match *)d_ovf, n_ovf,
                                        (const uint64_t *)d_ostart, (const uint32_t *)d_novf, d_ocur, d_ovfg, (uint64_t)0, (const unsigned long long *)nullptr);
            }
            unsigned long long *d_ms;                                  /* [0] reads handed on, [1] work counter, [2] matches seen, [3] survivors, [4..6] hand-over reasons */
            STCHK(ensure(c, "manystat", 16, &d_ms));
            HIPCHK(hipMemsetAsync(d_ms, 0, 16 * 8, st));
            const bool cap192 = c->opt.many_cap ? c->opt.many_cap <= 192 : stride <= 192u;
            const uint32_t gridm = std::min<uint32_t>(n_big, 256u * (cap192 ? 12u : 7u));
#define MTB_LAUNCH_MANY(K64, CAPV) hipLaunchKernelGGL((k_score_many<K64, CAPV>), dim3(gridm), dim3(64), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch, (const uint32_t *)d_rc, d_off_reads, \
            (const mtb_match *)d_ovfg, (const uint64_t *)d_ostart, (const uint32_t *)d_biglist, (const uint32_t *)(c->d_scal + 5), SL.d_qlen, SL.d_qlen2, tax_view(ix), SL.sp, SL.d_tcoff, SL.d_res, \
            SL.d_tc_tax, SL.d_tc_cnt, SL.tc_cap, SL.tc_base, d_rest, (uint32_t *)d_ms, d_cnt, d_ms + 1, d_ms + 2)
            if (cap192) { if (SL.key64) MTB_LAUNCH_MANY(true, 192); else MTB_LAUNCH_MANY(false, 192); }
            else { if (SL.key64) MTB_LAUNCH_MANY(true, 320); else MTB_LAUNCH_MANY(false, 320); }
#undef MTB_LAUNCH_MANY
            HIPCHK(hipGetLastError());
            uint64_t ms4[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            STCHK(d2h(c, ms4, d_ms, 64));
            const uint32_t n_rest = (uint32_t)(ms4[0] & 0xFFFFFFFFull);
            c->many_stats[1] = n_big - n_rest; c->many_stats[2] = ms4[2]; c->many_stats[3] = ms4[3];
            if (c->opt.many_verbose) fprintf(stderr, "mtb: k_score_many: %u reads listed, %u handed on (routed off / buckets / counts %llu, species table full %llu, survivors beyond the staging %llu); %llu matches, %llu survive the dead-species drop\n",
                                                    n_big, n_rest, (unsigned long long)ms4[4], (unsigned long long)ms4[5], (unsigned long long)ms4[6], (unsigned long long)ms4[2], (unsigned long long)ms4[3]);
            n_big = n_rest;
            const unsigned long long *d_nleft = d_ms;                   /* where the number of reads still unscored sits on the device */
            if (n_big && SL.key64 && !c->opt.no_many_sort) {
                /* ---- reads beyond the staging (thousands of matches: conserved genes of organisms that are not in the index): a workgroup per
                 * read gathers, drops dead species, sorts in LDS and writes the read's exact segment; k_score_long streams it ---- */
                uint32_t *d_bc3, *d_segcnt, *d_rest2; uint64_t *d_bs3; uint8_t *d_todo; mtb_match *d_big3;
                STCHK(ensure(c, "bigcnt", n_big, &d_bc3)); STCHK(ensure(c, "bigstart", (uint64_t)n_big + 1, &d_bs3)); STCHK(ensure(c, "bigseg", n_big, &d_segcnt));
                STCHK(ensure(c, "restlist2", n_reads, &d_rest2)); STCHK(ensure(c, "todoflag", n_reads, &d_todo));
                hipLaunchKernelGGL(k_big_count, dim3(std::min<uint32_t>(n_big, 4096)), dim3(64), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch,
                                   (const uint32_t *)d_rc, (const uint32_t *)d_rest, n_big, d_bc3, d_bigidx, (uint32_t *)(c->d_scal + 3), d_off_reads);
                scan_launch<uint32_t, uint64_t, false>(st, d_bc3, n_big, true, d_bs3, d_ws2);
                uint64_t total3 = 0;
                STCHK(d2h(c, &total3, d_bs3 + n_big, 8));
                STCHK(scratch(c, "bigm", total3 + 1, &d_big3));
                HIPCHK(hipMemsetAsync(d_todo, 0, n_reads, st));
                /* the small instantiation first (20 KB of LDS: eight workgroups per CU), the big one for the reads it flags 2 */
                hipLaunchKernelGGL((k_many_sort<11, 2048u>), dim3(std::min<uint32_t>(n_big, 256u * 8u)), dim3(MTB_MSORT_NT), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch, (const uint32_t *)d_rc, d_off_reads,
                                   (const mtb_match *)d_ovfg, (const uint64_t *)d_ostart, (const uint32_t *)d_rest, n_big, SL.d_qlen, SL.d_qlen2, SL.sp.dna_shift, (const uint64_t *)d_bs3, d_big3, d_segcnt, d_todo, 0u, 2u);
                hipLaunchKernelGGL((k_many_sort<12, 4096u>), dim3(std::min<uint32_t>(n_big, 256u * 3u)), dim3(MTB_MSORT_NT), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch, (const uint32_t *)d_rc, d_off_reads,
                                   (const mtb_match *)d_ovfg, (const uint64_t *)d_ostart, (const uint32_t *)d_rest, n_big, SL.d_qlen, SL.d_qlen2, SL.sp.dna_shift, (const uint64_t *)d_bs3, d_big3, d_segcnt, d_todo, 2u, 1u);
                hipLaunchKernelGGL((k_score_long<2048, 256, 256, 256>), dim3(std::min<uint32_t>(n_big, 256u * 5u)), dim3(MTB_LONG_NT), 0, st, (const mtb_match *)d_big3, (const uint64_t *)d_bs3, n_reads, SL.d_qlen, SL.d_qlen2,
                                   tax_view(ix), SL.sp, SL.d_tcoff, SL.d_res, SL.d_tc_tax, SL.d_tc_cnt, SL.tc_cap, SL.tc_base, d_todo, d_ms + 8, (const uint32_t *)d_segcnt, (const uint32_t *)d_rest, n_big);
                hipLaunchKernelGGL(k_list_flagged, dim3((n_big + 255) / 256), dim3(256), 0, st, (const uint32_t *)d_rest, n_big, (const uint8_t *)d_todo, (const uint32_t *)d_bc3, d_rest2, (uint32_t *)(d_ms + 9), d_cnt);
                HIPCHK(hipGetLastError());
                uint64_t left = 0;
                STCHK(d2h(c, &left, d_ms + 9, 8));
                if (c->opt.many_verbose) fprintf(stderr, "mtb: k_many_sort + k_score_long: %u reads (%llu records), %llu left for the exact-segment path\n", n_big, (unsigned long long)total3, (unsigned long long)(left & 0xFFFFFFFFull));
                n_big = (uint32_t)(left & 0xFFFFFFFFull);
                d_rest = d_rest2; d_nleft = d_ms + 9;
            }
            c->many_stats[1] = c->many_stats[0] - n_big;
            *go = n_big != 0;
            if (!n_big) return MTB_OK;
            /* what is left goes the old way: the list of those reads, its length where the launches below read it */
            d_biglist = d_rest;
            HIPCHK(hipMemcpyAsync(c->d_scal + 5, d_nleft, 8, hipMemcpyDeviceToDevice, st));
        }
        KTimer kt(c, MTB_K_SEGSORT);
        HIPCHK(hipMemsetAsync(d_bigidx, 0xFF, n_reads * 4, st));        /* reads that are not listed (k_score_many took them) own entries of the overflow list too: k_big_ovf skips them */
        STCHK(ensure(c, "bigcnt", n_big, &d_bigcnt)); STCHK(ensure(c, "bigstart", (uint64_t)n_big + 1, &d_bigstart)); STCHK(ensure(c, "bigcur", n_big, &d_bigcur));
        hipLaunchKernelGGL(k_big_count, dim3(std::min<uint32_t>(n_big, 4096)), dim3(64), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch,
                           (const uint32_t *)d_rc, (const uint32_t *)d_biglist, n_big, d_bigcnt, d_bigidx, (uint32_t *)(c->d_scal + 3), d_off_reads,
                           (const uint32_t *)(d_ovfg ? d_novf : nullptr), (unsigned long long *)(c->d_scal + 2));
        scan_launch<uint32_t, uint64_t, false>(st, d_bigcnt, n_big, true, d_bigstart, d_ws2);
        uint64_t mx = 0, n_ungrouped = 0;
        STCHK(d2h(c, &big_total, d_bigstart + n_big, 8));
        STCHK(d2h(c, &mx, c->d_scal + 3, 8));
        STCHK(d2h(c, &n_ungrouped, c->d_scal + 2, 8));
        STCHK(scratch(c, "bigm", big_total, &d_big));
        hipLaunchKernelGGL(k_big_copy, dim3(std::min<uint32_t>(n_big, 4096)), dim3(64), 0, st, (const mtb_slot16 *)d_segm, stride, direct, epoch,
                           (const uint32_t *)d_rc, (const uint32_t *)d_biglist, (const uint64_t *)d_bigstart, n_big, d_bigcur, d_big,
                           (const mtb_match *)d_ovfg, (const uint64_t *)d_ostart, (const uint32_t *)(d_ovfg ? d_novf : nullptr));
        /* the pass over the whole overflow list: only for listed reads whose entries are not in the grouped list (all of them when there is none) */
        const uint32_t *skip_grouped = d_ovfg ? d_novf : nullptr;
        if (d_ovfg && n_ungrouped == 0) { /* every listed read's entries came from its group */ }
        else if (ovf_region) {            /* striped list (the directory join's): every stripe's entries */
            if (c->ovf_max_region) hipLaunchKernelGGL(k_big_ovf, dim3((uint32_t)((c->ovf_max_region + 255) / 256), MTB_OVF_STRIPES), dim3(256), 0, st, (const mtb_match *)d_ovf, n_ovf,
                                                      (const uint32_t *)d_bigidx, (const uint64_t *)d_bigstart, d_bigcur, d_big, ovf_region, (const unsigned long long *)c->d_ovfctr, skip_grouped);
        } else
        if (n_ovf) hipLaunchKernelGGL(k_big_ovf, dim3((uint32_t)((n_ovf + 255) / 256)), dim3(256), 0, st, (const mtb_match *)d_ovf, n_ovf,
                                      (const uint32_t *)d_bigidx, (const uint64_t *)d_bigstart, d_bigcur, d_big, (uint64_t)0, (const unsigned long long *)nullptr, skip_grouped);
        /* segments of up to 512 matches (nearly all: a read of a conserved gene brings a few hundred) are sorted in LDS by one wave
         * each; only the ones beyond go through the HBM-resident bitonic network (36 global-memory stages for 256 records: it cost
         * ~70 us per segment, and 7 % of the reads of a realistic batch come here) */
        uint32_t *d_lg;
        STCHK(ensure(c, "large", n_big, &d_lg));
        HIPCHK(hipMemsetAsync(c->d_xscal + 2, 0, 8, st));
        hipLaunchKernelGGL(k_segsort_small, dim3(std::min<uint32_t>(n_big, 256u * 40u)), dim3(64), 0, st, d_big, (const uint64_t *)d_bigstart, (uint64_t)n_big, d_lg,
                           (uint32_t *)(c->d_xscal + 2), (uint32_t *)nullptr);
        hipLaunchKernelGGL((k_segsort_large<mtb_match>), dim3(std::min<uint32_t>(n_big, 1024)), dim3(256), 0, st, d_big, (const uint64_t *)d_bigstart,
                           (const uint32_t *)d_lg, (const uint32_t *)(c->d_xscal + 2));
        HIPCHK(hipGetLastError());
        b->m = d_big; b->seg = d_bigstart; b->list = d_biglist; b->n_list = (const uint32_t *)(c->d_scal + 5); b->seg_by_list = 1;
        /* one wave per workgroup, 14 of them resident per CU: a thousand workgroups left three quarters of the chip idle when hundreds of
         * thousands of reads come here (reads of conserved genes: hundreds of matches each); the slab pool stays bounded by dev_score */
        b->sort = false; b->max_seg = (uint32_t)mx; b->grid = std::min<uint32_t>(n_big, 256u * 14u); b->cap = 320;
        return MTB_OK;
    };
    STCHK(dev_score(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base, a, &second, max_len_deferred));
    /* number of matches (statistics): live records seen by the first launch + the deferred reads' segments */
    { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(st, d_cnt, n_reads, true, d_tot, d_ws2); }
    STCHK(d2h(c, &nm, d_tot + n_reads, 8));
    nm += big_total;
    *nm_out = nm;
    return MTB_OK;
}

/* results of a batch to the host with the taxID:count lists packed on the device first: only what they hold crosses PCIe */
static mtb_status download_packed(mtb_ctx *c, mtb_result *d_res, const int32_t *d_tt, const uint32_t *d_tc, uint64_t n_reads, mtb_result *results,
                                  int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t *n_taxcnt, uint64_t host_cap = ~0ull, int async_set = -1 /* >= 0: queue the copies on the download stream, packed lists in buffer set async_set */) {
    uint32_t *d_n; uint64_t *d_off, *d_ws; int32_t *d_tt2; uint32_t *d_tc2;
    STCHK(ensure(c, "tcn", n_reads, &d_n)); STCHK(ensure(c, "tcnewoff", n_reads + 1, &d_off)); STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    hipLaunchKernelGGL(k_taxcnt_n, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, c->stream, (const mtb_result *)d_res, n_reads, d_n);
    scan_launch<uint32_t, uint64_t, false>(c->stream, d_n, n_reads, true, d_off, d_ws);
    uint64_t total = 0;
    STCHK(d2h(c, &total, d_off + n_reads, 8));
    if (total > host_cap) { *n_taxcnt = total; return fail(MTB_ERR_CAPACITY, "taxcnt arrays too small for the batch's taxID:count lists"); }
    STCHK(ensure(c, async_set == 1 ? "tctax2b" : "tctax2", total, &d_tt2)); STCHK(ensure(c, async_set == 1 ? "tccnt2b" : "tccnt2", total, &d_tc2));
    hipLaunchKernelGGL(k_taxcnt_pack, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, c->stream, d_res, n_reads, (const uint64_t *)d_off, d_tt, d_tc, d_tt2, d_tc2);
    HIPCHK(hipGetLastError());
    if (async_set >= 0) {
        /* the rows and the packed lists leave on the download stream once the pack kernel is through; the compute stream is free for the next batch */
        HIPCHK(hipEventRecord(c->down_ready, c->stream));
        HIPCHK(hipStreamWaitEvent(c->down_stream, c->down_ready, 0));
        HIPCHK(hipMemcpyAsync(results, d_res, n_reads * sizeof(mtb_result), hipMemcpyDeviceToHost, c->down_stream));
        if (total) { HIPCHK(hipMemcpyAsync(taxcnt_tax, d_tt2, total * 4, hipMemcpyDeviceToHost, c->down_stream)); HIPCHK(hipMemcpyAsync(taxcnt_cnt, d_tc2, total * 4, hipMemcpyDeviceToHost, c->down_stream)); }
        HIPCHK(hipEventRecord(c->down_done, c->down_stream));
        c->down_pending = true;
        *n_taxcnt = total;
        return MTB_OK;
    }
    STCHK(d2h(c, results, d_res, n_reads * sizeof(mtb_result)));
    if (total) { STCHK(d2h(c, taxcnt_tax, d_tt2, total * 4)); STCHK(d2h(c, taxcnt_cnt, d_tc2, total * 4)); }
    *n_taxcnt = total;
    return MTB_OK;
}

/* one read range on one stream; taxcnt slots of this range start at tc_base of the caller's arrays */
static mtb_status classify_one(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const char *d_bases, const uint64_t *d_offs,
                               const char *d_bases2, const uint64_t *d_offs2, uint64_t n_reads, uint64_t n_bases_total,
                               mtb_result *d_results, int32_t *d_taxcnt_tax, uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap,
                               uint64_t *n_taxcnt, uint64_t tc_base) {
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    memset(&c->stats, 0, sizeof(c->stats));
    collect_kernel_times(c);          /* drop events of stage-level calls */
    memset(&c->stats, 0, sizeof(c->stats));
    if (n_reads == 0) return MTB_OK;
    hipStream_t st = c->stream;
    int32_t *d_ql, *d_ql2;
    STCHK(ensure(c, "qlen", n_reads, &d_ql)); STCHK(ensure(c, "qlen2", n_reads, &d_ql2));
    HIPCHK(hipEventRecord(c->ev[0], st));
    mtb_kmer *d_k; uint64_t nk; uint32_t max_len = 0, max_q = 0;
    uint64_t nk_real = 0;                 /* nk counts the blank tail records of the single-pass extraction too */
    /* short reads: per-read slot segments, the query's ordinal (tagged into qinfo by the extractor) is the slot of its
     * first match.  Needs positions < 2^12 (16-byte slot records) and a moderate number of metamers per read; otherwise
     * exact segments. */
    bool fixed = p->seq_mode != 3;
    /* long reads: ordinal slots too, with per-read slot ranges (k_join_dir<.., LONG>, kernels_seg_order.h) -- needs the directory join,
     * positions and ordinals below 2^16; otherwise (and after a failed attempt) exact segments via regroup + segment sort */
    bool lslot = p->seq_mode == 3 && ix->d_dir && !c->no_lslot && !c->opt.no_long_slots;
    uint32_t *d_dcnt = nullptr;
    if (lslot) { STCHK(ensure(c, "dcnt", n_reads, &d_dcnt)); HIPCHK(hipMemsetAsync(d_dcnt, 0, n_reads * 4, st)); }
    uint16_t *d_dig = nullptr;               /* first radix pass's digits, written by the single-pass extractor */
    const bool aa6 = p->kmer_format == 2;           /* 5-bit amino-acid letters: three base-21 pair passes order bits [34,64) */
    uint64_t n_bases_exact = n_bases_total;      /* the caller's figure is an estimate for a sub-batch of variable-length reads: the budget is kept on the exact one */
    /* short-read batches: reads whose positions do not fit a slot record (>= 4093 used bases) or that carry more metamers than a segment
     * has direct slots for are marked by the extractor and routed AROUND the slot segments one by one (join: overflow list; scoring:
     * exact segments) -- a handful of long reads in an Illumina batch no longer send the whole batch down the exact-segment path */
    uint8_t *d_off = nullptr; uint64_t off_stats[3] = {0, 0, 0};
    if (fixed && ix->d_dir) STCHK(ensure(c, "offreads", n_reads, &d_off));
    STCHK(dev_extract(c, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, &d_k, &nk, d_ql, d_ql2, &max_len, true, n_bases_total, &nk_real, fixed || lslot, &max_q, aa6 ? &d_dig : nullptr, d_dcnt, &n_bases_exact,
                      d_off, off_stats));
    uint32_t max_len_all = max_len;
    bool route_off = false;
    if (fixed && d_off && off_stats[1] > 0 && off_stats[1] * 4 <= n_reads && max_len + 3 < 65536u && max_q < 65535u && off_stats[1] < n_reads) {
        route_off = true;
        max_q = (uint32_t)off_stats[0]; max_len = (uint32_t)off_stats[2];       /* the slot geometry and the first scoring launch follow the reads that use the slots */
    }
    if ((fixed && !route_off && (max_len + 3 >= MTB_SLOT_MAX_POS || max_q > MTB_SLOT_MAX_Q)) || (lslot && (max_len + 3 >= 65536u || max_q >= 65535u))) {
        fixed = false; lslot = false;      /* tags would collide with positions / segments would be huge: extract again untagged */
        STCHK(dev_extract(c, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, &d_k, &nk, d_ql, d_ql2, &max_len, true, n_bases_total, &nk_real, false, &max_q, aa6 ? &d_dig : nullptr));
    }
    HIPCHK(hipEventRecord(c->ev[1], st));
    /* the join needs tiles with a narrow amino-acid range, not a total order: kmer_format 2 sorts on the first six
     * amino-acid letters (three base-21 pair passes = bits [34,64)), kmer_format 1 on the top 32 bits (four binary
     * passes; three binary passes make the tiles too wide for the LDS window, measured); the tile's target window
     * comes from k_join_bounds */
    mtb_kmer *d_s;
    int aa_first_shift = 34;
    if (aa6 && fixed && ix->d_dir && c->opt.sort_pairs > 0) aa_first_shift = 64 - 10 * std::max(1, std::min(3, c->opt.sort_pairs));
    const int low_bits = aa6 ? aa_first_shift : 32;          /* the bits the list is NOT sorted on (the join's tile windows follow the sort key) */
    STCHK(dev_sort(c, d_k, nk, aa6 ? MTB_SORT_AA6 : 32, &d_s, aa_first_shift == 34 ? d_dig : nullptr, aa_first_shift));
    HIPCHK(hipEventRecord(c->ev[2], st));
    c->last_sorted = (fixed && ix->d_dir) ? d_s : nullptr; c->last_sorted_n = nk;
    uint32_t *d_rc;
    STCHK(ensure(c, "readcnt", n_reads, &d_rc));
    HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, st));
    uint64_t nm = 0;
    if (fixed) {
        /* ---- join straight into per-read slot segments (d_rc = per-read tail cursor) ---- */
        uint32_t direct, stride;
        slot_geometry(c, max_q, &direct, &stride);
        mtb_slot16 *d_segm; mtb_match *d_ovf; uint64_t n_ovf = 0; uint32_t epoch = 0;
        STCHK(prepare_slots(c, n_reads, stride, &d_segm, &epoch));
        DevBuf &ob = c->bufs["ovf"];
        uint64_t ovf_cap = std::max<uint64_t>(ob.cap / sizeof(mtb_match), nk / 64 + 4096);
        for (int attempt = 0; attempt < 3; attempt++) {
            STCHK(ensure(c, "ovf", ovf_cap, &d_ovf));
            JoinSegArgs sa; memset(&sa, 0, sizeof(sa));
            sa.seg = d_segm; sa.stride = stride; sa.direct = direct; sa.cursor = d_rc; sa.ovf = d_ovf; sa.ovf_cap = ovf_cap;
            sa.ovf_counter = nullptr; sa.epoch = epoch; sa.off = route_off ? d_off : nullptr;
            sa.dense_ovf = attempt == 2 ? 1u : 0u;            /* last attempt: one list that holds the total, whatever the stripes' fill */
            sa.retry = attempt > 0 ? 1u : 0u;
            mtb_status s2 = dev_join(c, ix, d_s, nk, nullptr, 0, nullptr, &n_ovf, &sa, low_bits);
            if (s2 == MTB_OK) break;
            if (s2 != MTB_ERR_CAPACITY || attempt == 2) return s2;
            ovf_cap = n_ovf + n_ovf / 16 + 1024;
            HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, st));     /* ordinal slots are rewritten identically; tail slots and the overflow list are filled anew (their order follows the atomics) */
        }
        HIPCHK(hipEventRecord(c->ev[3], st));
        HIPCHK(hipEventRecord(c->ev[4], st));
        HIPCHK(hipEventRecord(c->ev[5], st));
        {   /* dead from here to the next extraction (the sorted list itself stays: mtb_ctx_join_footprint / mtb_ctx_query_runs look at it after the batch) */
            std::lock_guard<std::mutex> lk(c->bufs_mu);
            c->n_dead = 0;
            const size_t least = c->opt.scratch_alias > 0 ? 1 : (64u << 20);
            for (const char *nm : {"kmersA", "kmersB", "digA", "digB"}) {
                auto it = c->bufs.find(nm);
                if (c->opt.scratch_alias < 0 || it == c->bufs.end() || !it->second.p || it->second.p == (void *)d_s || it->second.cap < least || c->n_dead == 3) continue;
                c->dead[c->n_dead].p = (char *)it->second.p; c->dead[c->n_dead].cap = it->second.cap; c->dead[c->n_dead].used = 0; c->n_dead++;
            }
        }
        struct DeadEnd { mtb_ctx *c; ~DeadEnd() { c->n_dead = 0; } } dead_end{c};
        STCHK(score_fixed_slots(c, ix, p, n_reads, d_ql, d_ql2, max_len, nk_real, d_segm, d_rc, stride, direct, epoch, d_ovf, n_ovf,
                                d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base, &nm, route_off ? max_len_all : 0, route_off ? d_off : nullptr, c->ovf_region));
    } else if (lslot) {
        /* ---- long reads on ordinal slots: join into per-read slot ranges, order every range by a stable species partition, score ---- */
        uint32_t *d_sizes; uint64_t *d_rb, *d_ws2; uint32_t *d_live;
        STCHK(ensure(c, "lsizes", n_reads, &d_sizes)); STCHK(ensure(c, "lrb", n_reads + 1, &d_rb)); STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws2));
        STCHK(ensure(c, "livecnt", n_reads, &d_live));
        uint32_t *d_fail;
        STCHK(ensure(c, "lfail", n_reads, &d_fail));
        bool done = false;
        /* tail = a quarter of the read's metamers, then all of them; a context whose last batch needed the long tails starts with them
         * (a database with long candidate runs overflows the short ones batch after batch: the join ran twice every time) */
        for (uint32_t tf = c->lslot_tf_start; tf <= 4 && !done; tf *= 4) {
            hipLaunchKernelGGL(k_lslot_sizes, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, st, (const uint32_t *)d_dcnt, n_reads, tf, d_sizes);
            { KTimer kt(c, MTB_K_SCAN); scan_launch<uint32_t, uint64_t, false>(st, d_sizes, n_reads, true, d_rb, d_ws2); }
            uint64_t n_slots = 0;
            STCHK(d2h(c, &n_slots, d_rb + n_reads, 8));
            mtb_slot16 *d_segm; mtb_match *d_m;
            STCHK(ensure(c, "segm", n_slots, &d_segm));
            c->seg_epoch = MTB_SLOT_EPOCHS;                           /* (a later short-read batch starts from a cleared buffer) */
            HIPCHK(hipMemsetAsync(d_segm, 0, n_slots * sizeof(mtb_slot16), st));
            HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, st));
            JoinSegArgs sa; memset(&sa, 0, sizeof(sa));
            sa.seg = d_segm; sa.cursor = d_rc; sa.rb = d_rb; sa.dcnt = d_dcnt; sa.tf = tf; sa.ovf = nullptr; sa.ovf_cap = 0;
            uint64_t n_ovf = 0;
            mtb_status s2 = dev_join(c, ix, d_s, nk, nullptr, 0, nullptr, &n_ovf, &sa, low_bits);
            if (s2 == MTB_ERR_CAPACITY) {                              /* some read's tail overran: larger tails */
                c->lslot_tf_start = 4;
                if (c->opt.lslot_verbose) fprintf(stderr, "mtb: long-read slot path: %llu matches beyond the tails at tail factor %u/4\n", (unsigned long long)n_ovf, tf);
                continue;
            }
            if (s2 != MTB_OK) return s2;
            HIPCHK(hipEventRecord(c->ev[3], st));
            STCHK(ensure(c, "matches", n_slots, &d_m));
            HIPCHK(hipMemsetAsync(c->d_scal + 2, 0, 16, st));         /* [2] reads beyond the LDS tables, [3] matches of the range (incl. the dropped lonely ones) */
            unsigned long long *d_work = (unsigned long long *)(c->d_xscal + 7);
            HIPCHK(hipMemsetAsync(d_work, 0, 8, st));
            { KTimer kt(c, MTB_K_SEGSORT);
              hipLaunchKernelGGL(k_seg_order, dim3((uint32_t)std::min<uint64_t>(n_reads, 256ull * 2)), dim3(MTB_SO_NT), 0, st, (const mtb_slot16 *)d_segm, (const uint64_t *)d_rb,
                                 (const uint32_t *)d_dcnt, (const uint32_t *)d_rc, tf, n_reads, d_m, d_live, (uint32_t *)(c->d_scal + 2), d_fail, d_work,
                                 (unsigned long long *)(c->d_scal + 3)); }
            HIPCHK(hipGetLastError());
#ifdef MTB_SO_PHASE_CYCLES
            {   HIPCHK(hipStreamSynchronize(st));
                unsigned long long h[8], z[8] = {0};
                HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(mtb_so_cycles), sizeof(h)));
                HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mtb_so_cycles), z, sizeof(z)));
                static const char *nm[8] = {"claim + clear", "pass 1 (bits)", "pass 2 (counts, tail keys)", "single-match species", "tail sort", "species sort + offsets", "scatter pass", "tail pass"};
                unsigned long long tot = 0; for (int k = 0; k < 8; k++) tot += h[k];
                fprintf(stderr, "k_seg_order phases (%llu reads):", (unsigned long long)n_reads);
                for (int k = 0; k < 8; k++) fprintf(stderr, " %s %.1f %%;", nm[k], tot ? 100.0 * (double)h[k] / (double)tot : 0.0);
                fprintf(stderr, "\n");
            }
#endif
            uint64_t sc2[2] = {0, 0};
            STCHK(d2h(c, sc2, c->d_scal + 2, 16));
            if (c->opt.lslot_verbose) fprintf(stderr, "mtb: long-read slot path: %llu reads, tail factor %u/4, %llu slots, %llu matches, %llu reads beyond the LDS tables (sorted the general way)\n",
                                                     (unsigned long long)n_reads, tf, (unsigned long long)n_slots, (unsigned long long)sc2[1], (unsigned long long)(sc2[0] & 0xFFFFFFFFull));
            const uint32_t n_failed = (uint32_t)(sc2[0] & 0xFFFFFFFFull);
            /* reads beyond k_seg_order's LDS tables: their live slots -> exact segments, sorted the general way */
            mtb_match *d_big = nullptr; uint64_t *d_bigstart = nullptr; uint32_t big_max = 0; uint64_t big_total = 0;
            if (n_failed) {
                uint32_t *d_bigcnt; mtb_match *d_scr; uint64_t mx = 0;
                STCHK(ensure(c, "bigcnt", n_failed, &d_bigcnt)); STCHK(ensure(c, "bigstart", (uint64_t)n_failed + 1, &d_bigstart));
                HIPCHK(hipMemsetAsync(c->d_scal + 4, 0, 8, st));
                hipLaunchKernelGGL(k_lbig_count, dim3(std::min<uint32_t>(n_failed, 4096)), dim3(64), 0, st, (const mtb_slot16 *)d_segm, (const uint64_t *)d_rb, (const uint32_t *)d_dcnt,
                                   (const uint32_t *)d_rc, tf, (const uint32_t *)d_fail, n_failed, d_bigcnt, (uint32_t *)(c->d_scal + 4));
                scan_launch<uint32_t, uint64_t, false>(st, d_bigcnt, n_failed, true, d_bigstart, d_ws2);
                STCHK(d2h(c, &big_total, d_bigstart + n_failed, 8));
                STCHK(d2h(c, &mx, c->d_scal + 4, 8));
                big_max = (uint32_t)mx;
                STCHK(ensure(c, "bigm", big_total, &d_big)); STCHK(ensure(c, "jtemp", big_total, &d_scr));
                hipLaunchKernelGGL(k_lbig_copy, dim3(std::min<uint32_t>(n_failed, 4096)), dim3(64), 0, st, (const mtb_slot16 *)d_segm, (const uint64_t *)d_rb, (const uint32_t *)d_dcnt,
                                   (const uint32_t *)d_rc, tf, (const uint32_t *)d_fail, n_failed, (const uint64_t *)d_bigstart, d_big);
                const size_t lds = (size_t)MTB_SEGLDS_CHUNK * 14;
                HIPCHK(hipFuncSetAttribute((const void *)k_segsort_lds<mtb_match>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                { KTimer kt(c, MTB_K_SEGSORT);
                  hipLaunchKernelGGL((k_segsort_lds<mtb_match>), dim3(std::min<uint32_t>(n_failed, 2048)), dim3(MTB_SEGLDS_THREADS), lds, st, d_big, (const uint64_t *)d_bigstart,
                                     (const uint32_t *)nullptr, (const uint32_t *)(c->d_scal + 2), d_scr); }
                HIPCHK(hipGetLastError());
            }
            HIPCHK(hipEventRecord(c->ev[4], st));
            HIPCHK(hipEventRecord(c->ev[5], st));
            STCHK(dev_score_long(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base, d_m, d_rb, max_q + mtb_lslot_tail(max_q, tf), d_live));
            uint64_t n_generic = 0;
            if (n_failed) {
                STCHK(d2h(c, &n_generic, c->d_scal + 6, 8));        /* reads the first launch left to the generic kernel */
                uint64_t n_tc2 = 0;
                STCHK(dev_score_long(c, ix, p, n_reads, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, &n_tc2, tc_base, d_big, d_bigstart, big_max, nullptr, d_fail, n_failed));
                uint64_t g2 = 0;
                STCHK(d2h(c, &g2, c->d_scal + 6, 8));
                n_generic = (n_generic & 0xFFFFFFFFull) + (g2 & 0xFFFFFFFFull);
                STCHK(h2d(c, c->d_scal + 6, &n_generic, 8));
            }
            nm = sc2[1] + big_total;
            done = true;
        }
        if (!done) {
            /* redo this range without ordinal slots (tags stripped by a fresh extraction) */
            c->no_lslot = true;
            mtb_status s3 = classify_one(c, ix, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, n_bases_total, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base);
            c->no_lslot = false;
            return s3;
        }
    } else {
        /* ---- long reads: exact segments (temp buffer, per-read counters, scan, regroup) ---- */
        mtb_match *d_tmp;
        DevBuf &jb = c->bufs["jtemp"];
        uint64_t cap = std::max<uint64_t>(jb.cap / sizeof(mtb_match), nk + nk / 2 + 1024);
        for (int attempt = 0; attempt < 3; attempt++) {
            STCHK(ensure(c, "jtemp", cap, &d_tmp));
            mtb_status s2 = dev_join(c, ix, d_s, nk, d_tmp, cap, d_rc, &nm, nullptr, low_bits);
            if (s2 == MTB_OK) break;
            if (s2 != MTB_ERR_CAPACITY || attempt == 2) return s2;
            cap = nm + nm / 16 + 1024;                     /* the reference's retry (Classifier.cpp:127-131) with the exact size */
            HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, st));
        }
        HIPCHK(hipEventRecord(c->ev[3], st));
        STCHK(score_join_order(c, ix, p, d_tmp, nm, n_reads, d_rc, d_ql, d_ql2, max_len, d_results, d_taxcnt_tax, d_taxcnt_cnt, taxcnt_cap, n_taxcnt, tc_base));
    }
    HIPCHK(hipEventRecord(c->ev[6], st));
    HIPCHK(hipEventSynchronize(c->ev[6]));
    mtb_batch_stats &S = c->stats;
    HIPCHK(hipEventElapsedTime(&S.ms_extract, c->ev[0], c->ev[1]));
    HIPCHK(hipEventElapsedTime(&S.ms_sort, c->ev[1], c->ev[2]));
    HIPCHK(hipEventElapsedTime(&S.ms_join, c->ev[2], c->ev[3]));
    HIPCHK(hipEventElapsedTime(&S.ms_regroup, c->ev[3], c->ev[4]));
    HIPCHK(hipEventElapsedTime(&S.ms_segsort, c->ev[4], c->ev[5]));
    HIPCHK(hipEventElapsedTime(&S.ms_score, c->ev[5], c->ev[6]));
    HIPCHK(hipEventElapsedTime(&S.ms_total, c->ev[0], c->ev[6]));
    collect_kernel_times(c);
    S.n_reads = n_reads; S.n_bases = n_bases_exact; S.n_kmers = nk_real; S.n_matches = nm; S.n_targets = ix->T;
    if (c->fast_used) { uint64_t ns = 0; STCHK(d2h(c, &ns, c->d_scal + 6, 8)); S.n_generic_reads = ns & 0xFFFFFFFFull; c->fast_used = false; }
    else S.n_generic_reads = n_reads;
    S.n_slot_reads = (fixed || lslot) ? n_reads : 0;
    if (fixed) { S.n_deferred_reads = c->many_stats[0]; S.n_many_reads = c->many_stats[1]; S.n_many_matches = c->many_stats[2]; S.n_many_kept = c->many_stats[3]; }
    return MTB_OK;
}

static void merge_stats(mtb_batch_stats &S, const mtb_batch_stats &L) {
    S.ms_extract += L.ms_extract; S.ms_sort += L.ms_sort; S.ms_join += L.ms_join; S.ms_regroup += L.ms_regroup;
    S.ms_segsort += L.ms_segsort; S.ms_score += L.ms_score;
    S.n_reads += L.n_reads; S.n_bases += L.n_bases; S.n_kmers += L.n_kmers; S.n_matches += L.n_matches; S.n_targets = L.n_targets;
    S.n_generic_reads += L.n_generic_reads; S.n_slot_reads += L.n_slot_reads;
    S.n_deferred_reads += L.n_deferred_reads; S.n_many_reads += L.n_many_reads; S.n_many_matches += L.n_many_matches; S.n_many_kept += L.n_many_kept;
    for (int i = 0; i < MTB_NUM_KERNELS; i++) { S.ms_kernel[i] += L.ms_kernel[i]; S.n_launch[i] += L.n_launch[i]; }
    S.join_variant = L.join_variant; S.join_tuned = L.join_tuned;                 /* (of the last range) */
    for (int v = 0; v < 3; v++) S.join_tune_ms[v] = L.join_tune_ms[v];
    S.join_tiles += L.join_tiles; S.join_tiles_windowed += L.join_tiles_windowed; S.join_tiles_outside += L.join_tiles_outside;
}

/* HBM-budgeted batching (the reference sizes a QuerySplit from --max-ram, QueryIndexer.cpp:62,132, and redoes
 * it with a bigger match buffer on overflow, Classifier.cpp:127-131).  Here the budget is HBM: what hipMemGetInfo
 * reports free plus what the context's own buffers already hold (or mtb_ctx_set_workspace_limit), and the per-base
 * workspace is known -- two 16-byte metamer buffers + two 2-byte digit arrays per extracted metamer, one 16-byte slot per
 * metamer plus the tail for short reads (24-byte match records twice for long reads), ~100 bytes of per-read tables -- and
 * is re-measured on every sub-batch.  The batch is cut into contiguous read ranges of equal size that fit; a range that
 * still runs out of memory is halved and redone.  Results land at their final places (d_results + lo, taxcnt slots
 * appended), so the caller sees one batch. */
static mtb_status classify_budgeted(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const char *d_bases, const uint64_t *d_offs,
                                    const char *d_bases2, const uint64_t *d_offs2, uint64_t n_reads, uint64_t n_bases_total,
                                    mtb_result *d_results, int32_t *d_taxcnt_tax, uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap,
                                    uint64_t *n_taxcnt, uint64_t tc_base0) {
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    c->last_sub_batches = 0; c->last_scratch_bytes = 0;
    if (n_reads == 0) { memset(&c->stats, 0, sizeof(c->stats)); return MTB_OK; }
    if (c->ws_seq_mode != p->seq_mode) {             /* other buffer set: what the previous mode grew would only sit in the way */
        if (c->ws_seq_mode) { HIPCHK(hipStreamSynchronize(c->stream)); release_workspace(c); }
        c->ws_seq_mode = p->seq_mode;
    }
    size_t fr = 0, tot = 0;
    HIPCHK(hipMemGetInfo(&fr, &tot));
    const size_t held = held_bytes(c);
    uint64_t budget = c->ws_limit ? c->ws_limit : (uint64_t)fr + held;
    if (!c->ws_limit) budget -= std::min<uint64_t>(budget / 16, 4ull << 30);          /* allocator granularity, kernel scratch, other users of the device */
    if (p->seq_mode == 3 && !c->ws_limit) budget -= std::min<uint64_t>(budget / 4, MTB_SLAB_POOL_MAX);     /* the slab pool of the large-segment scorer does not scale with the batch */
    const double mean_len = (double)n_bases_total / (double)n_reads;
    double per_base = c->ws_per_base;
    if (per_base <= 0.0) {
        const double yield = (c->extract_yield > 0.0 ? c->extract_yield : (p->syncmer ? 1.0 : 2.0)) * 1.15;
        per_base = yield * (16 + 16 + 2 + 2) + (p->seq_mode == 3 ? yield * 1.5 * 48 : yield * 1.2 * 16 + 24.0 * 16 / std::max(mean_len, 1.0)) + 100.0 / std::max(mean_len, 1.0);
    }
    const double safety = c->ws_per_base > 0.0 ? 1.02 : 1.08;                           /* measured on the previous batch / first-call estimate */
    uint64_t fit = (uint64_t)((double)budget / (per_base * safety * mean_len));        /* reads per sub-batch */
    fit = std::max<uint64_t>(fit, 1);
    uint64_t n_sub = (n_reads + fit - 1) / fit;
    if (c->opt.host_timing)
        fprintf(stderr, "mtb budget: free %.1f GiB + held %.1f GiB -> budget %.1f GiB; %.1f bytes per base (%s) x %.2f -> %llu reads fit, %llu sub-batch(es) for %llu reads\n",
                (double)fr / 1073741824.0, (double)held / 1073741824.0, (double)budget / 1073741824.0, per_base, c->ws_per_base > 0.0 ? "measured" : "estimate", safety,
                (unsigned long long)fit, (unsigned long long)n_sub, (unsigned long long)n_reads);
    mtb_batch_stats S; memset(&S, 0, sizeof(S));
    uint64_t tc_used = 0, lo = 0;
    uint32_t done = 0;
    while (lo < n_reads) {
        const uint64_t left_sub = std::max<uint64_t>(1, n_sub > done ? n_sub - done : 1);
        uint64_t cnt = (n_reads - lo + left_sub - 1) / left_sub;
        uint64_t n_tc = 0;
        mtb_status st;
        for (;;) {
            st = classify_one(c, ix, p, d_bases, d_offs + lo, d_bases2, d_offs2 ? d_offs2 + lo : nullptr, cnt,
                              (uint64_t)(mean_len * (double)cnt), d_results + lo, d_taxcnt_tax + tc_used, d_taxcnt_cnt + tc_used,
                              taxcnt_cap - std::min(taxcnt_cap, tc_used), &n_tc, tc_base0 + tc_used);
            if (st != MTB_ERR_OOM || cnt <= 1024) break;
            /* the estimate was too optimistic for this range: give the memory back, halve, redo */
            HIPCHK(hipStreamSynchronize(c->stream));
            if (c->opt.host_timing) fprintf(stderr, "mtb budget: %llu reads ran out of memory (%s); halving\n", (unsigned long long)cnt, mtb_last_error());
            release_workspace(c);
            cnt = (cnt + 1) / 2; n_sub *= 2; done *= 2;
        }
        if (st == MTB_ERR_CAPACITY) {
            /* taxcnt arrays too small: report what the whole batch needs (slots of the ranges done + this one + the rest pro rata) */
            const uint64_t rest = n_reads - lo - cnt;
            *n_taxcnt = std::max<uint64_t>(taxcnt_cap + 1, tc_used + n_tc + (uint64_t)((double)n_tc / (double)cnt * (double)rest * 1.05) + 64);
            return st;
        }
        if (st != MTB_OK) return st;
        /* the buffers only grow: what they hold belongs to the largest sub-batch they have seen, not to this one (dividing by a
         * smaller one inflates the figure, more sub-batches follow, and the next measurement is worse still) */
        c->ws_max_sub_bases = std::max<uint64_t>(c->ws_max_sub_bases, c->stats.n_bases);
        if (c->ws_max_sub_bases) c->ws_per_base = (double)scaling_bytes(c) / (double)c->ws_max_sub_bases;
        if (c->opt.host_timing) {
            std::lock_guard<std::mutex> lk(c->bufs_mu);
            fprintf(stderr, "mtb budget: sub-batch of %llu reads (%llu bases) done; buffers (GiB):", (unsigned long long)cnt, (unsigned long long)c->stats.n_bases);
            for (auto &kv : c->bufs) if (kv.second.cap >= (256u << 20)) fprintf(stderr, " %s %.2f", kv.first.c_str(), (double)kv.second.cap / 1073741824.0);
            fprintf(stderr, "\n");
        }
        merge_stats(S, c->stats); S.ms_total += c->stats.ms_total;
        tc_used += n_tc; lo += cnt; done++;
    }
    c->stats = S;
    c->last_sub_batches = done;
    *n_taxcnt = tc_used;
    return MTB_OK;
}

extern "C" {

mtb_status mtb_classify_batch_device(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const char *d_bases, const uint64_t *d_offs,
                                     const char *d_bases2, const uint64_t *d_offs2, uint64_t n_reads, uint64_t n_bases_total,
                                     mtb_result *d_results, int32_t *d_taxcnt_tax, uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap,
                                     uint64_t *n_taxcnt) {
    if (!c || !ix || !p || !n_taxcnt) return fail(MTB_ERR_ARG, "NULL argument");
    const size_t L = c->lanes.size();
    if (L < 2 || n_reads < 4096 * L)
        return classify_budgeted(c, ix, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, n_bases_total, d_results, d_taxcnt_tax, d_taxcnt_cnt,
                                 taxcnt_cap, n_taxcnt, 0);
    /* L contiguous read ranges, one host thread + one non-blocking stream each */
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));          /* inputs produced on the caller's stream are complete */
    auto t0 = std::chrono::steady_clock::now();
    std::vector<mtb_status> st(L, MTB_OK);
    std::vector<std::string> errs(L);
    std::vector<uint64_t> ntc(L, 0);
    std::vector<std::thread> th;
    /* `chunks` read ranges per stream, interleaved (range k runs on stream k % L) */
    const int chunks = std::max(1, c->opt.chunks_per_stream);   /* sweep: profiles/r01_notes.md */
    const size_t NC = L * (size_t)chunks;
    const uint64_t tc_share = taxcnt_cap / NC;
    std::vector<mtb_batch_stats> lane_stats(L);
    for (size_t i = 0; i < L; i++) memset(&lane_stats[i], 0, sizeof(mtb_batch_stats));
    for (size_t i = 0; i < L; i++) {
        th.emplace_back([&, i]() {
            mtb_ctx *l = c->lanes[i];
            /* experiment switch: lane i starts i * MTB_LANE_STAGGER_MS late, so that the lanes are in DIFFERENT stages at any moment
             * (the join is bound by store transactions, the sort by HBM bandwidth, extraction and scoring by VALU issue) */
            const int stagger_ms = c->opt.lane_stagger_ms;
            if (stagger_ms > 0 && i > 0) std::this_thread::sleep_for(std::chrono::milliseconds((long)stagger_ms * (long)i));
            for (size_t k = i; k < NC; k += L) {
                uint64_t lo = n_reads * k / NC, hi = n_reads * (k + 1) / NC;
                uint64_t n_tc = 0;
                mtb_status s = classify_one(l, ix, p, d_bases, d_offs + lo, d_bases2, d_offs2 ? d_offs2 + lo : nullptr, hi - lo,
                                            n_bases_total * (hi - lo) / n_reads, d_results + lo, d_taxcnt_tax + tc_share * k,
                                            d_taxcnt_cnt + tc_share * k, tc_share, &n_tc, tc_share * k);
                ntc[i] = std::max(ntc[i], n_tc * NC / L);
                if (s != MTB_OK) { st[i] = s; errs[i] = g_err; break; }
                merge_stats(lane_stats[i], l->stats);
            }
        });
    }
    for (auto &t : th) t.join();
    memset(&c->stats, 0, sizeof(c->stats));
    uint64_t need = 0;
    for (size_t i = 0; i < L; i++) { merge_stats(c->stats, lane_stats[i]); need = std::max(need, ntc[i] * L); }
    c->stats.ms_total = std::chrono::duration<float, std::milli>(std::chrono::steady_clock::now() - t0).count();
    *n_taxcnt = taxcnt_cap;                             /* slots are spread over the whole array */
    for (size_t i = 0; i < L; i++)
        if (st[i] != MTB_OK) { if (st[i] == MTB_ERR_CAPACITY) *n_taxcnt = std::max<uint64_t>(need, taxcnt_cap + 1); return fail(st[i], errs[i]); }
    return MTB_OK;
}

/* Device-side taxcnt slots of a batch: one per position bucket of every read (k_taxcnt_bound), sum floor((len + 3) / shift) + 2 <=
 * (bases + 3 reads) / shift + 2 reads.  The host arrays of the single-stream calls only receive the packed lists (download_packed), so their
 * capacity may be far smaller (2-3 entries per read is typical); MTB_ERR_CAPACITY reports what the lists need. */
static uint64_t taxcnt_device_slots(const mtb_params *p, uint64_t n_reads, uint64_t n_bases) {
    mtb_score_params sp; mtb_make_score_params(p, &sp);
    return (n_bases + 3 * n_reads) / (uint64_t)std::max(1, sp.dna_shift) + 2 * n_reads + 64;
}

mtb_status mtb_classify_batch(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const char *bases, const uint64_t *offs, const char *bases2,
                              const uint64_t *offs2, uint64_t n_reads, mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt,
                              uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    if (!c || !ix || !p || !n_taxcnt) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    if (n_reads == 0) return MTB_OK;
    char *d_b, *d_b2; uint64_t *d_o, *d_o2; uint64_t nb;
    STCHK(upload_reads(c, p, bases, offs, bases2, offs2, n_reads, &d_b, &d_o, &d_b2, &d_o2, &nb));
    mtb_result *d_res; int32_t *d_tt; uint32_t *d_tc;
    /* one stream: the lists come back packed, the device arrays are sized here and taxcnt_cap only bounds what the host receives */
    const uint64_t dcap = c->lanes.size() < 2 ? std::max<uint64_t>(taxcnt_cap, taxcnt_device_slots(p, n_reads, nb)) : taxcnt_cap;
    STCHK(ensure(c, "results", n_reads, &d_res)); STCHK(ensure(c, "tctax", dcap, &d_tt)); STCHK(ensure(c, "tccnt", dcap, &d_tc));
    mtb_status st = mtb_classify_batch_device(c, ix, p, d_b, d_o, d_b2, d_o2, n_reads, nb, d_res, d_tt, d_tc, dcap, n_taxcnt);
    if (st != MTB_OK) return st;
    if (c->lanes.size() < 2 && *n_taxcnt) return download_packed(c, d_res, d_tt, d_tc, n_reads, results, taxcnt_tax, taxcnt_cnt, n_taxcnt, taxcnt_cap);
    STCHK(d2h(c, results, d_res, n_reads * sizeof(mtb_result)));
    if (*n_taxcnt) { STCHK(d2h(c, taxcnt_tax, d_tt, *n_taxcnt * 4)); STCHK(d2h(c, taxcnt_cnt, d_tc, *n_taxcnt * 4)); }
    return MTB_OK;
}

/* Host ingest at device rate: pinned host memory for the batch buffers, and the bases as 2-bit codes. */
void *mtb_host_alloc(size_t bytes) {
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    return p;
}
void mtb_host_free(void *p) { if (p) { hipError_t e = hipHostFree(p); (void)e; } }

static uint64_t packed_slots(const uint32_t *lens, uint64_t n_reads) {
    uint64_t slots = 0;
    for (uint64_t i = 0; i < n_reads; i++) slots += (lens[i] + 7u) >> 3;
    return slots;
}
/* the three packed arrays of one mate into input buffer set `set`, on stream `on` */
static mtb_status copy_packed(mtb_ctx *c, const char *tag, int set, hipStream_t on, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                              uint64_t n_reads, uint64_t slots, uint8_t **d_p2, uint8_t **d_nm, uint32_t **d_len) {
    const std::string t = std::string(tag) + (set ? "#1" : "#0");
    STCHK(ensure(c, ("pk2" + t).c_str(), slots * 2 + 8, d_p2)); STCHK(ensure(c, ("pkm" + t).c_str(), slots + 8, d_nm)); STCHK(ensure(c, ("pklen" + t).c_str(), n_reads, d_len));
    HIPCHK(hipMemcpyAsync(*d_p2, packed2, slots * 2, hipMemcpyHostToDevice, on));
    HIPCHK(hipMemcpyAsync(*d_nm, nmask, slots, hipMemcpyHostToDevice, on));
    HIPCHK(hipMemcpyAsync(*d_len, lens, n_reads * 4, hipMemcpyHostToDevice, on));
    return MTB_OK;
}
/* `prefetched`: the arrays already sit (or are arriving: the compute stream waits for the copy event) in set `set` */
static mtb_status upload_packed(mtb_ctx *c, const char *tag, int set, bool prefetched, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens, uint64_t n_reads,
                                char **d_bases, uint64_t **d_offs, uint64_t *n_bases) {
    const std::string t(tag);
    uint8_t *d_p2, *d_nm; uint32_t *d_len, *d_sl; uint64_t *d_so, *d_ws;
    if (prefetched) {
        const std::string ts = t + (set ? "#1" : "#0");
        d_p2 = (uint8_t *)c->bufs["pk2" + ts].p; d_nm = (uint8_t *)c->bufs["pkm" + ts].p; d_len = (uint32_t *)c->bufs["pklen" + ts].p;
    } else STCHK(copy_packed(c, tag, set, c->stream, packed2, nmask, lens, n_reads, packed_slots(lens, n_reads), &d_p2, &d_nm, &d_len));
    STCHK(ensure(c, ("pksl" + t).c_str(), n_reads, &d_sl)); STCHK(ensure(c, ("pkso" + t).c_str(), n_reads + 1, &d_so));
    STCHK(ensure(c, ("offs" + t).c_str(), n_reads + 1, d_offs)); STCHK(ensure(c, "scanws", scan_ws_elems(n_reads + 1), &d_ws));
    scan_launch<uint32_t, uint64_t, false>(c->stream, d_len, n_reads, true, *d_offs, d_ws);
    hipLaunchKernelGGL(k_pack_slots, dim3((uint32_t)((n_reads + 255) / 256)), dim3(256), 0, c->stream, (const uint32_t *)d_len, n_reads, d_sl);
    scan_launch<uint32_t, uint64_t, false>(c->stream, d_sl, n_reads, true, d_so, d_ws);
    uint64_t nb = 0;
    STCHK(d2h(c, &nb, *d_offs + n_reads, 8));
    STCHK(ensure(c, ("bases" + t).c_str(), nb + 8, d_bases));
    hipLaunchKernelGGL(k_unpack_reads, dim3((uint32_t)std::min<uint64_t>(n_reads, 256ull * 64)), dim3(64), 0, c->stream, (const uint8_t *)d_p2, (const uint8_t *)d_nm,
                       (const uint64_t *)*d_offs, (const uint64_t *)d_so, n_reads, *d_bases);
    HIPCHK(hipGetLastError());
    *n_bases = nb;
    return MTB_OK;
}

} // extern "C"

static mtb_status wait_results(mtb_ctx *c) {
    if (c->down_pending) { HIPCHK(hipEventSynchronize(c->down_done)); c->down_pending = false; }
    return MTB_OK;
}

static mtb_status classify_packed_impl(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                       const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n_reads,
                                       mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt, bool async) {
    if (!c || !ix || !p || !n_taxcnt) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    if (!async) STCHK(wait_results(c));          /* (a synchronous call after asynchronous ones: nothing stays in flight behind it) */
    if (n_reads == 0) return async ? wait_results(c) : MTB_OK;
    if (async && !c->down_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->down_stream, hipStreamNonBlocking));
        HIPCHK(hipEventCreateWithFlags(&c->down_ready, hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->down_done, hipEventDisableTiming));
    }
    /* asynchronous results: this call's rows and lists are built in the device buffer set the copies still in flight do not read */
    const int rset = async ? c->res_set : 0;
    if (!packed2 || !nmask || !lens) return fail(MTB_ERR_ARG, "packed2/nmask/lens NULL");
    char *d_b = nullptr, *d_b2 = nullptr; uint64_t *d_o = nullptr, *d_o2 = nullptr; uint64_t nb = 0, nb2 = 0;
    const bool timing = c->opt.host_timing != 0;      /* wall time of the call's three parts on stderr (the uploads are synchronised for it) */
    auto now = [] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t0 = now();
    if (p->seq_mode == 2 && (!packed2_mate || !nmask_mate || !lens_mate)) return fail(MTB_ERR_ARG, "seq_mode 2 needs the mates");
    /* this batch may already be on its way (mtb_prefetch_batch_packed): then the compute stream only waits for the copy */
    int set = -1;
    for (int k = 0; k < 2; k++)
        if (c->pre[k].valid && c->pre[k].key == (const void *)packed2 && c->pre[k].n_reads == n_reads && (p->seq_mode != 2 || c->pre[k].key2 == (const void *)packed2_mate)) set = k;
    const bool pre = set >= 0;
    if (!pre) {
        /* not prefetched: upload on the compute stream into a set no outstanding prefetch (of a LATER batch) is filling */
        if (!c->pre[c->pk_set].valid) set = c->pk_set;
        else if (!c->pre[c->pk_set ^ 1].valid) set = c->pk_set ^ 1;
        else {      /* both sets hold prefetches nobody came for: let them finish, drop them */
            HIPCHK(hipStreamSynchronize(c->copy_stream));
            c->pre[0].valid = c->pre[1].valid = false; set = c->pk_set;
        }
    } else {
        c->pre[set].valid = false; c->prefetch_used++;
        HIPCHK(hipStreamWaitEvent(c->stream, c->copy_done[set], 0));
    }
    c->pk_set = set;
    STCHK(upload_packed(c, "", set, pre, packed2, nmask, lens, n_reads, &d_b, &d_o, &nb));
    if (p->seq_mode == 2) STCHK(upload_packed(c, "2", set, pre, packed2_mate, nmask_mate, lens_mate, n_reads, &d_b2, &d_o2, &nb2));
    if (c->unpacked[set]) { HIPCHK(hipEventRecord(c->unpacked[set], c->stream)); c->unpacked_rec[set] = true; }      /* the set may be overwritten by a prefetch from here on (stream order) */
    mtb_result *d_res; int32_t *d_tt; uint32_t *d_tc;
    const uint64_t dcap = c->lanes.size() < 2 ? std::max<uint64_t>(taxcnt_cap, taxcnt_device_slots(p, n_reads, nb + nb2)) : taxcnt_cap;     /* (as in mtb_classify_batch) */
    STCHK(ensure(c, rset == 1 ? "resultsb" : "results", n_reads, &d_res)); STCHK(ensure(c, "tctax", dcap, &d_tt)); STCHK(ensure(c, "tccnt", dcap, &d_tc));
    if (timing) HIPCHK(hipStreamSynchronize(c->stream));
    const double t1 = now();
    mtb_status st = mtb_classify_batch_device(c, ix, p, d_b, d_o, d_b2, d_o2, n_reads, nb + nb2, d_res, d_tt, d_tc, dcap, n_taxcnt);
    if (st != MTB_OK) return st;
    const double t2 = now();
    if (c->lanes.size() < 2 && *n_taxcnt) {
        if (async) {
            /* the previous call's copies have had this whole batch to finish: its results are the caller's from here on (mtb.h) */
            STCHK(wait_results(c));
            st = download_packed(c, d_res, d_tt, d_tc, n_reads, results, taxcnt_tax, taxcnt_cnt, n_taxcnt, taxcnt_cap, rset);
            if (st == MTB_OK) c->res_set ^= 1;
            return st;
        }
        st = download_packed(c, d_res, d_tt, d_tc, n_reads, results, taxcnt_tax, taxcnt_cnt, n_taxcnt, taxcnt_cap);
        if (timing) fprintf(stderr, "mtb_classify_batch_packed: %llu reads: upload + unpack %.1f ms, classify %.1f ms (device %.1f ms), pack + download %.1f ms\n",
                            (unsigned long long)n_reads, t1 - t0, t2 - t1, (double)c->stats.ms_total, now() - t2);
        return st;
    }
    /* (several streams, or a batch without a single taxID:count entry: plain copies) */
    if (async) STCHK(wait_results(c));
    STCHK(d2h(c, results, d_res, n_reads * sizeof(mtb_result)));
    if (*n_taxcnt) { STCHK(d2h(c, taxcnt_tax, d_tt, *n_taxcnt * 4)); STCHK(d2h(c, taxcnt_cnt, d_tc, *n_taxcnt * 4)); }
    return MTB_OK;
}

extern "C" {

mtb_status mtb_classify_batch_packed(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                     const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n_reads,
                                     mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    return classify_packed_impl(c, ix, p, packed2, nmask, lens, packed2_mate, nmask_mate, lens_mate, n_reads, results, taxcnt_tax, taxcnt_cnt, taxcnt_cap, n_taxcnt, false);
}

/* The same with the results on their way back while the next batch computes: returns once the copies of the rows and of the packed
 * taxID:count lists into the caller's (pinned) arrays are QUEUED on a download stream of the context; *n_taxcnt is final on return
 * (MTB_ERR_CAPACITY as before, nothing queued).  The arrays of call k belong to the library until call k + 1 of this context returns --
 * whatever its status -- or until mtb_ctx_wait_results().  Two device-side result buffer sets alternate, so the batch in work never writes
 * what the copies in flight read.  (Classifier.cpp:81-125 hands a batch's results on when the batch is through; here the hand-over of
 * batch k happens one batch later and costs the device nothing.) */
mtb_status mtb_classify_batch_packed_async(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                           const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n_reads,
                                           mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    mtb_status st = classify_packed_impl(c, ix, p, packed2, nmask, lens, packed2_mate, nmask_mate, lens_mate, n_reads, results, taxcnt_tax, taxcnt_cnt, taxcnt_cap, n_taxcnt, true);
    if (st != MTB_OK && c && c->down_pending) {     /* the contract holds on every path: the previous call's copies are complete when this one returns */
        hipError_t e = hipEventSynchronize(c->down_done); (void)e; c->down_pending = false;
    }
    return st;
}
mtb_status mtb_ctx_wait_results(mtb_ctx *c) {
    if (!c) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    return wait_results(c);
}

/* Starts the upload of the NEXT batch (the arrays mtb_classify_batch_packed will be called with) on the context's copy stream, into
 * the input buffer set the running batch does not use, and returns at once: the PCIe transfer of batch k+1 overlaps the kernels of
 * batch k (the serial producer this replaces: KmerExtractor.cpp:117-173 fills a batch, then the batch is processed).  Call it from the
 * context's thread right before the mtb_classify_batch_packed of batch k; the arrays must stay valid (pinned: mtb_host_alloc) until
 * the classify call for batch k+1 returns.  A prefetch that is not followed by its classify call is discarded. */
mtb_status mtb_prefetch_batch_packed(mtb_ctx *c, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                     const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n_reads) {
    if (!c || !p || !packed2 || !nmask || !lens) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    if (n_reads == 0 || c->lanes.size() > 1) return MTB_OK;
    if (p->seq_mode == 2 && (!packed2_mate || !nmask_mate || !lens_mate)) return fail(MTB_ERR_ARG, "seq_mode 2 needs the mates");
    if (!c->copy_stream) {
        HIPCHK(hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        for (int k = 0; k < 2; k++) { HIPCHK(hipEventCreateWithFlags(&c->copy_done[k], hipEventDisableTiming)); HIPCHK(hipEventCreateWithFlags(&c->unpacked[k], hipEventDisableTiming)); }
    }
    /* the set that holds no outstanding prefetch: in protocol order (prefetch(k+1) right before classify(k)) batch k waits in one set and
     * the other was read by classify(k-1).  Both taken = the caller ran ahead of the protocol: nothing is prefetched, the batch is uploaded
     * by its own classify call. */
    int set = c->pk_set ^ 1;
    if (c->pre[set].valid) set ^= 1;
    if (c->pre[set].valid) return MTB_OK;
    if (c->unpacked_rec[set]) HIPCHK(hipStreamWaitEvent(c->copy_stream, c->unpacked[set], 0));     /* the call that last read the set has turned it into text */
    uint8_t *a, *b; uint32_t *l;
    const uint64_t s1 = packed_slots(lens, n_reads);
    STCHK(copy_packed(c, "", set, c->copy_stream, packed2, nmask, lens, n_reads, s1, &a, &b, &l));
    if (p->seq_mode == 2) STCHK(copy_packed(c, "2", set, c->copy_stream, packed2_mate, nmask_mate, lens_mate, n_reads, packed_slots(lens_mate, n_reads), &a, &b, &l));
    HIPCHK(hipEventRecord(c->copy_done[set], c->copy_stream));
    c->pre[set].key = packed2; c->pre[set].key2 = packed2_mate; c->pre[set].n_reads = n_reads; c->pre[set].valid = true;
    c->prefetch_issued++;
    return MTB_OK;
}

/* prefetches issued / prefetches a classify call found and used instead of uploading the batch itself (in protocol order: all of them) */
mtb_status mtb_ctx_prefetch_stats(mtb_ctx *c, uint64_t *issued, uint64_t *used) {
    if (!c) return fail(MTB_ERR_ARG, "NULL argument");
    if (issued) *issued = c->prefetch_issued;
    if (used) *used = c->prefetch_used;
    return MTB_OK;
}

/* ------------------------------------------------------------------ */
/* partitioned index: device-buffer stage calls  */
/* ------------------------------------------------------------------ */
} // extern "C"

/* pos[p] = number of entries of the sorted array with value < bounds[p]; stride in 8-byte words */
__global__ void k_lower_bounds(const uint64_t *__restrict__ v, uint64_t n, uint32_t stride, const uint64_t *__restrict__ bounds, uint32_t nb,
                               uint64_t *__restrict__ pos) {
    uint32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= nb) return;
    const uint64_t b = bounds[p];
    uint64_t lo = 0, hi = n;
    while (lo < hi) { uint64_t mid = (lo + hi) >> 1; if (v[mid * stride] < b) lo = mid + 1; else hi = mid; }
    pos[p] = lo;
}
static mtb_status lower_bounds(mtb_ctx *c, const uint64_t *d_v, uint64_t n, uint32_t stride, const uint64_t *bounds, uint32_t nb, uint64_t *pos) {
    uint64_t *d_b;
    STCHK(ensure(c, "lbounds", 2ull * nb, &d_b));
    STCHK(h2d(c, d_b, bounds, 8ull * nb));
    hipLaunchKernelGGL(k_lower_bounds, dim3((nb + 63) / 64), dim3(64), 0, c->stream, d_v, n, stride, (const uint64_t *)d_b, nb, d_b + nb);
    HIPCHK(hipGetLastError());
    return d2h(c, pos, d_b + nb, 8ull * nb);
}

extern "C" {

mtb_status mtb_index_slice(mtb_index *ix, uint64_t lo_value, uint64_t hi_value, int is_last, mtb_index **out) {
    if (!ix || !out) return fail(MTB_ERR_ARG, "NULL argument");
    mtb_ctx *c = ix->ctx;
    HIPCHK(hipSetDevice(c->device));
    uint64_t b[2] = {lo_value, hi_value}, pos[2] = {0, 0};
    if (ix->parent) return fail(MTB_ERR_ARG, "a view of a view: slice the parent instead");
    {   /* a view reads the parent's flat arrays: the parent goes flat now and stays flat while views are alive (ensure_packed /
           mtb_index_seal refuse, the fused join takes its flat-state kernel) */
        std::unique_lock<std::mutex> lk(ix->state_mu);
        ix->state_cv.wait(lk, [&] { return ix->users == 0; });
        STCHK(ensure_flat_locked(ix));
        ix->views++;
    }
    mtb_status st_lb = MTB_OK;
    if (ix->T) st_lb = lower_bounds(c, ix->d_values, ix->T, 1, b, 2, pos);
    if (st_lb != MTB_OK) { std::lock_guard<std::mutex> lk(ix->state_mu); ix->views--; return st_lb; }
    if (hi_value == UINT64_MAX) pos[1] = ix->T;
    mtb_index *sl = new mtb_index();
    sl->ctx = ix->ctx; sl->tax = ix->tax; sl->params = ix->params; sl->info_mask = ix->info_mask;
    sl->d_canon = ix->d_canon; sl->d_parent = ix->d_parent; sl->d_depth = ix->d_depth; sl->d_spparent = ix->d_spparent; sl->d_tax2species = ix->d_tax2species;
    sl->d_under = ix->d_under; sl->d_accleaf = ix->d_accleaf; sl->d_node = ix->d_node;
    sl->parent = ix;
    sl->own = false; sl->own_tax = false; sl->d_dir = nullptr; sl->d_dirbase = nullptr; sl->dir_L = 0;
    sl->d_values = ix->d_values + pos[0]; sl->d_info = ix->d_info + pos[0]; sl->T = pos[1] - pos[0];
    sl->match_last = !is_last || ix->match_last;
    {   /* a view gets a directory of its own (over its part of the parent's flat array): the owner side of the partitioned path joins through it */
        mtb_status sd = build_directory(c, sl);
        if (sd != MTB_OK) { mtb_index_close(sl); return sd; }
    }
    *out = sl;
    return MTB_OK;
}

mtb_status mtb_part_extract(mtb_ctx *c, const mtb_params *p, const char *d_bases, const uint64_t *d_offs, const char *d_bases2,
                            const uint64_t *d_offs2, uint64_t n_reads, const uint64_t *bounds, uint32_t n_parts, const mtb_kmer **d_sorted,
                            uint64_t *n_kmers, uint64_t *part_counts, uint64_t *part_starts) {
    if (!c || !p || !d_offs || !bounds || !d_sorted || !n_kmers || !part_counts || n_parts == 0) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *d_sorted = nullptr; *n_kmers = 0;
    for (uint32_t q = 0; q < n_parts; q++) { part_counts[q] = 0; if (part_starts) part_starts[q] = 0; }
    c->part_n_reads = n_reads; c->part_max_len = 0; c->part_mode = 0; c->part_max_q = 0; c->part_nk_real = 0;
    if (n_reads == 0) return MTB_OK;
    int32_t *d_ql, *d_ql2;
    STCHK(ensure(c, "qlen", n_reads, &d_ql)); STCHK(ensure(c, "qlen2", n_reads, &d_ql2));
    mtb_kmer *d_k, *d_s; uint64_t nk; uint32_t max_len = 0;
    if (part_starts && p->seq_mode != 3 && !c->opt.part_exact) {
        /* the product path of short reads: single-pass extraction with ordinals, the sort on the leading amino-acid letters only (it is
         * there for the locality of the owners' directory joins) -- so a run is cut at PREFIX granularity and the metamers that share
         * their prefix with a bound go to both neighbours: a query finds candidates only in the range that holds its amino-acid
         * group (no group straddles a cut), the other owner emits nothing for it */
        uint32_t max_q = 0; uint64_t nk_real = 0; uint16_t *d_dig = nullptr;
        const bool aa6 = p->kmer_format == 2;
        STCHK(dev_extract(c, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, &d_k, &nk, d_ql, d_ql2, &max_len, true, 0, &nk_real, true, &max_q, aa6 ? &d_dig : nullptr));
        if (max_len + 3 < MTB_SLOT_MAX_POS && max_q <= 384) {
            const int low_bits = aa6 ? 34 : 32;
            STCHK(dev_sort(c, d_k, nk, aa6 ? MTB_SORT_AA6 : 32, &d_s, d_dig));
            c->part_max_len = max_len; c->part_mode = 1; c->part_max_q = max_q; c->part_nk_real = nk_real;
            std::vector<uint64_t> key(2 * n_parts), pos(2 * n_parts);
            for (uint32_t q = 0; q < n_parts; q++) {
                const uint64_t pre = bounds[q] >> low_bits;
                key[2 * q] = pre << low_bits;                                                       /* first record whose prefix is >= the bound's */
                key[2 * q + 1] = pre + 1 >= (1ull << (64 - low_bits)) ? UINT64_MAX : (pre + 1) << low_bits;      /* first record behind the bound's prefix group */
            }
            if (nk) STCHK(lower_bounds(c, (const uint64_t *)d_s, nk, 2, key.data(), 2 * n_parts, pos.data()));
            for (uint32_t q = 0; q < n_parts; q++) {
                const uint64_t lo = q == 0 ? 0 : pos[2 * q];
                const uint64_t hi = q + 1 < n_parts ? (key[2 * (q + 1) + 1] == UINT64_MAX ? nk : pos[2 * (q + 1) + 1]) : nk;
                part_starts[q] = lo; part_counts[q] = hi > lo ? hi - lo : 0;
            }
            *d_sorted = d_s; *n_kmers = nk;
            return MTB_OK;
        }
    }
    STCHK(dev_extract(c, p, d_bases, d_offs, d_bases2, d_offs2, n_reads, &d_k, &nk, d_ql, d_ql2, &max_len));
    /* partition boundaries are amino-acid-part boundaries (bit 24), finer than the 32-bit tiles of the fused path: 5 passes */
    STCHK(dev_sort(c, d_k, nk, 24, &d_s));
    c->part_max_len = max_len;
    std::vector<uint64_t> pos(n_parts);
    if (nk) STCHK(lower_bounds(c, (const uint64_t *)d_s, nk, 2, bounds, n_parts, pos.data()));
    for (uint32_t q = 0; q < n_parts; q++) { uint64_t hi = q + 1 < n_parts ? pos[q + 1] : nk; part_counts[q] = hi >= pos[q] ? hi - pos[q] : 0; if (part_starts) part_starts[q] = pos[q]; }
    if (nk && pos[0] != 0) return fail(MTB_ERR_ARG, "bounds[0] must be 0");
    *d_sorted = d_s; *n_kmers = nk;
    return MTB_OK;
}

mtb_status mtb_part_join(mtb_ctx *c, mtb_index *ix, const mtb_kmer *d_kmers, uint64_t n, mtb_match *d_out, uint64_t cap, uint64_t *count) {
    if (!c || !ix || !count) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    *count = 0;
    if (n == 0 || ix->T == 0) return MTB_OK;
    if (ix->d_dir && !c->opt.part_exact) {
        /* the directory join, matches to a dense list: every record keeps its query's qinfo (with the ordinal tag of a slot-mode
         * batch) and says in `pad` whether it is the query's first match */
        JoinSegArgs sa; memset(&sa, 0, sizeof(sa));
        sa.list = 1; sa.ovf = d_out; sa.ovf_cap = cap;
        return dev_join(c, ix, d_kmers, n, nullptr, 0, nullptr, count, &sa);
    }
    /* no directory over this range (MTB_NO_DIR, no room, a letter >= 21, a tiny range): the bisection join.  The runs of a slot-mode
     * batch are ordered on the leading amino-acid letters only (bits [34, 64) for kmer_format 2, mtb_part_extract), and k_join_bounds
     * derives a tile's target window from its first and last query under exactly that assumption -- so the sender's granularity is
     * passed on (the exact-order runs of the other modes are ordered on those bits too).  Records leave with pad = 0: the home rank
     * places them through the tails / the overflow list, which is slower but exact. */
    return dev_join(c, ix, d_kmers, n, d_out, cap, nullptr, count, nullptr, ix->params.kmer_format == 2 ? 34 : 32);
}

mtb_status mtb_part_score(mtb_ctx *c, mtb_index *ix, const mtb_params *p, mtb_match *d_matches, uint64_t n_matches, uint64_t n_reads,
                          mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    if (!c || !ix || !p || !results || !n_taxcnt) return fail(MTB_ERR_ARG, "NULL argument");
    if (n_reads != c->part_n_reads) return fail(MTB_ERR_ARG, "mtb_part_score must follow mtb_part_extract of the same batch on the same context");
    HIPCHK(hipSetDevice(c->device));
    *n_taxcnt = 0;
    if (n_reads == 0) return MTB_OK;
    if (n_matches >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 matches in one batch; split the batch");
    int32_t *d_ql, *d_ql2; uint32_t *d_rc; mtb_result *d_res; int32_t *d_tt; uint32_t *d_tc;
    STCHK(ensure(c, "qlen", n_reads, &d_ql)); STCHK(ensure(c, "qlen2", n_reads, &d_ql2));
    STCHK(ensure(c, "readcnt", n_reads, &d_rc));
    STCHK(ensure(c, "results", n_reads, &d_res)); STCHK(ensure(c, "tctax", taxcnt_cap, &d_tt)); STCHK(ensure(c, "tccnt", taxcnt_cap, &d_tc));
    if (c->part_mode == 1) {
        /* slot mode: the matches are placed into this rank's slot segments by their ordinal (what the fused join does at once),
         * then the slot scorers run */
        hipStream_t st = c->stream;
        uint32_t direct, stride;
        slot_geometry(c, c->part_max_q, &direct, &stride);
        mtb_slot16 *d_segm; mtb_match *d_ovf; uint32_t epoch = 0; uint64_t n_ovf = 0;
        STCHK(prepare_slots(c, n_reads, stride, &d_segm, &epoch));
        DevBuf &ob = c->bufs["ovf"];
        uint64_t ovf_cap = std::max<uint64_t>(ob.cap / sizeof(mtb_match), n_matches / 64 + 4096);
        for (int attempt = 0; attempt < 3; attempt++) {
            STCHK(ensure(c, "ovf", ovf_cap, &d_ovf));
            HIPCHK(hipMemsetAsync(d_rc, 0, n_reads * 4, st));
            HIPCHK(hipMemsetAsync(c->d_scal, 0, 16, st));
            JoinSegArgs sa; memset(&sa, 0, sizeof(sa));
            sa.seg = d_segm; sa.stride = stride; sa.direct = direct; sa.cursor = d_rc; sa.ovf = d_ovf; sa.ovf_cap = ovf_cap; sa.ovf_counter = (unsigned long long *)c->d_scal; sa.epoch = epoch;
            if (n_matches) hipLaunchKernelGGL(k_slot_place, dim3((uint32_t)((n_matches + 255) / 256)), dim3(256), 0, st, (const mtb_match *)d_matches, n_matches, sa, n_reads, (uint32_t *)(c->d_scal + 1));
            HIPCHK(hipGetLastError());
            uint64_t sc[2];
            STCHK(d2h(c, sc, c->d_scal, 16));
            if (sc[1] & 0xFFFFFFFFull) return fail(MTB_ERR_ARG, "match records with a sequenceID outside 1..n_reads");
            n_ovf = sc[0];
            if (n_ovf <= ovf_cap) break;
            if (attempt == 2) return fail(MTB_ERR_CAPACITY, "overflow list too small");
            ovf_cap = n_ovf + n_ovf / 16 + 1024;          /* slots written by the failed attempt are rewritten identically */
        }
        uint64_t nm = 0;
        memset(&c->stats, 0, sizeof(c->stats));
        STCHK(score_fixed_slots(c, ix, p, n_reads, d_ql, d_ql2, c->part_max_len, c->part_nk_real, d_segm, d_rc, stride, direct, epoch, d_ovf, n_ovf,
                                d_res, d_tt, d_tc, taxcnt_cap, n_taxcnt, 0, &nm));
        c->stats.n_reads = n_reads; c->stats.n_matches = nm; c->stats.n_slot_reads = n_reads; c->stats.n_kmers = c->part_nk_real;
        if (c->fast_used) { uint64_t ns = 0; STCHK(d2h(c, &ns, c->d_scal + 6, 8)); c->stats.n_generic_reads = ns & 0xFFFFFFFFull; c->fast_used = false; }
        return download_packed(c, d_res, d_tt, d_tc, n_reads, results, taxcnt_tax, taxcnt_cnt, n_taxcnt);
    }
    STCHK(count_reads(c, d_matches, n_matches, n_reads, d_rc));
    STCHK(score_join_order(c, ix, p, d_matches, n_matches, n_reads, d_rc, d_ql, d_ql2, c->part_max_len, d_res, d_tt, d_tc, taxcnt_cap, n_taxcnt, 0));
    return download_packed(c, d_res, d_tt, d_tc, n_reads, results, taxcnt_tax, taxcnt_cnt, n_taxcnt);
}

/* One batch on a group of contexts of THIS process, context k holding range k of the database (mtb_index_open_part(.., k, n)):
 * the C++ host's form of SURVEY 8(e) row 2 (metabuli_amd/parallel.py is the torch.distributed form of the same steps).
 *   1. every context: its contiguous share of the reads -> device, mtb_part_extract (ordinal tags, prefix-granular runs);
 *   2. exchange #1: run p of every context -> the owner of range p (hipMemcpyPeerAsync, device to device over xGMI), which joins
 *      every received run through its directory (mtb_part_join: runs stay sorted, no merge);
 *   3. exchange #2: the matches of source k's reads -> context k, which places them into its slot segments and scores
 *      (mtb_part_score);
 *   4. rows in input order, the taxID:count lists packed back to back behind taxcnt_off.
 * One host thread per context inside each step; the steps are separated by joins (the exchange needs every sender's counts). */
static mtb_status run_parallel_status(uint32_t n, const std::function<mtb_status(uint32_t)> &f, std::string *err) {
    std::vector<mtb_status> st(n, MTB_OK); std::vector<std::string> errs(n);
    std::vector<std::thread> th;
    for (uint32_t k = 0; k < n; k++) th.emplace_back([&, k] { st[k] = f(k); if (st[k] != MTB_OK) errs[k] = g_err; });
    for (auto &t : th) t.join();
    for (uint32_t k = 0; k < n; k++) if (st[k] != MTB_OK) { *err = errs[k]; return st[k]; }
    return MTB_OK;
}

mtb_status mtb_classify_batch_partitioned(mtb_ctx **ctxs, mtb_index **parts, uint32_t n, const uint64_t *bounds, const mtb_params *p,
                                          const char *bases, const uint64_t *offs, const char *bases2, const uint64_t *offs2, uint64_t n_reads,
                                          mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt) {
    if (!ctxs || !parts || !bounds || !p || !n_taxcnt || n == 0) return fail(MTB_ERR_ARG, "NULL argument");
    *n_taxcnt = 0;
    if (n_reads == 0) return MTB_OK;
    if (!bases || !offs || !results) return fail(MTB_ERR_ARG, "NULL argument");
    const bool paired = p->seq_mode == 2;
    if (paired && (!bases2 || !offs2)) return fail(MTB_ERR_ARG, "seq_mode 2 needs bases2/offs2");
    struct Rank {
        uint64_t lo = 0, hi = 0;                          /* its reads */
        const mtb_kmer *d_sorted = nullptr; uint64_t nk = 0;
        std::vector<uint64_t> counts, starts;             /* runs by owner */
        std::vector<uint64_t> m_count, m_start;           /* as an owner: matches per source, their start in its match buffer */
        mtb_match *d_matches = nullptr;                   /* as an owner */
        std::vector<mtb_result> res; std::vector<int32_t> tt; std::vector<uint32_t> tc; uint64_t ntc = 0;
    };
    /* direct device-to-device copies over xGMI where the devices can reach each other (without peer access hipMemcpyPeerAsync stages
     * through host memory); enabling it twice is reported as an error that means "already on" */
    for (uint32_t i = 0; i < n; i++) for (uint32_t j = 0; j < n; j++) {
        if (!ctxs[i] || !ctxs[j] || ctxs[i]->device == ctxs[j]->device) continue;
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, ctxs[i]->device, ctxs[j]->device) != hipSuccess) { (void)hipGetLastError(); continue; }
        if (!can) continue;
        if (hipSetDevice(ctxs[i]->device) != hipSuccess) { (void)hipGetLastError(); continue; }
        const hipError_t e = hipDeviceEnablePeerAccess(ctxs[j]->device, 0);
        if (e != hipSuccess) (void)hipGetLastError();
    }
    std::vector<Rank> R(n);
    for (uint32_t k = 0; k < n; k++) { R[k].lo = n_reads * k / n; R[k].hi = n_reads * (k + 1) / n; R[k].counts.assign(n, 0); R[k].starts.assign(n, 0); R[k].m_count.assign(n, 0); R[k].m_start.assign(n, 0); }
    std::string err;
    /* 1. reads to their device, extraction */
    mtb_status st = run_parallel_status(n, [&](uint32_t k) -> mtb_status {
        mtb_ctx *c = ctxs[k]; Rank &r = R[k];
        const uint64_t m = r.hi - r.lo;
        if (m == 0) return MTB_OK;
        HIPCHK(hipSetDevice(c->device));
        std::vector<uint64_t> o(m + 1), o2(paired ? m + 1 : 0);
        for (uint64_t i = 0; i <= m; i++) o[i] = offs[r.lo + i] - offs[r.lo];
        if (paired) for (uint64_t i = 0; i <= m; i++) o2[i] = offs2[r.lo + i] - offs2[r.lo];
        char *d_b, *d_b2; uint64_t *d_o, *d_o2; uint64_t nb;
        STCHK(upload_reads(c, p, bases + offs[r.lo], o.data(), paired ? bases2 + offs2[r.lo] : nullptr, paired ? o2.data() : nullptr, m, &d_b, &d_o, &d_b2, &d_o2, &nb));
        return mtb_part_extract(c, p, d_b, d_o, d_b2, d_o2, m, bounds, n, &r.d_sorted, &r.nk, r.counts.data(), r.starts.data());
    }, &err);
    if (st != MTB_OK) return fail(st, err);
    /* 2. runs to their owners, joins */
    st = run_parallel_status(n, [&](uint32_t q) -> mtb_status {
        mtb_ctx *c = ctxs[q]; Rank &own = R[q];
        HIPCHK(hipSetDevice(c->device));
        uint64_t n_in = 0;
        for (uint32_t k = 0; k < n; k++) n_in += R[k].counts[q];
        if (n_in == 0) return MTB_OK;
        mtb_kmer *d_in;
        STCHK(ensure(c, "xkmers", n_in, &d_in));
        uint64_t at = 0;
        for (uint32_t k = 0; k < n; k++) {
            const uint64_t cnt = R[k].counts[q];
            if (cnt) HIPCHK(hipMemcpyPeerAsync(d_in + at, c->device, R[k].d_sorted + R[k].starts[q], ctxs[k]->device, cnt * sizeof(mtb_kmer), c->stream));
            at += cnt;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        uint64_t cap = n_in + n_in / 4 + 1024;
        for (int attempt = 0; attempt < 4; attempt++) {
            STCHK(ensure(c, "xmatches", cap, &own.d_matches));
            uint64_t used = 0, need = 0; bool again = false;
            at = 0;
            for (uint32_t k = 0; k < n && !again; k++) {
                const uint64_t cnt = R[k].counts[q];
                uint64_t got = 0;
                own.m_start[k] = used;
                if (cnt) {
                    mtb_status sj = mtb_part_join(c, parts[q], d_in + at, cnt, own.d_matches + used, cap - used, &got);
                    if (sj == MTB_ERR_CAPACITY) { again = true; need = used + got; }
                    else if (sj != MTB_OK) return sj;
                }
                own.m_count[k] = got; used += got; at += cnt;
            }
            if (!again) return MTB_OK;
            cap = need + need / 2 + 1024;                 /* (the runs behind the one that did not fit are not counted yet) */
        }
        return fail(MTB_ERR_CAPACITY, "match buffer of a range owner too small");
    }, &err);
    if (st != MTB_OK) return fail(st, err);
    /* 3. matches home, scoring */
    st = run_parallel_status(n, [&](uint32_t k) -> mtb_status {
        mtb_ctx *c = ctxs[k]; Rank &r = R[k];
        const uint64_t m = r.hi - r.lo;
        if (m == 0) return MTB_OK;
        HIPCHK(hipSetDevice(c->device));
        uint64_t nm = 0;
        for (uint32_t q = 0; q < n; q++) nm += R[q].m_count[k];
        mtb_match *d_home;
        STCHK(ensure(c, "xhome", nm, &d_home));
        uint64_t at = 0;
        for (uint32_t q = 0; q < n; q++) {
            const uint64_t cnt = R[q].m_count[k];
            if (cnt) HIPCHK(hipMemcpyPeerAsync(d_home + at, c->device, R[q].d_matches + R[q].m_start[k], ctxs[q]->device, cnt * sizeof(mtb_match), c->stream));
            at += cnt;
        }
        HIPCHK(hipStreamSynchronize(c->stream));
        r.res.resize(m);
        uint64_t cap = 24 * m + 4096;
        for (;;) {
            r.tt.resize(cap); r.tc.resize(cap);
            mtb_status ss = mtb_part_score(c, parts[k], p, d_home, nm, m, r.res.data(), r.tt.data(), r.tc.data(), cap, &r.ntc);
            if (ss == MTB_ERR_CAPACITY && r.ntc > cap) { cap = r.ntc; continue; }
            return ss;
        }
    }, &err);
    if (st != MTB_OK) return fail(st, err);
    /* 4. rows in input order, lists packed */
    uint64_t used = 0;
    for (uint32_t k = 0; k < n; k++) for (uint64_t i = 0; i < R[k].res.size(); i++) used += R[k].res[i].n_taxcnt;
    *n_taxcnt = used;
    if (used > taxcnt_cap) return fail(MTB_ERR_CAPACITY, "taxcnt buffers too small");
    if (used >= (1ull << 32)) return fail(MTB_ERR_ARG, "more than 2^32-1 taxcnt entries in one batch; split the batch");
    uint64_t at = 0;
    for (uint32_t k = 0; k < n; k++) {
        Rank &r = R[k];
        for (uint64_t i = 0; i < r.res.size(); i++) {
            mtb_result x = r.res[i];
            for (uint32_t j = 0; j < x.n_taxcnt; j++) { taxcnt_tax[at + j] = r.tt[x.taxcnt_off + j]; taxcnt_cnt[at + j] = r.tc[x.taxcnt_off + j]; }
            x.taxcnt_off = (uint32_t)at; at += x.n_taxcnt;
            results[r.lo + i] = x;
        }
    }
    return MTB_OK;
}

#ifdef MTB_SCORE_PHASE_CYCLES
/* profiling build only: read and reset the k_score phase cycle counters */
mtb_status mtb_debug_phase_cycles(mtb_ctx *c, unsigned long long *out4) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyFromSymbol(out4, HIP_SYMBOL(mtb_phase_cycles), 8 * MTB_NPHASE));      /* out4: MTB_NPHASE (16) counters */
    unsigned long long z[MTB_NPHASE] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mtb_phase_cycles), z, 8 * MTB_NPHASE));
    /* join phases ride in out4[16..23] */
    HIPCHK(hipMemcpyFromSymbol(out4 + MTB_NPHASE, HIP_SYMBOL(mtb_join_cycles), 64));
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mtb_join_cycles), z, 64));
    return MTB_OK;
}
#endif

#ifdef MTB_FAST_DEBUG
/* debugging build only: read and reset the k_score_fast exit counters */
mtb_status mtb_debug_fast_reasons(mtb_ctx *c, unsigned long long *out8) {
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipMemcpyFromSymbol(out8, HIP_SYMBOL(mtb_fast_reasons), 256));      /* 32 counters */
    unsigned long long z[32] = {0};
    HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(mtb_fast_reasons), z, 256));
    return MTB_OK;
}
#endif

#ifdef MTB_PLACEMENT_DEBUG
/* experiment build only (libmtb_place.so): release one workspace buffer and park `pad_bytes` of HBM so that its next allocation
 * lands somewhere else -- does the join's process-to-process spread follow the placement of the slot buffer? */
mtb_status mtb_debug_move_buffer(mtb_ctx *c, const char *name, unsigned long long pad_bytes) {
    HIPCHK(hipSetDevice(c->device));
    HIPCHK(hipStreamSynchronize(c->stream));
    auto it = c->bufs.find(name);
    void *old = nullptr;
    if (it != c->bufs.end()) { old = it->second.p; it->second.p = nullptr; it->second.cap = 0; }
    void *pad = nullptr;
    if (pad_bytes) { hipError_t e = hipMalloc(&pad, pad_bytes); if (e != hipSuccess) { (void)hipGetLastError(); pad = nullptr; } }   /* never freed: the process ends with the experiment */
    if (old) { hipError_t e = hipFree(old); (void)e; }
    return MTB_OK;
}
#endif

mtb_status mtb_ctx_join_footprint(mtb_ctx *c, mtb_index *ix, mtb_join_footprint *out) {
    if (!c || !ix || !out) return fail(MTB_ERR_ARG, "NULL argument");
    memset(out, 0, sizeof(*out));
    if (!c->last_sorted || !ix->d_dir || c->last_sub_batches != 1) return fail(MTB_ERR_UNSUPPORTED, "no directory join of a single sub-batch to look at");
    HIPCHK(hipSetDevice(c->device));
    const uint64_t nb = ix->dir_buckets, T = ix->T;
    const uint64_t w_b = (nb >> 5) + 1, w_d = (((nb + 1) * 4) >> 11) + 1, w_t = ((T * 8) >> 11) + 1;
    uint32_t *bm = nullptr; unsigned long long *d_cnt = nullptr;
    if (hipMalloc((void **)&bm, (w_b + w_d + w_t) * 4) != hipSuccess) { (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no HBM for the footprint bitmaps"); }
    if (hipMalloc((void **)&d_cnt, 32) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(bm); return fail(MTB_ERR_OOM, "no HBM for the footprint counters"); }
    mtb_status st = MTB_OK;
    hipError_t e = hipMemsetAsync(bm, 0, (w_b + w_d + w_t) * 4, c->stream);
    if (e == hipSuccess) e = hipMemsetAsync(d_cnt, 0, 32, c->stream);
    if (e == hipSuccess && c->last_sorted_n) {
        hipLaunchKernelGGL(k_join_footprint, dim3((uint32_t)((c->last_sorted_n + 255) / 256)), dim3(256), 0, c->stream, c->last_sorted, c->last_sorted_n, dir_view(ix), T,
                           bm, bm + w_b, bm + w_b + w_d, d_cnt);
        hipLaunchKernelGGL(k_popcount_words, dim3(2048), dim3(256), 0, c->stream, (const uint32_t *)bm, w_b, d_cnt + 1);
        hipLaunchKernelGGL(k_popcount_words, dim3(2048), dim3(256), 0, c->stream, (const uint32_t *)(bm + w_b), w_d, d_cnt + 2);
        hipLaunchKernelGGL(k_popcount_words, dim3(2048), dim3(256), 0, c->stream, (const uint32_t *)(bm + w_b + w_d), w_t, d_cnt + 3);
        e = hipGetLastError();
    }
    unsigned long long h[4] = {0, 0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, d_cnt, 32, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) st = fail(MTB_ERR_DEVICE, std::string("join footprint: ") + hipGetErrorString(e));
    (void)hipFree(bm); (void)hipFree(d_cnt);
    out->n_queries = h[0]; out->distinct_buckets = h[1]; out->dir_sectors = h[2]; out->target_sectors = h[3]; out->n_buckets = nb; out->n_targets = T;
    return st;
}

mtb_status mtb_index_run_histogram(mtb_index *ix, uint64_t *hist64) {
    if (!ix || !hist64) return fail(MTB_ERR_ARG, "NULL argument");
    memset(hist64, 0, 64 * sizeof(uint64_t));
    if (ix->T == 0) return MTB_OK;
    mtb_ctx *c = ix->ctx;
    HIPCHK(hipSetDevice(c->device));
    STCHK(ensure_flat(ix));
    unsigned long long *d_h = nullptr;
    if (hipMalloc((void **)&d_h, 64 * 8) != hipSuccess) { (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no HBM for the histogram"); }
    hipError_t e = hipMemsetAsync(d_h, 0, 64 * 8, c->stream);
    if (e == hipSuccess) { hipLaunchKernelGGL(k_index_run_hist, dim3((uint32_t)std::min<uint64_t>((ix->T + 255) / 256, 1u << 20)), dim3(256), 0, c->stream, (const uint64_t *)ix->d_values, ix->T, d_h); e = hipGetLastError(); }
    if (e == hipSuccess) e = hipMemcpyAsync(hist64, d_h, 64 * 8, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(d_h);
    if (e != hipSuccess) return fail(MTB_ERR_DEVICE, std::string("run histogram: ") + hipGetErrorString(e));
    return MTB_OK;
}

mtb_status mtb_ctx_join_run_histogram(mtb_ctx *c, mtb_index *ix, uint64_t *hist64) {
    if (!c || !ix || !hist64) return fail(MTB_ERR_ARG, "NULL argument");
    memset(hist64, 0, 64 * sizeof(uint64_t));
    if (!c->last_sorted || !ix->d_dir || c->last_sub_batches != 1) return fail(MTB_ERR_UNSUPPORTED, "no directory join of a single sub-batch to look at");
    HIPCHK(hipSetDevice(c->device));
    unsigned long long *d_h = nullptr;
    if (hipMalloc((void **)&d_h, 64 * 8) != hipSuccess) { (void)hipGetLastError(); return fail(MTB_ERR_OOM, "no HBM for the histogram"); }
    mtb_status st = MTB_OK;
    {
        IndexUse use;                                       /* the array keeps the state it is in while the kernel reads it */
        mtb_index *own = state_owner(ix);
        st = use.acquire(ix, own->packed);
        const uint64_t limit = ix->T ? ix->T - (ix->match_last ? 0 : 1) : 0;
        hipError_t e = st == MTB_OK ? hipMemsetAsync(d_h, 0, 64 * 8, c->stream) : hipSuccess;
        if (st == MTB_OK && e == hipSuccess && c->last_sorted_n) {
            const dim3 g((uint32_t)((c->last_sorted_n + 255) / 256));
            if (own->packed) hipLaunchKernelGGL((k_join_run_hist<true>), g, dim3(256), 0, c->stream, c->last_sorted, c->last_sorted_n, (const uint64_t *)ix->d_values, limit, dir_view(ix), d_h);
            else hipLaunchKernelGGL((k_join_run_hist<false>), g, dim3(256), 0, c->stream, c->last_sorted, c->last_sorted_n, (const uint64_t *)ix->d_values, limit, dir_view(ix), d_h);
            e = hipGetLastError();
        }
        if (st == MTB_OK && e == hipSuccess) e = hipMemcpyAsync(hist64, d_h, 64 * 8, hipMemcpyDeviceToHost, c->stream);
        if (st == MTB_OK && e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (st == MTB_OK && e != hipSuccess) st = fail(MTB_ERR_DEVICE, std::string("join run histogram: ") + hipGetErrorString(e));
    }
    (void)hipFree(d_h);
    return st;
}

/* db.parameters of a database directory applied to *p (what mtb_index_open does first), without opening anything */
mtb_status mtb_db_parameters(const char *dbdir, mtb_params *p) {
    if (!dbdir || !p) return fail(MTB_ERR_ARG, "NULL argument");
    int reduced = 0;
    mtbhost::load_db_parameters(dbdir, p, &reduced);
    return MTB_OK;
}

/* Grows the big workspace buffers of a short-read batch of that size ahead of time: the metamer buffers of the extractor and the sort,
 * their digit arrays, the slot segments (hipMalloc of ~35 GB for 4 M reads costs several hundred milliseconds -- the first batch of a
 * run used to pay them).  The sizes are those the first batch will ask for if its reads look like the estimate; a batch that needs
 * more simply grows a buffer as before.  MAY BE CALLED FROM ANOTHER THREAD while the context's thread is inside mtb_index_open (the
 * buffer table is locked per access): a driver hides the allocations behind the database load. */
mtb_status mtb_ctx_reserve(mtb_ctx *c, const mtb_params *p, uint64_t n_reads, uint64_t n_bases) {
    if (!c || !p) return fail(MTB_ERR_ARG, "NULL argument");
    if (n_reads == 0 || n_bases == 0 || p->seq_mode == 3) return MTB_OK;
    HIPCHK(hipSetDevice(c->device));
    std::lock_guard<std::mutex> rlk(c->reserve_mu);
    /* an index open on this context that runs short of device memory cancels the reservation (open_make_room): checked between buffers */
#define MTB_RESERVE_STEP(call) do { if (c->reserve_cancel) return MTB_OK; STCHK(call); } while (0)
    const uint32_t grid = (uint32_t)std::min<uint64_t>(n_reads, 256ull * 256);
    const uint64_t cap = extract_cap_guess(c, p, n_bases, grid);
    mtb_kmer *k; uint16_t *dg; mtb_slot16 *sg;
    MTB_RESERVE_STEP(ensure(c, "kmersA", cap, &k)); MTB_RESERVE_STEP(ensure(c, "kmersB", cap, &k));
    if (p->kmer_format == 2) { MTB_RESERVE_STEP(ensure(c, "digA", cap + 8, &dg)); MTB_RESERVE_STEP(ensure(c, "digB", cap + 8, &dg)); }
    /* slot segments: metamers of the longest read guessed from the mean length (six frames of L/3 - 7 windows per mate; syncmer
     * selection keeps a little over half), one direct slot each + the tail */
    const int mates = p->seq_mode == 2 ? 2 : 1;
    const double L = (double)n_bases / (double)n_reads / mates;
    const double per_read = std::max(0.0, L / 3.0 - 7.0) * 6.0 * mates * (p->syncmer ? 0.56 : 1.0) * 1.10;
    uint32_t direct, stride;
    slot_geometry(c, (uint32_t)std::min<double>(per_read, (double)MTB_SLOT_MAX_Q), &direct, &stride);
    MTB_RESERVE_STEP(ensure(c, "segm", n_reads * (uint64_t)stride, &sg));
#undef MTB_RESERVE_STEP
    return MTB_OK;
}

mtb_status mtb_last_batch_stats(mtb_ctx *c, mtb_batch_stats *out) {
    if (!c || !out) return fail(MTB_ERR_ARG, "NULL argument");
    *out = c->stats;
    return MTB_OK;
}

/* ------------------------------------------------------------------ */
/* synthetic index                                                     */
/* ------------------------------------------------------------------ */
mtb_status mtb_synth_index(mtb_ctx *c, uint64_t seed, uint64_t n_filler, int32_t filler_tax_lo, int32_t filler_tax_hi,
                           const uint64_t *real_values, const int32_t *real_taxids, uint64_t n_real, uint64_t *d_values,
                           uint32_t *d_info, uint64_t *n_out) {
    if (!c || !d_values || !d_info || !n_out) return fail(MTB_ERR_ARG, "NULL argument");
    HIPCHK(hipSetDevice(c->device));
    if (n_filler > MTB_AA_SPACE) return fail(MTB_ERR_ARG, "n_filler exceeds the amino-acid 8-mer space");
    if (n_filler && filler_tax_hi < filler_tax_lo) return fail(MTB_ERR_ARG, "empty filler taxon range");
    FillerParams P;
    memset(&P, 0, sizeof(P));
    P.seed = seed; P.n_filler = n_filler;
    uint64_t pairs = (n_filler + 1) / 2;
    P.stride = pairs ? MTB_AA_SPACE / pairs : 2;
    if (P.stride < 2) P.stride = 2;
    P.tax_lo = filler_tax_lo; P.tax_span = (uint32_t)(filler_tax_hi - filler_tax_lo + 1);
    for (int i = 0; i < 64; i++) {           /* valid codon ids per amino acid, ascending */
        uint32_t aa = c->h_tabs.codon[i] & 31u, cid = c->h_tabs.codon[i] >> 5;
        bool have = false;
        for (int k = 0; k < P.ncid[aa]; k++) have |= (P.cids[aa][k] == cid);
        if (!have) P.cids[aa][P.ncid[aa]++] = (uint8_t)cid;
    }
    for (int aa = 0; aa < 21; aa++) std::sort(P.cids[aa], P.cids[aa] + P.ncid[aa]);
    uint64_t *d_rv = nullptr, *d_pos = nullptr; int32_t *d_rt = nullptr;
    if (n_real) {
        for (uint64_t i = 1; i < n_real; i++) if (real_values[i] < real_values[i - 1]) return fail(MTB_ERR_ARG, "real_values must be sorted");
        STCHK(ensure(c, "synth_rv", n_real, &d_rv)); STCHK(ensure(c, "synth_rt", n_real, &d_rt)); STCHK(ensure(c, "synth_pos", n_real, &d_pos));
        STCHK(h2d(c, d_rv, real_values, n_real * 8)); STCHK(h2d(c, d_rt, real_taxids, n_real * 4));
        hipLaunchKernelGGL(k_synth_real_pos, dim3((uint32_t)((n_real + 255) / 256)), dim3(256), 0, c->stream, P, (const uint64_t *)d_rv, n_real, d_pos);
    }
    if (n_filler) hipLaunchKernelGGL(k_synth_fill, dim3((uint32_t)std::min<uint64_t>((n_filler + 255) / 256, 1ull << 20)), dim3(256), 0, c->stream, P, (const uint64_t *)d_rv, n_real, d_values, d_info);
    if (n_real) hipLaunchKernelGGL(k_synth_place_real, dim3((uint32_t)((n_real + 255) / 256)), dim3(256), 0, c->stream, (const uint64_t *)d_rv,
                                   (const int32_t *)d_rt, (const uint64_t *)d_pos, n_real, d_values, d_info);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    release(c, "synth_rv"); release(c, "synth_rt"); release(c, "synth_pos");
    *n_out = n_filler + n_real;
    return MTB_OK;
}

mtb_status mtb_extract_targets(mtb_ctx *c, const mtb_params *p, const char *genome, uint64_t len, uint64_t *values, uint64_t cap, uint64_t *count) {
    /* convenience wrapper: one sequence, long-read geometry, values only */
    if (!c || !p || !count) return fail(MTB_ERR_ARG, "NULL argument");
    mtb_params q = *p; q.seq_mode = 3;
    uint64_t offs[2] = {0, len};
    std::vector<mtb_kmer> tmp(cap);
    uint64_t n = 0;
    mtb_status st = mtb_extract(c, &q, genome, offs, nullptr, nullptr, 1, tmp.data(), cap, &n, nullptr, nullptr);
    *count = n;
    if (st != MTB_OK) return st;
    for (uint64_t i = 0; i < n; i++) values[i] = tmp[i].value;
    return MTB_OK;
}

} // extern "C"
