/* kernels_join_win.h -- the window join (round 6): k_join_dir's search and selection for the tiles whose target window fits LDS, written
 * for occupancy.
 *
 * What the measurements of rounds 5 and 6 say about the directory join (profiles/r06_notes.md): its waves spend two thirds of their time
 * waiting on DEPENDENT accesses (bisection step -> run end -> candidate -> species id -> tail cursor), VALU issue is at 42 %, the target
 * array streams at a quarter of the HBM rate -- the kernel runs as fast as the number of tiles a CU holds in flight.  Round 5's window form
 * kept 8-byte words in 31 KB of LDS (five workgroups per CU) behind 64-bit index arithmetic (89 registers).  Here
 *   - the window holds only the LOW 32 bits of every packed word -- all that the search and the evaluation read (29 bits tell the targets
 *     of a bucket apart, 24 of them are the DNA part): 15.5 KB per tile, staged by 4-byte direct-to-LDS loads (a lane per target);
 *   - every index is a 32-bit offset into the window; the query's value shrinks to its 29-bit comparable;
 *   - the full word is fetched from global memory (L2-warm: the window's load has just brought its sector) for SELECTED candidates only;
 * so that eight waves per SIMD fit without spilling.  The kernel handles ONLY tiles with a window (k_join_tile_win bounded it before the launch
 * and listed the others); the listed tiles -- spans beyond the capacity: sparse tiles, buckets of long candidate runs -- go to k_join_dir's
 * sector-random form in a launch of their own (tile_list).  A tile that finds a query outside its announced window (never, while the list is
 * sorted as announced) adds itself to that list and is redone there.
 *
 * Semantics are k_join_dir's, statement by statement (compareDna, KmerMatcher.cpp:363-416, 1117-1146): one bisection on the query's own
 * (eighth letter, DNA part); an equal block IS the selection (hamming sum 0); otherwise the run of the amino-acid part around the landing place,
 * minimum hamming sum, threshold min(2 x minimum, 7), selected candidates in index order -- the first to the query's ordinal slot, the others to
 * the read's tail, beyond that to the overflow list; runs longer than sa.coop_min are scanned by the wave.
 * Algorithmic HBM bytes per tile: 16 B per query + 8 B per window word (whole sectors are fetched, half of every word is kept) + 8 B per selected
 * candidate (L2) + 16 B per match slot. */
#ifndef MTB_KERNELS_JOIN_WIN_H
#define MTB_KERNELS_JOIN_WIN_H
#include "kernels_dir.h"

#define MTB_JW_CAP 4608               /* 32-bit words a window holds: 72 pieces of 64 = 18 KB -> eight workgroups (32 waves) per CU; a tile of 256 sorted queries spans
                                       * ~3200 targets at 10 M reads against 16 G targets */
#ifndef MTB_JW_WAVES
#define MTB_JW_WAVES 8                /* waves per SIMD the kernel is compiled for (64 registers); in-process A/B of the low-dword window inside k_join_dir
                                       * (89 -> 64 registers by spilling 11): 5 waves 73.0 ms, 6: 66.7, 7: 63.2, 8: 62.0 against 83.6 for the 8-byte window */
#endif

/* MODE 0: slot segments of fixed stride (short reads); 1: per-read slot ranges (long reads).
 * WINDOW false: the same search WITHOUT a window -- the sector-random form at eight waves per SIMD: every offset is relative to the query's own bucket
 * start (`base`, per lane), every access a 4-byte global load of a low dword.  For the tiles k_join_tile_win lists (tile_list: workgroup b takes
 * tile tile_list[b]) and for batches too sparse for windows (tile_list NULL: workgroup b takes tile b). */
template <int MODE, int WAVES = MTB_JW_WAVES, bool WINDOW = true>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WAVES))) void k_join_win(const mtb_kmer *__restrict__ q, uint64_t n, mtb_index_view ix, uint64_t limit, mtb_dir_view dv,
                                                   const mtb_tables *__restrict__ tabs, JoinSegArgs sa, uint32_t *__restrict__ overflow, uint32_t qt,
                                                   const mtb_tile_win *__restrict__ tile_win, unsigned long long *__restrict__ win_stat, uint32_t *__restrict__ redo_list,
                                                   const uint32_t *__restrict__ tile_list = nullptr) {
    constexpr bool LONG = MODE == 1;
    __shared__ __attribute__((aligned(16))) uint32_t s_win[WINDOW ? MTB_JW_CAP : 1];
    __shared__ uint32_t s_hr[8];                    /* hammingLookup rows as nibble words */
    uint64_t w0 = 0; uint32_t wn = 0;
    const uint32_t lane = threadIdx.x & 63u;
    if (WINDOW) {
        const mtb_tile_win tw = tile_win[blockIdx.x];   /* (a uniform address of read-only memory: scalar loads) */
        if (tw.words == 0) return;                      /* no window: the tile is on k_join_tile_win's list */
        w0 = tw.first; wn = (uint32_t)tw.words;
        /* the window's low dwords, 64 targets a piece, a wave each: straight into LDS, no wait between the pieces */
        const uint32_t n_piece = (wn + 63u) >> 6;
        for (uint32_t pc = threadIdx.x >> 6; pc < n_piece; pc += 4) {
            uint64_t idx = w0 + ((uint64_t)pc << 6) + lane;
            if (idx >= ix.n_targets) idx = ix.n_targets - 1;                /* (behind the window: never read) */
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ix.values + idx),
                                             (__attribute__((address_space(3))) void *)(s_win + (pc << 6)), 4, 0, MTB_WIN_AUX);
        }
    }
    /* the thread's query -> its bucket as a pair of offsets from `base`: the window's first word, or (no window) the bucket's own start */
    const uint32_t tile = (!WINDOW && tile_list) ? tile_list[blockIdx.x] : blockIdx.x;
    const uint64_t j = (uint64_t)tile * qt + threadIdx.x;
    uint64_t base = w0;
    /* what the search and the evaluation read of target `t` (offset from base b): its low 32 bits */
    auto rd = [&](uint64_t b, uint32_t t) -> uint32_t { return WINDOW ? s_win[t] : ((const uint32_t *)(ix.values + b))[2u * t]; };
    bool valid = j < n && threadIdx.x < qt;
    uint64_t qinfo_tagged = 0; uint32_t qc = 0, qk = 0, qdna = 0;
    uint32_t olo = 0, end = 0;
    bool outside = false;
    if (valid) {
        const mtb_kmer k = q[j];
        valid = mtb_q_seq(k.qinfo) != 0;            /* blank slots carry sequenceID 0 */
        if (valid) {
            qinfo_tagged = k.qinfo;
            qdna = (uint32_t)k.value & 0xFFFFFFu;
            qk = dv.kmer_format == 1 ? (uint32_t)(((k.value >> 24) % 21ull) << 24) : ((uint32_t)k.value & 0x1F000000u);
            qc = qk | qdna;
            const uint32_t b = mtb_dir_bucket(k.value, dv.L, dv.kmer_format);
            if (b < dv.n_buckets) {
                uint64_t lo = dv.base[b >> 16] + dv.dir[b], hi = dv.base[(b + 1) >> 16] + dv.dir[b + 1];
                if (hi > limit) hi = limit;         /* the last entry of the (whole) index is never a candidate */
                if (lo < hi) {
                    if (!WINDOW) { base = lo; olo = 0; end = (uint32_t)(hi - lo); }      /* (a bucket holds < 2^32 targets: the directory's rows are 32-bit) */
                    else if (lo < w0 || hi > w0 + wn) outside = true;
                    else { olo = (uint32_t)(lo - w0); end = (uint32_t)(hi - w0); }
                } else valid = false;               /* an empty bucket */
            } else valid = false;                   /* a metamer outside the directory's alphabet has no candidate */
        }
    }
    if (threadIdx.x < 8) s_hr[threadIdx.x] = tabs->hamrow[threadIdx.x];
    if (!WINDOW) __syncthreads();
    else {
    MTB_WAIT_VMEM();                                 /* the direct-to-LDS loads count in vmcnt; the barrier publishes them */
    if (__syncthreads_or(outside ? 1 : 0)) {         /* (never while the list is sorted as announced) the whole tile is redone by the sector-random form */
        if (threadIdx.x == 0) redo_list[atomicAdd(win_stat + 1, 1ull)] = blockIdx.x;
        return;
    }
    }
    /* ONE bisection per query on (eighth letter, DNA part): lands ON the block of targets equal to the query or inside / next to the run of
     * its amino-acid part (kernels_dir.h) */
    uint32_t p = olo;
    {
        uint32_t hi_ = valid ? end : olo;
        while (p < hi_) {
            const uint32_t mid = (p + hi_) >> 1;
            if ((rd(base, mid) & 0x1FFFFFFFu) < qc) p = mid + 1; else hi_ = mid;
        }
    }
    uint32_t s0 = p, e0 = p;
    if (valid) {
        if (p < end && (rd(base, p) & 0x1FFFFFFFu) == qc) {
            /* the block of targets equal to the query (several species may file the same metamer): hamming sum 0, threshold 0 -- the selection */
            e0 = p + 1;
            uint32_t c = 0;
            while (e0 < end && c < 8u && (rd(base, e0) & 0x1FFFFFFFu) == qc) { e0++; c++; }
            if (c == 8u && e0 < end && (rd(base, e0) & 0x1FFFFFFFu) == qc) {
                uint32_t y = end;
                while (e0 < y) { const uint32_t mid = (e0 + y) >> 1; if ((rd(base, mid) & 0x1FFFFFFFu) <= qc) e0 = mid + 1; else y = mid; }
            }
        } else {
            /* the run of the query's amino-acid part around the landing place */
            uint32_t c = 0;
            while (s0 > olo && c < 8u && (rd(base, s0 - 1) & 0x1F000000u) == qk) { s0--; c++; }
            if (c == 8u && s0 > olo && (rd(base, s0 - 1) & 0x1F000000u) == qk) {
                uint32_t x = olo, y = s0;
                while (x < y) { const uint32_t mid = (x + y) >> 1; if ((rd(base, mid) & 0x1F000000u) < qk) x = mid + 1; else y = mid; }
                s0 = x;
            }
            c = 0;
            while (e0 < end && c < 8u && (rd(base, e0) & 0x1F000000u) == qk) { e0++; c++; }
            if (c == 8u && e0 < end && (rd(base, e0) & 0x1F000000u) == qk) {
                uint32_t y = end;
                while (e0 < y) { const uint32_t mid = (e0 + y) >> 1; if ((rd(base, mid) & 0x1F000000u) <= qk) e0 = mid + 1; else y = mid; }
            }
        }
        if (s0 >= e0) valid = false;
    }
    /* runs still longer than sa.coop_min (no equal target in a long run) are scanned by the wave */
    const bool lng = valid && e0 - s0 > sa.coop_min;
    if (lng) valid = false;
    const uint64_t lt_mask = (1ull << lane) - 1ull;
    const uint32_t stripe = sa.ovf_stripes ? blockIdx.x & (sa.ovf_stripes - 1u) : 0u;
    /* a match beyond the read's tail: the overflow list (kernels_dir.h, ovf_put: its place behind the tail rides in the record) */
    auto ovf_put = [&](mtb_match mm, uint32_t beyond) {
        const unsigned long long o = atomicAdd(sa.ovf_counter + 8u * stripe, 1ull);
        if (LONG) return;                          /* counted only: the caller retries the join with a larger tail */
        mm.qinfo |= (uint64_t)(beyond < 0xFFFFu ? beyond : 0xFFFFu) << 16; mm.pad = 2;
        const unsigned long long room = sa.ovf_stripes ? sa.ovf_region : sa.ovf_cap;
        if (o < room) sa.ovf[(uint64_t)stripe * sa.ovf_region + o] = mm; else *overflow = 1;
    };
    /* where the matches of a query go: its read's segment, ordinal, tail capacity */
    struct Dest { mtb_slot16 *seg; uint32_t r, ord, direct, tcap; bool offr, first; };
    auto dest_of = [&](uint64_t qi) -> Dest {
        Dest d;
        d.r = mtb_q_seq(qi) - 1; d.ord = mtb_q_pos(qi) >> 16;
        if (LONG) { d.direct = sa.dcnt[d.r]; d.tcap = mtb_lslot_tail(d.direct, sa.tf); d.seg = sa.seg + sa.rb[d.r]; d.offr = false; }
        else { d.direct = sa.direct; d.tcap = sa.stride - sa.direct; d.seg = sa.seg + (uint64_t)d.r * sa.stride; d.offr = sa.off && sa.off[d.r]; }   /* offr: a read the slot records cannot hold */
        d.first = d.ord < d.direct && !d.offr;
        return d;
    };
    /* one selected candidate (window offset t, hamming sum h) of the query (qinfo: the reference's, tag stripped; qr: its rows) -> `at` = place in the
     * read's tail, or ~0u for the query's ordinal slot */
    auto put = [&](const Dest &d, const mtb_qrows &qr, uint64_t qinfo, bool rev, uint64_t b, uint32_t t, uint32_t h, uint32_t at) {
        const uint64_t v = ix.values[b + t];         /* the full word: info entry in the upper bits */
        const uint32_t td = (uint32_t)v & 0xFFFFFFu;
        const int32_t tid = (int32_t)((uint32_t)(v >> MTB_PACK_LOW) & ix.info_mask);
        const int32_t sp = (tid >= 0 && tid <= ix.max_taxid) ? ix.tax2species[tid] : 0;
        const uint16_t reh = mtb_hammings(&qr, td, rev);
        if (at == ~0u || at < d.tcap) {
            const mtb_slot16 sl = LONG ? mtb_lslot_pack(qinfo, tid, sp, td, reh, h) : mtb_slot_pack(qinfo, tid, sp, td, reh, h, sa.epoch);
            MTB_SLOT_STORE(sl, &d.seg[at == ~0u ? d.ord : d.direct + at]);
        } else {
            mtb_match mm; mm.qinfo = qinfo; mm.target_id = tid; mm.species_id = sp; mm.dna = td; mm.right_end_hamming = reh; mm.hamming = (uint8_t)h; mm.pad = 0;
            ovf_put(mm, at - d.tcap);
        }
    };
    /* ---- the lane's own candidates: minimum, threshold, emission in index order ---- */
    if (valid) {
        mtb_qrows qr; mtb_prepare_query_rows(s_hr, (uint64_t)qdna, &qr);
        uint32_t mn = 255u;
        for (uint32_t t = s0; t < e0; t++) { const uint32_t h = mtb_ham_sum(&qr, rd(base, t) & 0xFFFFFFu); mn = h < mn ? h : mn; }
        const uint32_t thr = mtb_ham_threshold(mn);
        const Dest d = dest_of(qinfo_tagged);
        const uint64_t qinfo = qinfo_tagged & ~0xFFFF0000ull;      /* the record carries the reference's qinfo */
        const bool rev = mtb_hammings_reversed(mtb_q_frame(qinfo), ix.kmer_format);
        bool first = d.first;
        for (uint32_t t = s0; t < e0; t++) {
            const uint32_t h = mtb_ham_sum(&qr, rd(base, t) & 0xFFFFFFu);
            if (h > thr) continue;
            if (first) { put(d, qr, qinfo, rev, base, t, h, ~0u); first = false; continue; }
            const uint32_t at = d.offr ? (atomicAdd(&sa.cursor[d.r], d.tcap + 1u), d.tcap) : atomicAdd(&sa.cursor[d.r], 1u);
            put(d, qr, qinfo, rev, base, t, h, at);
        }
    }
    /* ---- wave-scanned runs: one pass (minimum + the few candidates that can be selected, kept in registers), then emission -- the selected
     * candidate with the lowest index takes the query's ordinal slot, the others the read's tail (ONE returning atomic per step for all of
     * them), beyond that the overflow list: the contract of the per-lane loop below ---- */
    {
        uint64_t todo = __ballot(lng);
        while (todo) {
            const int src = __ffsll((unsigned long long)todo) - 1; todo &= todo - 1;
            const uint32_t rs = (uint32_t)__shfl((int)s0, src, 64), re = (uint32_t)__shfl((int)e0, src, 64);
            const uint64_t qi_t = wave_bcast64(qinfo_tagged, src);
            const uint64_t bs = WINDOW ? w0 : wave_bcast64(base, src);
            mtb_qrows qr; mtb_prepare_query_rows(s_hr, (uint64_t)(uint32_t)__shfl((int)qdna, src, 64), &qr);
            /* ONE pass over the run, four 64-candidate steps in flight: the minimum (-> the threshold) and, per lane, the candidates of its stripe
             * with a sum <= 7 (no other can be selected) as (offset in the run << 4 | sum): the last four are kept, n_c counts them all */
            uint32_t mn = 255u, n_c = 0, cb[4] = {0, 0, 0, 0};
            for (uint32_t t0 = rs + lane; t0 < re; t0 += 256) {
                uint32_t v[4];
#pragma unroll
                for (int u = 0; u < 4; u++) v[u] = t0 + 64 * u < re ? rd(bs, t0 + 64 * u) : 0u;
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    if (t0 + 64 * u < re) {
                        const uint32_t h = mtb_ham_sum(&qr, v[u] & 0xFFFFFFu);
                        mn = h < mn ? h : mn;
                        if (h <= 7u) { cb[3] = cb[2]; cb[2] = cb[1]; cb[1] = cb[0]; cb[0] = ((t0 + 64 * u - rs) << 4) | h; n_c++; }
                    }
                }
            }
            const uint32_t thr = mtb_ham_threshold(wave_min_shfl_u32(mn));
            const Dest d = dest_of(qi_t);
            const uint64_t qinfo = qi_t & ~0xFFFF0000ull;
            const bool rev = mtb_hammings_reversed(mtb_q_frame(qinfo), ix.kmer_format);
            const uint32_t inc = d.offr ? d.tcap + 1u : 1u;
            bool first = d.first;
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
                        if ((int)lane == leader) at0 = atomicAdd(&sa.cursor[d.r], (uint32_t)__popcll(m) * inc);
                        at0 = (uint32_t)__shfl((int)at0, leader, 64);
                    }
                    if (sel) put(d, qr, qinfo, rev, bs, rs + off, cb[b] & 15u, own ? ~0u : (d.offr ? d.tcap : at0 + (uint32_t)__popcll(m & lt_mask)));
                }
                continue;
            }
            for (uint32_t t0 = rs; t0 < re; t0 += 64) {      /* a lane met more than four possible candidates: second walk, 64 per step */
                const uint32_t t = t0 + lane;
                uint32_t h = 255u;
                if (t < re) h = mtb_ham_sum(&qr, rd(bs, t) & 0xFFFFFFu);
                const bool sel = h <= thr;
                const uint64_t m = __ballot(sel);
                if (!m) continue;
                const uint32_t rk = (uint32_t)__popcll(m & lt_mask), n_sel = (uint32_t)__popcll(m);
                const uint32_t skip = first ? 1u : 0u, n_tail = n_sel - skip;
                uint32_t at0 = 0;
                if (n_tail) {
                    const int leader = __ffsll((unsigned long long)m) - 1;
                    if ((int)lane == leader) at0 = atomicAdd(&sa.cursor[d.r], n_tail * inc);
                    at0 = (uint32_t)__shfl((int)at0, leader, 64);
                }
                if (sel) put(d, qr, qinfo, rev, bs, t, h, (first && rk == 0) ? ~0u : (d.offr ? d.tcap : at0 + rk - skip));
                first = false;
            }
        }
    }
    MTB_END_RELEASE();
}

#endif
