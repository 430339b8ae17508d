/* mtb_options.h -- the library's experiment / diagnosis switches in ONE place.
 *
 * Every MTB_* environment variable the library knows is read ONCE, when a context is created (mtb_ctx_create), into this struct; nothing
 * on the per-batch path calls getenv.  mtb_ctx_set_option(ctx, "MTB_...", "value") changes one switch of a live context afterwards
 * (value NULL or "" = back to the default): that is how bench.py's A/B legs and the tests compare variants inside one process.  An index
 * reads the switches of the context it is opened on, at that moment (directory depth, packed state, chunk size).
 *
 * None of them changes a result: they choose among exact variants, size buffers, or print.                                            */
#ifndef MTB_OPTIONS_H
#define MTB_OPTIONS_H
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <cstddef>

struct MtbOptions {
    /* join (kernels_dir.h) */
    int join_variant = 0;          /* MTB_JOIN_VARIANT: 0 auto (the context's tuner), else (Q << 4 | W) of q<Q>w<W>: 0x16, 0x25, 0x15, 0x26, 0x100 = window (8 waves per SIMD), 0x200 | W = the window form at W waves (A/B) */
    int join_win = -1;             /* MTB_JOIN_WIN: -1 by density, 0 off, 1 on */
    int join_win_qt = 0;           /* MTB_JOIN_WIN_QT: queries per window tile (0 = from the batch's density) */
    int join_coop_min = 0;         /* MTB_JOIN_COOP_MIN: candidate runs longer than this are scanned by the wave (0 = compiled default) */
    int join_verbose = 0;          /* MTB_JOIN_VERBOSE */
    /* sort */
    int sort_lsd = 0;              /* MTB_SORT_LSD: three LSD passes (round 2) */
    int sort_no_xcd = 0;           /* MTB_SORT_NO_XCD */
    int sort_pairs = 0;            /* MTB_SORT_PAIRS: letter pairs the fused path sorts on (1..3; 0 = 3) */
    /* scoring */
    int no_score_many = 0;         /* MTB_NO_SCORE_MANY: deferred reads through exact segments (round 4's path) */
    int no_many_sort = 0;          /* MTB_NO_MANY_SORT */
    int many_cap = 0;              /* MTB_MANY_CAP: staging of k_score_many, 192 or 320 records (0 = 192 up to 192 slots per read, else 320) */
    int many_verbose = 0;          /* MTB_MANY_VERBOSE */
    int no_fast_scorer = 0;        /* MTB_NO_FAST_SCORER */
    int no_fast_pairs = 0;         /* MTB_NO_FAST_PAIRS */
    int no_long_scorer = 0;        /* MTB_NO_LONG_SCORER */
    int no_long_slots = 0;         /* MTB_NO_LONG_SLOTS */
    int lslot_verbose = 0;         /* MTB_LSLOT_VERBOSE */
    int tail_min = 0;              /* MTB_TAIL_MIN: tail slots of a short read's segment (0 = 16) */
    int scratch_alias = 0;         /* MTB_SCRATCH_ALIAS: the scorer's big per-batch temporaries inside the buffers the join left dead: 0 = segments of >= 64 MiB, 1 = any size (tests), -1 = off (A/B) */
    /* index state */
    int no_dir = 0;                /* MTB_NO_DIR */
    int dir_depth = 0;             /* MTB_DIR_DEPTH: 1..7 (tests force depth 7 -- the packed state -- on toy indices; 0 = from the size) */
    int no_pack = 0;               /* MTB_NO_PACK */
    int open_packed = -1;          /* MTB_OPEN_PACKED: -1 by size, 0 / 1 */
    long long open_chunk = 0;      /* MTB_OPEN_CHUNK: 16-bit words of diffIdx per decode chunk (0 = default) */
    int part_exact = 0;            /* MTB_PART_EXACT */
    /* slot buffer placement / clearing */
    int segm_contig = 0;           /* MTB_SEGM_CONTIG */
    int segm_pad = 0;              /* MTB_SEGM_PAD */
    char segm_clear[16] = {0};     /* MTB_SEGM_CLEAR: kernel | sync | always */
    int no_placement_probe = 0;    /* MTB_NO_PLACEMENT_PROBE */
    int placement_probe = 0;       /* MTB_PLACEMENT_PROBE */
    int placement_verbose = 0;     /* MTB_PLACEMENT_VERBOSE */
    /* streams / host */
    int chunks_per_stream = 0;     /* MTB_CHUNKS_PER_STREAM (0 = 1) */
    int lane_stagger_ms = 0;       /* MTB_LANE_STAGGER_MS */
    int host_timing = 0;           /* MTB_HOST_TIMING */
};

namespace mtbopt {
enum Kind { FLAG, INT, LL, STR, VARIANT };
struct Entry { const char *name; Kind kind; size_t off; long long dflt; };
#define MTB_OPT(N, K, F, D) {N, K, offsetof(MtbOptions, F), D}
static const Entry kTable[] = {
    MTB_OPT("MTB_JOIN_VARIANT", VARIANT, join_variant, 0), MTB_OPT("MTB_JOIN_WIN", INT, join_win, -1), MTB_OPT("MTB_JOIN_WIN_QT", INT, join_win_qt, 0),
MTB_OPT("MTB_JOIN_COOP_MIN", INT, join_coop_min, 0), MTB_OPT("MTB_JOIN_VERBOSE", FLAG, join_verbose, 0),
    MTB_OPT("MTB_SORT_LSD", FLAG, sort_lsd, 0), MTB_OPT("MTB_SORT_NO_XCD", FLAG, sort_no_xcd, 0), MTB_OPT("MTB_SORT_PAIRS", INT, sort_pairs, 0),
    MTB_OPT("MTB_NO_SCORE_MANY", FLAG, no_score_many, 0), MTB_OPT("MTB_NO_MANY_SORT", FLAG, no_many_sort, 0), MTB_OPT("MTB_MANY_CAP", INT, many_cap, 0), MTB_OPT("MTB_MANY_VERBOSE", FLAG, many_verbose, 0),
    MTB_OPT("MTB_NO_FAST_SCORER", FLAG, no_fast_scorer, 0), MTB_OPT("MTB_NO_FAST_PAIRS", FLAG, no_fast_pairs, 0), MTB_OPT("MTB_NO_LONG_SCORER", FLAG, no_long_scorer, 0),
    MTB_OPT("MTB_NO_LONG_SLOTS", FLAG, no_long_slots, 0), MTB_OPT("MTB_LSLOT_VERBOSE", FLAG, lslot_verbose, 0), MTB_OPT("MTB_TAIL_MIN", INT, tail_min, 0), MTB_OPT("MTB_SCRATCH_ALIAS", INT, scratch_alias, 0),
    MTB_OPT("MTB_NO_DIR", FLAG, no_dir, 0), MTB_OPT("MTB_DIR_DEPTH", INT, dir_depth, 0), MTB_OPT("MTB_NO_PACK", FLAG, no_pack, 0), MTB_OPT("MTB_OPEN_PACKED", INT, open_packed, -1),
    MTB_OPT("MTB_OPEN_CHUNK", LL, open_chunk, 0), MTB_OPT("MTB_PART_EXACT", FLAG, part_exact, 0),
    MTB_OPT("MTB_SEGM_CONTIG", FLAG, segm_contig, 0), MTB_OPT("MTB_SEGM_PAD", FLAG, segm_pad, 0), MTB_OPT("MTB_SEGM_CLEAR", STR, segm_clear, 0),
    MTB_OPT("MTB_NO_PLACEMENT_PROBE", FLAG, no_placement_probe, 0), MTB_OPT("MTB_PLACEMENT_PROBE", FLAG, placement_probe, 0), MTB_OPT("MTB_PLACEMENT_VERBOSE", FLAG, placement_verbose, 0),
    MTB_OPT("MTB_CHUNKS_PER_STREAM", INT, chunks_per_stream, 0), MTB_OPT("MTB_LANE_STAGGER_MS", INT, lane_stagger_ms, 0), MTB_OPT("MTB_HOST_TIMING", FLAG, host_timing, 0),
};
#undef MTB_OPT

/* "q1w6" -> 0x16, "window" -> 0x100, "auto" / "" -> 0; anything else: -1 */
static inline int parse_variant(const char *v) {
    if (!v || !*v || !strcmp(v, "auto")) return 0;
    if (!strcmp(v, "window")) return 0x100;
    if (!strncmp(v, "windoww", 7) && v[7] >= '5' && v[7] <= '7' && !v[8]) return 0x200 | (v[7] - '0');      /* the window form compiled for 5..7 waves per SIMD (A/B; "window" is 8) */
    if (v[0] == 'q' && v[1] >= '1' && v[1] <= '2' && v[2] == 'w' && v[3] >= '5' && v[3] <= '6' && !v[4]) return ((v[1] - '0') << 4) | (v[3] - '0');
    return -1;
}
/* value NULL = the variable is unset.  A FLAG is on when the variable exists at all (getenv() != NULL was the test everywhere), off when unset. */
static inline bool set(MtbOptions *o, const char *name, const char *value) {
    for (const Entry &e : kTable) {
        if (strcmp(e.name, name)) continue;
        char *p = (char *)o + e.off;
        switch (e.kind) {
        case FLAG: *(int *)p = value ? 1 : 0; break;
        case INT: *(int *)p = value && *value ? atoi(value) : (int)e.dflt; break;
        case LL: *(long long *)p = value && *value ? strtoll(value, nullptr, 10) : e.dflt; break;
        case STR: { memset(p, 0, 16); if (value) strncpy(p, value, 15); break; }
        case VARIANT: { const int v = parse_variant(value); if (v < 0) return false; *(int *)p = v; break; }
        }
        return true;
    }
    return false;
}
static inline void from_environment(MtbOptions *o) {
    for (const Entry &e : kTable) if (const char *v = getenv(e.name)) (void)set(o, e.name, v);
}
}  // namespace mtbopt
#endif
