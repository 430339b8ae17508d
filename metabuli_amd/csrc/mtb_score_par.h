/*
 * mtb_score_par.h -- data-parallel formulation of the per-read scorer
 * (Taxonomer::chooseBestTaxon and callees, src/commons/Taxonomer.cpp:130-699),
 * written as PER-ELEMENT phases over shared arrays.  On the GPU one wavefront
 * owns one read: every phase is a lane-strided loop over the read's matches
 * followed by a barrier (kernels_score.h).  tests/emu runs the same phase
 * functions in plain loops, so the arithmetic is checked against the oracle
 * without a GPU.
 *
 * Phase list (n = matches of the read, sorted by compareMatches):
 *   flags    head flags of position groups / (species,frame) blocks / species
 *   ids      prefix sums of the flags -> gid, bid, sid; starts of every group
 *   links    per match: rank of its position group inside its block, codon
 *            shift to the previous group, bit mask of consecutive predecessors,
 *            "connected to next" flag, initial path      (getMatchPaths setup)
 *   rounds   r = 1..max rank: matches of rank r extend the best consecutive
 *            path of rank r-1                            (getMatchPaths DP)
 *   emit     a match emits its path if it was not extended and is deep enough
 *   combine  one lane per species: stable order + greedy combination
 *   select   best species / ties -> LCA                  (getBestSpeciesMatches)
 *   filter   redundancy filter by position bucket, match-parallel
 *   finish   taxCnt map, sub-species descent
 */
#ifndef MTB_SCORE_PAR_H
#define MTB_SCORE_PAR_H
#include "mtb_core.h"

#define MTB_F_GHEAD 1u      /* first match of its position group            */
#define MTB_F_BHEAD 2u      /* first match of its (species, frame) block    */
#define MTB_F_SHEAD 4u      /* first match of its species                   */
#define MTB_F_CONN  8u      /* connectedToNext                              */
#define MTB_F_MULTI 16u     /* its block has more than one position group   */
#define MTB_F_EUK   32u     /* species under Eukaryota                      */
#define MTB_F_EMIT  64u     /* path pushed to filteredMatchPaths            */
#define MTB_SHIFT_SLOW 0x80u /* previous group larger than 8: generic loop  */

template <typename IDX>
struct mtb_sws {
    mtb_match *m;            /* [n] sorted matches                                      */
    mtb_path  *path;         /* [n] (aliases the sort keys before the links phase)      */
    uint8_t   *flag;         /* [n]                                                     */
    uint8_t   *shift;        /* [n] codon shift to the previous group (0 = no link)     */
    uint8_t   *cmask;        /* [n] consecutive predecessors among the first 8          */
    IDX *gid, *bid, *sid;    /* [n] ids; after `links`: gid -> elist, bid -> pred_lo     */
    IDX *rk;                 /* [n+1] group rank in block; after rounds: emitted prefix  */
    IDX *grp_start;          /* [n+1]; after rounds (with blk_start): float sps[]        */
    IDX *blk_start;          /* [n+1]                                                   */
    IDX *sp_start;           /* [n+1]                                                   */
    IDX *acc;                /* [n] accepted paths (combine)                            */
    int32_t n;
};

/* bytes per element / fixed bytes of a workspace (for slab sizing) */
template <typename IDX> MTB_HD uint64_t mtb_sws_bytes(uint64_t n) {
    uint64_t b = n * (sizeof(mtb_match) + sizeof(mtb_path) + 3) + (n + 1) * 8 * sizeof(IDX) + 64;
    return (b + 15) & ~15ull;
}
template <typename IDX> MTB_HD void mtb_sws_carve(mtb_sws<IDX> *w, uint8_t *base, uint64_t cap) {
    uint64_t N = cap + 1;
    w->m = (mtb_match *)base; base += cap * sizeof(mtb_match);
    w->path = (mtb_path *)base; base += cap * sizeof(mtb_path);
    w->gid = (IDX *)base; base += N * sizeof(IDX);
    w->bid = (IDX *)base; base += N * sizeof(IDX);
    w->sid = (IDX *)base; base += N * sizeof(IDX);
    w->rk = (IDX *)base; base += N * sizeof(IDX);
    w->grp_start = (IDX *)base; base += N * sizeof(IDX);
    w->blk_start = (IDX *)base; base += N * sizeof(IDX);     /* must follow grp_start: sps[] spans both */
    w->sp_start = (IDX *)base; base += N * sizeof(IDX);
    w->acc = (IDX *)base; base += N * sizeof(IDX);
    w->flag = base; base += cap;
    w->shift = base; base += cap;
    w->cmask = base;
}

/* ---- sort keys (compareMatches inside one read: species, frame, pos, hamming, dna) ---- */
MTB_HD uint64_t mtb_key1(const mtb_match &x) {
    return ((uint64_t)(uint32_t)x.species_id << 32) | ((uint64_t)mtb_q_frame(x.qinfo) << 29) | (uint64_t)(mtb_q_pos(x.qinfo) & 0x1FFFFFFFu);
}
MTB_HD uint32_t mtb_key2(const mtb_match &x) { return ((uint32_t)x.hamming << 24) | (x.dna & 0xFFFFFFu); }

/* ---- flags ---- */
template <typename IDX>
MTB_HD void mtb_ph_flags(const mtb_sws<IDX> &w, int32_t i) {
    const mtb_match *m = w.m;
    uint32_t f = 0;
    if (i == 0) f = MTB_F_GHEAD | MTB_F_BHEAD | MTB_F_SHEAD;
    else {
        bool sp = m[i].species_id != m[i - 1].species_id;
        bool fr = sp || mtb_q_frame(m[i].qinfo) != mtb_q_frame(m[i - 1].qinfo);
        bool ps = fr || mtb_q_pos(m[i].qinfo) != mtb_q_pos(m[i - 1].qinfo);
        f = (sp ? MTB_F_SHEAD : 0u) | (fr ? MTB_F_BHEAD : 0u) | (ps ? MTB_F_GHEAD : 0u);
    }
    w.flag[i] = (uint8_t)f;
}
/* ---- starts (after gid/bid/sid hold the inclusive prefix count - 1) ---- */
template <typename IDX>
MTB_HD void mtb_ph_starts(const mtb_sws<IDX> &w, int32_t i, const mtb_tax_view *tx) {
    uint32_t f = w.flag[i];
    if (f & MTB_F_GHEAD) w.grp_start[w.gid[i]] = (IDX)i;
    if (f & MTB_F_BHEAD) w.blk_start[w.bid[i]] = (IDX)i;
    if (f & MTB_F_SHEAD) {          /* one global load per species: IsAncestor(eukaryota, species), Taxonomer.cpp:497-500 */
        int32_t s = w.sid[i];
        w.sp_start[s] = (IDX)i;
        int32_t spc = w.m[i].species_id;
        w.acc[s] = (IDX)((spc >= 0 && spc <= tx->max_taxid && tx->under_euk[spc]) ? 1 : 0);
    }
}
/* ---- links ---- */
template <typename IDX>
MTB_HD void mtb_ph_links(const mtb_sws<IDX> &w, int32_t i, const mtb_tax_view *tx, const mtb_score_params *sp,
                         int32_t n_groups, int32_t n_blocks) {
    const mtb_match *m = w.m;
    const int32_t n = w.n;
    int32_t g = w.gid[i], b = w.bid[i];
    int32_t bs = w.blk_start[b];
    int32_t be = (b + 1 < n_blocks) ? (int32_t)w.blk_start[b + 1] : n;
    int32_t g0 = w.gid[bs];
    int32_t rank = g - g0;
    int32_t gs = w.grp_start[g];
    int32_t ge = (g + 1 < n_groups) ? (int32_t)w.grp_start[g + 1] : n;
    bool fwd = mtb_q_frame(m[i].qinfo) < 3;
    uint32_t pos = mtb_q_pos(m[i].qinfo), dna = m[i].dna;
    uint32_t f = w.flag[i] & (MTB_F_GHEAD | MTB_F_BHEAD | MTB_F_SHEAD);
    /* block with a single position group: the reference's while loop never runs */
    if (rank > 0 || ge < be) f |= MTB_F_MULTI;
    if (w.acc[w.sid[i]]) f |= MTB_F_EUK;       /* acc[] holds the per-species flag until the combine phase */
    (void)tx;
    /* previous group */
    uint32_t sh = 0, cm = 0; int32_t pl = 0;
    if (rank > 0) {
        pl = w.grp_start[g - 1];
        int32_t s = (int32_t)(pos - mtb_q_pos(m[pl].qinfo)) / 3;
        if (s > 0 && s <= sp->max_codon_shift) {
            sh = (uint32_t)s;
            if (gs - pl > 8) sh |= MTB_SHIFT_SLOW;
            else for (int32_t cu = pl; cu < gs; cu++)
                if (mtb_consecutive(m[cu].dna, dna, s, fwd, sp->kmer_format)) cm |= 1u << (cu - pl);
        }
    }
    /* next group: connectedToNext (Taxonomer.cpp:536-541) */
    if (ge < be) {
        int32_t ne = (g + 2 < n_groups) ? (int32_t)w.grp_start[g + 2] : n;
        if (ne > be) ne = be;
        int32_t s = (int32_t)(mtb_q_pos(m[ge].qinfo) - pos) / 3;
        if (s > 0 && s <= sp->max_codon_shift)
            for (int32_t nx = ge; nx < ne; nx++)
                if (mtb_consecutive(dna, m[nx].dna, s, fwd, sp->kmer_format)) { f |= MTB_F_CONN; break; }
    }
    /* MatchPath(const Match*) (Taxonomer.h:39-46) */
    mtb_path p;
    p.start = (int32_t)pos; p.end = (int32_t)pos + 23; p.score = mtb_part_score(m[i].right_end_hamming, 8, false);
    p.ham = m[i].hamming; p.depth = 1; p.start_idx = i;
    /* NOTE: path aliases nothing the other lanes still read in this phase (keys are dead) */
    w.path[i] = p;
    w.flag[i] = (uint8_t)f; w.shift[i] = (uint8_t)sh; w.cmask[i] = (uint8_t)cm;
    w.rk[i] = (IDX)rank;
    w.bid[i] = (IDX)pl;            /* own slot only: bid[i] is read by nobody else */
}
/* ---- links when every position group holds exactly one match (n_groups == n: the usual read) ----
 * group g == match g: its block neighbours are the slots before and after it, so the group / block start tables
 * are not consulted (about a third of the LDS reads of mtb_ph_links).  Writes exactly what mtb_ph_links writes. */
template <typename IDX>
MTB_HD void mtb_ph_links_unit(const mtb_sws<IDX> &w, int32_t i, const mtb_score_params *sp) {
    const mtb_match *m = w.m;
    const int32_t n = w.n;
    const uint32_t fl = w.flag[i];
    const bool has_prev = !(fl & MTB_F_BHEAD);                               /* same block as slot i-1 */
    const bool has_next = i + 1 < n && !(w.flag[i + 1] & MTB_F_BHEAD);
    const uint64_t qi = m[i].qinfo;
    const bool fwd = mtb_q_frame(qi) < 3;
    const uint32_t pos = mtb_q_pos(qi), dna = m[i].dna;
    uint32_t f = fl & (MTB_F_GHEAD | MTB_F_BHEAD | MTB_F_SHEAD);
    if (has_prev || has_next) f |= MTB_F_MULTI;
    if (w.acc[w.sid[i]]) f |= MTB_F_EUK;
    uint32_t sh = 0, cm = 0; int32_t pl = 0;
    if (has_prev) {
        pl = i - 1;
        int32_t s = (int32_t)(pos - mtb_q_pos(m[pl].qinfo)) / 3;
        if (s > 0 && s <= sp->max_codon_shift) { sh = (uint32_t)s; if (mtb_consecutive(m[pl].dna, dna, s, fwd, sp->kmer_format)) cm = 1u; }
    }
    if (has_next) {
        int32_t s = (int32_t)(mtb_q_pos(m[i + 1].qinfo) - pos) / 3;
        if (s > 0 && s <= sp->max_codon_shift && mtb_consecutive(dna, m[i + 1].dna, s, fwd, sp->kmer_format)) f |= MTB_F_CONN;
    }
    mtb_path p;
    p.start = (int32_t)pos; p.end = (int32_t)pos + 23; p.score = mtb_part_score(m[i].right_end_hamming, 8, false);
    p.ham = m[i].hamming; p.depth = 1; p.start_idx = i;
    w.path[i] = p;
    w.flag[i] = (uint8_t)f; w.shift[i] = (uint8_t)sh; w.cmask[i] = (uint8_t)cm;
    w.rk[i] = (IDX)(i - (int32_t)w.blk_start[w.bid[i]]);                    /* rank of the group inside its block */
    w.bid[i] = (IDX)pl;
}
/* ---- one DP round (Taxonomer.cpp:528-560) ---- */
template <typename IDX>
MTB_HD void mtb_ph_round(const mtb_sws<IDX> &w, int32_t i, int32_t r, const mtb_score_params *sp) {
    if ((int32_t)w.rk[i] != r) return;
    uint32_t sh = w.shift[i];
    if (sh == 0) return;
    const mtb_match *m = w.m;
    int32_t pl = w.bid[i];
    int32_t shift = (int32_t)(sh & 0x7Fu);
    int32_t best = -1; float best_score = 0.0f;
    if (sh & MTB_SHIFT_SLOW) {
        int32_t gs = w.grp_start[w.gid[i]];
        bool fwd = mtb_q_frame(m[i].qinfo) < 3;
        for (int32_t cu = pl; cu < gs; cu++)
            if (mtb_consecutive(m[cu].dna, m[i].dna, shift, fwd, sp->kmer_format)) {
                float s = w.path[cu].score;
                if (s > best_score) { best = cu; best_score = s; }
            }
    } else {
        uint32_t cm = w.cmask[i];
        for (int32_t k = 0; cm; k++, cm >>= 1)
            if (cm & 1u) { float s = w.path[pl + k].score; if (s > best_score) { best = pl + k; best_score = s; } }
    }
    if (best < 0) return;
    mtb_path b = w.path[best];
    mtb_path p = w.path[i];
    uint32_t reh = m[i].right_end_hamming;
    p.start = b.start;
    p.score = b.score + mtb_part_score(reh, shift, false);
    p.ham = b.ham + mtb_part_ham(reh, shift, false);
    p.depth = b.depth + shift;
    p.start_idx = b.start_idx;
    w.path[i] = p;
}
/* ---- emit flag (Taxonomer.cpp:561-572) ---- */
template <typename IDX>
MTB_HD bool mtb_ph_emit(const mtb_sws<IDX> &w, int32_t i, const mtb_score_params *sp) {
    uint32_t f = w.flag[i];
    int32_t md = (f & MTB_F_EUK) ? sp->min_cons_cnt_euk : sp->min_cons_cnt;
    return (f & MTB_F_MULTI) && !(f & MTB_F_CONN) && w.path[i].depth >= md;
}
/* ---- combine (Taxonomer.cpp:410-468) on the emitted list el[lo..hi) ---- */
template <typename IDX>
MTB_HD float mtb_ph_combine(const mtb_sws<IDX> &w, IDX *el, int32_t lo, int32_t hi, int32_t read_len) {
    const mtb_match *m = w.m; mtb_path *path = w.path; IDX *acc = w.acc;
    /* stable insertion sort: score desc, hamming asc, start desc */
    for (int32_t a = lo + 1; a < hi; a++) {
        IDX x = el[a];
        mtb_path px = path[x];
        int32_t j = a;
        while (j > lo && mtb_path_before(px, path[el[j - 1]])) { el[j] = el[j - 1]; j--; }
        el[j] = x;
    }
    float score = 0.0f;
    int32_t na = 0;
    for (int32_t k = lo; k < hi; k++) {
        int32_t pi = el[k];
        mtb_path p = path[pi];
        bool drop = false;
        for (int32_t a = 0; a < na && !drop; a++) {
            mtb_path c = path[acc[lo + a]];
            if (!((p.end < c.start) || (c.end < p.start))) {
                int32_t ov = (p.end < c.end ? p.end : c.end) - (p.start > c.start ? p.start : c.start) + 1;
                if (ov == p.end - p.start + 1) { drop = true; break; }
                if (ov < 24) {
                    if (p.start < c.start) {
                        p.end = c.start - 1;
                        int32_t h = p.ham - mtb_part_ham(m[pi].right_end_hamming, ov / 3, false);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[pi].right_end_hamming, ov / 3, false) - (float)(ov % 3);
                    } else {
                        p.start = c.end + 1;
                        int32_t h = p.ham - mtb_part_ham(m[p.start_idx].right_end_hamming, ov / 3, true);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[p.start_idx].right_end_hamming, ov / 3, true) - (float)(ov % 3);
                    }
                } else drop = true;
            }
        }
        if (!drop) { path[pi] = p; acc[lo + na++] = (IDX)pi; score += p.score; }
    }
    return score / (float)read_len;
}
/* ---- combine in parallel pieces ------------------------------------------------------------------------
 * The insertion sort + greedy pass above is serial per species.  The same result comes from
 *   comb_lo     emitted entry e -> start `lo` of its species' range in the emitted list (binary search in sp_start)
 *   comb_rank   position of e in the stable (score desc, hamming asc, start desc) order of its species:
 *               lo + #{f in [lo,hi): f strictly before e, or equivalent with f < e}  (== stable insertion sort)
 *   comb_predrop  sorted entry k > lo is dropped by the greedy pass for certain when it overlaps the species'
 *               first path -- accepted first and untrimmed -- by its whole length or by >= 24 (that is the first
 *               test the greedy loop makes for it, Taxonomer.cpp:436-448)
 *   comb_greedy the greedy pass over the sorted list, skipping the pre-dropped entries (serial per species;
 *               for a typical read nothing but the best path survives the pre-drop)                          */
template <typename IDX>
MTB_HD int32_t mtb_ph_comb_species(const mtb_sws<IDX> &w, int32_t n_species, int32_t i) {     /* species index of match i */
    int32_t lo = 0, hi = n_species;            /* last s with sp_start[s] <= i */
    while (hi - lo > 1) { int32_t mid = (lo + hi) >> 1; if ((int32_t)w.sp_start[mid] <= i) lo = mid; else hi = mid; }
    return lo;
}
template <typename IDX>
MTB_HD int32_t mtb_ph_comb_rank(const mtb_sws<IDX> &w, const IDX *el, int32_t e, int32_t lo, int32_t hi) {
    const mtb_path *path = w.path;
    const mtb_path px = path[el[e]];
    int32_t r = lo;
    for (int32_t f = lo; f < hi; f++) {
        if (f == e) continue;
        const mtb_path pf = path[el[f]];
        if (mtb_path_before(pf, px) || (f < e && !mtb_path_before(px, pf))) r++;
    }
    return r;
}
/* comb_rank on packed keys: key = ~ordered(score) << 32 | hamming, so that "key smaller" == "score higher, then
 * hamming lower"; equal keys fall back to the start position (descending) and the emission index.  One
 * contiguous 8-byte load per comparison instead of two dependent loads (index, then the 24-byte path): the
 * slab scorer of long reads spent 30 % of its time here. */
MTB_HD uint64_t mtb_path_key(const mtb_path &p) {
    union { float f; uint32_t u; } c; c.f = p.score;
    uint32_t ord = (c.u & 0x80000000u) ? ~c.u : (c.u | 0x80000000u);      /* monotone float -> uint */
    if (p.score == 0.0f) ord = 0x80000000u;                               /* -0.0 == +0.0 */
    return ((uint64_t)(~ord) << 32) | (uint32_t)p.ham;
}
template <typename IDX>
MTB_HD int32_t mtb_ph_comb_rank_keys(const mtb_sws<IDX> &w, const IDX *el, const uint64_t *keys, int32_t e, int32_t lo, int32_t hi) {
    const uint64_t ke = keys[e];
    int32_t r = lo, start_e = 0; bool have = false;
    for (int32_t f = lo; f < hi; f++) {
        if (f == e) continue;
        const uint64_t kf = keys[f];
        bool before;
        if (kf != ke) before = kf < ke;
        else {
            if (!have) { start_e = w.path[el[e]].start; have = true; }
            const int32_t sf = w.path[el[f]].start;
            before = sf != start_e ? sf > start_e : f < e;
        }
        if (before) r++;
    }
    return r;
}
template <typename IDX>
MTB_HD bool mtb_ph_comb_predrop(const mtb_sws<IDX> &w, const IDX *sorted, int32_t k, int32_t lo) {
    if (k == lo) return false;
    const mtb_path c = w.path[sorted[lo]], p = w.path[sorted[k]];
    if ((p.end < c.start) || (c.end < p.start)) return false;
    int32_t ov = (p.end < c.end ? p.end : c.end) - (p.start > c.start ? p.start : c.start) + 1;
    return ov == p.end - p.start + 1 || ov >= 24;
}
template <typename IDX>
MTB_HD float mtb_ph_comb_greedy(const mtb_sws<IDX> &w, const IDX *sorted, const uint8_t *predrop, int32_t lo, int32_t hi, int32_t read_len) {
    const mtb_match *m = w.m; mtb_path *path = w.path; IDX *acc = w.acc;
    float score = 0.0f;
    int32_t na = 0;
    for (int32_t k = lo; k < hi; k++) {
        if (predrop[k]) continue;
        int32_t pi = sorted[k];
        mtb_path p = path[pi];
        bool drop = false;
        for (int32_t a = 0; a < na && !drop; a++) {
            mtb_path c = path[acc[lo + a]];
            if (!((p.end < c.start) || (c.end < p.start))) {
                int32_t ov = (p.end < c.end ? p.end : c.end) - (p.start > c.start ? p.start : c.start) + 1;
                if (ov == p.end - p.start + 1) { drop = true; break; }
                if (ov < 24) {
                    if (p.start < c.start) {
                        p.end = c.start - 1;
                        int32_t h = p.ham - mtb_part_ham(m[pi].right_end_hamming, ov / 3, false);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[pi].right_end_hamming, ov / 3, false) - (float)(ov % 3);
                    } else {
                        p.start = c.end + 1;
                        int32_t h = p.ham - mtb_part_ham(m[p.start_idx].right_end_hamming, ov / 3, true);
                        p.ham = h > 0 ? h : 0;
                        p.score = p.score - mtb_part_score(m[p.start_idx].right_end_hamming, ov / 3, true) - (float)(ov % 3);
                    }
                } else drop = true;
            }
        }
        if (!drop) { path[pi] = p; acc[lo + na++] = (IDX)pi; score += p.score; }
    }
    return score / (float)read_len;
}

/* ---- select over the species list (Taxonomer.cpp:354-407, 130-165) ---- */
template <typename IDX>
MTB_HD bool mtb_ph_select(const mtb_sws<IDX> &w, const float *sps, int32_t n_species, const mtb_tax_view *tx,
                          const mtb_score_params *sp, mtb_result *R, int32_t *best_s_out, int32_t *best_e_out, int32_t *species) {
    R->classification = 0; R->score = 0.0f; R->is_classified = 0; R->n_taxcnt = 0;
    float best_sp = 0.0f; int32_t best = -1; int32_t meaningful = 0;
    for (int32_t s = 0; s < n_species; s++) {
        float sc = sps[s];
        if (sc == -1.0f) continue;
        if (sc < sp->min_score) continue;
        if (sc > 0.0f) meaningful++;
        if (sc > best_sp) { best_sp = sc; best = s; }
    }
    if (meaningful == 0) return false;
    float sum = 0.0f; int32_t n_max = 0; int32_t lca = -1; int32_t only = 0, first_spc = 0;
    float cut = best_sp * sp->tie_ratio;
    for (int32_t s = 0; s < n_species; s++) {
        float sc = sps[s];
        if (sc == -1.0f || sc < sp->min_score) continue;
        if (sc >= cut) {
            int32_t spc = w.m[w.sp_start[s]].species_id;
            sum += sc; only = spc; n_max++;
            if (n_max == 1) first_spc = spc;
            else {
                if (n_max == 2) lca = mtb_tax_exists(tx, first_spc) ? mtb_tax_canon(tx, first_spc) : -1;
                if (mtb_tax_exists(tx, spc)) lca = lca < 0 ? mtb_tax_canon(tx, spc) : mtb_lca(tx, lca, spc);
            }
        }
    }
    float score = n_max > 1 ? sum / (float)n_max : sum;
    R->score = score;
    if (score == 0.0f || score < sp->min_score) return false;
    if (n_max > 1) { R->is_classified = 1; R->classification = lca < 0 ? 0 : lca; return false; }
    *best_s_out = w.sp_start[best];
    *best_e_out = (best + 1 < n_species) ? (int32_t)w.sp_start[best + 1] : w.n;
    *species = only;
    R->is_classified = 1;
    return true;
}

/* ---- redundancy filter, match-parallel (Taxonomer.cpp:205-241) ----------
 * hmin[q] (u32, init 255) = minimum hamming of bucket q; btax[q] (init -1) =
 * LCA-fold of the target ids of its minimum-hamming matches.  The fold of the
 * reference (first id raw, then LCA-merge) is order independent for ids that
 * exist in the taxonomy, so the lanes may merge in any order.               */
#if defined(__HIP_DEVICE_COMPILE__)
#define MTB_AMIN_U32(p, v) atomicMin((p), (v))
#define MTB_ACAS_I32(p, o, n) atomicCAS((p), (o), (n))
#else
static inline uint32_t mtb_host_amin(uint32_t *p, uint32_t v) { uint32_t o = *p; if (v < o) *p = v; return o; }
static inline int32_t mtb_host_acas(int32_t *p, int32_t o, int32_t n) { int32_t c = *p; if (c == o) *p = n; return c; }
#define MTB_AMIN_U32(p, v) mtb_host_amin((p), (v))
#define MTB_ACAS_I32(p, o, n) mtb_host_acas((p), (o), (n))
#endif

MTB_HD void mtb_ph_filter_min(const mtb_match *m, int32_t i, int32_t dna_shift, int32_t nb, uint32_t *hmin) {
    int32_t q = (int32_t)(mtb_q_pos(m[i].qinfo) / (uint32_t)dna_shift);
    if (q < nb) MTB_AMIN_U32(&hmin[q], (uint32_t)m[i].hamming);
}
MTB_HD void mtb_ph_filter_merge(const mtb_match *m, int32_t i, int32_t dna_shift, int32_t nb, const uint32_t *hmin,
                                int32_t *btax, const mtb_tax_view *tx) {
    int32_t q = (int32_t)(mtb_q_pos(m[i].qinfo) / (uint32_t)dna_shift);
    if (q >= nb || (uint32_t)m[i].hamming != hmin[q]) return;
    int32_t tid = m[i].target_id;
    int32_t old = MTB_ACAS_I32(&btax[q], -1, tid);         /* first id of the bucket stays raw */
    while (old != -1) {
        int32_t merged = mtb_lca(tx, old, tid);
        if (merged == old) break;
        int32_t seen = MTB_ACAS_I32(&btax[q], old, merged);
        if (seen == old) break;
        old = seen;
    }
}

/* ---- sub-species descent on pre-climbed chains (Taxonomer.cpp:252-314) ----
 * entry i of the taxCnt map: lev[i] = depth below the species (0 = the species
 * itself, -1 = not under it), anc[i*K + k] = its ancestor k+1 levels below the
 * species.  mtb_lr_climb fills them with (parallel) global loads; mtb_lr_bfs
 * then walks only these arrays.                                             */
#define MTB_LR_K 4
#define MTB_LR_MAXE 32
MTB_HD void mtb_lr_climb(const mtb_tax_view *tx, int32_t tax, int32_t species, int32_t *lev_out, int32_t *anc /* [K] */) {
    int32_t cs = mtb_tax_canon(tx, species), c = mtb_tax_canon(tx, tax);
    if (cs < 0 || c < 0) { *lev_out = -1; return; }
    int32_t dsp = tx->depth[cs];
    int32_t L = tx->depth[c] - dsp;
    if (L < 0) { *lev_out = -1; return; }
    if (L > MTB_LR_K) { *lev_out = MTB_LR_K + 1; return; }      /* too deep: caller falls back */
    int32_t a = c;
    for (int32_t k = L - 1; k >= 0; k--) { anc[k] = a; a = tx->parent[a]; }
    *lev_out = (a == cs) ? L : -1;
}
MTB_HD int32_t mtb_lr_bfs(const int32_t *lev, const int32_t *anc, const uint32_t *cnt, int32_t n, int32_t species_canon,
                          int32_t read_len, int32_t denominator, const mtb_tax_view *tx = 0, int32_t accession_level = 0) {
    uint32_t thr = (uint32_t)((read_len - 1) / denominator);
    int32_t root = species_canon;
    for (int32_t level = 0; level < MTB_LR_K; level++) {
        /* entries in root's subtree with at least one more level: their child of root is anc[level] */
        uint32_t max_cnt = thr; int32_t best = -1, n_best = 0; bool any = false;
        for (int32_t i = 0; i < n; i++) {
            if (lev[i] <= level) continue;
            if (level > 0 && anc[i * MTB_LR_K + level - 1] != root) continue;
            int32_t c = anc[i * MTB_LR_K + level];
            if (accession_level == 2 && tx && tx->acc_leaf && tx->acc_leaf[c]) continue;      /* Taxonomer.cpp:256-267 */
            any = true;
            bool first = true;
            for (int32_t j = 0; j < i && first; j++)
                if (lev[j] > level && (level == 0 || anc[j * MTB_LR_K + level - 1] == root) && anc[j * MTB_LR_K + level] == c) first = false;
            if (!first) continue;
            uint32_t clade = 0;
            for (int32_t j = i; j < n; j++)
                if (lev[j] > level && (level == 0 || anc[j * MTB_LR_K + level - 1] == root) && anc[j * MTB_LR_K + level] == c) clade += cnt[j];
            if (clade > max_cnt) { best = c; n_best = 1; max_cnt = clade; }
            else if (clade == max_cnt) { if (n_best == 0) best = c; n_best++; }
        }
        if (!any) return root;
        if (n_best == 1) root = best; else return root;
    }
    return root;
}

/* ---- chain DP by pointer doubling -----------------------------------------
 * When every match has at most one consecutive predecessor (cmask has <= 1 bit
 * and no SLOW group) the DP of getMatchPaths is a forest of chains:
 *   V[i] = V[pred[i]] + (score, hamming, depth) increment of i,   V[root] = fresh path.
 * Scores are sums of multiples of 0.5 (exact in fp32), hamming / depth are
 * integers, so the fold is associative: ceil(log2(max rank + 1)) doubling steps
 * give bit-identical paths.  jump[] lives in the storage of path[] (16 B <= 24 B
 * per match); the final paths are rebuilt from the roots' matches.          */
typedef struct { int32_t ptr; float score; int32_t ham; int32_t depth; } mtb_jump;

/* true if match i allows the chain formulation */
template <typename IDX>
MTB_HD bool mtb_chain_simple(const mtb_sws<IDX> &w, int32_t i) {
    uint32_t sh = w.shift[i], cm = w.cmask[i];
    if (sh & MTB_SHIFT_SLOW) return false;
    return (cm & (cm - 1u)) == 0;
}
template <typename IDX>
MTB_HD void mtb_ph_jump_init(const mtb_sws<IDX> &w, int32_t i, mtb_jump *jump) {
    uint32_t sh = w.shift[i], cm = w.cmask[i];
    mtb_jump j; j.ptr = -1; j.score = 0.0f; j.ham = 0; j.depth = 0;
    if (sh && cm) {
        int32_t k = 0; while (!((cm >> k) & 1u)) k++;
        int32_t shift = (int32_t)(sh & 0x7Fu);
        uint32_t reh = w.m[i].right_end_hamming;
        j.ptr = (int32_t)w.bid[i] + k;
        j.score = mtb_part_score(reh, shift, false);
        j.ham = mtb_part_ham(reh, shift, false);
        j.depth = shift;
    }
    jump[i] = j;
}
/* one doubling step, read half: returns the new value of jump[i] */
MTB_HD mtb_jump mtb_ph_jump_step(const mtb_jump *jump, int32_t i) {
    mtb_jump j = jump[i];
    if (j.ptr >= 0) {
        mtb_jump p = jump[j.ptr];
        if (p.ptr >= 0) { j.score += p.score; j.ham += p.ham; j.depth += p.depth; j.ptr = p.ptr; }
    }
    return j;
}
/* final path of match i from its resolved jump record (ptr = root or -1) */
template <typename IDX>
MTB_HD mtb_path mtb_ph_jump_final(const mtb_sws<IDX> &w, int32_t i, const mtb_jump &j) {
    const mtb_match *m = w.m;
    int32_t r = j.ptr >= 0 ? j.ptr : i;
    mtb_path p;
    p.start = (int32_t)mtb_q_pos(m[r].qinfo);
    p.end = (int32_t)mtb_q_pos(m[i].qinfo) + 23;
    p.score = mtb_part_score(m[r].right_end_hamming, 8, false) + (j.ptr >= 0 ? j.score : 0.0f);
    p.ham = (int32_t)m[r].hamming + (j.ptr >= 0 ? j.ham : 0);
    p.depth = 1 + (j.ptr >= 0 ? j.depth : 0);
    p.start_idx = r;
    return p;
}
#endif
