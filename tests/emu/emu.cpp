/*
 * tests/emu/emu.cpp -- TEST-ONLY host build of metabuli_amd/csrc/mtb_core.h.
 *
 * The HIP kernels call the per-lane functions of mtb_core.h; this file calls
 * the same functions from plain sequential loops so that the kernel
 * arithmetic can be checked against the oracle in a container without a GPU.
 * It is not a CPU fallback: nothing in the product library or in bench.py
 * links or loads it.
 */
#include <cstdlib>
#include <cstring>
#include <vector>
#include <algorithm>
#include "../../metabuli_amd/csrc/mtb_core.h"
#include "../../metabuli_amd/csrc/mtb_score_par.h"
#include "../../metabuli_amd/csrc/host_db.h"

extern "C" {

void emu_tables(mtb_tables *t) { mtb_build_tables(t); }

// mirrors kernels_extract.hip: per read, per frame, per window
size_t emu_extract_batch(const char *bases, const uint64_t *offs, const char *bases2, const uint64_t *offs2,
                         size_t n_reads, const mtb_params *p, mtb_kmer *out, size_t cap, int32_t *qlen, int32_t *qlen2) {
    mtb_tables t; mtb_build_tables(&t);
    size_t n = 0;
    for (size_t r = 0; r < n_reads; r++) {
        int32_t len1 = (int32_t)(offs[r + 1] - offs[r]);
        int32_t len2 = p->seq_mode == 2 ? (int32_t)(offs2[r + 1] - offs2[r]) : 0;
        qlen[r] = mtb_used_len(len1); qlen2[r] = p->seq_mode == 2 ? mtb_used_len(len2) : 0;
        bool skip = mtb_read_too_short(len1) || (p->seq_mode == 2 && mtb_read_too_short(len2));
        if (skip) continue;
        for (int mate = 0; mate < (p->seq_mode == 2 ? 2 : 1); mate++) {
            const char *seq = mate ? bases2 + offs2[r] : bases + offs[r];
            int32_t len = mate ? len2 : len1;
            uint32_t off = mate ? (uint32_t)(qlen[r] + 3) : 0;
            int32_t used = mtb_used_len(len);
            int32_t n_cod = used / 3, n_win = n_cod - 7;
            for (int f = 0; f < 6; f++) {
                bool fwd = f < 3;
                int32_t begin = mtb_frame_begin(len, f);
                std::vector<uint8_t> cod((size_t)n_cod);
                const bool old_fmt = p->kmer_format == 1;
                for (int j = 0; j < n_cod; j++)
                    cod[(size_t)j] = old_fmt ? mtb_codon_byte_old(&t, seq, mtb_codon_ci_old(begin, used, j, fwd), fwd)
                                             : mtb_codon_byte(&t, seq, mtb_codon_ci(begin, used, j, fwd), fwd);
                for (int w = 0; w < n_win; w++) {
                    uint64_t v;
                    bool ok = old_fmt ? mtb_window_metamer_old(&cod[(size_t)w], &v) : mtb_window_metamer(&cod[(size_t)w], p->syncmer, p->smer_len, &v);
                    if (ok) {
                        uint32_t wp = old_fmt ? mtb_window_pos_old(begin, used, w, fwd) : mtb_window_pos(begin, used, w, fwd);
                        if (n < cap) out[n] = {v, mtb_qinfo((uint32_t)(r + 1), wp + off, (uint32_t)f)};
                        n++;
                    }
                }
            }
        }
    }
    return n;
}

size_t emu_join(const uint64_t *values, const uint32_t *info, uint64_t T, const int32_t *tax2species, int32_t max_taxid,
                uint32_t info_mask, int kmer_format, const mtb_kmer *q, size_t n, mtb_match *out, size_t cap) {
    mtb_tables t; mtb_build_tables(&t);
    mtb_index_view ix{values, info, T, tax2species, max_taxid, info_mask, kmer_format};
    size_t m = 0;
    for (size_t j = 0; j < n; j++) {
        uint64_t rs = 0; uint32_t rl = 0;
        uint32_t c = mtb_join_query(&t, &ix, q[j].value, q[j].qinfo, nullptr, 0, 0, &rs, &rl);
        if (c == 0) continue;
        if (m + c <= cap) mtb_join_query(&t, &ix, q[j].value, q[j].qinfo, out + m, c, 1, &rs, &rl);
        m += c;
    }
    return m;
}

/* The search of k_join_dir (kernels_dir.h) restated sequentially, without the directory (the whole index is one bucket): ONE lower bound on
 * the query's whole value; a target equal to the query ends the search -- the selection is the block of equal targets (hamming 0 <=> equal
 * DNA parts, threshold min(2 x 0, 7) = 0) --, otherwise the run of the amino-acid part is found by stepping from the landing place and
 * evaluated as the reference does.  Must give emu_join's (= the reference loop's) matches, in the same order. */
size_t emu_join_one_bisection(const uint64_t *values, const uint32_t *info, uint64_t T, const int32_t *tax2species, int32_t max_taxid,
                              uint32_t info_mask, int kmer_format, const mtb_kmer *q, size_t n, mtb_match *out, size_t cap, uint64_t *n_exact) {
    mtb_tables t; mtb_build_tables(&t);
    const uint64_t AAM = ~0xFFFFFFull, limit = T ? T - 1 : 0;
    size_t m = 0; uint64_t exact = 0;
    for (size_t j = 0; j < n; j++) {
        const uint64_t qv = q[j].value, aa = qv & AAM;
        const uint64_t p = mtb_lower_bound(values, limit, qv);
        uint64_t s = p, e = p;
        if (p < limit && values[p] == qv) { e = p + 1; while (e < limit && values[e] == qv) e++; exact++; }
        else { while (s > 0 && (values[s - 1] & AAM) == aa) s--; while (e < limit && (values[e] & AAM) == aa) e++; }
        if (s >= e) continue;
        const uint32_t c = mtb_join_select(&t, values, s, (uint32_t)(e - s), qv, q[j].qinfo, info, 0, tax2species, max_taxid, info_mask, kmer_format, (mtb_match *)nullptr, 0);
        if (m + c <= cap) mtb_join_select(&t, values, s, (uint32_t)(e - s), qv, q[j].qinfo, info, 0, tax2species, max_taxid, info_mask, kmer_format, out + m, c);
        m += c;
    }
    if (n_exact) *n_exact = exact;
    return m;
}

void emu_sort_matches(mtb_match *m, size_t n) { std::sort(m, m + n, [](const mtb_match &a, const mtb_match &b) { return mtb_match_less(a, b); }); }

// mirrors kernels_score.hip score_read(): sequential over the sf blocks / species blocks
size_t emu_score(const uint8_t *acc_leaf, const int32_t *canon, const int32_t *parent, const int32_t *depth, const uint8_t *under_euk, const int32_t *sp_parent, int32_t max_taxid,
                 const mtb_params *p, const mtb_match *ml, size_t nM, size_t n_reads, const int32_t *qlen, const int32_t *qlen2,
                 mtb_result *res, int32_t *tc_tax, uint32_t *tc_cnt, size_t cap) {
    mtb_tax_view tx{acc_leaf, canon, parent, depth, under_euk, sp_parent, max_taxid};
    mtb_score_params sp; mtb_make_score_params(p, &sp);
    for (size_t r = 0; r < n_reads; r++) { res[r] = mtb_result{0, 0.f, qlen[r], qlen2 ? qlen2[r] : 0, 0, 0, 0, 0}; }
    size_t w = 0, idx = 0;
    while (idx < nM) {
        uint32_t seq = mtb_q_seq(ml[idx].qinfo);
        size_t s0 = idx; while (idx < nM && mtb_q_seq(ml[idx].qinfo) == seq) idx++;
        int32_t n = (int32_t)(idx - s0);
        const mtb_match *m = ml + s0;
        size_t r = seq - 1;
        int32_t read_len = qlen[r] + (qlen2 ? qlen2[r] : 0);
        std::vector<mtb_path> path((size_t)n); std::vector<uint8_t> flag((size_t)n, 0);
        std::vector<int32_t> order((size_t)n), acc((size_t)n);
        // phase 1: paths per (species, frame) block
        int32_t i = 0;
        while (i < n) {
            int32_t s = i; int32_t spc = m[i].species_id; uint32_t fr = mtb_q_frame(m[i].qinfo);
            while (i < n && m[i].species_id == spc && mtb_q_frame(m[i].qinfo) == fr) i++;
            if (i - s > 1) {
                int32_t md = (spc >= 0 && spc <= max_taxid && under_euk[spc]) ? sp.min_cons_cnt_euk : sp.min_cons_cnt;
                mtb_sf_block_paths(m, s, i, path.data(), flag.data(), &sp, md);
            }
        }
        // phase 2: per species combine (score at the species' first slot)
        std::vector<float> sps((size_t)n, -1.0f);
        i = 0;
        while (i < n) {
            int32_t s = i; int32_t spc = m[i].species_id;
            while (i < n && m[i].species_id == spc) i++;
            int32_t np = 0;
            float sc = mtb_species_combine(m, s, i, path.data(), flag.data(), order.data(), acc.data(), read_len, &np);
            if (np > 0) sps[(size_t)s] = sc < 1.0f ? sc : 1.0f;
        }
        // phase 3: decision
        int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
        std::vector<int32_t> btax((size_t)nb); std::vector<uint8_t> bham((size_t)nb);
        std::vector<int32_t> ot((size_t)nb); std::vector<uint32_t> oc((size_t)nb);
        mtb_result R = res[r];
        mtb_read_decide(m, n, sps.data(), &tx, &sp, read_len, btax.data(), bham.data(), nb, ot.data(), oc.data(), nb, &R);
        R.taxcnt_off = (uint32_t)w;
        for (int32_t k = 0; k < (int32_t)R.n_taxcnt; k++) { if (w < cap) { tc_tax[w] = ot[(size_t)k]; tc_cnt[w] = oc[(size_t)k]; } w++; }
        res[r] = R;
    }
    return w;
}

// mirrors kernels_score.h score_read_par(): the per-element phases of
// mtb_score_par.h run in plain loops (one loop == one lane-strided phase + barrier)
size_t emu_score_par(const uint8_t *acc_leaf, const int32_t *canon, const int32_t *parent, const int32_t *depth, const uint8_t *under_euk, const int32_t *sp_parent,
                     int32_t max_taxid, const mtb_params *p, const mtb_match *ml_in, size_t nM, size_t n_reads, const int32_t *qlen,
                     const int32_t *qlen2, mtb_result *res, int32_t *tc_tax, uint32_t *tc_cnt, size_t cap, int presorted, int use_chain,
                     size_t *n_chain_out) {
    size_t n_chain_reads = 0;
    mtb_tax_view tx{acc_leaf, canon, parent, depth, under_euk, sp_parent, max_taxid};
    mtb_score_params sp; mtb_make_score_params(p, &sp);
    for (size_t r = 0; r < n_reads; r++) { res[r] = mtb_result{0, 0.f, qlen[r], qlen2 ? qlen2[r] : 0, 0, 0, 0, 0}; }
    size_t wout = 0, idx = 0;
    typedef uint16_t IDX;
    while (idx < nM) {
        uint32_t seq = mtb_q_seq(ml_in[idx].qinfo);
        size_t s0 = idx; while (idx < nM && mtb_q_seq(ml_in[idx].qinfo) == seq) idx++;
        int32_t n = (int32_t)(idx - s0);
        size_t r = seq - 1;
        int32_t read_len = qlen[r] + (qlen2 ? qlen2[r] : 0);
        std::vector<uint8_t> slab(mtb_sws_bytes<IDX>((uint64_t)n) + 64);
        mtb_sws<IDX> w; mtb_sws_carve<IDX>(&w, slab.data(), (uint64_t)n); w.n = n;
        // rank sort (keys alias the path area)
        const mtb_match *src = ml_in + s0;
        if (presorted) { for (int32_t i = 0; i < n; i++) w.m[i] = src[i]; }
        else {
            uint64_t *k1 = (uint64_t *)w.path; uint32_t *k2 = (uint32_t *)(k1 + n);
            for (int32_t i = 0; i < n; i++) { k1[i] = mtb_key1(src[i]); k2[i] = mtb_key2(src[i]); }
            std::vector<int32_t> rank((size_t)n);
            for (int32_t i = 0; i < n; i++) {
                int32_t c = 0;
                for (int32_t j = 0; j < n; j++) c += (k1[j] < k1[i]) || (k1[j] == k1[i] && (k2[j] < k2[i] || (k2[j] == k2[i] && j < i)));
                rank[(size_t)i] = c;
            }
            for (int32_t i = 0; i < n; i++) w.m[rank[(size_t)i]] = src[i];
        }
        for (int32_t i = 0; i < n; i++) mtb_ph_flags(w, i);
        int32_t ng = 0, nbk = 0, nsp = 0;
        for (int32_t i = 0; i < n; i++) {
            uint32_t f = w.flag[i];
            ng += (f & MTB_F_GHEAD) ? 1 : 0; nbk += (f & MTB_F_BHEAD) ? 1 : 0; nsp += (f & MTB_F_SHEAD) ? 1 : 0;
            w.gid[i] = (IDX)(ng - 1); w.bid[i] = (IDX)(nbk - 1); w.sid[i] = (IDX)(nsp - 1);
        }
        for (int32_t i = 0; i < n; i++) mtb_ph_starts(w, i, &tx);
        int32_t maxrank = 0;
        // links: every element reads shared arrays written by EARLIER phases only, except its own slots
        {
            // emulate "all lanes read, then write" per chunk is unnecessary: reads touch gid/grp_start/blk_start/acc/sid/m, writes touch path/flag/shift/cmask/rk/bid[own]
            if (ng == n) {      // every position group is a single match: the kernel's shortcut must write the same workspace
                std::vector<uint8_t> slab2(slab);
                mtb_sws<IDX> w2; mtb_sws_carve<IDX>(&w2, slab2.data(), (uint64_t)n); w2.n = n;
                for (int32_t i = 0; i < n; i++) mtb_ph_links_unit(w2, i, &sp);
                for (int32_t i = 0; i < n; i++) { mtb_ph_links(w, i, &tx, &sp, ng, nbk); }
                // padding bytes of mtb_path (none) / untouched fields are identical by construction
                if (memcmp(slab.data(), slab2.data(), slab.size()) != 0) { fprintf(stderr, "emu: links_unit mismatch\n"); abort(); }
            } else
            for (int32_t i = 0; i < n; i++) { mtb_ph_links(w, i, &tx, &sp, ng, nbk); }
            for (int32_t i = 0; i < n; i++) maxrank = std::max<int32_t>(maxrank, w.rk[i]);
        }
        bool simple = use_chain != 0;
        for (int32_t i = 0; i < n; i++) simple = simple && mtb_chain_simple(w, i);
        if (simple) {       // pointer doubling (jump[] aliases the path storage, as in the kernel)
            mtb_jump *jump = (mtb_jump *)w.path;
            for (int32_t i = 0; i < n; i++) mtb_ph_jump_init(w, i, jump);
            std::vector<mtb_jump> tmp((size_t)n);
            for (int32_t span = 1; span <= maxrank; span <<= 1) {
                for (int32_t i = 0; i < n; i++) tmp[(size_t)i] = mtb_ph_jump_step(jump, i);
                for (int32_t i = 0; i < n; i++) jump[i] = tmp[(size_t)i];
            }
            for (int32_t i = 0; i < n; i++) tmp[(size_t)i] = jump[i];
            for (int32_t i = 0; i < n; i++) w.path[i] = mtb_ph_jump_final(w, i, tmp[(size_t)i]);
            n_chain_reads++;
        } else
        for (int32_t rr = 1; rr <= maxrank; rr++) for (int32_t i = 0; i < n; i++) mtb_ph_round(w, i, rr, &sp);
        // emit + compaction (elist = gid array, prefix = rk array)
        IDX *elist = w.gid, *ec = w.rk;
        int32_t ne = 0;
        {
            std::vector<uint8_t> em((size_t)n);
            for (int32_t i = 0; i < n; i++) em[(size_t)i] = mtb_ph_emit(w, i, &sp) ? 1 : 0;
            for (int32_t i = 0; i < n; i++) { ec[i] = (IDX)ne; if (em[(size_t)i]) elist[ne++] = (IDX)i; }
        }
        float *sps = (float *)w.grp_start;
        {
            // combine in the kernel's parallel pieces: rank -> sorted list (bid), range start (sid), pre-drop flags (shift), greedy
            IDX *sorted = w.bid, *elo = w.sid; uint8_t *predrop = w.shift;
            std::vector<int32_t> pos((size_t)ne);
            for (int32_t e = 0; e < ne; e++) {
                int32_t s = mtb_ph_comb_species(w, nsp, (int32_t)elist[e]);
                int32_t lo = ec[w.sp_start[s]], hi = (s + 1 < nsp) ? (int32_t)ec[w.sp_start[s + 1]] : ne;
                pos[(size_t)e] = mtb_ph_comb_rank(w, elist, e, lo, hi);
                {   // the packed-key variant used by the slab scorer must rank identically
                    static thread_local std::vector<uint64_t> keys;
                    keys.resize((size_t)ne);
                    for (int32_t q = 0; q < ne; q++) keys[(size_t)q] = mtb_path_key(w.path[elist[q]]);
                    if (mtb_ph_comb_rank_keys(w, elist, keys.data(), e, lo, hi) != pos[(size_t)e]) { fprintf(stderr, "emu: comb_rank_keys mismatch\n"); abort(); }
                }
                elo[e] = (IDX)lo;
            }
            for (int32_t e = 0; e < ne; e++) sorted[pos[(size_t)e]] = elist[e];
            for (int32_t k = 0; k < ne; k++) predrop[k] = mtb_ph_comb_predrop(w, sorted, k, (int32_t)elo[k]) ? 1 : 0;
            std::vector<float> tmp((size_t)nsp);
            for (int32_t s = 0; s < nsp; s++) {
                int32_t lo = ec[w.sp_start[s]], hi = (s + 1 < nsp) ? (int32_t)ec[w.sp_start[s + 1]] : ne;
                float sc = -1.0f;
                if (hi > lo) { sc = mtb_ph_comb_greedy(w, sorted, predrop, lo, hi, read_len); sc = sc < 1.0f ? sc : 1.0f; }
                tmp[(size_t)s] = sc;
            }
            for (int32_t s = 0; s < nsp; s++) sps[s] = tmp[(size_t)s];
        }
        mtb_result R = res[r];
        int32_t bs = 0, be = 0, species = 0;
        bool go = mtb_ph_select(w, sps, nsp, &tx, &sp, &R, &bs, &be, &species);
        R.taxcnt_off = (uint32_t)wout;
        if (go) {
            int32_t nb = mtb_num_buckets(read_len, sp.dna_shift);
            std::vector<int32_t> btax((size_t)nb, -1), otax((size_t)nb); std::vector<uint32_t> hmin((size_t)nb, 255), ocnt((size_t)nb); std::vector<uint8_t> bham((size_t)nb);
            for (int32_t i = bs; i < be; i++) mtb_ph_filter_min(w.m, i, sp.dna_shift, nb, hmin.data());
            for (int32_t i = be - 1; i >= bs; i--) mtb_ph_filter_merge(w.m, i, sp.dna_shift, nb, hmin.data(), btax.data(), &tx);   // reverse order on purpose
            for (int32_t q = 0; q < nb; q++) bham[(size_t)q] = hmin[(size_t)q] == 255 ? 255 : 0;
            int32_t ntc = mtb_taxcnt_gather(btax.data(), bham.data(), nb, otax.data(), ocnt.data(), nb);
            R.n_taxcnt = (uint16_t)ntc;
            if (R.score < sp.min_sp_score) R.classification = (species >= 0 && species <= max_taxid) ? sp_parent[species] : 0;
            else {
                bool slow = ntc > MTB_LR_MAXE;
                std::vector<int32_t> lev((size_t)std::max(ntc, 1)), anc((size_t)std::max(ntc, 1) * MTB_LR_K);
                if (!slow) for (int32_t i = 0; i < ntc; i++) { mtb_lr_climb(&tx, otax[(size_t)i], species, &lev[(size_t)i], &anc[(size_t)i * MTB_LR_K]); if (lev[(size_t)i] > MTB_LR_K) slow = true; }
                int32_t cs = mtb_tax_canon(&tx, species);
                if (slow || cs < 0) R.classification = mtb_lower_rank(&tx, otax.data(), ocnt.data(), ntc, species, read_len, sp.denominator, sp.accession_level);
                else R.classification = mtb_lr_bfs(lev.data(), anc.data(), ocnt.data(), ntc, cs, read_len, sp.denominator, &tx, sp.accession_level);
            }
            for (int32_t k = 0; k < ntc; k++) { if (wout < cap) { tc_tax[wout] = otax[(size_t)k]; tc_cnt[wout] = ocnt[(size_t)k]; } wout++; }
        }
        res[r] = R;
    }
    if (n_chain_out) *n_chain_out = n_chain_reads;
    return wout;
}

// host_db.h (what mtb_index_open does on the host) exposed for CPU tests
int emu_load_taxonomy(const char *dir, const int32_t *taxid_list, size_t n_ids, int32_t cap, int32_t *max_id, int32_t *canon, int32_t *parent,
                      int32_t *depth, uint8_t *under_euk, int32_t *sp_parent, int32_t *tax2species, uint8_t *acc_leaf) {
    mtbhost::Taxonomy t; std::string err;
    if (!mtbhost::load_taxonomy(dir, &t, &err)) return 1;
    mtbhost::build_tax2species(&t, taxid_list, n_ids);
    *max_id = t.max_id;
    if (t.max_id + 1 > cap) return 2;
    size_t n = (size_t)t.max_id + 1;
    memcpy(canon, t.canon.data(), n * 4); memcpy(parent, t.parent.data(), n * 4); memcpy(depth, t.depth.data(), n * 4);
    memcpy(acc_leaf, t.acc_leaf.data(), n); memcpy(under_euk, t.under_euk.data(), n); memcpy(sp_parent, t.sp_parent.data(), n * 4); memcpy(tax2species, t.tax2species.data(), n * 4);
    return 0;
}
/* host_db.h::load_taxonomy_db: the same arrays from a binary taxonomyDB file, plus the internal -> original id map */
int emu_load_taxonomy_db(const char *path, const int32_t *taxid_list, size_t n_ids, int32_t cap, int32_t *max_id, int32_t *canon, int32_t *parent,
                         int32_t *depth, uint8_t *under_euk, int32_t *sp_parent, int32_t *tax2species, uint8_t *acc_leaf, int32_t *orig,
                         int32_t *eukaryota, char *err_out, size_t err_cap) {
    mtbhost::Taxonomy t; std::string err;
    if (!mtbhost::load_taxonomy_db(path, &t, &err)) { if (err_out && err_cap) { strncpy(err_out, err.c_str(), err_cap - 1); err_out[err_cap - 1] = 0; } return 1; }
    mtbhost::build_tax2species(&t, taxid_list, n_ids);
    *max_id = t.max_id; *eukaryota = t.eukaryota;
    if (t.max_id + 1 > cap) return 2;
    size_t n = (size_t)t.max_id + 1;
    memcpy(canon, t.canon.data(), n * 4); memcpy(parent, t.parent.data(), n * 4); memcpy(depth, t.depth.data(), n * 4);
    memcpy(acc_leaf, t.acc_leaf.data(), n); memcpy(under_euk, t.under_euk.data(), n); memcpy(sp_parent, t.sp_parent.data(), n * 4); memcpy(tax2species, t.tax2species.data(), n * 4);
    memcpy(orig, t.orig.data(), n * 4);
    return 0;
}
int emu_load_db_parameters(const char *dir, mtb_params *p) { return mtbhost::load_db_parameters(dir, p) ? 0 : 1; }

} // extern "C"

// 16-byte slot form of a match (mtb_core.h): pack + unpack must return the record (pad = 0), the epoch must survive
extern "C" int emu_slot_roundtrip(const mtb_match *m, size_t n, uint32_t epoch, mtb_match *out) {
    for (size_t i = 0; i < n; i++) {
        mtb_slot16 s = mtb_slot_pack(m[i].qinfo, m[i].target_id, m[i].species_id, m[i].dna, m[i].right_end_hamming, m[i].hamming, epoch);
        if (mtb_slot_epoch(s) != (epoch & 31u)) return 1;
        out[i] = mtb_slot_unpack(s, mtb_q_seq(m[i].qinfo));
    }
    return 0;
}
