/*
 * null_mtb.cpp -- TEST INFRASTRUCTURE ONLY: a host-only stand-in for the C ABI of libmtb.so (include/mtb.h), for tests of the
 * DRIVER (metabuli_amd/csrc/host/classify_main.cpp: parsing, batching, 2-bit packing, the cut of a batch over several engines,
 * the order of the rows, formatting, the report, error handling) on a machine without a GPU, and for timing the host pipeline alone.
 *
 * It classifies NOTHING.  Every read gets a pseudo-result computed from a checksum of its (normalised) bases, so that a test can
 * predict the driver's output from the input file alone (tests/test_driver_null_engine.py restates the rule in Python).  The
 * taxonomy services are the real host code (csrc/host_db.h), because the driver's formatting goes through them.
 *
 * Never shipped and never linked by the product: csrc/Makefile builds mtb_classify against libmtb.so, which has no CPU path
 * (mtb_ctx_create fails without a HIP device).  tests/ builds this file into a temporary directory as libmtb_null.so and links a
 * separate test binary against it.
 *
 * Knobs (environment): MTB_NULL_DEVICES (device ordinals accepted, default 4), MTB_NULL_TC_MAX (taxID:count entries per classified
 * read: 1..MAX, default 3), MTB_NULL_FAIL_CALL=k (the k-th classify call of the process, counted from 1, fails with MTB_ERR_DEVICE),
 * MTB_NULL_DELAY_MS (sleep per classify call: a stand-in for device time), MTB_NULL_VERBOSE=1 (call counters on stderr at exit),
 * MTB_NULL_FAST=1 (timing the host pipeline: the packed entry point does not look at the bases -- read i of a call is classified as
 * taxon i mod #taxa with two list entries; MTB_NULL_US_PER_KREAD = simulated device time per 1000 reads of a call).
 */
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <thread>
#include <vector>
#include <sys/stat.h>
#include <zlib.h>

#include "../../include/mtb.h"
#include "../../metabuli_amd/csrc/mtb_core.h"
#include "../../metabuli_amd/csrc/host_db.h"

namespace {
thread_local std::string g_err;
mtb_status fail(mtb_status s, const std::string &m) { g_err = m; return s; }
int env_int(const char *k, int d) { const char *v = getenv(k); return v && *v ? atoi(v) : d; }

std::atomic<unsigned long> g_calls{0}, g_prefetched{0}, g_prefetch_calls{0}, g_retries{0};
struct AtExit { ~AtExit() { if (env_int("MTB_NULL_VERBOSE", 0)) fprintf(stderr, "null engine: %lu classify calls, %lu capacity retries, %lu prefetch calls, %lu batches found prefetched\n",
                                                                     g_calls.load(), g_retries.load(), g_prefetch_calls.load(), g_prefetched.load()); } } g_at_exit;
}  // namespace

struct mtb_ctx {
    int device = 0;
    mtb_tables tabs;
    mtb_batch_stats stats;
    /* the prefetch protocol of mtb.h: prefetch(k+1), classify(k), prefetch(k+2), classify(k+1) ... with the same pointers */
    const void *pf_p2 = nullptr, *pf_nm = nullptr, *pf_len = nullptr; uint64_t pf_n = 0; bool pf_pending = false, pf_ready = false;
    const void *rd_p2 = nullptr, *rd_nm = nullptr, *rd_len = nullptr; uint64_t rd_n = 0;
    /* mtb_classify_batch_packed_async: the results of a call reach the caller's arrays when the NEXT call returns or at mtb_ctx_wait_results;
     * until then the arrays are filled with 0xEE, so that a driver that hands a batch on too early writes garbage rows */
    std::vector<mtb_result> as_res; std::vector<int32_t> as_tt; std::vector<uint32_t> as_tc;
    mtb_result *as_dst = nullptr; int32_t *as_dtt = nullptr; uint32_t *as_dtc = nullptr; bool as_pending = false;
    void deliver() {
        if (!as_pending) return;
        if (!as_res.empty()) memcpy(as_dst, as_res.data(), as_res.size() * sizeof(mtb_result));
        if (!as_tt.empty()) { memcpy(as_dtt, as_tt.data(), as_tt.size() * 4); memcpy(as_dtc, as_tc.data(), as_tc.size() * 4); }
        as_pending = false;
    }
};
struct mtb_index {
    mtbhost::Taxonomy tax;
    std::vector<int32_t> ids;
    mtb_params params;
    uint64_t T = 0;
};

namespace {
/* the pseudo-result of one read (mate text appended after a '|') */
void pseudo(const mtb_index *ix, const std::string &text, uint32_t len1, uint32_t len2, int tc_max, mtb_result *r, int32_t *tt, uint32_t *tc) {
    const uint64_t lo = crc32(0L, (const Bytef *)text.data(), (uInt)text.size());
    const uint64_t hi = crc32(0x5bd1e995UL, (const Bytef *)text.data(), (uInt)text.size());
    const uint64_t h = lo | (hi << 32), n = ix->ids.size();
    memset(r, 0, sizeof(*r));
    const bool cls = (h & 7) != 0 && n > 0;
    r->is_classified = cls;
    r->classification = cls ? ix->ids[(size_t)((h >> 3) % n)] : 0;
    r->score = (float)((double)((h >> 24) % 100001) / 100000.0);
    r->query_length = (int32_t)len1; r->query_length2 = (int32_t)len2;
    r->n_taxcnt = cls ? (uint16_t)(1 + (h >> 44) % (uint64_t)tc_max) : 0;
    for (uint32_t k = 0; k < r->n_taxcnt; k++) {
        tt[k] = ix->ids[(size_t)(((h >> 48) + 7u * k) % n)];
        tc[k] = 1 + (uint32_t)((h >> (52 + 3 * (k % 4))) & 7);
    }
}

/* one batch from normalised texts; the lists arrive packed (taxcnt_off = running total); too small a capacity: the size needed, nothing copied */
mtb_status run_batch(mtb_ctx *c, mtb_index *ix, const std::vector<std::string> &texts, const std::vector<uint32_t> &l1, const std::vector<uint32_t> &l2,
                     mtb_result *results, int32_t *tt, uint32_t *tc, uint64_t cap, uint64_t *ntc) {
    const unsigned long call = ++g_calls;
    if ((int)call == env_int("MTB_NULL_FAIL_CALL", -1)) return fail(MTB_ERR_DEVICE, "injected failure of classify call " + std::to_string(call));
    const int tc_max = std::max(1, std::min(16, env_int("MTB_NULL_TC_MAX", 3)));
    const int delay = env_int("MTB_NULL_DELAY_MS", 0);
    if (delay > 0) std::this_thread::sleep_for(std::chrono::milliseconds(delay));
    const size_t n = texts.size();
    std::vector<mtb_result> res(n);
    std::vector<int32_t> t_all; std::vector<uint32_t> c_all;
    int32_t tb[16]; uint32_t cb[16];
    uint64_t total = 0, bases = 0;
    for (size_t i = 0; i < n; i++) {
        pseudo(ix, texts[i], l1[i], l2.empty() ? 0 : l2[i], tc_max, &res[i], tb, cb);
        res[i].taxcnt_off = (uint32_t)total;
        for (uint32_t k = 0; k < res[i].n_taxcnt; k++) { t_all.push_back(tb[k]); c_all.push_back(cb[k]); }
        total += res[i].n_taxcnt; bases += l1[i] + (l2.empty() ? 0 : l2[i]);
    }
    if (ntc) *ntc = total;
    if (total > cap) { ++g_retries; return fail(MTB_ERR_CAPACITY, "taxcnt capacity"); }
    memcpy(results, res.data(), n * sizeof(mtb_result));
    if (total) { memcpy(tt, t_all.data(), total * 4); memcpy(tc, c_all.data(), total * 4); }
    memset(&c->stats, 0, sizeof(c->stats));
    c->stats.ms_total = (float)delay; c->stats.n_reads = n; c->stats.n_bases = bases; c->stats.n_targets = ix->T;
    return MTB_OK;
}

std::string normalise(const mtb_tables &t, const char *s, size_t n) {
    std::string o(n, 'N');
    for (size_t i = 0; i < n; i++) { const uint8_t c = t.base[(uint8_t)s[i]]; if (c < 4) o[i] = "ACTG"[c]; }
    return o;
}
std::string unpack(const uint8_t *p2, const uint8_t *nm, uint64_t group, uint32_t len) {
    std::string o(len, 'N');
    for (uint32_t k = 0; k < len; k++) {
        const uint64_t g = group + k / 8; const uint32_t b = k % 8;
        const uint32_t w = p2[2 * g] | ((uint32_t)p2[2 * g + 1] << 8);
        if (!((nm[g] >> b) & 1)) o[k] = "ACTG"[(w >> (2 * b)) & 3];
    }
    return o;
}
mtb_status open_common(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, mtb_index **out) {
    if (!c || !dbdir || !params || !out) return fail(MTB_ERR_ARG, "NULL argument");
    const std::string d(dbdir);
    int reduced = 0;
    mtbhost::load_db_parameters(d, params, &reduced);
    std::string taxdir = taxonomy_dir && *taxonomy_dir ? std::string(taxonomy_dir) : d + "/taxonomy";
    const bool have_bin = mtbhost::file_exists(d + "/taxonomyDB");
    if (!have_bin && !mtbhost::file_exists(taxdir + "/nodes.dmp")) return fail(MTB_ERR_IO, "no taxonomy: neither " + d + "/taxonomyDB nor dump files in " + taxdir);
    mtb_index *ix = new mtb_index();
    std::string err;
    bool ok = have_bin && mtbhost::load_taxonomy_db(d + "/taxonomyDB", &ix->tax, &err);
    if (!ok && !mtbhost::load_taxonomy(have_bin ? d + "/taxonomy" : taxdir, &ix->tax, &err)) { delete ix; return fail(MTB_ERR_IO, err); }
    if (!mtbhost::read_taxid_list(d + "/taxID_list", &ix->ids)) { delete ix; return fail(MTB_ERR_IO, "cannot open " + d + "/taxID_list"); }
    mtbhost::build_tax2species(&ix->tax, ix->ids.data(), ix->ids.size());
    struct stat sb;
    ix->T = stat((d + "/info").c_str(), &sb) == 0 ? (uint64_t)sb.st_size / 4 : 0;
    ix->params = *params;
    *out = ix;
    return MTB_OK;
}
}  // namespace

extern "C" {

const char *mtb_version(void) { return "metabuli_amd null engine (tests only)"; }
const char *mtb_last_error(void) { return g_err.c_str(); }
void mtb_default_params(mtb_params *p) {       /* (the values of libmtb.so: setClassifyDefaults, classify.cpp:10-37) */
    p->seq_mode = 2; p->syncmer = 0; p->smer_len = 5; p->kmer_format = 1; p->min_cons_cnt = 4; p->min_cons_cnt_euk = 9;
    p->min_score = 0.0f; p->min_sp_score = 0.0f; p->tie_ratio = 0.95f; p->accession_level = 0; p->skip_redundancy = 0;
}
mtb_status mtb_ctx_create(int device, void *, mtb_ctx **out) {
    if (!out) return fail(MTB_ERR_ARG, "out is NULL");
    if (device < 0 || device >= env_int("MTB_NULL_DEVICES", 4)) return fail(MTB_ERR_ARG, "bad device ordinal");
    mtb_ctx *c = new mtb_ctx(); c->device = device; mtb_build_tables(&c->tabs); memset(&c->stats, 0, sizeof(c->stats));
    *out = c; return MTB_OK;
}
void mtb_ctx_destroy(mtb_ctx *c) { delete c; }
mtb_status mtb_ctx_sync(mtb_ctx *) { return MTB_OK; }
mtb_status mtb_ctx_set_profiling(mtb_ctx *, int) { return MTB_OK; }
mtb_status mtb_ctx_set_streams(mtb_ctx *, int) { return MTB_OK; }
mtb_status mtb_ctx_set_workspace_limit(mtb_ctx *, uint64_t) { return MTB_OK; }
mtb_status mtb_ctx_set_placement_probe(mtb_ctx *, int) { return MTB_OK; }
mtb_status mtb_ctx_set_join_variant(mtb_ctx *, int) { return MTB_OK; }
mtb_status mtb_index_export(mtb_index *, mtb_index_share *) { return MTB_ERR_UNSUPPORTED; }
mtb_status mtb_index_import(mtb_ctx *, const mtb_index_share *, const char *, const int32_t *, size_t, const mtb_params *, mtb_index **) { return MTB_ERR_UNSUPPORTED; }
mtb_status mtb_ctx_set_option(mtb_ctx *, const char *, const char *) { return MTB_OK; }
uint32_t mtb_ctx_last_sub_batches(const mtb_ctx *) { return 1; }
uint64_t mtb_ctx_last_scratch_bytes(const mtb_ctx *) { return 0; }
mtb_status mtb_ctx_reserve(mtb_ctx *c, const mtb_params *p, uint64_t, uint64_t) { return c && p ? MTB_OK : fail(MTB_ERR_ARG, "NULL argument"); }
mtb_status mtb_db_parameters(const char *dbdir, mtb_params *p) {
    if (!dbdir || !p) return fail(MTB_ERR_ARG, "NULL argument");
    int reduced = 0; mtbhost::load_db_parameters(dbdir, p, &reduced); return MTB_OK;
}
mtb_status mtb_last_batch_stats(mtb_ctx *c, mtb_batch_stats *out) { if (!c || !out) return fail(MTB_ERR_ARG, "NULL argument"); *out = c->stats; return MTB_OK; }

mtb_status mtb_index_open(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, mtb_index **out) { return open_common(c, dbdir, taxonomy_dir, params, out); }
mtb_status mtb_index_open_part(mtb_ctx *c, const char *dbdir, const char *taxonomy_dir, mtb_params *params, uint32_t part, uint32_t n_parts, mtb_index **out) {
    if (n_parts == 0 || part >= n_parts) return fail(MTB_ERR_ARG, "bad part");
    return open_common(c, dbdir, taxonomy_dir, params, out);
}
mtb_status mtb_index_part_bounds(const char *dbdir, uint32_t n_parts, uint64_t *bounds) {
    if (!dbdir || !bounds || n_parts == 0) return fail(MTB_ERR_ARG, "bad argument");
    for (uint32_t p = 0; p < n_parts; p++) bounds[p] = p ? (~0ull / n_parts) * p & ~0xFFFFFFull : 0;
    return MTB_OK;
}
mtb_status mtb_index_open_stats(const mtb_index *ix, uint64_t *o) { if (!ix || !o) return fail(MTB_ERR_ARG, "NULL argument"); o[0] = 1; o[1] = 0; o[2] = 0; o[3] = 0; return MTB_OK; }
mtb_status mtb_index_clone(mtb_index *src, mtb_ctx *dst, mtb_index **out) {
    if (!src || !dst || !out) return fail(MTB_ERR_ARG, "NULL argument");
    mtb_index *ix = new mtb_index(); ix->tax = src->tax; ix->ids = src->ids; ix->params = src->params; ix->T = src->T; *out = ix; return MTB_OK;
}
mtb_status mtb_index_seal(mtb_index *) { return MTB_OK; }
void mtb_index_close(mtb_index *ix) { delete ix; }
uint64_t mtb_index_num_targets(const mtb_index *ix) { return ix ? ix->T : 0; }
mtb_status mtb_index_state(const mtb_index *ix, int32_t *d, int32_t *p, int32_t *s) { if (!ix) return fail(MTB_ERR_ARG, "NULL argument"); if (d) *d = 0; if (p) *p = 0; if (s) *s = 0; return MTB_OK; }

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

void *mtb_host_alloc(size_t bytes) { void *p = nullptr; return posix_memalign(&p, 4096, bytes ? bytes : 1) == 0 ? p : nullptr; }
void mtb_host_free(void *p) { free(p); }

mtb_status mtb_classify_batch(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const char *bases, const uint64_t *offs, const char *bases2, const uint64_t *offs2,
                              uint64_t n, mtb_result *results, int32_t *tt, uint32_t *tc, uint64_t cap, uint64_t *ntc) {
    if (!c || !ix || !p || !bases || !offs || !results) return fail(MTB_ERR_ARG, "NULL argument");
    const bool paired = p->seq_mode == 2;
    if (paired && (!bases2 || !offs2)) return fail(MTB_ERR_ARG, "seq_mode 2 without mates");
    std::vector<std::string> texts(n); std::vector<uint32_t> l1(n), l2(paired ? n : 0);
    for (uint64_t i = 0; i < n; i++) {
        l1[i] = (uint32_t)(offs[i + 1] - offs[i]);
        texts[i] = normalise(c->tabs, bases + offs[i], l1[i]);
        if (paired) { l2[i] = (uint32_t)(offs2[i + 1] - offs2[i]); texts[i] += '|'; texts[i] += normalise(c->tabs, bases2 + offs2[i], l2[i]); }
    }
    return run_batch(c, ix, texts, l1, l2, results, tt, tc, cap, ntc);
}

mtb_status mtb_prefetch_batch_packed(mtb_ctx *c, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                     const uint8_t *, const uint8_t *, const uint32_t *, uint64_t n) {
    if (!c || !p || !packed2 || !nmask || !lens) return fail(MTB_ERR_ARG, "NULL argument");
    ++g_prefetch_calls;
    if (c->pf_pending) return fail(MTB_ERR_ARG, "prefetch protocol: two prefetches without a classify call between them");
    c->pf_p2 = packed2; c->pf_nm = nmask; c->pf_len = lens; c->pf_n = n; c->pf_pending = true;
    return MTB_OK;
}

mtb_status mtb_classify_batch_packed(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                     const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n,
                                     mtb_result *results, int32_t *tt, uint32_t *tc, uint64_t cap, uint64_t *ntc) {
    if (!c || !ix || !p || !packed2 || !nmask || !lens || !results) return fail(MTB_ERR_ARG, "NULL argument");
    const bool paired = p->seq_mode == 2;
    if (paired && (!packed2_mate || !nmask_mate || !lens_mate)) return fail(MTB_ERR_ARG, "seq_mode 2 without mates");
    /* what was prefetched during the previous call is this call's batch; what is prefetched now (before this call) is the next one's */
    if (c->pf_ready) {
        if (c->rd_p2 != packed2 || c->rd_nm != nmask || c->rd_len != lens || c->rd_n != n)
            return fail(MTB_ERR_ARG, "prefetch protocol: the classify call does not get the arrays that were prefetched for it");
    }
    if (env_int("MTB_NULL_FAST", 0)) {
        ++g_calls;
        const uint64_t total = 2 * n, nid = ix->ids.size();
        if (ntc) *ntc = total;
        if (total > cap) { ++g_retries; return fail(MTB_ERR_CAPACITY, "taxcnt capacity"); }
        const auto t0 = std::chrono::steady_clock::now();
        for (uint64_t i = 0; i < n; i++) {
            mtb_result &r = results[i];
            r.classification = ix->ids[(size_t)(i % nid)]; r.score = 0.5f; r.query_length = (int32_t)lens[i]; r.query_length2 = paired ? (int32_t)lens_mate[i] : 0;
            r.is_classified = 1; r.reserved = 0; r.n_taxcnt = 2; r.taxcnt_off = (uint32_t)(2 * i);
            tt[2 * i] = r.classification; tc[2 * i] = 20; tt[2 * i + 1] = ix->ids[(size_t)((i + 1) % nid)]; tc[2 * i + 1] = 3;
        }
        const int us = env_int("MTB_NULL_US_PER_KREAD", 0);
        if (us > 0) std::this_thread::sleep_until(t0 + std::chrono::microseconds((long long)us * (long long)n / 1000));
        memset(&c->stats, 0, sizeof(c->stats));
        c->stats.ms_total = us * 1e-6f * (float)n; c->stats.n_reads = n;
        c->pf_ready = c->pf_pending; c->rd_p2 = c->pf_p2; c->rd_nm = c->pf_nm; c->rd_len = c->pf_len; c->rd_n = c->pf_n; c->pf_pending = false;
        return MTB_OK;
    }
    std::vector<std::string> texts(n); std::vector<uint32_t> l1(lens, lens + n), l2;
    if (paired) l2.assign(lens_mate, lens_mate + n);
    uint64_t g1 = 0, g2 = 0;
    for (uint64_t i = 0; i < n; i++) {
        texts[i] = unpack(packed2, nmask, g1, l1[i]); g1 += (l1[i] + 7u) / 8u;
        if (paired) { texts[i] += '|'; texts[i] += unpack(packed2_mate, nmask_mate, g2, l2[i]); g2 += (l2[i] + 7u) / 8u; }
    }
    mtb_status st = run_batch(c, ix, texts, l1, l2, results, tt, tc, cap, ntc);
    if (st == MTB_ERR_CAPACITY) return st;          /* the same batch comes again: the prefetch state stays */
    if (st == MTB_OK && c->pf_ready) ++g_prefetched;
    c->pf_ready = c->pf_pending; c->rd_p2 = c->pf_p2; c->rd_nm = c->pf_nm; c->rd_len = c->pf_len; c->rd_n = c->pf_n; c->pf_pending = false;
    return st;
}

mtb_status mtb_classify_batch_packed_async(mtb_ctx *c, mtb_index *ix, const mtb_params *p, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                           const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n,
                                           mtb_result *results, int32_t *tt, uint32_t *tc, uint64_t cap, uint64_t *ntc) {
    if (!c || !results || !ntc) return fail(MTB_ERR_ARG, "NULL argument");
    std::vector<mtb_result> r(n); std::vector<int32_t> a(cap); std::vector<uint32_t> b(cap);
    mtb_status st = mtb_classify_batch_packed(c, ix, p, packed2, nmask, lens, packed2_mate, nmask_mate, lens_mate, n, r.data(), a.data(), b.data(), cap, ntc);
    if (st == MTB_ERR_CAPACITY) return st;      /* nothing queued, nothing delivered: the same batch comes again */
    c->deliver();                               /* the previous call's results are the caller's now, whatever this call's status */
    if (st != MTB_OK) return st;
    a.resize(*ntc); b.resize(*ntc);
    c->as_res.swap(r); c->as_tt.swap(a); c->as_tc.swap(b); c->as_dst = results; c->as_dtt = tt; c->as_dtc = tc; c->as_pending = true;
    memset(results, 0xEE, n * sizeof(mtb_result)); if (cap) { memset(tt, 0xEE, cap * 4); memset(tc, 0xEE, cap * 4); }
    return MTB_OK;
}
mtb_status mtb_ctx_prefetch_stats(mtb_ctx *c, uint64_t *issued, uint64_t *used) {
    if (!c) return fail(MTB_ERR_ARG, "NULL argument");
    if (issued) *issued = g_prefetch_calls.load();
    if (used) *used = g_prefetched.load();
    return MTB_OK;
}
mtb_status mtb_ctx_wait_results(mtb_ctx *c) { if (!c) return fail(MTB_ERR_ARG, "NULL argument"); c->deliver(); return MTB_OK; }

/* one process, several engines, every engine owning a value range: here simply the text entry point on the first engine */
mtb_status mtb_classify_batch_partitioned(mtb_ctx **ctxs, mtb_index **parts, uint32_t n, const uint64_t *bounds, const mtb_params *p, const char *bases,
                                          const uint64_t *offs, const char *bases2, const uint64_t *offs2, uint64_t n_reads, mtb_result *results,
                                          int32_t *tt, uint32_t *tc, uint64_t cap, uint64_t *ntc) {
    if (!ctxs || !parts || !n || !bounds) return fail(MTB_ERR_ARG, "NULL argument");
    return mtb_classify_batch(ctxs[0], parts[0], p, bases, offs, bases2, offs2, n_reads, results, tt, tc, cap, ntc);
}

/* everything else of the ABI: not part of what the driver's pipeline needs */
#define NULL_UNSUPPORTED { return fail(MTB_ERR_UNSUPPORTED, "null engine (tests only): not implemented"); }
mtb_status mtb_index_from_device(mtb_ctx *, uint64_t *, uint32_t *, uint64_t, const char *, const int32_t *, size_t, const mtb_params *, mtb_index **) NULL_UNSUPPORTED
mtb_status mtb_index_download(mtb_index *, uint64_t *, uint32_t *, uint64_t) NULL_UNSUPPORTED
mtb_status mtb_extract(mtb_ctx *, const mtb_params *, const char *, const uint64_t *, const char *, const uint64_t *, uint64_t, mtb_kmer *, uint64_t, uint64_t *, int32_t *, int32_t *) NULL_UNSUPPORTED
mtb_status mtb_sort_kmers(mtb_ctx *, mtb_kmer *, uint64_t) NULL_UNSUPPORTED
mtb_status mtb_match_kmers(mtb_ctx *, mtb_index *, const mtb_kmer *, uint64_t, mtb_match *, uint64_t, uint64_t *) NULL_UNSUPPORTED
mtb_status mtb_sort_matches(mtb_ctx *, mtb_match *, uint64_t, uint64_t) NULL_UNSUPPORTED
mtb_status mtb_score(mtb_ctx *, mtb_index *, const mtb_params *, const mtb_match *, uint64_t, uint64_t, const int32_t *, const int32_t *, mtb_result *, int32_t *, uint32_t *, uint64_t, uint64_t *) NULL_UNSUPPORTED
mtb_status mtb_classify_batch_device(mtb_ctx *, mtb_index *, const mtb_params *, const char *, const uint64_t *, const char *, const uint64_t *, uint64_t, uint64_t, mtb_result *, int32_t *, uint32_t *, uint64_t, uint64_t *) NULL_UNSUPPORTED
mtb_status mtb_index_write(const mtb_index *, const char *, int) NULL_UNSUPPORTED
mtb_status mtb_index_slice(mtb_index *, uint64_t, uint64_t, int, mtb_index **) NULL_UNSUPPORTED

}  // extern "C"
