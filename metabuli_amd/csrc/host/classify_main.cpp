/*
 * classify_main.cpp -- `mtb_classify`: the host loop of `metabuli classify`
 * (src/workflow/classify.cpp:39-199 -> Classifier::startClassify,
 * src/commons/Classifier.cpp:44-164) over the C ABI, with the reference's
 * positional arguments, the subset of its flags that the hot path reads, and
 * its two TSV outputs (Reporter.cpp:35-80, 115-193).  Everything that computes
 * runs on the GPU through libmtb.so; this file only parses reads, batches them
 * and formats results -- as a three-stage pipeline: a
 * reader thread parses batch k+1 (fastx.h: block-parallel FASTA/FASTQ/gzip
 * parser, flat buffers), the main thread runs batch k on the GPU, a writer
 * thread formats batch k-1 with --threads workers and appends it in order.
 *
 *   mtb_classify [flags] <FASTA/Q> [<FASTA/Q mate>] <DBDIR> <OUTDIR> <JobID>
 *   flags read on the path: --seq-mode 1|2|3  --min-score F  --min-sp-score F  --min-cons-cnt N  --min-cons-cnt-euk N
 *          --tie-ratio F  --taxonomy-path DIR  --syncmer 0|1  --smer-len N  --kmer-format 1|2  --accession-level 0|1|2
 *          --lineage 0|1  --threads N (host parsing / formatting)
 *   accepted for command-line compatibility, no effect here (one warning each): --max-ram (batches are bounded by HBM inside
 *          the library; the host batch is --max-reads), --match-per-kmer (exact-size retry), --hamming-margin and --max-gap (stored
 *          but never read by the reference's classify path either), --mask 0, --mask-prob, --validate-input, --validate-db,
 *          --print-log, -v   (LocalParameters.cpp:631-654).  Refused: --mask 1, --reduced-aa 1 (they change the answers)
 *   filter mode (`metabuli filter`, src/workflow/filter.cpp:5-45, QueryFilter.cpp:75-186): --filter 1 [--print-mode 1|2] <FASTA/Q> [<mate>] <DBDIR>
 *          classifies with the filter command's defaults (--min-score 0.5 unless given) and writes, next to the input,
 *          <base>_filtered.fna (reads NOT classified = not contamination), with --print-mode 2 also <base>_removed.fna (classified
 *          reads), and <base>_classifications.tsv / <base>_report.tsv; <base> = LocalUtil::getQueryBaseName (LocalUtil.cpp:5-20).
 *          (In the reference snapshot the match loop of filterReads is stubbed out, QueryFilter.cpp:172-175, so it keeps every
 *          read; this implements the documented behaviour of the command.)
 *   own flags: --max-reads N (host batch)  --gpu-workers W (default 1; W batches inside the GPU stage at once, each on contexts of its own, so that
 *          one batch's PCIe transfers overlap another's kernels -- measured on one MI355X: no gain, the kernels of two batches slow each other
 *          by what the overlap wins, profiles/r03_notes.md)  --pack-reads 0|1 (default 1: the reads cross PCIe as 2-bit codes + invalid mask out of
 *          pinned buffers, mtb_classify_batch_packed; 0: as text)  --async-results 1 (default 0; one device, packed reads: a batch's results are copied out
 *          while the next batch computes -- mtb_classify_batch_packed_async -- and reach the formatter one batch later)  --partitioned 1 (with --devices: engine d holds value range d of the database -- for
 *          databases larger than one GPU's HBM; metamers and matches are exchanged between the GPUs)
 *          --device N | --devices 0,1,... (one engine per GPU: every host batch is cut into
 *          contiguous read ranges, one per device, classified concurrently, results concatenated in input order and
 *          the per-taxon counts summed -- reads are independent, Classifier.cpp:187-203)
 */
#include <cstdio>
#include <sys/mman.h>
#include <unistd.h>
#include <cstring>
#include <cmath>
#include <fstream>
#include <algorithm>
#include <functional>
#include <memory>
#include <iostream>
#include <sstream>
#include <unordered_map>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <thread>

#include "../../../include/mtb.hpp"
#include "../mtb_core.h"         /* mtb_build_tables: the extractor's base classes, for the 2-bit packing of the reads */
#include "fastx.h"
#include "format.h"

namespace {

/* one batch travelling through the pipeline */
struct Job {
    mtbhost::FlatBatch r1, r2;
    mtbhost::PodVec<mtb_result> res; mtbhost::PodVec<int32_t> tt; mtbhost::PodVec<uint32_t> tc;      /* never zero-filled */
    bool last = false;
    /* what crosses PCIe lives in pinned host memory (mtb_host_alloc): 2-bit reads up, results and taxID:count lists down */
    void pin() {
        for (mtbhost::FlatBatch *b : {&r1, &r2}) { b->packed2.set_allocator(mtb_host_alloc, mtb_host_free); b->nmask.set_allocator(mtb_host_alloc, mtb_host_free); b->lens.set_allocator(mtb_host_alloc, mtb_host_free); }
        res.set_allocator(mtb_host_alloc, mtb_host_free); tt.set_allocator(mtb_host_alloc, mtb_host_free); tc.set_allocator(mtb_host_alloc, mtb_host_free);
    }
    void reset() { r1.clear(); r2.clear(); res.clear(); tt.clear(); tc.clear(); last = false; }
    /* the pinned buffers at the size a batch of max_reads reads of ~est_len bases will need, allocated up front (while the database
     * loads): pinning half a GB costs 0.1 - 0.3 s, and a buffer that grows on its first use does that inside the parse or the GPU stage */
    void prealloc(size_t max_reads, bool paired, size_t est_len) {
        const size_t groups = max_reads * ((est_len + 7) / 8 + 1);
        for (mtbhost::FlatBatch *b : {&r1, &r2}) {
            if (b == &r2 && !paired) continue;
            b->packed2.reserve(groups * 2); b->nmask.reserve(groups); b->lens.reserve(max_reads + 16);
        }
        res.reserve(max_reads + max_reads / 8 + 16); tt.reserve(7 * max_reads + 4096); tc.reserve(7 * max_reads + 4096);
    }
};

/* bounded single-producer / single-consumer hand-over */
template <class T> class Channel {
public:
    explicit Channel(size_t cap) : cap_(cap) {}
    void put(std::unique_ptr<T> v) { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return q_.size() < cap_; }); q_.push_back(std::move(v)); cv_.notify_all(); }
    std::unique_ptr<T> get() { std::unique_lock<std::mutex> l(m_); cv_.wait(l, [&] { return !q_.empty(); }); auto v = std::move(q_.front()); q_.erase(q_.begin()); cv_.notify_all(); return v; }
    std::unique_ptr<T> try_get() { std::unique_lock<std::mutex> l(m_); if (q_.empty()) return nullptr; auto v = std::move(q_.front()); q_.erase(q_.begin()); cv_.notify_all(); return v; }
private:
    std::mutex m_; std::condition_variable cv_; std::vector<std::unique_ptr<T>> q_; size_t cap_;
};

/* TaxonomyWrapper::taxLineage2 (TaxonomyWrapper.cpp:431-454) with findShortRank2 (:423-429, table TaxonomyWrapper.h:9-26):
 * "<short rank>_<name>" of the node and its ancestors below the root, root side first, ';' separated */
const char *short_rank(const char *rank) {
    static const std::map<std::string, const char *> M = {
        {"subspecies", "ss"}, {"species", "s"}, {"subgenus", "sg"}, {"genus", "g"}, {"subfamily", "sf"}, {"family", "f"},
        {"suborder", "so"}, {"order", "o"}, {"subclass", "sc"}, {"class", "c"}, {"subphylum", "sp"}, {"phylum", "p"},
        {"subkingdom", "sk"}, {"kingdom", "k"}, {"superkingdom", "d"}, {"domain", "d"}, {"realm", "r"}};
    auto it = M.find(rank);
    return it == M.end() ? "-" : it->second;
}
void append_lineage(const mtb_index *ix, int32_t taxid, std::string &out) {
    int32_t chain[128]; int n = 0;
    int32_t node = taxid;
    do {                                        /* the node itself, then its ancestors; the root ends the walk unlisted */
        if (n < 128) chain[n++] = node;
        int32_t p = mtb_tax_parent(ix, node);
        if (p < 0) break;
        node = p;
    } while (mtb_tax_parent(ix, node) != node);
    for (int i = n - 1; i >= 0; i--) {
        out += short_rank(mtb_tax_rank(ix, chain[i])); out += '_'; out += mtb_tax_name(ix, chain[i]);
        if (i > 0) out += ';';
    }
}

/* Reporter::writeReadClassification (Reporter.cpp:35-80), one line per read, reads [lo, hi) of the job.  The score is
 * printed like an ostream prints a float (6 significant digits; format.h).  The piece's reads per classification (Classifier.cpp:201-203)
 * are counted on the way: `counts` gets (taxon, reads) pairs, merged by the caller. */
void format_reads(const Job &j, size_t lo, size_t hi, const mtb_index *ix, bool lineage, std::string &out, std::vector<std::pair<int32_t, uint32_t>> *counts = nullptr) {
    mtbhost::RowBuf o(out, (hi - lo) * 96);
    std::unordered_map<int32_t, uint32_t> cnt;
    int32_t last_cls = -1; uint32_t last_run = 0;             /* reads of a sample often repeat the taxon of their neighbour */
    const char *names = j.r1.names.data();
    std::string lin;
    for (size_t i = lo; i < hi; i++) {
        const mtb_result &r = j.res[i];
        if (counts) { if (r.classification == last_cls) last_run++; else { if (last_run) cnt[last_cls] += last_run; last_cls = r.classification; last_run = 1; } }
        const size_t nlen = (size_t)(j.r1.name_offs[i + 1] - j.r1.name_offs[i]);
        o.need(nlen + 160 + (size_t)r.n_taxcnt * 24);
        o.ch(r.is_classified ? '1' : '0'); o.ch('\t');
        o.bytes(names + j.r1.name_offs[i], nlen); o.ch('\t');
        o.p = mtbhost::put_int(o.p, mtb_tax_original_id(ix, r.classification)); o.ch('\t');
        o.p = mtbhost::put_int(o.p, (long long)r.query_length + r.query_length2); o.ch('\t');
        o.p = mtbhost::put_float_g6(o.p, r.score);
        o.ch('\t');
        if (r.is_classified) {
            o.cstr(mtb_tax_rank(ix, r.classification)); o.ch('\t');
            if (lineage) { lin.clear(); append_lineage(ix, r.classification, lin); o.need(lin.size() + 64 + (size_t)r.n_taxcnt * 24); o.bytes(lin.data(), lin.size()); o.ch('\t'); }
            for (uint32_t k = 0; k < r.n_taxcnt; k++) {
                o.p = mtbhost::put_int(o.p, mtb_tax_original_id(ix, j.tt[r.taxcnt_off + k])); o.ch(':');
                o.p = mtbhost::put_uint(o.p, j.tc[r.taxcnt_off + k]); o.ch(' ');
            }
            o.ch('\n');
        } else { if (lineage) o.bytes("-\t-\t-\t\n", 7); else o.bytes("-\t-\t\n", 5); }
    }
    o.finish();
    if (counts) {
        if (last_run) cnt[last_cls] += last_run;
        counts->assign(cnt.begin(), cnt.end());
    }
}

/* The rows of a batch (one string per formatting piece) appended to `out`, in order.  One buffered writer: 3.6 GB of rows per 60 M reads
 * took 0.3 - 0.5 s this way on the box's local disk (page cache).  Mapping the grown file and copying the pieces into the mapping in
 * parallel -- what this function tried before -- pays a page fault per 4 KiB of output and measured 1.7 - 2.5 s for the same rows
 * (round 4, once the file was opened readable and the mapping stopped failing); parallel pwrite()s serialise on the inode lock. */
void append_parts(FILE *out, const std::vector<std::string> &parts, mtbhost::WorkerPool &, std::string &err) {
    for (auto &p : parts) if (!p.empty() && fwrite(p.data(), 1, p.size(), out) != p.size()) err = "short write";
}

/* QueryFilter::printFilteredReads (QueryFilter.cpp:102-118): ">name\nsequence\n" of the reads [lo, hi) whose is_classified flag equals `classified` */
void format_fasta(const mtbhost::FlatBatch &r, const mtbhost::PodVec<mtb_result> &res, size_t lo, size_t hi, bool classified, std::string &out) {
    out.clear();
    for (size_t i = lo; i < hi; i++) {
        if ((res[i].is_classified != 0) != classified) continue;
        out += '>'; out.append(r.names.data() + r.name_offs[i], r.names.data() + r.name_offs[i + 1]); out += '\n';
        out.append(r.bases.data() + r.offs[i], r.bases.data() + r.offs[i + 1]); out += '\n';
    }
}

/* LocalUtil::getQueryBaseName (LocalUtil.cpp:5-20): the path without its last extension (without the last two for .gz) */
std::string query_base_name(const std::string &path) {
    std::vector<std::string> parts;
    size_t a = 0;
    for (;;) { size_t b = path.find('.', a); if (b == std::string::npos) { parts.push_back(path.substr(a)); break; } parts.push_back(path.substr(a, b - a)); a = b + 1; }
    const size_t drop = path.size() >= 3 && path.compare(path.size() - 3, 3, ".gz") == 0 ? 2 : 1;
    std::string base;
    for (size_t i = 0; i + drop < parts.size(); i++) { if (i) base += '.'; base += parts[i]; }
    return base;
}

/* Reporter::writeReportFile / writeReport / kronaReport (Reporter.cpp:86-193): clade counts as NcbiTaxonomy::getCladeCounts
 * builds them (every counted taxon adds its count to itself and to all of its ancestors; a node's children are the
 * taxonomy's children), the report as a depth-first walk with the children ordered by clade count, descending -- the
 * reference's order among equal counts is unspecified (unstable sort); here ties go by taxon id.
 * Printed ids are the original ones (getOriginalTaxID, Reporter.cpp:181). */
struct CladeTable {
    std::unordered_map<int, unsigned> clade, own;
    const mtb_index *ix;
    CladeTable(const std::map<int, unsigned> &taxCounts, const mtb_index *ix_) : ix(ix_) {
        for (const auto &kv : taxCounts) {
            own[kv.first] = kv.second;
            clade[kv.first] += kv.second;
            if (kv.first == 0) continue;
            int t = kv.first;
            for (int guard = 0; guard < 1000; guard++) {
                int p = mtb_tax_parent(ix, t);
                if (p < 0 || p == t) break;
                clade[p] += kv.second;
                t = p;
            }
        }
    }
    unsigned clade_of(int t) const { auto it = clade.find(t); return it == clade.end() ? 0u : it->second; }
    unsigned own_of(int t) const { auto it = own.find(t); return it == own.end() ? 0u : it->second; }
    std::vector<int> children(int t) const {                 /* counted children, by clade count */
        std::vector<int> ch;
        const int n = mtb_tax_num_children(ix, t);
        for (int k = 0; k < n; k++) { int c = mtb_tax_child(ix, t, k); if (clade_of(c)) ch.push_back(c); }
        std::sort(ch.begin(), ch.end(), [&](int a, int b) { unsigned ca = clade_of(a), cb = clade_of(b); return ca != cb ? ca > cb : a < b; });
        return ch;
    }
};

void write_report(FILE *fp, const CladeTable &ct, unsigned long total) {
    fprintf(fp, "#clade_proportion\tclade_count\ttaxon_count\trank\ttaxID\tname\n");
    if (ct.clade_of(0) > 0)
        fprintf(fp, "%.4f\t%i\t%i\tno rank\t0\tunclassified\n", 100 * ct.clade_of(0) / double(total), (int)ct.clade_of(0), (int)ct.own_of(0));
    std::function<void(int, int)> rec = [&](int t, int depth) {
        const unsigned c = ct.clade_of(t);
        if (c == 0) return;
        fprintf(fp, "%.4f\t%i\t%i\t%s\t%i\t%s%s\n", 100 * c / double(total), (int)c, (int)ct.own_of(t), mtb_tax_rank(ct.ix, t),
                mtb_tax_original_id(ct.ix, t), std::string(2 * (size_t)depth, ' ').c_str(), mtb_tax_name(ct.ix, t));
        for (int ch : ct.children(t)) rec(ch, depth + 1);
    };
    rec(1, 0);
}

std::string escape_attribute(const char *s) {
    std::string o;
    for (; *s; s++) switch (*s) {
        case '&': o += "&amp;"; break; case '"': o += "&quot;"; break; case '\'': o += "&apos;"; break;
        case '<': o += "&lt;"; break; case '>': o += "&gt;"; break; default: o += *s;
    }
    return o;
}
/* <JobID>_krona.html (Reporter.cpp:143-158).  The <node> tree is kronaReport's (:86-113).  The HTML prelude of the reference
 * is MMseqs2's generated resource krona_prelude_html, absent from the snapshot: a minimal prelude with the same
 * <krona> data island stands in, so the file opens in KronaTools' importer / any XML reader but not as the interactive chart. */
void write_krona(FILE *fp, const CladeTable &ct, unsigned long total) {
    fputs("<!DOCTYPE html><html><head><meta charset=\"utf-8\"/><title>Krona</title></head><body><div style=\"display:none\">"
          "<krona collapse=\"false\" key=\"true\"><attributes magnitude=\"magnitude\"><attribute display=\"Count\">magnitude</attribute></attributes>", fp);
    fprintf(fp, "<node name=\"all\"><magnitude><val>%zu</val></magnitude>", (size_t)total);
    if (ct.clade_of(0) > 0) fprintf(fp, "<node name=\"unclassified\"><magnitude><val>%d</val></magnitude></node>", (int)ct.clade_of(0));
    std::function<void(int)> rec = [&](int t) {
        const unsigned c = ct.clade_of(t);
        if (c == 0) return;
        fprintf(fp, "<node name=\"%s\"><magnitude><val>%d</val></magnitude>", escape_attribute(mtb_tax_name(ct.ix, t)).c_str(), (int)c);
        for (int ch : ct.children(t)) rec(ch);
        fputs("</node>", fp);
    };
    rec(1);
    fputs("</node></krona></div></body></html>", fp);
}

} // namespace

int main(int argc, char **argv) {
    mtb_params par; mtb_default_params(&par);
    std::string taxdir; std::vector<int> devices(1, 0); size_t max_reads = 2000000;
    int threads = (int)std::max(1u, std::min(128u, std::thread::hardware_concurrency()));
    bool lineage = false, filter = false, min_score_given = false, partitioned = false, pack = true, async_results = false; int print_mode = 1, gpu_workers = 1;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return std::string(argv[++i]); };
        if (a == "--seq-mode") par.seq_mode = atoi(val().c_str());
        else if (a == "--min-score") { par.min_score = (float)atof(val().c_str()); min_score_given = true; }
        else if (a == "--filter") filter = atoi(val().c_str()) != 0;
        else if (a == "--print-mode") print_mode = atoi(val().c_str());
        else if (a == "--min-sp-score") par.min_sp_score = (float)atof(val().c_str());
        else if (a == "--min-cons-cnt") par.min_cons_cnt = atoi(val().c_str());
        else if (a == "--min-cons-cnt-euk") par.min_cons_cnt_euk = atoi(val().c_str());
        else if (a == "--tie-ratio") par.tie_ratio = (float)atof(val().c_str());
        else if (a == "--accession-level") par.accession_level = atoi(val().c_str());
        else if (a == "--kmer-format") par.kmer_format = atoi(val().c_str());
        else if (a == "--taxonomy-path") taxdir = val();
        else if (a == "--syncmer") par.syncmer = atoi(val().c_str());
        else if (a == "--smer-len") par.smer_len = atoi(val().c_str());
        else if (a == "--max-reads") max_reads = (size_t)atoll(val().c_str());
        else if (a == "--device") { devices.assign(1, atoi(val().c_str())); }
        else if (a == "--partitioned") partitioned = atoi(val().c_str()) != 0;
        else if (a == "--pack-reads") pack = atoi(val().c_str()) != 0;
        else if (a == "--async-results") async_results = atoi(val().c_str()) != 0;
        else if (a == "--gpu-workers") gpu_workers = std::max(1, std::min(4, atoi(val().c_str())));
        else if (a == "--devices") { devices.clear(); std::stringstream ss(val()); std::string tok; while (std::getline(ss, tok, ',')) if (!tok.empty()) devices.push_back(atoi(tok.c_str())); }
        else if (a == "--reduced-aa") { if (atoi(val().c_str()) != 0) { fprintf(stderr, "mtb_classify: --reduced-aa 1 is not implemented\n"); return 1; } }
        else if (a == "--mask") {       /* tantan masking of the reads before extraction (KmerExtractor.cpp:308-314) changes the answers: refuse it rather than ignore it */
            if (atoi(val().c_str()) != 0) { fprintf(stderr, "mtb_classify: --mask 1 (low-complexity masking of the reads) is not implemented\n"); return 1; } }
        else if (a == "--max-ram" || a == "--match-per-kmer" || a == "--hamming-margin" || a == "--mask-prob" ||
                 a == "--validate-input" || a == "--validate-db" || a == "--print-log" || a == "-v" || a == "--max-gap") {
            std::string v = val();
            fprintf(stderr, "mtb_classify: %s %s accepted for compatibility with `metabuli classify`, it has no effect here\n", a.c_str(), v.c_str());
        }
        else if (a == "--threads") threads = std::max(1, atoi(val().c_str()));
        else if (a == "--lineage") lineage = atoi(val().c_str()) != 0;
        else if (a.rfind("--", 0) == 0) { fprintf(stderr, "mtb_classify: unknown flag %s\n", a.c_str()); return 1; }
        else pos.push_back(a);
    }
    const bool paired = par.seq_mode == 2;
    size_t need = (paired ? 5 : 4) - (filter ? 2 : 0);
    if (pos.size() != need) {
        fprintf(stderr, filter ? "usage: mtb_classify --filter 1 [flags] <FASTA/Q>%s <DBDIR>\n" : "usage: mtb_classify [flags] <FASTA/Q>%s <DBDIR> <OUTDIR> <JobID>\n", paired ? " <FASTA/Q>" : "");
        return 1;
    }
    if (filter && !min_score_given) par.min_score = 0.5f;     /* setFilterDefaults, filter.cpp:8 */
    if (partitioned) { pack = false; gpu_workers = 1; }        /* (the partitioned batch takes the text; its engines work on one batch together) */
    if (!pack || partitioned || devices.size() != 1) async_results = false;      /* (one engine per worker, packed reads: mtb_classify_batch_packed_async) */
    const std::string dbdir = pos[paired ? 2 : 1];
    /* classify: <OUTDIR>/<JobID>_*; filter: <base of the first input>_* (QueryFilter.cpp:75-93) */
    const std::string base1 = filter ? query_base_name(pos[0]) : std::string(), base2 = filter && paired ? query_base_name(pos[1]) : std::string();
    const std::string prefix = filter ? base1 : pos[paired ? 3 : 2] + "/" + pos[paired ? 4 : 3];
    try {
        auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
        const double t_start = now();
        if (devices.empty()) throw std::runtime_error("--devices: empty list");
        /* the parser starts before the database is opened: the first batches are parsed (and their pinned buffers allocated) while the
         * index streams into HBM, so the GPU stage finds work waiting when the open returns */
        double t_parse = 0, t_gpu = 0, t_write = 0, t_dev = 0, t_fmt = 0, t_app = 0;  /* busy time of the three stages; device time inside the GPU stage */
        Channel<Job> parsed(2), scored(2), idle((size_t)gpu_workers + 6);
        /* a few batches are in flight (parse / GPU workers / format); their buffers are recycled, so that after the first round no stage
         * touches fresh pages, and what crosses PCIe sits in pinned memory */
        const int n_jobs = gpu_workers + 5;
        std::thread job_maker([&, n_jobs] {      /* (the first batch can be parsed as soon as the first job's buffers are pinned; the rest follow while the database loads) */
            for (int k = 0; k < n_jobs; k++) {
                std::unique_ptr<Job> j(new Job());
                if (pack) { j->pin(); try { j->prealloc(max_reads, paired, 152); } catch (const std::exception &) { /* the buffers then grow on first use */ } }
                idle.put(std::move(j));
            }
        });
        /* the output files are opened while the database loads: a job that is run again truncates GBs of rows of its previous run, which
         * took several tenths of a second between the open and the first batch */
        FILE *out = nullptr, *flt[2] = {nullptr, nullptr}, *rmv[2] = {nullptr, nullptr};
        std::string out_err;
        std::thread out_opener([&] {
            auto open_w = [&](const std::string &p) { FILE *f = fopen(p.c_str(), "w"); if (!f && out_err.empty()) out_err = "cannot write " + p; return f; };
            out = open_w(prefix + "_classifications.tsv");
            if (filter) {
                flt[0] = open_w(base1 + "_filtered.fna"); if (paired) flt[1] = open_w(base2 + "_filtered.fna");
                if (print_mode == 2) { rmv[0] = open_w(base1 + "_removed.fna"); if (paired) rmv[1] = open_w(base2 + "_removed.fna"); }
            }
        });
        mtbhost::WorkerPool parse_pool(threads), format_pool(threads);
        mtbhost::PackTable pack_table;
        { static mtb_tables tabs; mtb_build_tables(&tabs); for (int c = 0; c < 256; c++) pack_table.code[c] = tabs.base[c] < 4 ? tabs.base[c] : 0xFF; }
        std::string reader_err, writer_err;
        /* stage 1: parse */
        std::thread reader([&] {
            try {
                mtbhost::FastxReader r1(pos[0], threads, 64u << 20, &parse_pool);
                std::unique_ptr<mtbhost::FastxReader> r2;
                if (paired) r2.reset(new mtbhost::FastxReader(pos[1], threads, 64u << 20, &parse_pool));
                if (pack) { r1.set_pack(&pack_table, filter); if (r2) r2->set_pack(&pack_table, filter); }      /* the text is only kept for the filter command's FASTA output */
                for (;;) {
                    std::unique_ptr<Job> j = idle.get();
                    j->reset();
                    const double t0 = now();
                    r1.next_batch(max_reads, j->r1);
                    if (paired) { r2->next_batch(j->r1.size(), j->r2); if (j->r2.size() != j->r1.size()) throw std::runtime_error("mate file is shorter"); }
                    j->last = j->r1.size() == 0;
                    t_parse += now() - t0;
                    bool last = j->last;
                    parsed.put(std::move(j));
                    if (last) break;
                }
            } catch (const std::exception &e) { reader_err = e.what(); std::unique_ptr<Job> j(new Job()); j->last = true; parsed.put(std::move(j)); }
        });
        /* one engine (context + resident copy of the index) per GPU; db.parameters overrides the flags (common.cpp:88-133) */
        std::vector<std::unique_ptr<mtb::Engine>> engs;
        std::vector<uint64_t> bounds(devices.size(), 0);
        try {
        if (partitioned) mtb::check(mtb_index_part_bounds(dbdir.c_str(), (uint32_t)devices.size(), bounds.data()));
        for (size_t d = 0; d < devices.size(); d++) {
            mtb_params pd = par;
            /* --partitioned 1: engine d holds range d of the database (databases larger than one HBM) */
            if (partitioned) engs.emplace_back(new mtb::Engine(devices[d], dbdir, taxdir, d == 0 ? par : pd, (uint32_t)d, (uint32_t)devices.size()));
            else if (d == 0) {
                /* while the database streams into HBM on this thread, a helper grows the workspace of the first batches (tens of GB of
                 * hipMalloc: several hundred milliseconds that the first batch used to wait for) */
                engs.emplace_back(new mtb::Engine(devices[d]));
                mtb_params guess = par;
                mtb_db_parameters(dbdir.c_str(), &guess);
                std::thread reserve([&, guess] { if (par.seq_mode != 3) mtb_ctx_reserve(engs[0]->ctx, &guess, max_reads, (uint64_t)max_reads * 152u * (paired ? 2u : 1u)); });
                try { engs[0]->open(dbdir, taxdir, par); } catch (...) { reserve.join(); throw; }
                reserve.join();
            }
            else engs.emplace_back(nullptr);                                         /* cloned below, all destinations at once */
        }
        if (!partitioned && devices.size() > 1) {
            /* the files are read and decoded once: the other GPUs get peer copies (mtb_index_clone) -- ALL AT ONCE, one host thread per
             * destination: every destination pulls over its own xGMI link to GPU 0 (the fabric is point-to-point: seven copies issued one
             * after another took seven times one copy where they can share the source's seven links), into its own HBM */
            const double tc0 = now();
            std::vector<std::string> cerr_(devices.size());
            std::vector<std::thread> cl;
            for (size_t d = 1; d < devices.size(); d++)
                cl.emplace_back([&, d] { try { engs[d].reset(new mtb::Engine(devices[d], *engs[0])); } catch (const std::exception &e) { cerr_[d] = e.what(); if (cerr_[d].empty()) cerr_[d] = "clone failed"; } });
            for (auto &t : cl) t.join();
            for (size_t d = 1; d < devices.size(); d++) if (!cerr_[d].empty()) throw std::runtime_error("engine " + std::to_string(d) + ": " + cerr_[d]);
            const double tc = now() - tc0;
            int32_t depth = 0, pk = 0, sealed = 0;
            (void)mtb_index_state(engs[0]->index, &depth, &pk, &sealed);
            const double gb = ((double)mtb_index_num_targets(engs[0]->index) * (sealed ? 8.0 : 12.0) + (depth ? 4.0 * pow(21.0, depth) : 0.0)) / 1e9;
            fprintf(stderr, "mtb_classify: resident index cloned to %zu more device(s) concurrently in %.2f s (%.1f GB each, %.1f GB/s aggregate)\n",
                    devices.size() - 1, tc, gb, gb * (double)(devices.size() - 1) / std::max(tc, 1e-9));
        }
        } catch (const std::exception &e) {                   /* (the parser thread is running: leave at once) */
            fprintf(stderr, "mtb_classify: %s\n", e.what()); fflush(stderr); _exit(1);
        }
        mtb::Engine &eng = *engs[0];                          /* taxonomy services for formatting */
        const size_t ND = engs.size();
        const double t_open = now() - t_start;
        {   uint64_t os4[4] = {0, 0, 0, 0}; int32_t depth = 0, pk = 0, sealed = 0;
            if (mtb_index_open_stats(eng.index, os4) == MTB_OK && mtb_index_state(eng.index, &depth, &pk, &sealed) == MTB_OK)
                fprintf(stderr, "mtb_classify: database of %llu targets open in %.2f s: %llu chunks of %llu 16-bit words, peak device memory during the open %.2f GiB, directory depth %d%s\n",
                        (unsigned long long)mtb_index_num_targets(eng.index), t_open, (unsigned long long)os4[0], (unsigned long long)os4[1], (double)os4[2] / 1073741824.0, depth,
                        os4[3] ? ", packed on load (8-byte words, info folded in)" : "");
        }

        out_opener.join();
        if (!out_err.empty()) { fprintf(stderr, "mtb_classify: %s\n", out_err.c_str()); fflush(stderr); _exit(1); }      /* (the parser thread is running: leave at once) */
        fputs(lineage ? "#is_classified\tname\ttaxID\tquery_length\tscore\trank\tlineage\ttaxID:match_count\n"
                      : "#is_classified\tname\ttaxID\tquery_length\tscore\trank\ttaxID:match_count\n", out);
        /* stage 3: format + append, per-taxon read counts (Classifier.cpp:201-203) */
        std::vector<uint64_t> tax_counts((size_t)mtb_tax_max_id(eng.index) + 2, 0);
        unsigned long total = 0;
        std::thread writer([&] {
            const size_t NP = (size_t)threads * 2;                       /* pieces: a few per worker, reads differ in their row length */
            std::vector<std::string> parts(NP);
            std::vector<std::vector<std::pair<int32_t, uint32_t>>> piece_counts(NP);
            for (;;) {
                std::unique_ptr<Job> j = scored.get();
                if (j->last) break;
                const double t0 = now();
                const size_t n = j->r1.size();
                format_pool.run(NP, [&](size_t t) { format_reads(*j, n * t / NP, n * (t + 1) / NP, eng.index, lineage, parts[t], &piece_counts[t]); });
                try { format_pool.rethrow(); } catch (const std::exception &e) { if (writer_err.empty()) writer_err = std::string("formatting failed: ") + e.what(); }
                if (!writer_err.empty()) { idle.put(std::move(j)); continue; }      /* nothing is written after a failure; the buffers keep the pipeline draining */
                const double t1 = now();
                append_parts(out, parts, format_pool, writer_err);
                t_fmt += t1 - t0; t_app += now() - t1;
                for (int mate = 0; mate < 2; mate++) for (int cls = 0; cls < 2; cls++) {
                    FILE *f = cls ? rmv[mate] : flt[mate];
                    if (!f) continue;
                    const mtbhost::FlatBatch &rb = mate ? j->r2 : j->r1;
                    format_pool.run(NP, [&](size_t t) { format_fasta(rb, j->res, n * t / NP, n * (t + 1) / NP, cls != 0, parts[t]); });
                    for (auto &p : parts) if (fwrite(p.data(), 1, p.size(), f) != p.size()) writer_err = "short write";
                }
                for (auto &pc : piece_counts) for (auto &kv : pc) if (kv.first >= 0 && (size_t)kv.first < tax_counts.size()) tax_counts[(size_t)kv.first] += kv.second;
                total += n;
                t_write += now() - t0;
                std::cout << "The number of processed sequences: " << total << std::endl;
                idle.put(std::move(j));                                  /* its buffers serve the next batch */
            }
        });
        /* stage 2: the GPUs.  A host batch is cut into ND contiguous read ranges; range d runs on engine d from its own host
         * thread (one thread per mtb_ctx); rows land at their places in j->res, the taxcnt lists are appended range by range.
         * --gpu-workers W (default 1): W batches are in this stage at once, each on contexts of its own (same resident index), so
         * that one batch's PCIe transfers and host-side bookkeeping overlap another's kernels; batches leave the stage in input order. */
        std::string gpu_err; std::mutex gpu_err_mu;
        struct Range { std::vector<uint64_t> offs, offs2; mtbhost::PodVec<int32_t> tt; mtbhost::PodVec<uint32_t> tc; uint64_t ntc = 0; std::string err; double dev_ms = 0; };
        const int W = gpu_workers;
        std::atomic<double> tc_per_read{6.0};
        std::vector<std::vector<mtb_ctx *>> wctx((size_t)W, std::vector<mtb_ctx *>(ND, nullptr));
        for (int w = 0; w < W; w++) for (size_t d = 0; d < ND; d++) { if (w == 0) wctx[0][d] = engs[d]->ctx; else if (mtb_ctx_create(devices[d], nullptr, &wctx[(size_t)w][d]) != MTB_OK) { fprintf(stderr, "mtb_classify: mtb: %s\n", mtb_last_error()); fflush(stderr); _exit(1); } }
        std::vector<double> w_busy((size_t)W, 0.0), w_dev((size_t)W, 0.0);
        auto process = [&](Job &job, int w) {
            Job *j = &job;
            const double t0 = now();
            const size_t n = j->r1.size();
            j->res.resize_uninit(n);
            std::vector<Range> rg(ND);
            /* 2-bit reads: the group (8 bases) at which every engine's read range starts in the packed arrays */
            std::vector<uint64_t> slot_lo(ND + 1, 0), slot2_lo(ND + 1, 0);
            if (pack && ND > 1) {
                size_t d = 1; uint64_t a1 = 0, a2 = 0;
                for (size_t i = 0; i < n && d <= ND; i++) {
                    while (d <= ND && i == n * d / ND) { slot_lo[d] = a1; slot2_lo[d] = a2; d++; }
                    a1 += (j->r1.lens[i] + 7u) >> 3; if (paired) a2 += (j->r2.lens[i] + 7u) >> 3;
                }
            }
            auto run = [&](size_t d) {
                Range &R = rg[d];
                mtb_ctx *cx = wctx[(size_t)w][d];
                const size_t lo = n * d / ND, hi = n * (d + 1) / ND, m = hi - lo;
                if (m == 0) return;
                uint64_t b0 = 0, c0 = 0;
                if (!pack) {
                    b0 = j->r1.offs[lo];
                    R.offs.resize(m + 1);
                    for (size_t i = 0; i <= m; i++) R.offs[i] = j->r1.offs[lo + i] - b0;
                    if (paired) { c0 = j->r2.offs[lo]; R.offs2.resize(m + 1); for (size_t i = 0; i <= m; i++) R.offs2[i] = j->r2.offs[lo + i] - c0; }
                }
                mtb_params pd = par;
                /* the taxID:count lists arrive packed: a few entries per read (the factor follows what the batches so far needed; a
                 * batch that needs more is redone with the exact size) -- these are pinned buffers, 24 entries per read cost the
                 * first batches of a run more in page pinning than their classification */
                size_t cap = (size_t)(tc_per_read.load() * (double)m) + 4096;
                if (ND == 1) { R.tt = std::move(j->tt); R.tc = std::move(j->tc); }       /* the job's own (pinned, recycled) buffers */
                for (;;) {
                    R.tt.resize_uninit(cap); R.tc.resize_uninit(cap);
                    mtb_status st = pack
                        ? (async_results ? mtb_classify_batch_packed_async : mtb_classify_batch_packed)(cx, engs[d]->index, &pd, j->r1.packed2.data() + 2 * slot_lo[d], j->r1.nmask.data() + slot_lo[d], j->r1.lens.data() + lo,
                                                    paired ? j->r2.packed2.data() + 2 * slot2_lo[d] : nullptr, paired ? j->r2.nmask.data() + slot2_lo[d] : nullptr,
                                                    paired ? j->r2.lens.data() + lo : nullptr, m, j->res.data() + lo, R.tt.data(), R.tc.data(), cap, &R.ntc)
                        : mtb_classify_batch(cx, engs[d]->index, &pd, j->r1.bases.data() + b0, R.offs.data(),
                                             paired ? j->r2.bases.data() + c0 : nullptr, paired ? R.offs2.data() : nullptr, m,
                                             j->res.data() + lo, R.tt.data(), R.tc.data(), cap, &R.ntc);
                    if (st == MTB_ERR_CAPACITY && R.ntc > cap) { cap = R.ntc + R.ntc / 8; continue; }
                    if (st != MTB_OK) R.err = mtb_last_error();
                    else { mtb_batch_stats bs; if (mtb_last_batch_stats(cx, &bs) == MTB_OK) R.dev_ms = bs.ms_total; }
                    if (st == MTB_OK && m) { const double need = 1.5 * (double)R.ntc / (double)m; double cur = tc_per_read.load(); while (need > cur && !tc_per_read.compare_exchange_weak(cur, need)) {} }
                    break;
                }
            };
            std::string err;
            if (partitioned) {
                /* every engine owns a value range: its share of the reads is extracted there, the sorted metamers travel to the range
                 * owners and the matches back (peer copies inside the library), rows come back in input order */
                std::vector<mtb_ctx *> cs(ND); std::vector<mtb_index *> is(ND);
                for (size_t d = 0; d < ND; d++) { cs[d] = engs[d]->ctx; is[d] = engs[d]->index; }
                Range &R = rg[0];
                mtb_params pd = par;
                size_t cap = 8 * n + 4096;
                for (;;) {
                    R.tt.resize_uninit(cap); R.tc.resize_uninit(cap);
                    mtb_status st = mtb_classify_batch_partitioned(cs.data(), is.data(), (uint32_t)ND, bounds.data(), &pd, j->r1.bases.data(), j->r1.offs.data(),
                                                                   paired ? j->r2.bases.data() : nullptr, paired ? j->r2.offs.data() : nullptr, n,
                                                                   j->res.data(), R.tt.data(), R.tc.data(), cap, &R.ntc);
                    if (st == MTB_ERR_CAPACITY && R.ntc > cap) { cap = R.ntc; continue; }
                    if (st != MTB_OK) R.err = mtb_last_error();
                    break;
                }
                if (R.err.empty()) { j->tt = std::move(R.tt); j->tc = std::move(R.tc); }
                err = R.err;
            } else {
                if (ND == 1) run(0);
                else { std::vector<std::thread> th; for (size_t d = 0; d < ND; d++) th.emplace_back(run, d); for (auto &x : th) x.join(); }
                uint64_t tot_tc = 0;
                for (size_t d = 0; d < ND; d++) { if (!rg[d].err.empty() && err.empty()) err = rg[d].err; tot_tc += rg[d].ntc; }
                if (err.empty()) {
                    if (tot_tc >= (1ull << 32)) err = "taxcnt lists of one host batch exceed 2^32 entries; lower --max-reads";
                    else if (ND == 1) { j->tt = std::move(rg[0].tt); j->tc = std::move(rg[0].tc); }
                    else {
                        j->tt.resize_uninit(tot_tc); j->tc.resize_uninit(tot_tc);
                        uint64_t base = 0;
                        for (size_t d = 0; d < ND; d++) {
                            const size_t lo = n * d / ND, hi = n * (d + 1) / ND;
                            if (rg[d].ntc) { memcpy(j->tt.data() + base, rg[d].tt.data(), rg[d].ntc * 4); memcpy(j->tc.data() + base, rg[d].tc.data(), rg[d].ntc * 4); }
                            for (size_t i = lo; i < hi; i++) j->res[i].taxcnt_off += (uint32_t)base;
                            base += rg[d].ntc;
                        }
                    }
                }
                double mx = 0; for (size_t d = 0; d < ND; d++) mx = std::max(mx, rg[d].dev_ms);
                w_dev[(size_t)w] += mx * 1e-3;
            }
            w_busy[(size_t)w] += now() - t0;
            if (!err.empty()) { std::lock_guard<std::mutex> l(gpu_err_mu); if (gpu_err.empty()) gpu_err = err; }
        };
        auto failed = [&] { std::lock_guard<std::mutex> l(gpu_err_mu); return !gpu_err.empty(); };
        /* worker w takes batches w, w + W, ...; the collector hands them on in that order */
        std::vector<std::unique_ptr<Channel<Job>>> win, wout;
        for (int w = 0; w < W; w++) { win.emplace_back(new Channel<Job>(2)); wout.emplace_back(new Channel<Job>(1)); }      /* (two in: the batch behind the one in work must already be there for its prefetch) */
        std::vector<std::thread> workers;
        /* the batch behind the one in work starts crossing PCIe before that one's kernels are launched (mtb_prefetch_batch_packed: copy
         * stream + second input buffer set inside the context), so its upload is hidden behind them */
        const bool can_prefetch = pack && !partitioned && ND == 1;
        for (int w = 0; w < W; w++) workers.emplace_back([&, w] {
            std::unique_ptr<Job> j = win[(size_t)w]->get();
            std::unique_ptr<Job> held;           /* --async-results 1: the batch whose results are still being copied out */
            for (;;) {
                const bool last = j->last;
                std::unique_ptr<Job> nxt;
                if (async_results) {
                    const bool run_it = !last && !failed();
                    if (run_it) {
                        nxt = win[(size_t)w]->try_get();
                        if (nxt && !nxt->last && can_prefetch && nxt->r1.size()) {
                            mtb_params pd = par;
                            (void)mtb_prefetch_batch_packed(wctx[(size_t)w][0], &pd, nxt->r1.packed2.data(), nxt->r1.nmask.data(), nxt->r1.lens.data(), paired ? nxt->r2.packed2.data() : nullptr,
                                                            paired ? nxt->r2.nmask.data() : nullptr, paired ? nxt->r2.lens.data() : nullptr, nxt->r1.size());
                        }
                        process(*j, w);          /* returns with this batch's copies queued; the previous batch's are complete (whatever the status) */
                    } else if (mtb_ctx_wait_results(wctx[(size_t)w][0]) != MTB_OK) { std::lock_guard<std::mutex> l(gpu_err_mu); if (gpu_err.empty()) gpu_err = mtb_last_error(); }
                    if (held) wout[(size_t)w]->put(std::move(held));
                    if (last) { wout[(size_t)w]->put(std::move(j)); break; }
                    if (run_it && !failed()) held = std::move(j);
                    else wout[(size_t)w]->put(std::move(j));       /* nothing is in flight for this batch (the collector recycles it after a failure) */
                    j = nxt ? std::move(nxt) : win[(size_t)w]->get();
                    continue;
                }
                if (!last) {
                    nxt = win[(size_t)w]->try_get();
                    if (nxt && !nxt->last && can_prefetch && nxt->r1.size() && !failed()) {
                        mtb_params pd = par;
                        const mtb_status ps = mtb_prefetch_batch_packed(wctx[(size_t)w][0], &pd, nxt->r1.packed2.data(), nxt->r1.nmask.data(), nxt->r1.lens.data(),
                                                                        paired ? nxt->r2.packed2.data() : nullptr, paired ? nxt->r2.nmask.data() : nullptr,
                                                                        paired ? nxt->r2.lens.data() : nullptr, nxt->r1.size());
                        (void)ps;               /* a failed prefetch only means that the batch is uploaded by its own call */
                    }
                    if (!failed()) process(*j, w);
                }
                wout[(size_t)w]->put(std::move(j));
                if (last) break;
                j = nxt ? std::move(nxt) : win[(size_t)w]->get();
            }
        });
        std::thread collector([&] {
            for (size_t k = 0;; k++) {
                std::unique_ptr<Job> j = wout[k % (size_t)W]->get();
                if (j->last) { scored.put(std::move(j)); break; }
                if (failed()) idle.put(std::move(j));        /* after a failure nothing is written any more; the buffers keep the reader going until the input ends */
                else scored.put(std::move(j));
            }
        });
        const double t_stage0 = now();
        for (size_t k = 0;; k++) {
            std::unique_ptr<Job> j = parsed.get();
            const bool last = j->last;
            if (last) for (int w = 0; w < W; w++) if ((size_t)w != k % (size_t)W) { std::unique_ptr<Job> s2(new Job()); s2->last = true; win[(size_t)w]->put(std::move(s2)); }
            win[k % (size_t)W]->put(std::move(j));
            if (last) break;
        }
        for (auto &t : workers) t.join();
        collector.join();
        const double t_stage = now() - t_stage0;
        for (int w = 0; w < W; w++) { t_gpu = std::max(t_gpu, w_busy[(size_t)w]); t_dev += w_dev[(size_t)w]; }
        uint64_t pf_issued = 0, pf_used = 0;           /* uploads that crossed PCIe behind the previous batch's kernels (mtb_prefetch_batch_packed) */
        for (int w = 0; w < W; w++) { uint64_t a_ = 0, b_ = 0; if (mtb_ctx_prefetch_stats(wctx[(size_t)w][0], &a_, &b_) == MTB_OK) { pf_issued += a_; pf_used += b_; } }
        for (int w = 1; w < W; w++) for (size_t d = 0; d < ND; d++) mtb_ctx_destroy(wctx[(size_t)w][d]);
        job_maker.join(); reader.join(); writer.join();
        fclose(out);
        for (int k = 0; k < 2; k++) { if (flt[k]) fclose(flt[k]); if (rmv[k]) fclose(rmv[k]); }
        if (!reader_err.empty()) throw std::runtime_error(reader_err);
        if (!gpu_err.empty()) throw std::runtime_error("mtb: " + gpu_err);
        if (!writer_err.empty()) throw std::runtime_error(writer_err);
        std::map<int, unsigned> counts;
        for (size_t t = 0; t < tax_counts.size(); t++) if (tax_counts[t]) counts[(int)t] = (unsigned)tax_counts[t];
        CladeTable ct(counts, eng.index);
        FILE *fp = fopen((prefix + "_report.tsv").c_str(), "w");
        if (!fp) throw std::runtime_error("cannot write the report");
        write_report(fp, ct, total);
        fclose(fp);
        if (!filter) {                                        /* the filter command writes the default report only (QueryFilter.cpp:198) */
            fp = fopen((prefix + "_krona.html").c_str(), "w");
            if (!fp) throw std::runtime_error("cannot write the krona file");
            write_krona(fp, ct, total);
            fclose(fp);
        }
        fprintf(stderr, "mtb_classify: %lu reads in %.2f s on %zu GPU(s) (index open %.2f s; stage busy time: parse %.2f s, GPU incl. PCIe %.2f s busiest of %d worker(s) over %.2f s (device time of all batches %.2f s), format+write %.2f s (rows %.2f s, file %.2f s); %d host threads; %llu of %llu prefetched uploads used%s)\n",
                total, now() - t_start, ND, t_open, t_parse, t_gpu, W, t_stage, t_dev, t_write, t_fmt, t_app, threads, (unsigned long long)pf_used, (unsigned long long)pf_issued,
                async_results ? "; results copied out while the next batch runs" : "");
    } catch (const std::exception &e) {
        fprintf(stderr, "mtb_classify: %s\n", e.what());
        return 1;
    }
    return 0;
}
