/*
 * classify_main.cpp -- `mtb_classify`: the host loop of `metabuli classify`
 * (src/workflow/classify.cpp:39-199 -> Classifier::startClassify,
 * src/commons/Classifier.cpp:44-164) over the C ABI, with the reference's
 * positional arguments, the subset of its flags that the hot path reads, and
 * its two TSV outputs (Reporter.cpp:35-80, 115-193).  Everything that computes
 * runs on the GPU through libmtb.so; this file only parses reads, batches them
 * and formats results.
 *
 *   mtb_classify [flags] <FASTA/Q> [<FASTA/Q mate>] <DBDIR> <OUTDIR> <JobID>
 *   flags: --seq-mode 1|2|3  --min-score F  --min-sp-score F  --min-cons-cnt N
 *          --min-cons-cnt-euk N  --tie-ratio F  --taxonomy-path DIR
 *          --syncmer 0|1  --smer-len N  --kmer-format 1|2  --accession-level 0|1|2
 *          --max-reads N (batch size)  --device N
 */
#include <cstdio>
#include <cstring>
#include <fstream>
#include <algorithm>
#include <functional>
#include <memory>
#include <iostream>
#include <sstream>
#include <unordered_map>

#include "../../../include/mtb.hpp"

namespace {

/* FASTA / FASTQ records; name = header up to the first whitespace (kseq) */
struct SeqReader {
    std::ifstream in; std::string pending; bool fastq = false; bool started = false;
    explicit SeqReader(const std::string &p) : in(p) { if (!in) throw std::runtime_error("cannot open " + p); }
    bool next(std::string &name, std::string &seq) {
        std::string line;
        if (!started) {
            while (std::getline(in, line)) if (!line.empty()) break;
            if (line.empty()) return false;
            fastq = line[0] == '@'; pending = line; started = true;
        }
        if (pending.empty()) return false;
        size_t e = pending.find_first_of(" \t", 1);
        name = pending.substr(1, e == std::string::npos ? std::string::npos : e - 1);
        seq.clear(); pending.clear();
        if (fastq) {
            if (!std::getline(in, seq)) return false;
            if (!seq.empty() && seq.back() == '\r') seq.pop_back();
            std::getline(in, line); std::getline(in, line);       /* '+' and qualities */
            while (std::getline(in, line)) if (!line.empty()) { pending = line; break; }
        } else {
            while (std::getline(in, line)) {
                if (!line.empty() && line[0] == '>') { pending = line; break; }
                if (!line.empty() && line.back() == '\r') line.pop_back();
                seq += line;
            }
        }
        return true;
    }
};

/* Reporter::writeReadClassification (Reporter.cpp:35-80) */
void write_classifications(std::ostream &os, const std::vector<mtb::Query> &q, const mtb_index *ix, bool header) {
    if (header) os << "#is_classified\tname\ttaxID\tquery_length\tscore\trank\ttaxID:match_count\n";
    for (const auto &r : q) {
        if (r.isClassified) {
            os << r.isClassified << "\t" << r.name << "\t" << r.classification << "\t" << r.queryLength + r.queryLength2 << "\t"
               << r.score << "\t" << mtb_tax_rank(ix, r.classification) << "\t";
            for (const auto &kv : r.taxCnt) os << kv.first << ":" << kv.second << " ";
            os << "\n";
        } else {
            os << r.isClassified << "\t" << r.name << "\t" << r.classification << "\t" << r.queryLength + r.queryLength2 << "\t"
               << r.score << "\t-\t-\t\n";
        }
    }
}

/* Reporter::writeReportFile / writeReport (Reporter.cpp:115-193); clade counts as in
 * NcbiTaxonomy::getCladeCounts (every ancestor of a counted taxon accumulates it).
 * Children are ordered by clade count (descending), ties by taxid: the reference's
 * order among equal counts is unspecified (unstable sort). */
void write_report(FILE *fp, const std::map<int, unsigned> &taxCounts, const mtb_index *ix, unsigned long total) {
    std::unordered_map<int, unsigned> clade, own;
    std::unordered_map<int, std::vector<int>> children;
    for (const auto &kv : taxCounts) {
        own[kv.first] = kv.second;
        if (kv.first == 0) { clade[0] += kv.second; continue; }
        int t = kv.first;
        for (int guard = 0; guard < 1000; guard++) {
            bool fresh = clade.find(t) == clade.end();
            clade[t] += kv.second;
            int p = mtb_tax_parent(ix, t);
            if (p < 0 || p == t) break;
            if (fresh) children[p].push_back(t);
            t = p;
        }
    }
    fprintf(fp, "#clade_proportion\tclade_count\ttaxon_count\trank\ttaxID\tname\n");
    if (clade.count(0) && clade[0] > 0)
        fprintf(fp, "%.4f\t%i\t%i\tno rank\t0\tunclassified\n", 100 * clade[0] / double(total), (int)clade[0], (int)own[0]);
    std::function<void(int, int)> rec = [&](int t, int depth) {
        auto it = clade.find(t);
        if (it == clade.end() || it->second == 0) return;
        fprintf(fp, "%.4f\t%i\t%i\t%s\t%i\t%s%s\n", 100 * it->second / double(total), (int)it->second, (int)(own.count(t) ? own[t] : 0),
                mtb_tax_rank(ix, t), t, std::string(2 * (size_t)depth, ' ').c_str(), mtb_tax_name(ix, t));
        std::vector<int> ch = children[t];
        std::sort(ch.begin(), ch.end(), [&](int a, int b) { return clade[a] != clade[b] ? clade[a] > clade[b] : a < b; });
        for (int c : ch) rec(c, depth + 1);
    };
    rec(1, 0);
}

} // namespace

int main(int argc, char **argv) {
    mtb_params par; mtb_default_params(&par);
    std::string taxdir; int device = 0; size_t max_reads = 2000000;
    std::vector<std::string> pos;
    for (int i = 1; i < argc; i++) {
        std::string a = argv[i];
        auto val = [&]() { if (i + 1 >= argc) { fprintf(stderr, "missing value for %s\n", a.c_str()); exit(1); } return std::string(argv[++i]); };
        if (a == "--seq-mode") par.seq_mode = atoi(val().c_str());
        else if (a == "--min-score") par.min_score = (float)atof(val().c_str());
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
        else if (a == "--device") device = atoi(val().c_str());
        else if (a.rfind("--", 0) == 0) { fprintf(stderr, "unsupported flag %s\n", a.c_str()); return 1; }
        else pos.push_back(a);
    }
    size_t need = par.seq_mode == 2 ? 5 : 4;
    if (pos.size() != need) {
        fprintf(stderr, "usage: mtb_classify [flags] <FASTA/Q>%s <DBDIR> <OUTDIR> <JobID>\n", par.seq_mode == 2 ? " <FASTA/Q>" : "");
        return 1;
    }
    const bool paired = par.seq_mode == 2;
    const std::string dbdir = pos[paired ? 2 : 1], outdir = pos[paired ? 3 : 2], job = pos[paired ? 4 : 3];
    try {
        mtb::Engine eng(device, dbdir, taxdir, par);          /* db.parameters overrides the flags (common.cpp:88-133) */
        mtb::Classifier cls(eng, par);
        SeqReader r1(pos[0]);
        std::unique_ptr<SeqReader> r2;
        if (paired) r2.reset(new SeqReader(pos[1]));
        std::ofstream out(outdir + "/" + job + "_classifications.tsv");
        if (!out) throw std::runtime_error("cannot write to " + outdir);
        unsigned long total = 0; bool first = true;
        for (;;) {
            mtb::ReadBatch b;
            std::string name, seq, n2, s2;
            while (b.size() < max_reads && r1.next(name, seq)) {
                b.add(name, seq);
                if (paired) { if (!r2->next(n2, s2)) throw std::runtime_error("mate file is shorter"); b.add_mate(s2); }
            }
            if (b.size() == 0) break;
            std::vector<mtb::Query> q;
            cls.classifyBatch(b, q);
            write_classifications(out, q, eng.index, first);
            first = false;
            total += b.size();
            std::cout << "The number of processed sequences: " << total << std::endl;
        }
        FILE *fp = fopen((outdir + "/" + job + "_report.tsv").c_str(), "w");
        if (!fp) throw std::runtime_error("cannot write the report");
        write_report(fp, cls.getTaxCounts(), eng.index, total);
        fclose(fp);
    } catch (const std::exception &e) {
        fprintf(stderr, "mtb_classify: %s\n", e.what());
        return 1;
    }
    return 0;
}
