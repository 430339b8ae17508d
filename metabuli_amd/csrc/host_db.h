/* host_db.h -- host-side loading of what the hot path needs from a Metabuli
 * database directory: db.parameters (src/commons/common.cpp:88-133),
 * taxonomy dumps (loadTaxonomy, common.cpp:50-86 -> *.dmp branch), taxID_list
 * (KmerMatcher::loadTaxIdList, KmerMatcher.cpp:93-117) and the raw diffIdx /
 * info files.  Produces dense arrays indexed by taxonomy id for the device.
 * The binary `taxonomyDB` (TaxonomyWrapper::unserialize) is not supported yet:
 * its layout depends on MMseqs2 types absent from the reference snapshot.     */
#ifndef MTB_HOST_DB_H
#define MTB_HOST_DB_H
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <string>
#include <vector>
#include "../../include/mtb.h"

namespace mtbhost {

inline bool file_exists(const std::string &p) { FILE *f = fopen(p.c_str(), "rb"); if (!f) return false; fclose(f); return true; }

/* loadDbParameters (common.cpp:88-133): the DB overrides the flags */
inline bool load_db_parameters(const std::string &dbdir, mtb_params *p, int *reduced_aa = nullptr) {
    std::ifstream in(dbdir + "/db.parameters");
    if (!in) return false;
    std::string line;
    while (std::getline(in, line)) {
        size_t tab = line.find('\t');
        if (tab == std::string::npos) continue;
        std::string k = line.substr(0, tab), v = line.substr(tab + 1);
        size_t t2 = v.find('\t'); if (t2 != std::string::npos) v = v.substr(0, t2);
        if (k == "Accession_level") {
            if (v == "0" && p->accession_level == 1) p->accession_level = 0;
            if (v == "1" && p->accession_level == 0) p->accession_level = 2;
        } else if (k == "Skip_redundancy") { if (v == "1") p->skip_redundancy = 1; }
        else if (k == "Syncmer") { if (v == "1" && p->syncmer == 0) p->syncmer = 1; }
        else if (k == "S-mer_len") p->smer_len = atoi(v.c_str());
        else if (k == "Kmer_format") p->kmer_format = atoi(v.c_str());
        else if (k == "Reduced_alphabet") { if (reduced_aa) *reduced_aa = atoi(v.c_str()); }
    }
    return true;
}

/* rank table of MMseqs2's NcbiTaxonomy::findRankIndex, mirrored in
 * TaxonomyWrapper.h:229-267; unknown / "no rank" -> -1 */
inline int find_rank_index(const std::string &r) {
    static const std::map<std::string, int> M = {
        {"forma", 1}, {"varietas", 2}, {"subspecies", 3}, {"species", 4}, {"species subgroup", 5},
        {"species group", 6}, {"subgenus", 7}, {"genus", 8}, {"subtribe", 9}, {"tribe", 10},
        {"subfamily", 11}, {"family", 12}, {"superfamily", 13}, {"parvorder", 14}, {"infraorder", 15},
        {"suborder", 16}, {"order", 17}, {"superorder", 18}, {"infraclass", 19}, {"subclass", 20},
        {"class", 21}, {"superclass", 22}, {"subphylum", 23}, {"phylum", 24}, {"superphylum", 25},
        {"subkingdom", 26}, {"kingdom", 27}, {"superkingdom", 28}, {"domain", 28}};
    auto it = M.find(r);
    return it == M.end() ? -1 : it->second;
}

struct Taxonomy {
    int32_t max_id = 0;
    std::vector<int32_t> canon, parent, depth, rank_idx, sp_parent, tax2species;
    std::vector<uint8_t> under_euk, acc_leaf;
    std::vector<std::string> rank, name;     /* by canonical id (reporting only) */
    int32_t eukaryota = 0;

    int32_t cn(int32_t t) const { return (t >= 0 && t <= max_id) ? canon[(size_t)t] : -1; }
    int32_t lca(int32_t a, int32_t b) const {
        int32_t ca = cn(a), cb = cn(b);
        if (ca < 0) return b;
        if (cb < 0) return a;
        a = ca; b = cb;
        while (depth[(size_t)a] > depth[(size_t)b]) a = parent[(size_t)a];
        while (depth[(size_t)b] > depth[(size_t)a]) b = parent[(size_t)b];
        while (a != b) { a = parent[(size_t)a]; b = parent[(size_t)b]; }
        return a;
    }
    /* TaxonomyWrapper::getTaxIdAtRank (TaxonomyWrapper.cpp:479-498) */
    int32_t at_rank(int32_t t, int target) const {
        if (t == 0 || cn(t) < 0 || t == 1) return 0;
        int32_t cur = cn(t);
        int cnt = 0;
        while (cnt < 30 && rank_idx[(size_t)cur] < target) { cur = parent[(size_t)cur]; cnt++; }
        if (cnt == 30) return t;
        return cur;
    }
};

inline std::vector<std::string> split_dmp(const std::string &line) {
    std::vector<std::string> out;
    size_t prev = 0;
    while (true) {
        size_t pos = line.find("\t|", prev);
        if (pos == std::string::npos) { if (prev < line.size()) out.push_back(line.substr(prev)); break; }
        out.push_back(line.substr(prev, pos - prev));
        prev = pos + 2;
        if (prev < line.size() && line[prev] == '\t') prev++;
    }
    return out;
}

inline bool load_taxonomy(const std::string &dir, Taxonomy *t, std::string *err) {
    std::ifstream fn(dir + "/nodes.dmp");
    if (!fn) { *err = "cannot open " + dir + "/nodes.dmp"; return false; }
    struct N { int32_t id, parent; int rank; std::string rank_name; };
    std::vector<N> nodes;
    std::string line;
    int32_t mx = 1;
    while (std::getline(fn, line)) {
        auto c = split_dmp(line);
        if (c.size() < 3) continue;
        N n{(int32_t)atoi(c[0].c_str()), (int32_t)atoi(c[1].c_str()), find_rank_index(c[2]), c[2]};
        mx = std::max(mx, std::max(n.id, n.parent));
        nodes.push_back(n);
    }
    std::vector<std::pair<int32_t, int32_t>> merged;
    {
        std::ifstream fm(dir + "/merged.dmp");
        while (fm && std::getline(fm, line)) {
            auto c = split_dmp(line);
            if (c.size() < 2) continue;
            merged.push_back({(int32_t)atoi(c[0].c_str()), (int32_t)atoi(c[1].c_str())});
            mx = std::max(mx, std::max(merged.back().first, merged.back().second));
        }
    }
    t->max_id = mx;
    size_t sz = (size_t)mx + 1;
    t->canon.assign(sz, -1); t->parent.assign(sz, -1); t->depth.assign(sz, 0); t->rank_idx.assign(sz, -1);
    t->sp_parent.assign(sz, 0); t->tax2species.assign(sz, 0); t->under_euk.assign(sz, 0);
    t->rank.assign(sz, std::string()); t->name.assign(sz, std::string()); t->acc_leaf.assign(sz, 0);
    for (auto &n : nodes) { t->canon[(size_t)n.id] = n.id; t->parent[(size_t)n.id] = n.parent; t->rank_idx[(size_t)n.id] = n.rank; t->rank[(size_t)n.id] = n.rank_name;
                            t->acc_leaf[(size_t)n.id] = (n.rank_name.empty() || n.rank_name == "accession") ? 1 : 0; }
    for (auto &n : nodes) if (t->canon[(size_t)n.parent] < 0) { *err = "nodes.dmp: missing parent taxon"; return false; }
    for (auto &m : merged) if (t->canon[(size_t)m.first] < 0 && t->canon[(size_t)m.second] >= 0) t->canon[(size_t)m.first] = m.second;
    for (auto &n : nodes) {
        int32_t d = 0, c = n.id;
        while (t->parent[(size_t)c] != c && d < 100000) { c = t->parent[(size_t)c]; d++; }
        t->depth[(size_t)n.id] = d;
    }
    {   /* setEukaryoteTaxID (TaxonomyWrapper.h:89-100): node named "Eukaryota" */
        std::ifstream fnm(dir + "/names.dmp");
        while (fnm && std::getline(fnm, line)) {
            if (line.find("scientific name") == std::string::npos) continue;
            auto c = split_dmp(line);
            if (c.size() < 2) continue;
            int32_t id = (int32_t)atoi(c[0].c_str());
            if (id >= 0 && id <= mx && t->canon[(size_t)id] == id) t->name[(size_t)id] = c[1];
            if (c[1] == "Eukaryota" && t->eukaryota == 0) t->eukaryota = id;
        }
    }
    const int SPECIES = find_rank_index("species");
    for (auto &n : nodes) {
        if (t->eukaryota > 0) {         /* IsAncestor(eukaryota, n) incl. equality */
            int32_t c = n.id;
            while (true) { if (c == t->eukaryota) { t->under_euk[(size_t)n.id] = 1; break; } if (t->parent[(size_t)c] == c) break; c = t->parent[(size_t)c]; }
        }
        int32_t sp = t->at_rank(n.id, SPECIES);
        int32_t csp = t->cn(sp);
        t->sp_parent[(size_t)n.id] = csp >= 0 ? t->parent[(size_t)csp] : 0;
    }
    return true;
}

/* KmerMatcher::loadTaxIdList (KmerMatcher.cpp:93-117) as a dense table */
inline void build_tax2species(Taxonomy *t, const int32_t *ids, size_t n) {
    const int SPECIES = find_rank_index("species");
    for (size_t i = 0; i < n; i++) {
        int32_t tax = ids[i];
        if (tax < 0 || tax > t->max_id) continue;
        int32_t sp = t->at_rank(tax, SPECIES);
        int32_t cur = t->cn(tax);
        if (cur < 0) continue;
        if (tax != cur) t->tax2species[(size_t)tax] = sp;
        int guard = 0;
        while (cur != sp && guard++ < 100000) {
            t->tax2species[(size_t)cur] = sp;
            int32_t par = t->parent[(size_t)cur];
            if (par == cur) break;
            cur = par;
        }
        if (sp >= 0 && sp <= t->max_id) t->tax2species[(size_t)sp] = sp;
    }
}

inline bool read_taxid_list(const std::string &path, std::vector<int32_t> *out) {
    std::ifstream in(path);
    if (!in) return false;
    std::string line;
    while (std::getline(in, line)) { if (line.empty()) continue; out->push_back((int32_t)strtoul(line.c_str(), nullptr, 10)); }
    return true;
}

template <class T> inline bool read_whole(const std::string &path, std::vector<T> *v) {
    FILE *f = fopen(path.c_str(), "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long long sz = ftell(f); fseek(f, 0, SEEK_SET);
    v->resize((size_t)sz / sizeof(T));
    size_t r = fread(v->data(), sizeof(T), v->size(), f); fclose(f);
    return r == v->size();
}

} // namespace mtbhost
#endif
