/* host_db.h -- host-side loading of what the hot path needs from a Metabuli
 * database directory: db.parameters (src/commons/common.cpp:88-133),
 * taxonomy dumps (loadTaxonomy, common.cpp:50-86 -> *.dmp branch), taxID_list
 * (KmerMatcher::loadTaxIdList, KmerMatcher.cpp:93-117) and the raw diffIdx /
 * info files.  Produces dense arrays indexed by taxonomy id for the device.
 * The binary `taxonomyDB` (TaxonomyWrapper::unserialize) is read by load_taxonomy_db;
 * the layout of its MMseqs2 parts is restated from the published sources, unpinned.   */
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

template <class T> inline bool read_whole(const std::string &path, std::vector<T> *v) {
    FILE *f = fopen(path.c_str(), "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long long sz = ftell(f); fseek(f, 0, SEEK_SET);
    v->resize((size_t)sz / sizeof(T));
    size_t r = fread(v->data(), sizeof(T), v->size(), f); fclose(f);
    return r == v->size();
}

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
    std::vector<int32_t> orig;               /* internal -> original taxonomy id (identity for dump-file taxonomies) */
    int32_t eukaryota = 0;
    mutable std::vector<std::vector<int32_t>> kids;      /* built on first use (reporting only) */
    const std::vector<int32_t> &children_of(int32_t c) const {
        if (kids.empty()) {
            kids.assign((size_t)max_id + 1, std::vector<int32_t>());
            for (int32_t t = 0; t <= max_id; t++) if (canon[(size_t)t] == t && parent[(size_t)t] != t && parent[(size_t)t] >= 0) kids[(size_t)parent[(size_t)t]].push_back(t);
        }
        return kids[(size_t)c];
    }

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

/* one node as either source delivers it */
struct RawNode { int32_t id, parent; std::string rank, name; };

/* Dense arrays from a node list: `aliases` = (old id, current id) pairs (merged.dmp / the D table of taxonomyDB),
 * `eukaryota` = id of the node named "Eukaryota" (0 if none; setEukaryoteTaxID, TaxonomyWrapper.h:89-100),
 * `orig` = internal -> original id (getOriginalTaxID, TaxonomyWrapper.h:70-79), empty = identity. */
inline bool finalize_taxonomy(Taxonomy *t, const std::vector<RawNode> &nodes, const std::vector<std::pair<int32_t, int32_t>> &aliases,
                              int32_t max_id, int32_t eukaryota, const std::vector<int32_t> &orig, std::string *err) {
    int32_t mx = std::max<int32_t>(max_id, 1);
    for (auto &n : nodes) mx = std::max(mx, std::max(n.id, n.parent));
    for (auto &m : aliases) mx = std::max(mx, std::max(m.first, m.second));
    /* the tables are dense in the taxon id (NCBI ids are below 2^22 today): a damaged file must not ask for tens of GB of them */
    if (mx > (1 << 28)) { *err = "taxonomy: implausible taxon id " + std::to_string(mx); return false; }
    t->max_id = mx;
    size_t sz = (size_t)mx + 1;
    t->canon.assign(sz, -1); t->parent.assign(sz, -1); t->depth.assign(sz, 0); t->rank_idx.assign(sz, -1);
    t->sp_parent.assign(sz, 0); t->tax2species.assign(sz, 0); t->under_euk.assign(sz, 0);
    t->rank.assign(sz, std::string()); t->name.assign(sz, std::string()); t->acc_leaf.assign(sz, 0);
    t->orig.assign(sz, 0);
    for (size_t i = 0; i < sz; i++) t->orig[i] = i < orig.size() ? orig[i] : (orig.empty() ? (int32_t)i : 0);
    t->eukaryota = eukaryota;
    for (auto &n : nodes) {
        if (n.id < 0 || n.parent < 0) { *err = "taxonomy: negative taxon id"; return false; }
        t->canon[(size_t)n.id] = n.id; t->parent[(size_t)n.id] = n.parent; t->rank_idx[(size_t)n.id] = find_rank_index(n.rank);
        t->rank[(size_t)n.id] = n.rank; t->name[(size_t)n.id] = n.name;
        t->acc_leaf[(size_t)n.id] = (n.rank.empty() || n.rank == "accession") ? 1 : 0;
    }
    for (auto &n : nodes) if (t->canon[(size_t)n.parent] < 0) { *err = "taxonomy: missing parent taxon"; return false; }
    {   /* one tree: the walks of two taxa towards each other (lca, here and on the device) end at a common root */
        int32_t root = -1;
        for (auto &n : nodes) if (n.parent == n.id) { if (root >= 0 && root != n.id) { *err = "taxonomy: more than one root (" + std::to_string(root) + ", " + std::to_string(n.id) + ")"; return false; } root = n.id; }
    }
    for (auto &m : aliases) if (m.first >= 0 && m.second >= 0 && t->canon[(size_t)m.first] < 0 && t->canon[(size_t)m.second] >= 0) t->canon[(size_t)m.first] = t->canon[(size_t)m.second];
    for (auto &n : nodes) {
        int32_t d = 0, c = n.id;
        while (t->parent[(size_t)c] != c && d < 100000) { c = t->parent[(size_t)c]; d++; }
        /* every walk towards the root below (and on the device) relies on reaching a node that is its own parent */
        if (d >= 100000) { *err = "taxonomy: the parent links of taxon " + std::to_string(n.id) + " do not lead to a root (cycle)"; return false; }
        t->depth[(size_t)n.id] = d;
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

/* names / nodes / merged.dmp (the `else` branches of loadTaxonomy, common.cpp:76-86; ids are used as they are) */
inline bool load_taxonomy(const std::string &dir, Taxonomy *t, std::string *err) {
    std::ifstream fn(dir + "/nodes.dmp");
    if (!fn) { *err = "cannot open " + dir + "/nodes.dmp"; return false; }
    std::vector<RawNode> nodes;
    std::string line;
    int32_t mx = 1;
    while (std::getline(fn, line)) {
        auto c = split_dmp(line);
        if (c.size() < 3) continue;
        RawNode n{(int32_t)atoi(c[0].c_str()), (int32_t)atoi(c[1].c_str()), c[2], std::string()};
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
    int32_t eukaryota = 0;
    {
        std::vector<int32_t> slot((size_t)mx + 1, -1);
        for (size_t i = 0; i < nodes.size(); i++) if (nodes[i].id >= 0) slot[(size_t)nodes[i].id] = (int32_t)i;
        std::ifstream fnm(dir + "/names.dmp");
        while (fnm && std::getline(fnm, line)) {
            if (line.find("scientific name") == std::string::npos) continue;
            auto c = split_dmp(line);
            if (c.size() < 2) continue;
            int32_t id = (int32_t)atoi(c[0].c_str());
            if (id >= 0 && id <= mx && slot[(size_t)id] >= 0) nodes[(size_t)slot[(size_t)id]].name = c[1];
            if (c[1] == "Eukaryota" && eukaryota == 0) eukaryota = id;
        }
    }
    return finalize_taxonomy(t, nodes, merged, mx, eukaryota, std::vector<int32_t>(), err);
}

/* The binary `taxonomyDB` every database written by the current `build` carries (TaxonomyWrapper::serialize,
 * TaxonomyWrapper.cpp:289-361; reader restated from ::unserialize, :363-421).  loadTaxonomy prefers it over the dump files
 * (common.cpp:52-75).  Layout, little-endian, no alignment padding between the sections:
 *     int32   SERIALIZATION_VERSION                  (MMseqs2 NcbiTaxonomy.h; 2 in the releases Metabuli vendors)
 *   [ uint64  1 ]                                    present iff internal ids are used (unserialize peeks: a size_t equal
 *                                                    to 1 at offset 4 means "internal ids", anything else is maxNodes)
 *     uint64  maxNodes;  int32 maxTaxID
 *     TaxonNode[maxNodes]   = { int32 id; int32 taxId; int32 parentTaxId; (4 bytes padding) uint64 rankIdx; uint64 nameIdx } = 32 B
 *     int32   D[maxTaxID+1]                          taxon id -> node index, -1 = absent; a merged id shares its target's index
 *   [ int32   internal2orgTaxId[maxTaxID+1] ]        iff internal ids
 *     int32   E[2 maxNodes], L[2 maxNodes], H[maxNodes], M[2 maxNodes][K]     Euler tour / RMQ tables (not needed here: the
 *                                                    device works on parent / depth arrays), K = (int)MathUtil::flog2(2 maxNodes) + 1
 *     StringBlock<unsigned int>: uint64 byteCapacity, entryCapacity, entryCount; char data[byteCapacity]; uint32 offsets[entryCapacity]
 * TaxonNode and StringBlock belong to MMseqs2, which is absent from the reference snapshot: their layout is restated from
 * the published MMseqs2 sources and is NOT pinned against a file written by the reference ("layout unpinned").
 * Because K comes from a float approximation of log2, the StringBlock is located by trying K = floor(log2(2 maxNodes)) + 1
 * and its neighbours and keeping the one whose header accounts for the rest of the file exactly. */
inline bool load_taxonomy_db(const std::string &path, Taxonomy *t, std::string *err) {
    std::vector<char> buf;
    if (!read_whole(path, &buf)) { *err = "cannot read " + path; return false; }
    const size_t N = buf.size();
    size_t p = 0;
    auto need = [&](size_t n) { return p + n <= N; };
    auto rd32 = [&](size_t at) { int32_t v; memcpy(&v, buf.data() + at, 4); return v; };
    auto rd64 = [&](size_t at) { uint64_t v; memcpy(&v, buf.data() + at, 8); return v; };
    if (!need(4 + 8 + 8 + 4)) { *err = path + ": truncated header"; return false; }
    const int32_t version = rd32(p); p += 4;
    if (version != 2) { *err = path + ": unsupported serialization version " + std::to_string(version) + " (the reference answers \"Outdated taxonomy information\")"; return false; }
    bool internal = false;
    if (rd64(p) == 1) { internal = true; p += 8; }
    const uint64_t max_nodes = rd64(p); p += 8;
    const int32_t max_taxid = rd32(p); p += 4;
    if (max_nodes == 0 || max_nodes > (1ull << 31) || max_taxid < 0) { *err = path + ": implausible header"; return false; }
    const size_t nodes_at = p;
    if (!need(max_nodes * 32)) { *err = path + ": truncated node table"; return false; }
    p += max_nodes * 32;
    const size_t d_at = p;
    if (!need(((size_t)max_taxid + 1) * 4)) { *err = path + ": truncated D table"; return false; }
    p += ((size_t)max_taxid + 1) * 4;
    size_t i2o_at = 0;
    if (internal) { i2o_at = p; if (!need(((size_t)max_taxid + 1) * 4)) { *err = path + ": truncated id map"; return false; } p += ((size_t)max_taxid + 1) * 4; }
    p += (max_nodes * 2) * 4 * 2 + max_nodes * 4;                     /* E, L, H */
    if (p > N) { *err = path + ": truncated Euler tour tables"; return false; }
    const uint64_t dim = max_nodes * 2;
    int k0 = 0; while ((2ull << k0) <= dim) k0++;                     /* floor(log2(dim)) */
    size_t block_at = 0; uint64_t byte_cap = 0, entry_cap = 0, entry_cnt = 0;
    for (int dk : {1, 2, 0, 3}) {
        const size_t at = p + (size_t)dim * (size_t)(k0 + dk) * 4;
        if (at + 24 > N) continue;
        const uint64_t bc = rd64(at), ec = rd64(at + 8), en = rd64(at + 16);
        if (bc > N || ec > N / 4 + 1 || en > ec) continue;
        /* serialize() always reserves (maxTaxID + 1) ints for internal2orgTaxId in memSize and writes the whole buffer
         * (TaxonomyWrapper.cpp:296-310), but only fills them in when internal ids are used (:341-344): a file written with
         * useInternalTaxID == false ends in that many unused bytes */
        const size_t slack = internal ? 0 : ((size_t)max_taxid + 1) * 4;
        if (at + 24 + bc + ec * 4 != N && at + 24 + bc + ec * 4 + slack != N) continue;
        block_at = at; byte_cap = bc; entry_cap = ec; entry_cnt = en;
        break;
    }
    if (!block_at) { *err = path + ": cannot locate the string block (layout differs from the one restated in host_db.h)"; return false; }
    const char *data = buf.data() + block_at + 24;
    const size_t offs_at = block_at + 24 + byte_cap;
    auto str = [&](uint64_t idx) -> std::string {
        if (idx >= entry_cnt) return std::string();
        uint32_t o; memcpy(&o, buf.data() + offs_at + idx * 4, 4);
        if (o >= byte_cap) return std::string();
        return std::string(data + o, strnlen(data + o, byte_cap - o));
    };
    (void)entry_cap;
    std::vector<RawNode> nodes((size_t)max_nodes);
    int32_t eukaryota = 0;
    for (uint64_t i = 0; i < max_nodes; i++) {
        const size_t at = nodes_at + i * 32;
        RawNode &n = nodes[(size_t)i];
        n.id = rd32(at + 4); n.parent = rd32(at + 8);
        const uint64_t rank_idx = rd64(at + 16), name_idx = rd64(at + 24);
        n.rank = str(rank_idx); n.name = str(name_idx);
        if (n.id < 0 || n.id > max_taxid || n.parent < 0 || n.parent > max_taxid) { *err = path + ": taxon id outside 0..maxTaxID"; return false; }
        if (eukaryota == 0 && name_idx != 0 && n.name == "Eukaryota") eukaryota = n.id;      /* setEukaryoteTaxID skips nameIdx 0 */
    }
    std::vector<std::pair<int32_t, int32_t>> aliases;
    for (int32_t id = 0; id <= max_taxid; id++) {
        const int32_t d = rd32(d_at + (size_t)id * 4);
        if (d < 0) continue;
        if ((uint64_t)d >= max_nodes) { *err = path + ": D table points outside the node table"; return false; }
        if (nodes[(size_t)d].id != id) aliases.push_back({id, nodes[(size_t)d].id});
    }
    std::vector<int32_t> orig;
    if (internal) { orig.resize((size_t)max_taxid + 1); memcpy(orig.data(), buf.data() + i2o_at, orig.size() * 4); }
    return finalize_taxonomy(t, nodes, aliases, max_taxid, eukaryota, orig, err);
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

} // namespace mtbhost
#endif
