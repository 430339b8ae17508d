/*
 * oracle.cpp -- CPU restatement of the `metabuli classify` hot path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Every function cites the
 * reference file:line (relative to /root/reference/src/commons) it follows.
 * The code is written from scratch; control flow mirrors the reference so
 * that data-dependent quirks are reproduced.
 */
#include "oracle.h"

#include <algorithm>
#include <parallel/algorithm>      /* __gnu_parallel::sort = the reference's SORT_PARALLEL (FastSort.h of MMseqs2) under OpenMP */
#include <omp.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <fstream>
#include <map>
#include <string>
#include <unordered_map>
#include <vector>

/* Host threads of the OpenMP layer (the reference parallelises the same four places with `#pragma omp`:
 * KmerExtractor.cpp:83-160 per read chunk, SORT_PARALLEL, KmerMatcher.cpp:206-470 per query split,
 * Classifier.cpp:187-203 per read block).  1 = the plain serial restatement. */
static int g_orc_threads = 1;
extern "C" void orc_set_threads(int n) { g_orc_threads = n < 1 ? 1 : n; omp_set_num_threads(g_orc_threads); }
extern "C" int orc_get_threads(void) { return g_orc_threads; }
/* uninitialised storage: big scratch arrays must not be zero-filled by one thread before the parallel stages touch them */
template <class T> struct RawBuf {
    T *p = nullptr; size_t n = 0;
    explicit RawBuf(size_t n_ = 0) { if (n_) resize(n_); }
    ~RawBuf() { free(p); }
    RawBuf(const RawBuf &) = delete; RawBuf &operator=(const RawBuf &) = delete;
    void resize(size_t n_) { free(p); p = (T *)malloc(std::max<size_t>(n_, 1) * sizeof(T)); n = p ? n_ : 0; }
    T *data() { return p; } size_t size() const { return n; }
};

namespace {

/* ------------------------------------------------------------------ */
/* Tables                                                              */
/* ------------------------------------------------------------------ */

// Base canonicalisation: common.cpp:13-23 (`atcg`, `iRCT`) followed by
// nuc2int (GeneticCode.h:6).  Result: A,R,W->0  C,M,S->1  T,H,Y->2
// G,B,D,K,U->3 (either case), everything else 7.
struct BaseTables {
    uint8_t fwd[256];
    uint8_t rev[256];
    BaseTables() {
        memset(fwd, 7, sizeof(fwd));
        memset(rev, 7, sizeof(rev));
        const char *cls[4] = {"ARW", "CMS", "HTY", "BDGKU"};
        for (int c = 0; c < 4; c++) {
            for (const char *p = cls[c]; *p; ++p) {
                fwd[(unsigned char)*p] = (uint8_t)c;
                fwd[(unsigned char)(*p + 32)] = (uint8_t)c;   // lower case
            }
        }
        // iRCT complements A<->T, C<->G and leaves N/'.' invalid
        // (KmerScanner.h:95-97): codes 0<->2, 1<->3.
        for (int i = 0; i < 256; i++) rev[i] = fwd[i] == 7 ? 7 : (uint8_t)(fwd[i] ^ 2);
    }
};
const BaseTables BT;

// GeneticCode.h:34-193 (standard alphabet).  Amino acids are indexed into
// "ARNDCQEGHILKMFPSTWYV", stop = 20.  Codon id = code of the third base
// (A0 C1 T2 G3) with the extra ids of the six-codon families.
struct CodonTables {
    int aa[8][8][8];
    int num[8][8][8];
    CodonTables() {
        for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) for (int c = 0; c < 8; c++) {
            aa[a][b][c] = -1; num[a][b][c] = -1;
        }
        // standard genetic code, bases ordered T C A G (the classic table)
        const char *tcag = "FFLLSSSSYY**CC*WLLLLPPPPHHQQRRRRIIIMTTTTNNKKSSRRVVVVAAAADDEEGGGG";
        const char *letters = "ARNDCQEGHILKMFPSTWYV";
        const int tcag2code[4] = {2, 1, 0, 3};          // T C A G -> our codes
        for (int i = 0; i < 4; i++) for (int j = 0; j < 4; j++) for (int k = 0; k < 4; k++) {
            char r = tcag[i * 16 + j * 4 + k];
            int idx = 20;
            if (r != '*') idx = (int)(strchr(letters, r) - letters);
            int a = tcag2code[i], b = tcag2code[j], c = tcag2code[k];
            aa[a][b][c] = idx;
            num[a][b][c] = c;
        }
        // six-codon families + third stop codon (GeneticCode.h arginine,
        // leucine, serine and stop blocks)
        num[0][3][3] = 4;  // AGG  Arg
        num[0][3][0] = 5;  // AGA  Arg
        num[2][2][3] = 4;  // TTG  Leu
        num[2][2][0] = 5;  // TTA  Leu
        num[0][3][2] = 6;  // AGT  Ser
        num[0][3][1] = 7;  // AGC  Ser
        num[2][3][0] = 5;  // TGA  stop
    }
};
const CodonTables CT;

// KmerMatcher.h:66-70
const uint8_t HAMMING_LOOKUP[8][8] = {
    {0, 1, 1, 1, 2, 1, 3, 3}, {1, 0, 1, 1, 2, 2, 3, 2},
    {1, 1, 0, 1, 2, 2, 2, 3}, {1, 1, 1, 0, 1, 2, 3, 3},
    {2, 2, 2, 1, 0, 1, 4, 4}, {1, 2, 2, 2, 1, 0, 4, 4},
    {3, 3, 2, 3, 4, 4, 0, 1}, {3, 2, 3, 3, 4, 4, 1, 0}};

// KmerMatcher.h:72-158.  LUTk[a*8+b] = (h<4 ? h : 0) << 2k, except that the
// reference's LUT7 holds 1<<14 in the cells [4|5][6|7].
struct HammingLuts {
    uint16_t lut[8][64];
    HammingLuts() {
        for (int k = 0; k < 8; k++) for (int a = 0; a < 8; a++) for (int b = 0; b < 8; b++) {
            int h = HAMMING_LOOKUP[a][b];
            lut[k][a * 8 + b] = (uint16_t)((h < 4 ? h : 0) << (2 * k));
        }
        for (int a = 4; a <= 5; a++) for (int b = 6; b <= 7; b++) lut[7][a * 8 + b] = (uint16_t)(1u << 14);
    }
};
const HammingLuts HL;

inline uint64_t qinfo_pack(uint32_t seqId, uint32_t pos, uint32_t frame) {
    // Kmer.h:11-16 bit-field order on x86-64: pos:32 | sequenceID:29 | frame:3
    return (uint64_t)pos | ((uint64_t)(seqId & 0x1FFFFFFFu) << 32) | ((uint64_t)(frame & 7u) << 61);
}
inline uint32_t qi_pos(uint64_t q)   { return (uint32_t)q; }
inline uint32_t qi_seq(uint64_t q)   { return (uint32_t)((q >> 32) & 0x1FFFFFFFu); }
inline uint32_t qi_frame(uint64_t q) { return (uint32_t)(q >> 61); }

// LocalUtil.h:51-59
inline int max_covered_length(int len) {
    if (len % 3 == 2) return len - 2;
    if (len % 3 == 1) return len - 4;
    return len - 3;
}
// LocalUtil.h:46-48 (spaceNum = 0, k = 8)
inline int query_kmer_number(int len) { return (max_covered_length(len) / 3 - 8 + 1) * 6; }

/* ------------------------------------------------------------------ */
/* Scanners                                                            */
/* ------------------------------------------------------------------ */

struct ScanOut { uint64_t value; uint32_t pos; };

// Common codon fetch of KmerScanner.h:89-98 / SyncmerScanner.h:46-52
struct CodonReader {
    const char *seq; int seqStart; int seqEnd; bool fwd;
    inline void codes(int aaPos, int &a, int &b, int &c) const {
        if (fwd) {
            int ci = seqStart + aaPos * 3;
            a = BT.fwd[(unsigned char)seq[ci]]; b = BT.fwd[(unsigned char)seq[ci + 1]]; c = BT.fwd[(unsigned char)seq[ci + 2]];
        } else {
            int ci = seqEnd - aaPos * 3;
            a = BT.rev[(unsigned char)seq[ci]]; b = BT.rev[(unsigned char)seq[ci - 1]]; c = BT.rev[(unsigned char)seq[ci - 2]];
        }
    }
    inline int aa(int aaPos) const { int a, b, c; codes(aaPos, a, b, c); return CT.aa[a][b][c]; }
    inline int cid(int aaPos) const { int a, b, c; codes(aaPos, a, b, c); return CT.num[a][b][c]; }
};

// MetamerScanner (KmerScanner.h:49-117)
struct MetamerScanner {
    CodonReader rd; int aaLen; int posStart; int loaded; uint64_t dnaPart, aaPart;
    void init(const char *seq, int seqStart, int seqEnd, bool fwd) {
        rd = {seq, seqStart, seqEnd, fwd};
        aaLen = (seqEnd - seqStart + 1) / 3; posStart = 0; loaded = 0; dnaPart = aaPart = 0;
    }
    bool next(ScanOut &o) {
        while (posStart <= aaLen - 8) {
            bool sawN = false;
            loaded -= (loaded == 8);
            while (loaded < 8) {
                int aa = rd.aa(posStart + loaded);
                if (aa < 0) { sawN = true; break; }
                dnaPart = (dnaPart << 3) | (uint64_t)rd.cid(posStart + loaded);
                aaPart = (aaPart << 5) | (uint64_t)aa;
                loaded++;
            }
            if (sawN) { posStart += loaded + 1; dnaPart = aaPart = 0; loaded = 0; continue; }
            o.value = (aaPart << 24) | (dnaPart & 0xFFFFFFull);
            if (rd.fwd) o.pos = (uint32_t)(rd.seqStart + posStart * 3);
            else        o.pos = (uint32_t)(rd.seqEnd - (posStart + 8) * 3 + 1);
            posStart++;
            return true;
        }
        return false;
    }
};

// OldMetamerScanner (KmerScanner.h:119-181): kmerFormat 1.  Forward frames consume
// codons from the END of the window, reverse frames from its start; the amino-acid
// part is a base-21 number kept with a deque of positional weights.
struct OldMetamerScanner {
    const char *seq; int seqStart, seqEnd; bool fwd; int aaLen, posStart, loaded; uint64_t dnaPart, aaPart;
    std::deque<uint64_t> dq;
    void init(const char *s, int a, int b, bool f) { seq = s; seqStart = a; seqEnd = b; fwd = f; aaLen = (b - a + 1) / 3; posStart = 0; loaded = 0; dnaPart = aaPart = 0; dq.clear(); }
    bool next(ScanOut &o) {
        while (posStart <= aaLen - 8) {
            bool sawN = false;
            loaded -= (loaded == 8);
            while (loaded < 8) {
                int a, b, c;
                if (fwd) {
                    int ci = seqEnd - (posStart + loaded) * 3;
                    a = BT.fwd[(unsigned char)seq[ci - 2]]; b = BT.fwd[(unsigned char)seq[ci - 1]]; c = BT.fwd[(unsigned char)seq[ci]];
                } else {
                    int ci = seqStart + (posStart + loaded) * 3;
                    a = BT.rev[(unsigned char)seq[ci + 2]]; b = BT.rev[(unsigned char)seq[ci + 1]]; c = BT.rev[(unsigned char)seq[ci]];
                }
                int aa = CT.aa[a][b][c], codon = CT.num[a][b][c];
                if (aa < 0) { sawN = true; break; }
                if (dq.size() == 8) { aaPart = aaPart - dq.back(); dq.pop_back(); }
                for (auto &x : dq) x *= 21;
                dq.push_front((uint64_t)aa);
                aaPart = aaPart * 21 + (uint64_t)aa;
                dnaPart = (dnaPart << 3) | (uint64_t)codon;
                loaded++;
            }
            if (sawN) { posStart += loaded + 1; dnaPart = aaPart = 0; loaded = 0; dq.clear(); continue; }
            o.value = (aaPart << 24) | (dnaPart & 0xFFFFFFull);
            if (fwd) o.pos = (uint32_t)(seqEnd - (posStart + 8) * 3 + 1);
            else     o.pos = (uint32_t)(seqStart + posStart * 3);
            posStart++;
            return true;
        }
        return false;
    }
};

// SyncmerScanner (SyncmerScanner.h:9-101)
struct SyncmerScanner {
    CodonReader rd; int aaLen; int posStart; int loaded; uint64_t dnaPart, aaPart;
    int smerLen; uint64_t smerMask; int smerCnt; uint64_t smer; int prevPos;
    struct Item { uint64_t v; int pos; };
    std::deque<Item> dq;
    void init(const char *seq, int seqStart, int seqEnd, bool fwd, int s) {
        rd = {seq, seqStart, seqEnd, fwd};
        aaLen = (seqEnd - seqStart + 1) / 3; posStart = 0; loaded = 0; dnaPart = aaPart = 0;
        smerLen = s; smerMask = (1ull << (5 * s)) - 1; smerCnt = 0; smer = 0; prevPos = -8; dq.clear();
    }
    bool next(ScanOut &o) {
        bool found = false;
        while (posStart <= aaLen - 8 && !found) {
            bool sawN = false;
            smerCnt -= (smerCnt > 0);
            while (smerCnt < 8 - smerLen + 1) {
                loaded -= (loaded == smerLen);
                while (loaded < smerLen) {
                    int aa = rd.aa(posStart + smerCnt + loaded);
                    if (aa < 0) { sawN = true; break; }
                    smer = (smer << 5) | (uint64_t)aa;
                    loaded++;
                }
                if (sawN) break;
                smer &= smerMask;
                while (!dq.empty() && dq.back().v > smer) dq.pop_back();
                dq.push_back({smer, posStart + smerCnt});
                smerCnt++;
            }
            if (sawN) {
                posStart += smerCnt + loaded + 1;
                prevPos = posStart - 8;
                dq.clear(); smerCnt = loaded = 0; smer = 0;
                continue;
            }
            if (!dq.empty() && dq.front().pos < posStart) dq.pop_front();
            int anchor1 = posStart, anchor2 = posStart + (8 - smerLen);
            if (!dq.empty() && (dq.front().pos == anchor1 || dq.front().pos == anchor2)) {
                int shifts = posStart - prevPos;
                for (int i = 0; i < shifts; ++i) {
                    aaPart = (aaPart << 5) | (uint64_t)rd.aa(prevPos + 8 + i);
                    dnaPart = (dnaPart << 3) | (uint64_t)rd.cid(prevPos + 8 + i);
                }
                prevPos = posStart;
                found = true;
            }
            ++posStart;
        }
        if (!found) return false;
        o.value = (aaPart << 24) | (dnaPart & 0xFFFFFFull);
        if (rd.fwd) o.pos = (uint32_t)(rd.seqStart + prevPos * 3);
        else        o.pos = (uint32_t)(rd.seqEnd - (prevPos + 8) * 3 + 1);
        return true;
    }
};

// KmerExtractor::fillQueryKmerBuffer (KmerExtractor.cpp:342-373)
size_t fill_query_kmers(const char *seq, int seqLen, const orc_params &p, uint32_t seqID,
                        uint32_t offset, orc_kmer *out, size_t cap) {
    size_t n = 0;
    int usedLen = max_covered_length(seqLen);
    MetamerScanner ms; SyncmerScanner ss; OldMetamerScanner os;
    for (int frame = 0; frame < 6; frame++) {
        bool fwd = frame < 3;
        int begin;
        if (fwd) begin = frame % 3;
        else { begin = (seqLen % 3) - (frame % 3); if (begin < 0) begin += 3; }
        ScanOut o;
        if (p.kmer_format == 1) {          // KmerExtractor.cpp:13-17: format 1 always uses OldMetamerScanner
            os.init(seq, begin, begin + usedLen - 1, fwd);
            while (os.next(o)) { if (n < cap) out[n] = {o.value, qinfo_pack(seqID, o.pos + offset, (uint32_t)frame)}; n++; }
        } else if (p.syncmer) {
            ss.init(seq, begin, begin + usedLen - 1, fwd, p.smer_len);
            while (ss.next(o)) { if (n < cap) out[n] = {o.value, qinfo_pack(seqID, o.pos + offset, (uint32_t)frame)}; n++; }
        } else {
            ms.init(seq, begin, begin + usedLen - 1, fwd);
            while (ms.next(o)) { if (n < cap) out[n] = {o.value, qinfo_pack(seqID, o.pos + offset, (uint32_t)frame)}; n++; }
        }
    }
    return n;
}

/* ------------------------------------------------------------------ */
/* Taxonomy (what the path needs from MMseqs2 NcbiTaxonomy + wrapper)  */
/* ------------------------------------------------------------------ */

// TaxonomyWrapper.h:229-267 mirror of MMseqs2's rank table; unknown -> -1.
int find_rank_index(const std::string &r) {
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

std::vector<std::string> split_dmp(const std::string &line) {
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

} // namespace

struct orc_taxonomy {
    int maxTaxId = 0;
    std::vector<int> parent;       // by taxid; -1 = absent
    std::vector<int> alias;        // merged.dmp: old -> current
    std::vector<int> rankIdx;
    std::vector<int> depth;
    std::vector<std::string> rank, name;
    int eukaryota = 0;

    int canon(int t) const { if (t < 0 || t > maxTaxId) return -1; if (parent[t] >= 0) return t; if (alias[t] > 0) return alias[t]; return -1; }
    bool exists(int t) const { return canon(t) >= 0; }
    // NcbiTaxonomy::LCA(a,b): a missing node yields the other one.
    int lca(int a, int b) const {
        int ca = canon(a), cb = canon(b);
        if (ca < 0) return b;
        if (cb < 0) return a;
        a = ca; b = cb;
        while (depth[a] > depth[b]) a = parent[a];
        while (depth[b] > depth[a]) b = parent[b];
        while (a != b) { a = parent[a]; b = parent[b]; }
        return a;
    }
    // NcbiTaxonomy::LCA(vector): skips ids that do not exist.
    int lca(const std::vector<int> &v) const {
        int cur = -1;
        for (int t : v) { if (!exists(t)) continue; cur = cur < 0 ? canon(t) : lca(cur, t); }
        return cur < 0 ? 0 : cur;
    }
    // NcbiTaxonomy::IsAncestor: true when equal; false when either is absent/0.
    bool isAncestor(int anc, int child) const {
        if (anc == child) return true;
        if (anc == 0 || child == 0) return false;
        int a = canon(anc), c = canon(child);
        if (a < 0 || c < 0) return false;
        while (depth[c] > depth[a]) c = parent[c];
        return c == a;
    }
    // TaxonomyWrapper.cpp:479-498
    int atRank(int taxId, const std::string &r) const {
        if (taxId == 0 || !exists(taxId) || taxId == 1) return 0;
        int target = find_rank_index(r);
        int cur = canon(taxId);
        int cnt = 0;
        while (cnt < 30 && rankIdx[cur] < target) { cur = parent[cur]; cnt++; }
        if (cnt == 30) return taxId;
        return cur;
    }
};

extern "C" {

void orc_codon_tables(int *a, int *n) {
    memcpy(a, CT.aa, sizeof(CT.aa)); memcpy(n, CT.num, sizeof(CT.num));
}
void orc_base_codes(uint8_t *f, uint8_t *r) { memcpy(f, BT.fwd, 256); memcpy(r, BT.rev, 256); }
void orc_hamming_tables(uint8_t *lookup64, uint16_t *lut8x64) {
    memcpy(lookup64, HAMMING_LOOKUP, 64); memcpy(lut8x64, HL.lut, sizeof(HL.lut));
}
// KmerMatcher.h:348-360
uint8_t orc_hamming_sum(uint64_t a, uint64_t b) {
    uint8_t s = 0;
    for (int i = 0; i < 8; i++) s += HAMMING_LOOKUP[(a >> (3 * i)) & 7][(b >> (3 * i)) & 7];
    return s;
}
// KmerMatcher.h:386-400
uint16_t orc_hammings(uint64_t a, uint64_t b) {
    uint16_t h = 0;
    for (int i = 0; i < 8; i++) h |= HL.lut[i][(((a >> (3 * i)) & 7) << 3) | ((b >> (3 * i)) & 7)];
    return h;
}
// KmerMatcher.h:402-416
uint16_t orc_hammings_reverse(uint64_t a, uint64_t b) {
    uint16_t h = 0;
    for (int i = 0; i < 8; i++) h |= HL.lut[7 - i][(((a >> (3 * i)) & 7) << 3) | ((b >> (3 * i)) & 7)];
    return h;
}

size_t orc_extract_read(const char *seq, int len, const orc_params *p, uint32_t seq_id,
                        uint32_t offset, orc_kmer *out, size_t cap) {
    if (query_kmer_number(len) < 1) return 0;     // KmerExtractor.cpp:468-473
    return fill_query_kmers(seq, len, *p, seq_id, offset, out, cap);
}

// loadChunkOfReads + processSequence (KmerExtractor.cpp:292-340, 429-481) for reads [lo, hi)
static size_t extract_range(const char *bases, const uint64_t *offs, const char *bases2, const uint64_t *offs2, size_t lo, size_t hi,
                            const orc_params *p, orc_kmer *out, size_t cap, int32_t *qlen, int32_t *qlen2) {
    size_t n = 0;
    std::string buf;
    for (size_t i = lo; i < hi; i++) {
        int len1 = (int)(offs[i + 1] - offs[i]);
        qlen[i] = max_covered_length(len1);
        qlen2[i] = 0;
        bool empty = query_kmer_number(len1) < 1;
        int len2 = 0;
        if (p->seq_mode == 2) {
            len2 = (int)(offs2[i + 1] - offs2[i]);
            qlen2[i] = max_covered_length(len2);
            if (query_kmer_number(len2) < 1) empty = true;    // either mate too short -> skip pair
        }
        if (empty) continue;
        buf.assign(bases + offs[i], (size_t)len1);             // NUL-terminated like std::string
        n += fill_query_kmers(buf.c_str(), len1, *p, (uint32_t)(i + 1), 0, out + std::min(n, cap), cap - std::min(n, cap));
        if (p->seq_mode == 2) {
            buf.assign(bases2 + offs2[i], (size_t)len2);
            n += fill_query_kmers(buf.c_str(), len2, *p, (uint32_t)(i + 1), (uint32_t)(qlen[i] + 3),
                                  out + std::min(n, cap), cap - std::min(n, cap));
        }
    }
    return n;
}
size_t orc_extract_batch(const char *bases, const uint64_t *offs, const char *bases2,
                         const uint64_t *offs2, size_t n_reads, const orc_params *p,
                         orc_kmer *out, size_t cap, int32_t *qlen, int32_t *qlen2) {
    const int T = g_orc_threads;
    if (T <= 1 || n_reads < (size_t)T * 64) return extract_range(bases, offs, bases2, offs2, 0, n_reads, p, out, cap, qlen, qlen2);
    /* KmerExtractor.cpp:117-200: every thread extracts a chunk of reads into a private buffer and reserves its slice of the
     * shared buffer with one atomic add (kmerBuffer.reserveMemory); the order of the slices is arbitrary there too */
    const size_t CH = 1024, n_chunks = (n_reads + CH - 1) / CH;
    size_t total = 0;
#pragma omp parallel num_threads(T)
    {
        std::vector<orc_kmer> buf;
#pragma omp for schedule(dynamic, 1)
        for (size_t c = 0; c < n_chunks; c++) {
            size_t lo = c * CH, hi = std::min(n_reads, lo + CH);
            size_t nb = (size_t)(offs[hi] - offs[lo]) + (offs2 ? (size_t)(offs2[hi] - offs2[lo]) : 0);
            if (buf.size() < 2 * nb + 64) buf.resize(2 * nb + 64);
            size_t cnt = extract_range(bases, offs, bases2, offs2, lo, hi, p, buf.data(), buf.size(), qlen, qlen2);
            size_t at;
#pragma omp atomic capture
            { at = total; total += cnt; }
            if (at < cap) memcpy(out + at, buf.data(), std::min(cnt, cap - at) * sizeof(orc_kmer));
        }
    }
    return total;
}

void orc_sort_kmers(orc_kmer *k, size_t n) {
    auto cmp = [](const orc_kmer &a, const orc_kmer &b) {
        if (a.value != b.value) return a.value < b.value;
        return qi_seq(a.qinfo) < qi_seq(b.qinfo);
    };
    if (g_orc_threads > 1) __gnu_parallel::sort(k, k + n, cmp); else std::sort(k, k + n, cmp);
}

// IndexCreator::getDiffIdx (IndexCreator.cpp:874-892)
size_t orc_diffidx_encode(const uint64_t *values, size_t n, uint16_t *out) {
    size_t w = 0; uint64_t last = 0;
    for (size_t i = 0; i < n; i++) {
        uint64_t d = values[i] - last;
        uint16_t buf[5]; int idx = 3;
        buf[4] = (uint16_t)(0x8000u | (d & 0x7FFFu));
        d >>= 15;
        while (d) { buf[idx--] = (uint16_t)(d & 0x7FFFu); d >>= 15; }
        for (int j = idx + 1; j <= 4; j++) out[w++] = buf[j];
        last = values[i];
    }
    return w;
}
// KmerMatcher::getNextTargetKmer (KmerMatcher.h:282-297)
static inline uint64_t next_target(uint64_t cur, const uint16_t *buf, size_t &idx) {
    uint64_t d = 0;
    uint16_t frag = buf[idx++];
    while (!(frag & 0x8000u)) { d |= frag; d <<= 15; frag = buf[idx++]; }
    d |= (frag & 0x7FFFu);
    return d + cur;
}
size_t orc_diffidx_decode(const uint16_t *in, size_t n16, uint64_t *values) {
    size_t idx = 0, n = 0; uint64_t cur = 0;
    while (idx < n16) { cur = next_target(cur, in, idx); values[n++] = cur; }
    return n;
}

// IndexCreator::writeTargetFilesAndSplits + writeDbParameters
// (IndexCreator.cpp:817-872, 1251-1272), taxID_list (:329-333)
int orc_write_db(const char *dir, const uint64_t *values, const int32_t *taxids, size_t n,
                 int split_num, const orc_params *p) {
    std::string d(dir);
    std::vector<uint16_t> diff; diff.reserve(n * 3);
    struct Split { uint64_t ad; uint64_t diffOff; uint64_t infoOff; };
    std::vector<Split> splits((size_t)split_num, Split{0, 0, 0});
    const uint64_t AAMASK = ~0xFFFFFFull;
    uint64_t aaOfTemp = UINT64_MAX;
    size_t sizeOfSplit = n / (size_t)(split_num - 1);
    std::vector<uint64_t> offsetList((size_t)split_num + 1);
    for (int os = 0; os < split_num; os++) offsetList[(size_t)os] = (uint64_t)os * sizeOfSplit;
    offsetList[(size_t)split_num] = UINT64_MAX;
    int offsetListIdx = 1, splitListIdx = 1, splitCheck = 0;
    uint64_t last = 0;
    for (size_t j = 0; j < n; j++) {
        uint64_t v = values[j];
        // getDiffIdx: delta vs the previous entry, big-endian 15-bit groups
        uint64_t dlt = v - last; uint16_t buf[5]; int idx = 3;
        buf[4] = (uint16_t)(0x8000u | (dlt & 0x7FFFu)); dlt >>= 15;
        while (dlt) { buf[idx--] = (uint16_t)(dlt & 0x7FFFu); dlt >>= 15; }
        for (int q = idx + 1; q <= 4; q++) diff.push_back(buf[q]);
        last = v;
        size_t infoCnt = j + 1;
        if ((last & AAMASK) != aaOfTemp && splitCheck == 1) {
            if (splitListIdx < split_num) splits[(size_t)splitListIdx++] = {last, (uint64_t)diff.size(), (uint64_t)infoCnt};
            splitCheck = 0;
        }
        if (infoCnt == offsetList[(size_t)offsetListIdx]) {
            aaOfTemp = last & AAMASK; splitCheck = 1; offsetListIdx++;
        }
    }
    FILE *f = fopen((d + "/diffIdx").c_str(), "wb"); if (!f) return 1;
    fwrite(diff.data(), 2, diff.size(), f); fclose(f);
    f = fopen((d + "/info").c_str(), "wb"); if (!f) return 1;
    fwrite(taxids, 4, n, f); fclose(f);
    f = fopen((d + "/split").c_str(), "wb"); if (!f) return 1;
    fwrite(splits.data(), sizeof(Split), splits.size(), f); fclose(f);
    // taxID_list: distinct taxids, one per line
    // (distinct ids through a byte map when they are small non-negative numbers -- sorting a copy of 10^9 entries took longer than
    //  everything else of this function; the sorted distinct list is the same)
    std::vector<int32_t> u;
    {
        int32_t mn = 0, mx = 0;
        for (size_t j = 0; j < n; j++) { mn = std::min(mn, taxids[j]); mx = std::max(mx, taxids[j]); }
        if (mn >= 0 && mx < (1 << 28)) {
            std::vector<uint8_t> seen((size_t)mx + 1, 0);
            for (size_t j = 0; j < n; j++) seen[(size_t)taxids[j]] = 1;
            for (size_t t = 0; t < seen.size(); t++) if (seen[t]) u.push_back((int32_t)t);
        } else {
            u.assign(taxids, taxids + n);
            std::sort(u.begin(), u.end()); u.erase(std::unique(u.begin(), u.end()), u.end());
        }
    }
    f = fopen((d + "/taxID_list").c_str(), "w"); if (!f) return 1;
    for (int32_t t : u) fprintf(f, "%d\n", t);
    fclose(f);
    f = fopen((d + "/db.parameters").c_str(), "w"); if (!f) return 1;
    fprintf(f, "DB_name\tsynthetic\nCreation_date\t2026-01-01\nReduced_alphabet\t0\nAccession_level\t0\n");
    fprintf(f, "Mask_mode\t0\nMask_prob\t0.900000\nSkip_redundancy\t1\nSyncmer\t%d\n", p->syncmer);
    if (p->syncmer == 1) fprintf(f, "Syncmer_len\t%d\n", p->smer_len);
    fprintf(f, "Kmer_format\t%d\n", p->kmer_format);
    fclose(f);
    return 0;
}

/* ---- taxonomy ---- */
orc_taxonomy *orc_taxonomy_load(const char *names, const char *nodes, const char *merged) {
    auto *t = new orc_taxonomy();
    std::ifstream fn(nodes);
    if (!fn) { delete t; return nullptr; }
    std::string line;
    struct N { int id, parent; std::string rank; };
    std::vector<N> tmp;
    int mx = 1;
    while (std::getline(fn, line)) {
        auto c = split_dmp(line);
        if (c.size() < 3) continue;
        N n{atoi(c[0].c_str()), atoi(c[1].c_str()), c[2]};
        mx = std::max(mx, std::max(n.id, n.parent));
        tmp.push_back(n);
    }
    std::vector<std::pair<int, int>> mg;
    if (merged && *merged) {
        std::ifstream fm(merged);
        while (fm && std::getline(fm, line)) {
            auto c = split_dmp(line);
            if (c.size() < 2) continue;
            int o = atoi(c[0].c_str()), m = atoi(c[1].c_str());
            mg.push_back({o, m}); mx = std::max(mx, std::max(o, m));
        }
    }
    t->maxTaxId = mx;
    t->parent.assign((size_t)mx + 1, -1); t->alias.assign((size_t)mx + 1, 0);
    t->rankIdx.assign((size_t)mx + 1, -1); t->depth.assign((size_t)mx + 1, 0);
    t->rank.assign((size_t)mx + 1, ""); t->name.assign((size_t)mx + 1, "");
    for (auto &n : tmp) { t->parent[(size_t)n.id] = n.parent; t->rank[(size_t)n.id] = n.rank; t->rankIdx[(size_t)n.id] = find_rank_index(n.rank); }
    for (auto &m : mg) if (t->parent[(size_t)m.first] < 0 && t->parent[(size_t)m.second] >= 0) t->alias[(size_t)m.first] = m.second;
    // depth (root = 1 is its own parent)
    for (auto &n : tmp) {
        int d = 0, c = n.id;
        while (t->parent[(size_t)c] != c && d < 1000) { c = t->parent[(size_t)c]; d++; }
        t->depth[(size_t)n.id] = d;
    }
    std::ifstream fnm(names);
    while (fnm && std::getline(fnm, line)) {
        if (line.find("scientific name") == std::string::npos) continue;
        auto c = split_dmp(line);
        if (c.size() < 2) continue;
        int id = atoi(c[0].c_str());
        if (id >= 0 && id <= mx) { t->name[(size_t)id] = c[1]; if (c[1] == "Eukaryota") t->eukaryota = id; }
    }
    return t;
}
void orc_taxonomy_free(orc_taxonomy *t) { delete t; }
int orc_tax_lca(const orc_taxonomy *t, int a, int b) { return t->lca(a, b); }
int orc_tax_at_rank(const orc_taxonomy *t, int taxid, const char *rank) { return t->atRank(taxid, rank); }
int orc_tax_is_ancestor(const orc_taxonomy *t, int a, int c) { return t->isAncestor(a, c) ? 1 : 0; }
int orc_tax_max_id(const orc_taxonomy *t) { return t->maxTaxId; }
int orc_tax_parent(const orc_taxonomy *t, int x) { int c = t->canon(x); return c < 0 ? -1 : t->parent[(size_t)c]; }

} // extern "C"

/* ------------------------------------------------------------------ */
/* Matcher                                                             */
/* ------------------------------------------------------------------ */
struct orc_db {
    std::vector<uint16_t> diff;
    std::vector<int32_t> info;
    struct Split { uint64_t ad; uint64_t diffOff; uint64_t infoOff; };
    std::vector<Split> splits;
    std::unordered_map<int, int> taxId2speciesId;
    orc_params par;
    const orc_taxonomy *tax;
};

namespace {
template <class T> bool read_file(const std::string &path, std::vector<T> &v) {
    FILE *f = fopen(path.c_str(), "rb"); if (!f) return false;
    fseek(f, 0, SEEK_END); long sz = ftell(f); fseek(f, 0, SEEK_SET);
    v.resize((size_t)sz / sizeof(T));
    size_t r = fread(v.data(), sizeof(T), v.size(), f); fclose(f);
    return r == v.size();
}

// KmerMatcher::compareDna (KmerMatcher.cpp:1117-1146)
void compare_dna(uint64_t query, const std::vector<uint64_t> &targets, std::vector<uint8_t> &hd,
                 std::vector<size_t> &sel, std::vector<uint8_t> &selHam, std::vector<uint16_t> &selHams,
                 size_t &selCnt, uint8_t frame, int kmerFormat) {
    hd.resize(targets.size());
    uint8_t minHam = UINT8_MAX;
    for (size_t i = 0; i < targets.size(); i++) { hd[i] = orc_hamming_sum(query, targets[i]); minHam = std::min(minHam, hd[i]); }
    selCnt = 0;
    uint8_t maxHam = (uint8_t)std::min((int)minHam * 2, 7);
    for (size_t h = 0; h < targets.size(); h++) {
        if (hd[h] <= maxHam) {
            selHam[selCnt] = hd[h];
            selHams[selCnt] = !((frame < 3) ^ (kmerFormat == 2)) ? orc_hammings(query, targets[h])
                                                                  : orc_hammings_reverse(query, targets[h]);
            sel[selCnt++] = h;
        }
    }
}
} // namespace

extern "C" {

// KmerMatcher ctor + loadTaxIdList (KmerMatcher.cpp:19-37, 93-117)
orc_db *orc_db_open(const char *dir, const orc_taxonomy *tax, const orc_params *p) {
    auto *db = new orc_db();
    std::string d(dir);
    db->par = *p; db->tax = tax;
    if (!read_file(d + "/diffIdx", db->diff) || !read_file(d + "/info", db->info) || !read_file(d + "/split", db->splits)) {
        delete db; return nullptr;
    }
    std::ifstream in(d + "/taxID_list");
    std::string line;
    while (in && std::getline(in, line)) {
        if (line.empty()) continue;
        int taxId = (int)strtoul(line.c_str(), nullptr, 10);
        int sp = tax->atRank(taxId, "species");
        int cur = tax->canon(taxId);
        if (cur < 0) continue;
        if (taxId != cur) db->taxId2speciesId[taxId] = sp;
        int guard = 0;
        while (cur != sp && guard++ < 1000) {
            db->taxId2speciesId[cur] = sp;
            int par = tax->parent[(size_t)cur];
            if (par == cur) break;
            cur = par;
        }
        db->taxId2speciesId[sp] = sp;
    }
    return db;
}
void orc_db_close(orc_db *db) { delete db; }
size_t orc_db_num_kmers(const orc_db *db) { return db->info.size(); }

} // extern "C"

// The per-thread body of KmerMatcher::matchKmers (KmerMatcher.cpp:206-470) for the query split [startIdx, endIdx]:
// the thread finds its start checkpoint in `split` and streams diffIdx/info from there.
static size_t match_split(orc_db *db, const orc_kmer *q, size_t startIdx, size_t endIdx, orc_match *out, size_t cap) {
    const uint64_t AAMASK = ~0xFFFFFFull;
    const size_t numOfDiffIdx = db->diff.size();
    // usable splits (:157-164)
    std::vector<orc_db::Split> sp = db->splits;
    size_t use = sp.size();
    for (size_t i = 1; i < sp.size(); i++) {
        if (sp[i].ad == 0 || sp[i].ad == UINT64_MAX) { sp[i] = {UINT64_MAX, UINT64_MAX, UINT64_MAX}; use--; }
    }
    orc_db::Split start = sp[0];
    {
        uint64_t queryAA = q[startIdx].value & AAMASK;
        bool needLast = true;
        for (size_t j = 0; j < use; j++) {
            if (queryAA <= (sp[j].ad & AAMASK)) { j = j - (j != 0); start = sp[j]; needLast = false; break; }
        }
        // The reference indexes [use-2], undefined for all-zero split files of
        // tiny DBs (Appendix B.14); fall back to the stream start there.
        if (needLast) start = use >= 2 ? sp[use - 2] : sp[0];
    }
    int redundancyStored = (db->par.skip_redundancy == 0);
    uint32_t mask = ~((uint32_t)redundancyStored << 31);

    const uint16_t *diffBuf = db->diff.data();
    uint64_t currentTarget = start.ad;
    size_t diffIdxPos = (size_t)start.diffOff;
    size_t infoIdx = (size_t)start.infoOff - (start.ad != 0);
    if (start.ad == 0 && start.diffOff == 0 && start.infoOff == 0) currentTarget = next_target(currentTarget, diffBuf, diffIdxPos);

    uint64_t currentQuery = UINT64_MAX, currentQueryAA = UINT64_MAX;
    uint64_t currentQueryInfo = 0;
    std::vector<uint64_t> candTargets; std::vector<int32_t> candInfos; std::vector<uint8_t> hd;
    std::vector<uint8_t> selHam(1024); std::vector<size_t> sel(1024); std::vector<uint16_t> selHams(1024);
    size_t selCnt = 0, m = 0;
    const int kf = db->par.kmer_format;

    auto emit = [&](size_t j) {
        for (size_t k = 0; k < selCnt; k++) {
            size_t idx = sel[k];
            auto it = db->taxId2speciesId.find(candInfos[idx]);
            int spId = it == db->taxId2speciesId.end() ? 0 : it->second;
            if (m < cap) out[m] = {q[j].qinfo, candInfos[idx], spId, (uint32_t)(candTargets[idx] & 0xFFFFFFu), selHams[k], selHam[k], 0};
            m++;
        }
    };

    for (size_t j = startIdx; j < endIdx + 1; j++) {
        // (a) identical value on the same strand: reuse selection (:277-311)
        if (currentQuery == q[j].value && (qi_frame(currentQueryInfo) / 3 == qi_frame(q[j].qinfo) / 3)) { emit(j); continue; }
        selCnt = 0;
        // (b) same amino-acid part: reuse candidates (:315-353)
        if (currentQueryAA == (q[j].value & AAMASK)) {
            compare_dna(q[j].value, candTargets, hd, sel, selHam, selHams, selCnt, (uint8_t)qi_frame(q[j].qinfo), kf);
            emit(j);
            currentQuery = q[j].value; currentQueryAA = currentQuery & AAMASK; currentQueryInfo = q[j].qinfo;
            continue;
        }
        candTargets.clear(); candInfos.clear();
        currentQuery = q[j].value; currentQueryAA = currentQuery & AAMASK; currentQueryInfo = q[j].qinfo;
        // (c) advance the target stream (:363-371)
        while (diffIdxPos != numOfDiffIdx && currentQueryAA > (currentTarget & AAMASK)) {
            currentTarget = next_target(currentTarget, diffBuf, diffIdxPos);
            infoIdx++;
        }
        if (currentQueryAA != (currentTarget & AAMASK)) continue;
        while (diffIdxPos != numOfDiffIdx && currentQueryAA == (currentTarget & AAMASK)) {
            candTargets.push_back(currentTarget);
            candInfos.push_back((int32_t)((uint32_t)db->info[infoIdx] & mask));
            currentTarget = next_target(currentTarget, diffBuf, diffIdxPos);
            infoIdx++;
        }
        if (candTargets.size() > sel.size()) { sel.resize(candTargets.size()); selHam.resize(candTargets.size()); selHams.resize(candTargets.size()); }
        compare_dna(currentQuery, candTargets, hd, sel, selHam, selHams, selCnt, (uint8_t)qi_frame(q[j].qinfo), kf);
        emit(j);
    }
    return m;
}

extern "C" {

// KmerMatcher::matchKmers (KmerMatcher.cpp:123-481).  One thread: the whole list is one split.  T threads: the sorted
// list is cut into T query splits at amino-acid-part changes (:157-193 cut at the `split` checkpoints' amino-acid
// parts; any amino-acid boundary gives the same matches since a query's candidates depend on its own amino-acid run
// only), each thread fills a private buffer, the buffers are concatenated (the reference's threads reserve slices of
// one shared buffer; match order is unspecified either way).
size_t orc_match_kmers(orc_db *db, const orc_kmer *q, size_t queryKmerNum, orc_match *out, size_t cap) {
    const uint64_t AAMASK = ~0xFFFFFFull;
    size_t blank = 0;
    for (size_t i = 0; i < queryKmerNum; i++) { if (qi_seq(q[i].qinfo) == 0) blank++; else break; }
    queryKmerNum -= blank;
    if (queryKmerNum == 0) return 0;
    const int T = g_orc_threads;
    if (T <= 1 || queryKmerNum < (size_t)T * 1024) return match_split(db, q, blank, blank + queryKmerNum - 1, out, cap);
    std::vector<size_t> cut;
    cut.push_back(blank);
    for (int t = 1; t < T; t++) {
        size_t c = blank + queryKmerNum * (size_t)t / (size_t)T;
        while (c < blank + queryKmerNum && (q[c].value & AAMASK) == (q[c - 1].value & AAMASK)) c++;
        if (c > cut.back() && c < blank + queryKmerNum) cut.push_back(c);
    }
    cut.push_back(blank + queryKmerNum);
    const int S = (int)cut.size() - 1;
    std::vector<RawBuf<orc_match>> part((size_t)S);
    std::vector<size_t> cnt((size_t)S, 0), first((size_t)S + 1, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(T)
    for (int t = 0; t < S; t++) {
        RawBuf<orc_match> &v = part[(size_t)t];
        v.resize((cut[(size_t)t + 1] - cut[(size_t)t]) * 2 + 1024);
        size_t m = match_split(db, q, cut[(size_t)t], cut[(size_t)t + 1] - 1, v.data(), v.size());
        if (m > v.size()) { v.resize(m); m = match_split(db, q, cut[(size_t)t], cut[(size_t)t + 1] - 1, v.data(), v.size()); }
        cnt[(size_t)t] = m;
    }
    for (int t = 0; t < S; t++) first[(size_t)t + 1] = first[(size_t)t] + cnt[(size_t)t];
#pragma omp parallel for schedule(static, 1) num_threads(T)
    for (int t = 0; t < S; t++) {
        size_t n = first[(size_t)t], c = cnt[(size_t)t];
        if (n < cap) memcpy(out + n, part[(size_t)t].data(), std::min(c, cap - n) * sizeof(orc_match));
    }
    return first[(size_t)S];
}

// KmerMatcher::compareMatches / sortMatches (KmerMatcher.cpp:1071-1078, 1149-1166)
void orc_sort_matches(orc_match *m, size_t n) {
    auto cmp = [](const orc_match &a, const orc_match &b) {
        if (qi_seq(a.qinfo) != qi_seq(b.qinfo)) return qi_seq(a.qinfo) < qi_seq(b.qinfo);
        if (a.species_id != b.species_id) return a.species_id < b.species_id;
        if (qi_frame(a.qinfo) != qi_frame(b.qinfo)) return qi_frame(a.qinfo) < qi_frame(b.qinfo);
        if (qi_pos(a.qinfo) != qi_pos(b.qinfo)) return qi_pos(a.qinfo) < qi_pos(b.qinfo);
        if (a.hamming != b.hamming) return a.hamming < b.hamming;
        return a.dna < b.dna;
    };
    if (g_orc_threads > 1) __gnu_parallel::sort(m, m + n, cmp); else std::sort(m, m + n, cmp);
}

} // extern "C"

/* ------------------------------------------------------------------ */
/* Taxonomer                                                           */
/* ------------------------------------------------------------------ */
namespace {

// Match.h:32-87
inline float codon_score(int h) { return h == 0 ? 3.0f : 2.0f - 0.5f * h; }
float match_score(const orc_match &m) { float s = 0; for (int c = 0; c < 8; c++) s += codon_score((m.right_end_hamming >> (c * 2)) & 3); return s; }
float right_part_score(const orc_match &m, int range) { float s = 0; for (int c = 0; c < range; c++) s += codon_score((m.right_end_hamming >> (c * 2)) & 3); return s; }
float left_part_score(const orc_match &m, int range) { float s = 0; for (int c = 0; c < range; c++) s += codon_score((m.right_end_hamming >> (14 - c * 2)) & 3); return s; }
int right_part_ham(const orc_match &m, int range) { int s = 0; for (int i = 0; i < range; i++) s += (m.right_end_hamming >> (i * 2)) & 3; return s; }
int left_part_ham(const orc_match &m, int range) { int s = 0; for (int i = 0; i < range; i++) s += (m.right_end_hamming >> (14 - i * 2)) & 3; return s; }

struct MatchPath {            // Taxonomer.h:34-57
    int start, end; float score; int hammingDist; int depth; const orc_match *startMatch, *endMatch;
    MatchPath() : start(0), end(0), score(0.f), hammingDist(0), depth(0), startMatch(nullptr), endMatch(nullptr) {}
    explicit MatchPath(const orc_match *m) : start((int)qi_pos(m->qinfo)), end((int)qi_pos(m->qinfo) + 23), score(match_score(*m)),
        hammingDist(m->hamming), depth(1), startMatch(m), endMatch(m) {}
};

struct Taxonomer {
    const orc_params &par; const orc_taxonomy *tax; int kmerFormat;
    int minConsCnt, minConsCntEuk, eukaryota; float tieRatio;
    int denominator, bitsPerCodon = 3, totalDnaBits = 24, maxCodonShift, dnaShift;
    std::vector<MatchPath> matchPaths, combined, local;
    std::vector<bool> connectedToNext;
    std::vector<int> maxSpecies;
    bool ambiguous = false;

    // Taxonomer ctor (Taxonomer.cpp:12-84)
    Taxonomer(const orc_params &p, const orc_taxonomy *t) : par(p), tax(t), kmerFormat(p.kmer_format) {
        minConsCnt = p.min_cons_cnt; minConsCntEuk = p.min_cons_cnt_euk; eukaryota = t->eukaryota; tieRatio = p.tie_ratio;
        if (p.syncmer) { dnaShift = (8 - p.smer_len) * 3; maxCodonShift = 8 - p.smer_len; }
        else { dnaShift = 3; maxCodonShift = 1; }
        denominator = (p.seq_mode == 1 || p.seq_mode == 2) ? 100 : 1000;
    }
    // Taxonomer.cpp:650-669
    static float scoreInc(uint16_t h, int shift) { float s = 0; for (int i = 0; i < shift; i++) s += codon_score((h >> (i * 2)) & 3); return s; }
    static int hamInc(uint16_t h, int shift) { int s = 0; for (int i = 0; i < shift; i++) s += (h >> (i * 2)) & 3; return s; }
    // Taxonomer.cpp:684-699
    bool isConsecutive(const orc_match *a, const orc_match *b, int shift) const {
        return (a->dna >> (bitsPerCodon * shift)) == (b->dna & ((1U << (totalDnaBits - bitsPerCodon * shift)) - 1));
    }
    bool isConsecutive2(const orc_match *a, const orc_match *b, int shift) const {
        return (a->dna & ((1U << (totalDnaBits - bitsPerCodon * shift)) - 1)) == (b->dna >> (bitsPerCodon * shift));
    }

    // Taxonomer::getMatchPaths (Taxonomer.cpp:487-648); the forward and the
    // reverse branch differ only in the argument order of the overlap test.
    void getMatchPaths(const orc_match *ml, size_t start, size_t end, std::vector<MatchPath> &outPaths, int speciesId) {
        size_t i = start;
        uint32_t currPos = qi_pos(ml[start].qinfo);
        bool fwd = qi_frame(ml[start].qinfo) < 3;
        int MIN_DEPTH = minConsCnt;
        if (tax->isAncestor(eukaryota, speciesId)) MIN_DEPTH = minConsCntEuk;
        connectedToNext.assign(end - start + 1, false);
        local.clear(); local.resize(end - start + 1);
        size_t curS = i;
        while (i < end && qi_pos(ml[i].qinfo) == currPos) { local[i - start] = MatchPath(ml + i); ++i; }
        size_t curE = i;
        while (i < end) {
            uint32_t nextPos = qi_pos(ml[i].qinfo);
            size_t nxtS = i;
            while (i < end && nextPos == qi_pos(ml[i].qinfo)) { local[i - start] = MatchPath(ml + i); ++i; }
            size_t nxtE = i;
            int shift = (int)(nextPos - currPos) / 3;
            if (shift > 0 && shift <= maxCodonShift) {
                for (size_t nx = nxtS; nx < nxtE; nx++) {
                    float inc = scoreInc(ml[nx].right_end_hamming, shift);
                    const MatchPath *best = nullptr; float bestScore = 0;
                    for (size_t cu = curS; cu < curE; ++cu) {
                        bool cons;
                        if (kmerFormat == 2) cons = fwd ? isConsecutive2(ml + cu, ml + nx, shift) : isConsecutive2(ml + nx, ml + cu, shift);
                        else                 cons = fwd ? isConsecutive(ml + cu, ml + nx, shift) : isConsecutive(ml + nx, ml + cu, shift);
                        if (cons) {
                            connectedToNext[cu - start] = true;
                            if (local[cu - start].score > bestScore) { best = &local[cu - start]; bestScore = local[cu - start].score; }
                        }
                    }
                    if (best != nullptr) {
                        MatchPath &p = local[nx - start];
                        p.start = best->start; p.score = best->score + inc;
                        p.hammingDist = best->hammingDist + hamInc(ml[nx].right_end_hamming, shift);
                        p.depth = best->depth + shift; p.startMatch = best->startMatch;
                    }
                }
            }
            for (size_t cu = curS; cu < curE; ++cu)
                if (!connectedToNext[cu - start] && local[cu - start].depth >= MIN_DEPTH) outPaths.push_back(local[cu - start]);
            if (i == end)
                for (size_t nx = nxtS; nx < nxtE; ++nx)
                    if (local[nx - start].depth >= MIN_DEPTH) outPaths.push_back(local[nx - start]);
            curS = nxtS; curE = nxtE; currPos = nextPos;
        }
    }

    // Taxonomer.cpp:470-485
    static bool overlapped(const MatchPath &a, const MatchPath &b) { return !((a.end < b.start) || (b.end < a.start)); }
    static void trim(MatchPath &p1, const MatchPath &p2, int ov) {
        if (p1.start < p2.start) {
            p1.end = p2.start - 1;
            p1.hammingDist = std::max(0, p1.hammingDist - right_part_ham(*p1.endMatch, ov / 3));
            p1.score = p1.score - right_part_score(*p1.endMatch, ov / 3) - (ov % 3);
        } else {
            p1.start = p2.end + 1;
            p1.hammingDist = std::max(0, p1.hammingDist - left_part_ham(*p1.startMatch, ov / 3));
            p1.score = p1.score - left_part_score(*p1.startMatch, ov / 3) - (ov % 3);
        }
    }
    // Taxonomer::combineMatchPaths (Taxonomer.cpp:410-468)
    float combineMatchPaths(size_t pathStart, size_t combStart, int readLength) {
        auto cmp = [](const MatchPath &a, const MatchPath &b) {
            if (a.score != b.score) return a.score > b.score;
            if (a.hammingDist != b.hammingDist) return a.hammingDist < b.hammingDist;
            return a.start > b.start;
        };
        // flag inputs on which std::sort's instability could matter (Appendix B.13)
        if (matchPaths.size() - pathStart > 16) {
            std::vector<MatchPath> tmp(matchPaths.begin() + (long)pathStart, matchPaths.end());
            std::stable_sort(tmp.begin(), tmp.end(), cmp);
            for (size_t i = 1; i < tmp.size(); i++)
                if (!cmp(tmp[i - 1], tmp[i]) && !cmp(tmp[i], tmp[i - 1]) &&
                    (tmp[i - 1].end != tmp[i].end || tmp[i - 1].endMatch->right_end_hamming != tmp[i].endMatch->right_end_hamming ||
                     tmp[i - 1].startMatch->right_end_hamming != tmp[i].startMatch->right_end_hamming)) ambiguous = true;
        }
        std::sort(matchPaths.begin() + (long)pathStart, matchPaths.end(), cmp);
        float score = 0;
        for (size_t i = pathStart; i < matchPaths.size(); i++) {
            if (combStart == combined.size()) { combined.push_back(matchPaths[i]); score += matchPaths[i].score; }
            else {
                bool isOv = false;
                for (size_t j = combStart; j < combined.size(); j++) {
                    if (overlapped(matchPaths[i], combined[j])) {
                        int ovLen = std::min(matchPaths[i].end, combined[j].end) - std::max(matchPaths[i].start, combined[j].start) + 1;
                        if (ovLen == matchPaths[i].end - matchPaths[i].start + 1) { isOv = true; break; }
                        if (ovLen < 24) { trim(matchPaths[i], combined[j], ovLen); continue; }
                        else { isOv = true; break; }
                    }
                }
                if (!isOv) { combined.push_back(matchPaths[i]); score += matchPaths[i].score; }
            }
        }
        return score / readLength;
    }

    struct TaxonScore { int taxId = 0; float score = 0.f; int hammingDist = 0; bool LCA = false; };

    // Taxonomer::getBestSpeciesMatches (Taxonomer.cpp:316-408)
    TaxonScore getBestSpeciesMatches(std::pair<size_t, size_t> &bestRange, const orc_match *ml, size_t end, size_t offset, int queryLength) {
        matchPaths.clear(); combined.clear();
        std::vector<std::pair<int, float>> sp2score;
        TaxonScore best; float bestSp = 0;
        size_t i = offset, meaningful = 0;
        while (i < end + 1) {
            int cur = ml[i].species_id;
            size_t start = i;
            size_t prevPathSize = matchPaths.size();
            while ((i < end + 1) && cur == ml[i].species_id) {
                uint32_t fr = qi_frame(ml[i].qinfo);
                size_t fs = i;
                while ((i < end + 1) && cur == ml[i].species_id && fr == qi_frame(ml[i].qinfo)) i++;
                if (i - fs > 1) getMatchPaths(ml, fs, i, matchPaths, cur);
            }
            size_t pathSize = matchPaths.size();
            if (pathSize > prevPathSize) {
                float score = combineMatchPaths(prevPathSize, combined.size(), queryLength);
                score = std::min(score, 1.0f);
                if (score < par.min_score) continue;
                sp2score.emplace_back(cur, score);
                if (score > 0.f) meaningful++;
                if (score > bestSp) { bestSp = score; bestRange = std::make_pair(start, i); }
            }
        }
        if (meaningful == 0) { best.score = 0; return best; }
        maxSpecies.clear();
        for (size_t k = 0; k < sp2score.size(); k++)
            if (sp2score[k].second >= bestSp * tieRatio) { maxSpecies.push_back(sp2score[k].first); best.score += sp2score[k].second; }
        if (maxSpecies.size() > 1) {
            best.LCA = true; best.taxId = tax->lca(maxSpecies); best.score /= maxSpecies.size();
            return best;
        }
        best.taxId = maxSpecies[0];
        return best;
    }

    // Taxonomer::filterRedundantMatches (Taxonomer.cpp:205-241)
    void filterRedundantMatches(const orc_match *ml, const std::pair<size_t, size_t> &range, std::map<int, unsigned> &taxCnt, int queryLength) {
        size_t maxQ = (size_t)(queryLength + 3) / (size_t)dnaShift;
        std::vector<const orc_match *> bestM(maxQ + 1, nullptr); std::vector<int> bestTax(maxQ + 1, 0); std::vector<uint8_t> minH(maxQ + 1, 255);
        for (size_t i = range.first; i < range.second; i++) {
            size_t qt = qi_pos(ml[i].qinfo) / (size_t)dnaShift;
            if (qt > maxQ) { bestM.resize(qt + 1, nullptr); bestTax.resize(qt + 1, 0); minH.resize(qt + 1, 255); maxQ = qt; }
            uint8_t h = ml[i].hamming;
            if (bestM[qt] == nullptr) { bestM[qt] = ml + i; bestTax[qt] = ml[i].target_id; minH[qt] = h; }
            else if (h < minH[qt]) { bestM[qt] = ml + i; bestTax[qt] = ml[i].target_id; minH[qt] = h; }
            else if (h == minH[qt]) bestTax[qt] = tax->lca(bestTax[qt], ml[i].target_id);
        }
        for (size_t i = 0; i <= maxQ; ++i) if (bestM[i] != nullptr) taxCnt[bestTax[i]]++;
    }

    struct Clade { unsigned taxCount = 0, cladeCount = 0; std::vector<int> children; };
    // Taxonomer.cpp:273-314
    void speciesCladeCounts(const std::map<int, unsigned> &taxCnt, std::unordered_map<int, Clade> &cc, int species) {
        for (auto &kv : taxCnt) {
            int t = tax->canon(kv.first);
            cc[t].taxCount = kv.second; cc[t].cladeCount += kv.second;
            int guard = 0;
            while (t != species && guard++ < 1000) {
                int par = tax->parent[(size_t)t];
                auto &ch = cc[par].children;
                if (std::find(ch.begin(), ch.end(), t) == ch.end()) ch.push_back(t);
                cc[par].cladeCount += kv.second;
                if (par == t) break;
                t = par;
            }
        }
    }
    int BFS(const std::unordered_map<int, Clade> &cc, int root, unsigned maxCnt) {
        unsigned maxCnt2 = maxCnt;
        if (cc.at(root).children.empty()) return root;
        std::vector<int> bestCh;
        for (int c : cc.at(root).children) {
            unsigned cur = cc.at(c).cladeCount;
            if (cur > maxCnt) { bestCh.clear(); bestCh.push_back(c); maxCnt = cur; }
            else if (cur == maxCnt) bestCh.push_back(c);
        }
        if (bestCh.size() == 1) return BFS(cc, bestCh[0], maxCnt2);
        return root;
    }
    // Taxonomer.cpp:252-271
    int lowerRankClassification(const std::map<int, unsigned> &taxCnt, int sp, int queryLength) {
        unsigned minSub = (unsigned)((queryLength - 1) / denominator);
        std::unordered_map<int, Clade> cc;
        speciesCladeCounts(taxCnt, cc, sp);
        if (par.accession_level == 2) {
            for (auto it = cc.begin(); it != cc.end(); it++) {
                const std::string &r = tax->rank[(size_t)it->first];
                if (r == "" || r == "accession") {
                    auto &ch = cc[tax->parent[(size_t)it->first]].children;
                    auto f = std::find(ch.begin(), ch.end(), it->first);
                    if (f != ch.end()) ch.erase(f);
                }
            }
        }
        return BFS(cc, sp, minSub);
    }

    // Taxonomer::chooseBestTaxon (Taxonomer.cpp:130-202)
    void chooseBestTaxon(size_t offset, size_t end, const orc_match *ml, int qlen, int qlen2,
                         orc_result &r, std::map<int, unsigned> &taxCnt) {
        ambiguous = false;
        std::pair<size_t, size_t> range;
        TaxonScore ss = getBestSpeciesMatches(range, ml, end, offset, qlen + qlen2);
        r.ambiguous = ambiguous;
        if (ss.score == 0 || ss.score < par.min_score) { r.is_classified = 0; r.classification = 0; r.score = ss.score; return; }
        if (ss.LCA) { r.is_classified = 1; r.classification = ss.taxId; r.score = ss.score; return; }
        taxCnt.clear();
        filterRedundantMatches(ml, range, taxCnt, qlen + qlen2);
        if (ss.score < par.min_sp_score) {
            r.is_classified = 1;
            int spNode = tax->atRank(ss.taxId, "species");
            r.classification = tax->parent[(size_t)tax->canon(spNode)];
            r.score = ss.score; return;
        }
        r.is_classified = 1; r.score = ss.score;
        r.classification = lowerRankClassification(taxCnt, ss.taxId, qlen + qlen2);
    }
};
} // namespace

extern "C" size_t orc_score(const orc_db *db, const orc_taxonomy *tax, const orc_params *p,
                            const orc_match *ml, size_t nM, size_t nReads, const int32_t *qlen,
                            const int32_t *qlen2, orc_result *res, int32_t *tcTax, uint32_t *tcCnt, size_t cap) {
    (void)db;
    for (size_t i = 0; i < nReads; i++) {
        res[i].classification = 0; res[i].score = 0; res[i].query_length = qlen[i]; res[i].query_length2 = qlen2 ? qlen2[i] : 0;
        res[i].is_classified = 0; res[i].ambiguous = 0; res[i].n_taxcnt = 0; res[i].taxcnt_off = 0;
    }
    // Classifier::assignTaxonomy block cutting (Classifier.cpp:166-186)
    struct Block { size_t s, e; uint32_t id; };
    std::vector<Block> blocks;
    size_t idx = 0;
    while (idx < nM) {
        uint32_t cur = qi_seq(ml[idx].qinfo);
        size_t s = idx;
        while (idx < nM && qi_seq(ml[idx].qinfo) == cur) ++idx;
        blocks.push_back({s, idx - 1, cur});
    }
    // blocks are independent (Classifier.cpp:187-203: omp for schedule(dynamic, 1), one Taxonomer per thread)
    const int T = std::max(1, std::min<int>(g_orc_threads, (int)(blocks.size() / 64 + 1)));
    std::vector<std::vector<std::pair<int, unsigned>>> tcs(blocks.size());
#pragma omp parallel num_threads(T)
    {
        Taxonomer tx(*p, tax);
#pragma omp for schedule(dynamic, 64)
        for (size_t b = 0; b < blocks.size(); b++) {
            size_t rIdx = (size_t)blocks[b].id - 1;
            std::map<int, unsigned> taxCnt;
            tx.chooseBestTaxon(blocks[b].s, blocks[b].e, ml, qlen[rIdx], qlen2 ? qlen2[rIdx] : 0, res[rIdx], taxCnt);
            tcs[b].assign(taxCnt.begin(), taxCnt.end());
        }
    }
    size_t w = 0;
    for (size_t b = 0; b < blocks.size(); b++) {
        size_t rIdx = (size_t)blocks[b].id - 1;
        res[rIdx].taxcnt_off = (uint32_t)w; res[rIdx].n_taxcnt = (uint16_t)std::min<size_t>(tcs[b].size(), 65535);
        for (auto &kv : tcs[b]) { if (w < cap) { tcTax[w] = kv.first; tcCnt[w] = kv.second; } w++; }
    }
    return w;
}

/* The whole loop body of Classifier::startClassify (Classifier.cpp:81-125) for one batch, buffers kept inside (what
 * bench.py's cpu_baseline times: no Python copies between the stages).  stage_s[5] = seconds of extract, sort, match,
 * sortMatches, assignTaxonomy.  Returns the number of taxcnt entries (needed size if > cap). */
#include <chrono>
extern "C" size_t orc_classify_batch(orc_db *db, const orc_taxonomy *tax, const orc_params *p, const char *bases, const uint64_t *offs,
                                     const char *bases2, const uint64_t *offs2, size_t n_reads, orc_result *res, int32_t *tcTax,
                                     uint32_t *tcCnt, size_t cap, double *stage_s, size_t *n_kmers, size_t *n_matches) {
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t0 = now();
    size_t nb = (size_t)offs[n_reads] + (offs2 ? (size_t)offs2[n_reads] : 0);
    RawBuf<orc_kmer> k(2 * nb + 64);
    std::vector<int32_t> ql(n_reads), ql2(n_reads);
    size_t nk = orc_extract_batch(bases, offs, bases2, offs2, n_reads, p, k.data(), k.size(), ql.data(), ql2.data());
    double t1 = now();
    orc_sort_kmers(k.data(), nk);
    double t2 = now();
    RawBuf<orc_match> m(nk + nk / 2 + 1024);
    size_t nm = orc_match_kmers(db, k.data(), nk, m.data(), m.size());
    if (nm > m.size()) { m.resize(nm); nm = orc_match_kmers(db, k.data(), nk, m.data(), m.size()); }   // Classifier.cpp:127-131 retry
    double t3 = now();
    orc_sort_matches(m.data(), nm);
    double t4 = now();
    size_t w = orc_score(db, tax, p, m.data(), nm, n_reads, ql.data(), ql2.data(), res, tcTax, tcCnt, cap);
    double t5 = now();
    if (stage_s) { stage_s[0] = t1 - t0; stage_s[1] = t2 - t1; stage_s[2] = t3 - t2; stage_s[3] = t4 - t3; stage_s[4] = t5 - t4; }
    if (n_kmers) *n_kmers = nk;
    if (n_matches) *n_matches = nm;
    return w;
}

/* ------------------------------------------------------------------ */
/* Reporter                                                            */
/* ------------------------------------------------------------------ */
#include <functional>
#include <iostream>
#include <sstream>
extern "C" {
// Reporter::writeReadClassification (Reporter.cpp:35-80), lineage off.
// names: '\n'-separated read names.  Returns 0 on success.
// TaxonomyWrapper::taxLineage2(node, infoAsName = true) (TaxonomyWrapper.cpp:431-454); short ranks of
// ExtendedShortRanks (TaxonomyWrapper.h:9-26), "-" for every other rank (findShortRank2, :423-429)
static std::string orc_lineage(const orc_taxonomy *tax, int taxid) {
    static const std::map<std::string, std::string> shortRanks = {
        {"subspecies", "ss"}, {"species", "s"}, {"subgenus", "sg"}, {"genus", "g"}, {"subfamily", "sf"}, {"family", "f"},
        {"suborder", "so"}, {"order", "o"}, {"subclass", "sc"}, {"class", "c"}, {"subphylum", "sp"}, {"phylum", "p"},
        {"subkingdom", "sk"}, {"kingdom", "k"}, {"superkingdom", "d"}, {"domain", "d"}, {"realm", "r"}};
    std::vector<int> vec;
    int node = tax->canon(taxid);
    if (node < 0) return std::string();
    do {
        vec.push_back(node);
        node = tax->parent[(size_t)node];
    } while (tax->parent[(size_t)node] != node);
    std::string out;
    for (int i = (int)vec.size() - 1; i >= 0; --i) {
        auto it = shortRanks.find(tax->rank[(size_t)vec[(size_t)i]]);
        out += it == shortRanks.end() ? std::string("-") : it->second;
        out += '_';
        out += tax->name[(size_t)vec[(size_t)i]];
        if (i > 0) out += ";";
    }
    return out;
}
int orc_write_classifications2(const char *path, const orc_taxonomy *tax, const char *names, size_t n_reads, const orc_result *res,
                               const int32_t *tcTax, const uint32_t *tcCnt, int printLineage) {
    std::ofstream f(path);
    if (!f) return 1;
    f << "#is_classified\tname\ttaxID\tquery_length\tscore\trank";
    if (printLineage) f << "\tlineage";
    f << "\ttaxID:match_count\n";
    std::istringstream nm(names);
    std::string name;
    for (size_t i = 0; i < n_reads; i++) {
        std::getline(nm, name);
        const orc_result &r = res[i];
        bool cls = r.is_classified != 0;
        if (cls) {
            int c = tax->canon(r.classification);
            f << cls << "\t" << name << "\t" << r.classification << "\t" << r.query_length + r.query_length2 << "\t" << r.score << "\t"
              << (c >= 0 ? tax->rank[(size_t)c] : std::string()) << "\t";
            if (printLineage) f << orc_lineage(tax, r.classification) << "\t";
            for (uint32_t k = 0; k < r.n_taxcnt; k++) f << tcTax[r.taxcnt_off + k] << ":" << tcCnt[r.taxcnt_off + k] << " ";
            f << "\n";
        } else {
            f << cls << "\t" << name << "\t" << r.classification << "\t" << r.query_length + r.query_length2 << "\t" << r.score << "\t-\t";
            if (printLineage) f << "-\t";
            f << "-\t\n";
        }
    }
    return 0;
}
int orc_write_classifications(const char *path, const orc_taxonomy *tax, const char *names, size_t n_reads, const orc_result *res,
                              const int32_t *tcTax, const uint32_t *tcCnt) {
    return orc_write_classifications2(path, tax, names, n_reads, res, tcTax, tcCnt, 0);
}
// Reporter::writeReportFile / writeReport (Reporter.cpp:115-193) with clade counts as
// NcbiTaxonomy::getCladeCounts; children ordered by (clade count desc, taxid asc).
int orc_write_report(const char *path, const orc_taxonomy *tax, size_t n_reads, const orc_result *res) {
    std::map<int, unsigned> taxCounts;
    for (size_t i = 0; i < n_reads; i++) ++taxCounts[res[i].classification];      // Classifier.cpp:201-203
    std::unordered_map<int, unsigned> clade, own;
    std::unordered_map<int, std::vector<int>> children;
    for (auto &kv : taxCounts) {
        own[kv.first] = kv.second;
        if (kv.first == 0) { clade[0] += kv.second; continue; }
        int t = tax->canon(kv.first);
        int guard = 0;
        while (t >= 0 && guard++ < 1000) {
            bool fresh = clade.find(t) == clade.end();
            clade[t] += kv.second;
            int p = tax->parent[(size_t)t];
            if (p == t) break;
            if (fresh) children[p].push_back(t);
            t = p;
        }
    }
    FILE *fp = fopen(path, "w");
    if (!fp) return 1;
    fprintf(fp, "#clade_proportion\tclade_count\ttaxon_count\trank\ttaxID\tname\n");
    if (clade.count(0) && clade[0] > 0) fprintf(fp, "%.4f\t%i\t%i\tno rank\t0\tunclassified\n", 100 * clade[0] / double(n_reads), (int)clade[0], (int)own[0]);
    std::function<void(int, int)> rec = [&](int t, int depth) {
        auto it = clade.find(t);
        if (it == clade.end() || it->second == 0) return;
        fprintf(fp, "%.4f\t%i\t%i\t%s\t%i\t%s%s\n", 100 * it->second / double(n_reads), (int)it->second, (int)(own.count(t) ? own[t] : 0),
                tax->rank[(size_t)t].c_str(), t, std::string(2 * (size_t)depth, ' ').c_str(), tax->name[(size_t)t].c_str());
        std::vector<int> ch = children[t];
        std::sort(ch.begin(), ch.end(), [&](int a, int b) { return clade[a] != clade[b] ? clade[a] > clade[b] : a < b; });
        for (int c : ch) rec(c, depth + 1);
    };
    rec(1, 0);
    fclose(fp);
    return 0;
}
}
