/*
 * mtb.hpp -- C++ host-side mirror of the reference's stage interface on top of
 * the C ABI (mtb.h).  Same names, argument meaning and error behaviour as the
 * three member calls Classifier::startClassify makes (src/commons/
 * Classifier.cpp:105-119), so that the reference's caller reads unchanged:
 *
 *   kmerExtractor->extractQueryKmers(queryKmerBuffer, queryList, ...)    KmerExtractor.h:79-85
 *   kmerMatcher->matchKmers(&queryKmerBuffer, &matchBuffer)  -> bool     KmerMatcher.h:228-230
 *   kmerMatcher->sortMatches(&matchBuffer)                               KmerMatcher.h:244
 *   assignTaxonomy(matchBuffer.buffer, n, queryList, par)                Classifier.h:77-80
 *
 * Header-only; link with libmtb.so.  No CPU implementation lives here.
 */
#ifndef MTB_HPP
#define MTB_HPP
#include <cstdint>
#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "mtb.h"

namespace mtb {

typedef mtb_kmer Kmer;      /* Kmer.h:24-47  */
typedef mtb_match Match;    /* Match.h:9-25  */
typedef mtb_params LocalParameters;

/* Buffer<T> (common.h:141-215): malloc'd array + fill count */
template <typename T> struct Buffer {
    T *buffer = nullptr;
    size_t startIndexOfReserve = 0;
    size_t bufferSize = 0;
    explicit Buffer(size_t n = 0) { if (n) reallocateMemory(n); }
    ~Buffer() { free(buffer); }
    void reallocateMemory(size_t n) {
        if (n > bufferSize) { buffer = (T *)realloc(buffer, n * sizeof(T)); bufferSize = n; }
    }
};

/* Query (common.h:94-122): the fields the path assigns */
struct Query {
    int classification = 0;
    float score = 0.f;
    int hammingDist = 0;         /* always 0 in the reference (Appendix B.10) */
    int queryLength = 0, queryLength2 = 0;
    bool isClassified = false;
    std::string name;
    std::map<int, int> taxCnt;
};

/* reads of one batch, concatenated (what KSeqWrapper + loadChunkOfReads hand over) */
struct ReadBatch {
    std::string bases, bases2;
    std::vector<uint64_t> offs{0}, offs2{0};
    std::vector<std::string> names;
    size_t size() const { return offs.size() - 1; }
    void add(const std::string &name, const std::string &seq) { names.push_back(name); bases += seq; offs.push_back(bases.size()); }
    void add_mate(const std::string &seq) { bases2 += seq; offs2.push_back(bases2.size()); }
};

inline void check(mtb_status s) { if (s != MTB_OK) throw std::runtime_error(std::string("mtb: ") + mtb_last_error()); }

class Engine {      /* one GPU: context + resident index (replaces the per-call file streaming) */
public:
    Engine(int device, const std::string &dbDir, const std::string &taxonomyDir, LocalParameters &par) {
        check(mtb_ctx_create(device, nullptr, &ctx));
        open(dbDir, taxonomyDir, par);
    }
    /* two steps, for a host that does something with the context (mtb_ctx_reserve on a helper thread) while the database loads */
    explicit Engine(int device) { check(mtb_ctx_create(device, nullptr, &ctx)); }
    void open(const std::string &dbDir, const std::string &taxonomyDir, LocalParameters &par) {
        check(mtb_index_open(ctx, dbDir.c_str(), taxonomyDir.empty() ? nullptr : taxonomyDir.c_str(), &par, &index));
    }
    /* range `part` of `n_parts` of a database larger than one HBM (mtb_index_open_part) */
    Engine(int device, const std::string &dbDir, const std::string &taxonomyDir, LocalParameters &par, uint32_t part, uint32_t n_parts) {
        check(mtb_ctx_create(device, nullptr, &ctx));
        check(mtb_index_open_part(ctx, dbDir.c_str(), taxonomyDir.empty() ? nullptr : taxonomyDir.c_str(), &par, part, n_parts, &index));
    }
    /* a further GPU of the node: the index is copied from an engine that already holds it (peer copies, mtb_index_clone) */
    Engine(int device, Engine &loaded) {
        check(mtb_ctx_create(device, nullptr, &ctx));
        check(mtb_index_clone(loaded.index, ctx, &index));
    }
    ~Engine() { mtb_index_close(index); mtb_ctx_destroy(ctx); }
    mtb_ctx *ctx = nullptr;
    mtb_index *index = nullptr;
};

class KmerExtractor {
public:
    KmerExtractor(Engine &e, const LocalParameters &par) : eng(e), par(par) {}
    /* KmerExtractor::extractQueryKmers: fills kmerBuffer with the SORTED query k-mers and
     * queryList with name / queryLength / queryLength2 */
    void extractQueryKmers(Buffer<Kmer> &kmerBuffer, std::vector<Query> &queryList, const ReadBatch &reads) {
        size_t n = reads.size();
        queryList.assign(n, Query());
        std::vector<int32_t> ql(n), ql2(n);
        uint64_t cnt = 0;
        const bool paired = par.seq_mode == 2;
        for (;;) {
            mtb_status s = mtb_extract(eng.ctx, &par, reads.bases.data(), reads.offs.data(), paired ? reads.bases2.data() : nullptr,
                                       paired ? reads.offs2.data() : nullptr, n, kmerBuffer.buffer, kmerBuffer.bufferSize, &cnt,
                                       ql.data(), ql2.data());
            if (s == MTB_ERR_CAPACITY) { kmerBuffer.reallocateMemory(cnt); continue; }
            check(s);
            break;
        }
        kmerBuffer.startIndexOfReserve = cnt;
        check(mtb_sort_kmers(eng.ctx, kmerBuffer.buffer, cnt));
        for (size_t i = 0; i < n; i++) { queryList[i].name = reads.names[i]; queryList[i].queryLength = ql[i]; queryList[i].queryLength2 = ql2[i]; }
    }
private:
    Engine &eng; LocalParameters par;
};

class KmerMatcher {
public:
    explicit KmerMatcher(Engine &e) : eng(e) {}
    /* returns false when matchBuffer is too small, like the reference (KmerMatcher.cpp:474-476);
     * neededSize() then tells the exact capacity */
    bool matchKmers(Buffer<Kmer> *queryKmerBuffer, Buffer<Match> *matchBuffer) {
        uint64_t cnt = 0;
        mtb_status s = mtb_match_kmers(eng.ctx, eng.index, queryKmerBuffer->buffer, queryKmerBuffer->startIndexOfReserve,
                                       matchBuffer->buffer, matchBuffer->bufferSize, &cnt);
        needed = cnt;
        if (s == MTB_ERR_CAPACITY) return false;
        check(s);
        matchBuffer->startIndexOfReserve = cnt;
        totalMatchCnt += cnt;
        return true;
    }
    void sortMatches(Buffer<Match> *matchBuffer, size_t numOfReads) {
        check(mtb_sort_matches(eng.ctx, matchBuffer->buffer, matchBuffer->startIndexOfReserve, numOfReads));
    }
    size_t neededSize() const { return needed; }
    size_t getTotalMatchCnt() const { return totalMatchCnt; }
private:
    Engine &eng; size_t needed = 0, totalMatchCnt = 0;
};

class Classifier {
public:
    Classifier(Engine &e, const LocalParameters &par) : eng(e), par(par) {}
    /* Classifier::assignTaxonomy: sorted matches -> queryList; also taxCounts[classification]++ */
    void assignTaxonomy(const Match *matchList, size_t numOfMatches, std::vector<Query> &queryList) {
        size_t n = queryList.size();
        std::vector<int32_t> ql(n), ql2(n);
        for (size_t i = 0; i < n; i++) { ql[i] = queryList[i].queryLength; ql2[i] = queryList[i].queryLength2; }
        std::vector<mtb_result> res(n);
        std::vector<int32_t> tt(numOfMatches + 16); std::vector<uint32_t> tc(numOfMatches + 16);
        uint64_t ntc = 0;
        check(mtb_score(eng.ctx, eng.index, &par, matchList, numOfMatches, n, ql.data(), ql2.data(), res.data(), tt.data(), tc.data(), tt.size(), &ntc));
        fill(queryList, res, tt, tc);
    }
    /* the fused loop body of startClassify (Classifier.cpp:81-125) for one batch */
    void classifyBatch(const ReadBatch &reads, std::vector<Query> &queryList) {
        size_t n = reads.size();
        queryList.assign(n, Query());
        std::vector<mtb_result> res(n);
        const bool paired = par.seq_mode == 2;
        size_t cap = 64 * n + 1024;
        std::vector<int32_t> tt; std::vector<uint32_t> tc;
        uint64_t ntc = 0;
        for (;;) {
            tt.resize(cap); tc.resize(cap);
            mtb_status s = mtb_classify_batch(eng.ctx, eng.index, &par, reads.bases.data(), reads.offs.data(), paired ? reads.bases2.data() : nullptr,
                                              paired ? reads.offs2.data() : nullptr, n, res.data(), tt.data(), tc.data(), cap, &ntc);
            if (s == MTB_ERR_CAPACITY && ntc > cap) { cap = ntc; continue; }
            check(s);
            break;
        }
        for (size_t i = 0; i < n; i++) queryList[i].name = reads.names[i];
        fill(queryList, res, tt, tc);
    }
    std::map<int, unsigned> &getTaxCounts() { return taxCounts; }
private:
    void fill(std::vector<Query> &q, const std::vector<mtb_result> &res, const std::vector<int32_t> &tt, const std::vector<uint32_t> &tc) {
        for (size_t i = 0; i < q.size(); i++) {
            q[i].classification = res[i].classification; q[i].score = res[i].score; q[i].isClassified = res[i].is_classified != 0;
            q[i].queryLength = res[i].query_length; q[i].queryLength2 = res[i].query_length2;
            q[i].taxCnt.clear();
            for (uint32_t k = 0; k < res[i].n_taxcnt; k++) q[i].taxCnt[tt[res[i].taxcnt_off + k]] = (int)tc[res[i].taxcnt_off + k];
            ++taxCounts[q[i].classification];
        }
    }
    Engine &eng; LocalParameters par; std::map<int, unsigned> taxCounts;
};

} // namespace mtb
#endif
