/*
 * oracle.h -- C API of the CPU oracle (TEST INFRASTRUCTURE ONLY).
 *
 * The oracle is an independent CPU restatement of the `metabuli classify`
 * hot path of steineggerlab/Metabuli.  It follows the *structure* of the
 * reference (streaming scanners, streaming merge over the delta-coded index,
 * the Taxonomer decision tree) so that agreement with the HIP engine -- which
 * is built from a functional specification instead -- is meaningful.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The product library (libmtb.so) never links it.
 *
 * PARITY PINNING STATUS (see DESIGN.md "Oracle"):
 *   - codon tables / base maps: PINNED against the reference's own
 *     GeneticCode.h compiled verbatim (oracle/_ref/ref_codon_dump).
 *   - Hamming tables: PINNED by parsing the numeric literals of
 *     KmerMatcher.h:66-158 in tests/test_oracle_tables.py (container only).
 *   - scanners, diffIdx codec, join, scorer: "parity unpinned" -- the
 *     reference ships no golden vectors for this path and cannot be built
 *     here without stand-in headers for the absent MMseqs2 submodule.
 */
#ifndef MTB_ORACLE_H
#define MTB_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { uint64_t value; uint64_t qinfo; } orc_kmer;   /* Kmer.h:24-47 */

typedef struct {                 /* Match.h:9-25 payload, packed to 24 B */
    uint64_t qinfo;
    int32_t  target_id;
    int32_t  species_id;
    uint32_t dna;
    uint16_t right_end_hamming;
    uint8_t  hamming;
    uint8_t  pad;
} orc_match;

typedef struct {
    int   seq_mode;          /* 1 single, 2 paired, 3 long     (classify.cpp:20) */
    int   syncmer;           /* db.parameters "Syncmer"                          */
    int   smer_len;          /* classify.cpp:12 default 5                        */
    int   kmer_format;       /* only 2 supported by the oracle                   */
    int   min_cons_cnt;      /* classify.cpp default 4                           */
    int   min_cons_cnt_euk;  /* 9                                                */
    float min_score;         /* 0                                                */
    float min_sp_score;      /* 0                                                */
    float tie_ratio;         /* 0.95                                             */
    int   accession_level;   /* 0                                                */
    int   skip_redundancy;   /* db.parameters "Skip_redundancy"                  */
} orc_params;

typedef struct {
    int32_t  classification;   /* internal taxid (0 = unclassified)  */
    float    score;
    int32_t  query_length;     /* used(L1)                            */
    int32_t  query_length2;    /* used(L2) or 0                       */
    uint8_t  is_classified;
    uint8_t  ambiguous;        /* 1 if a std::sort tie could change the result (Appendix B.13) */
    uint16_t n_taxcnt;         /* entries in the taxCnt map           */
    uint32_t taxcnt_off;       /* offset into the taxcnt arrays       */
} orc_result;

/* ---- tables ---------------------------------------------------------- */
/* nuc2aa / nuc2num as [8][8][8] ints (GeneticCode.h:34-193)              */
void orc_codon_tables(int *nuc2aa512, int *nuc2num512);
/* base -> 0..3 / 7 code for forward and reverse-complement strands       */
void orc_base_codes(uint8_t *fwd256, uint8_t *rev256);
/* hammingLookup[8][8] and HAMMING_LUT0..7 (KmerMatcher.h:66-158)          */
void orc_hamming_tables(uint8_t *lookup64, uint16_t *lut8x64);
uint8_t  orc_hamming_sum(uint64_t a, uint64_t b);
uint16_t orc_hammings(uint64_t a, uint64_t b);
uint16_t orc_hammings_reverse(uint64_t a, uint64_t b);

/* ---- extraction (KmerExtractor.cpp:342-373, scanners) ---------------- */
/* one sequence, six frames; returns number of k-mers written (<= cap)    */
size_t orc_extract_read(const char *seq, int len, const orc_params *p,
                        uint32_t seq_id, uint32_t offset,
                        orc_kmer *out, size_t cap);
/* whole batch: reads concatenated in `bases`, offs[n+1]; for seq_mode 2
 * mates in bases2/offs2.  Fills qlen/qlen2 (used lengths).  Returns count. */
size_t orc_extract_batch(const char *bases, const uint64_t *offs,
                         const char *bases2, const uint64_t *offs2,
                         size_t n_reads, const orc_params *p,
                         orc_kmer *out, size_t cap,
                         int32_t *qlen, int32_t *qlen2);
/* sort by (value, seqID)  (Kmer.h:89-94) */
void orc_sort_kmers(orc_kmer *k, size_t n);

/* ---- on-disk DB (IndexCreator.cpp:817-892, 1251-1272) ----------------- */
size_t orc_diffidx_encode(const uint64_t *values, size_t n, uint16_t *out /* cap 5n */);
size_t orc_diffidx_decode(const uint16_t *in, size_t n16, uint64_t *values /* cap n16 */);
/* values must be sorted by (value, species, taxid); writes diffIdx, info,
 * split, taxID_list, db.parameters into dir.  Returns 0 on success.       */
int orc_write_db(const char *dir, const uint64_t *values, const int32_t *taxids,
                 size_t n, int split_num, const orc_params *p);

/* ---- taxonomy (dmp files) --------------------------------------------- */
typedef struct orc_taxonomy orc_taxonomy;
orc_taxonomy *orc_taxonomy_load(const char *names, const char *nodes, const char *merged);
void orc_taxonomy_free(orc_taxonomy *);
int  orc_tax_lca(const orc_taxonomy *, int a, int b);
int  orc_tax_at_rank(const orc_taxonomy *, int taxid, const char *rank);
int  orc_tax_is_ancestor(const orc_taxonomy *, int anc, int child);
int  orc_tax_max_id(const orc_taxonomy *);
int  orc_tax_parent(const orc_taxonomy *, int t);

/* ---- matcher (KmerMatcher.cpp:56-120, 123-481, 1071-1166) ------------- */
typedef struct orc_db orc_db;
orc_db *orc_db_open(const char *dir, const orc_taxonomy *tax, const orc_params *p);
void    orc_db_close(orc_db *);
size_t  orc_db_num_kmers(const orc_db *);
/* sorted query k-mers -> unsorted matches; returns count (needed size if > cap) */
size_t orc_match_kmers(orc_db *, const orc_kmer *sorted, size_t n, orc_match *out, size_t cap);
void   orc_sort_matches(orc_match *m, size_t n);

/* ---- scorer (Classifier.cpp:166-208, Taxonomer.cpp) ------------------- */
/* sorted matches -> per read result; taxcnt arrays sized by caller (cap);
 * returns number of taxcnt entries written.                               */
size_t orc_score(const orc_db *, const orc_taxonomy *, const orc_params *p,
                 const orc_match *sorted, size_t n_matches,
                 size_t n_reads, const int32_t *qlen, const int32_t *qlen2,
                 orc_result *res, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, size_t cap);

/* ---- Reporter (Reporter.cpp:35-80, 115-193) ---------------------------- */
int orc_write_classifications(const char *path, const orc_taxonomy *, const char *names_nl, size_t n_reads,
                              const orc_result *res, const int32_t *taxcnt_tax, const uint32_t *taxcnt_cnt);
/* the same with the --lineage column (TaxonomyWrapper::taxLineage2, TaxonomyWrapper.cpp:431-454) when print_lineage != 0 */
int orc_write_classifications2(const char *path, const orc_taxonomy *, const char *names_nl, size_t n_reads,
                               const orc_result *res, const int32_t *taxcnt_tax, const uint32_t *taxcnt_cnt, int print_lineage);
int orc_write_report(const char *path, const orc_taxonomy *, size_t n_reads, const orc_result *res);

#ifdef __cplusplus
}
#endif
#endif
