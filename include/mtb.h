/*
 * mtb.h -- C ABI of the MI355X-native metamer classification engine.
 *
 * Drop-in boundary for the hot path of `metabuli classify`
 * (steineggerlab/Metabuli).  The reference has no FFI; its seam is three C++
 * member calls made by Classifier::startClassify (src/commons/Classifier.cpp:
 * 105-119).  Every entry point below names the reference interface it
 * replaces.  Plain C types only: pointers + sizes, caller-allocated outputs
 * with capacity and a required-size return, status codes instead of exit().
 *
 * Memory spaces: every pointer parameter documented as "host" is ordinary
 * host memory; parameters documented as "device" are HIP device pointers
 * (e.g. torch.Tensor.data_ptr()).  The stage-level entry points take host
 * buffers (they are the parity seam); mtb_classify_batch_device takes device
 * buffers so that a step can be timed with inputs already resident in HBM.
 *
 * Threading: one host thread per mtb_ctx; one mtb_ctx per GPU.
 */
#ifndef MTB_H
#define MTB_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    MTB_OK = 0,
    MTB_ERR_ARG = 1,        /* bad argument                                            */
    MTB_ERR_IO = 2,         /* database / taxonomy file missing or malformed           */
    MTB_ERR_DEVICE = 3,     /* HIP runtime error (message in mtb_last_error)           */
    MTB_ERR_CAPACITY = 4,   /* output buffer too small; required size was returned     */
    MTB_ERR_OOM = 5,        /* not enough HBM for the request                          */
    MTB_ERR_UNSUPPORTED = 6 /* feature of the reference not implemented yet            */
} mtb_status;

/* Kmer (src/commons/Kmer.h:24-47): 16 bytes.
 * qinfo = pos[0:31] | sequenceID[32:60] | frame[61:63]  (Kmer.h:11-16). */
typedef struct { uint64_t value; uint64_t qinfo; } mtb_kmer;

/* Match payload (src/commons/Match.h:9-25) packed to 24 bytes (the reference
 * object is 32 bytes because of its vptr).                                  */
typedef struct {
    uint64_t qinfo;
    int32_t  target_id;          /* taxonomy id stored in `info`              */
    int32_t  species_id;         /* taxId2speciesId[target_id]                */
    uint32_t dna;                /* target value & 0xFFFFFF                   */
    uint16_t right_end_hamming;  /* 2 bits per codon                          */
    uint8_t  hamming;            /* sum of per-codon Hamming distances        */
    uint8_t  pad;
} mtb_match;

/* The subset of LocalParameters (src/commons/LocalParameters.h:157-248) read
 * on the path; defaults are `classify`'s (src/workflow/classify.cpp:10-37),
 * overridden by DBDIR/db.parameters (src/commons/common.cpp:88-133).        */
typedef struct {
    int32_t seq_mode;          /* 1 single-end, 2 paired-end, 3 long read     */
    int32_t syncmer;           /* 0/1                                         */
    int32_t smer_len;          /* 5                                           */
    int32_t kmer_format;       /* 1 (legacy, the CLI default) or 2 (db.parameters) */
    int32_t min_cons_cnt;      /* 4                                           */
    int32_t min_cons_cnt_euk;  /* 9                                           */
    float   min_score;         /* 0                                           */
    float   min_sp_score;      /* 0                                           */
    float   tie_ratio;         /* 0.95                                        */
    int32_t accession_level;   /* 0                                           */
    int32_t skip_redundancy;   /* 1 for DBs written by `build`                */
} mtb_params;

/* Per-read outcome: the fields of Query (src/commons/common.h:94-122) that
 * Taxonomer::chooseBestTaxon assigns (Taxonomer.cpp:130-202).               */
typedef struct {
    int32_t  classification;   /* taxonomy id, 0 = unclassified               */
    float    score;
    int32_t  query_length;     /* getMaxCoveredLength(L1) (LocalUtil.h:51-59) */
    int32_t  query_length2;    /* same for mate 2, else 0                     */
    uint8_t  is_classified;
    uint8_t  reserved;
    uint16_t n_taxcnt;         /* entries of Query::taxCnt                    */
    uint32_t taxcnt_off;       /* first entry in the taxcnt arrays            */
} mtb_result;

typedef struct mtb_ctx mtb_ctx;
typedef struct mtb_index mtb_index;

/* kernel ids for mtb_batch_stats.ms_kernel / n_launch */
enum {
    MTB_K_EXTRACT_COUNT = 0, MTB_K_EXTRACT_EMIT = 1, MTB_K_RADIX_HIST = 2, MTB_K_RADIX_SCATTER = 3,
    MTB_K_JOIN = 4, MTB_K_REGROUP = 5, MTB_K_SEGSORT = 6, MTB_K_SCORE = 7, MTB_K_SCAN = 8, MTB_K_SCORE_FAST = 9,
    MTB_K_SCORE_MANY = 10,     /* the reads of conserved genes: overflow grouping + k_score_many (kernels_score_many.h) */
    MTB_NUM_KERNELS = 11
};

/* Per-stage device time of the last mtb_classify_batch* call (HIP events on
 * the context's stream) and the run-time counters of SURVEY.md 8(d).        */
typedef struct {
    float    ms_extract, ms_sort, ms_join, ms_regroup, ms_segsort, ms_score, ms_total;
    uint64_t n_reads, n_bases, n_kmers, n_matches, n_targets;
    /* per-kernel sums over the batch (filled when profiling is on, see
     * mtb_ctx_set_profiling): HIP events recorded on the context's stream
     * immediately before and after every launch of that kernel.            */
    float    ms_kernel[MTB_NUM_KERNELS];
    uint32_t n_launch[MTB_NUM_KERNELS];
    uint64_t n_generic_reads;   /* reads the register-resident scorer (k_score_fast) / the workgroup-per-read scorer of long reads (k_score_long) handed to the generic k_score (short reads whose tails overflowed are NOT in here: n_deferred_reads) */
    uint64_t n_slot_reads;      /* reads whose matches went through per-read ordinal slots (short reads: fixed segments; long reads: per-read
                                 * ranges ordered by k_seg_order) instead of regroup + segment sort */
    /* short reads on slot segments: reads the first scoring launches deferred (tail overflow, more live records than the staging); of
     * those, the reads k_score_many scored straight from slots + overflow entries; the matches of these reads; and how many of them
     * survived the dead-species drop (Taxonomer.cpp:342: a species without a (species, frame) group of two never scores) */
    uint64_t n_deferred_reads, n_many_reads, n_many_matches, n_many_kept;
    /* the short-read join on packed words: which exact instantiation ran on the (last sub-)batch (mtb_join_variant; 0 = another join path,
     * negative = an A/B-only instantiation), whether the context's tuner chose it (1) or it was the default / pinned / still being tuned (0),
     * and the tuner's timings of {q1w6, q2w5, window} for this (index, batch size) so far (0 = not timed).  Window variant: tiles launched, tiles
     * whose target window was staged in LDS, tiles that found a query outside the announced window (0 while the list is sorted). */
    int32_t  join_variant, join_tuned;
    float    join_tune_ms[3];
    uint32_t join_tiles, join_tiles_windowed, join_tiles_outside;
} mtb_batch_stats;

/* the exact instantiations of the short-read join (kernels_dir.h) */
typedef enum { MTB_JOIN_AUTO = 0, MTB_JOIN_Q1W6 = 1, MTB_JOIN_Q2W5 = 2, MTB_JOIN_WINDOW = 3 } mtb_join_variant;

const char *mtb_version(void);
const char *mtb_last_error(void);
void mtb_default_params(mtb_params *p);          /* classify.cpp:10-37       */

/* One context per GPU.  `stream` is a hipStream_t (or NULL for the default
 * stream); all kernels of the context are launched on it.                   */
mtb_status mtb_ctx_create(int device, void *stream, mtb_ctx **out);
void       mtb_ctx_destroy(mtb_ctx *);
mtb_status mtb_ctx_sync(mtb_ctx *);
/* 1: bracket every kernel launch of mtb_classify_batch* with HIP events
 * (adds a few microseconds per launch); 0 (default): stage-level events only. */
mtb_status mtb_ctx_set_profiling(mtb_ctx *, int on);
/* Number of HIP streams mtb_classify_batch* pipelines a batch over (default 1).  With n > 1
 * the batch is cut into n contiguous read ranges that run concurrently, each on its own
 * non-blocking stream and workspace (one host thread per stream inside the call), so that
 * latency-bound kernels of one range overlap bandwidth-bound kernels of another.  Results are
 * identical; per-read taxcnt slots of range i live in the i-th n-th of the taxcnt arrays.   */
mtb_status mtb_ctx_set_streams(mtb_ctx *, int n);
/* HBM-budgeted batching (replaces QueryIndexer's --max-ram split rule, QueryIndexer.cpp:62,132, and the
 * matchPerKmer += 4 redo, Classifier.cpp:92-99,127-131): mtb_classify_batch* cut a batch into contiguous read ranges
 * whose workspace (metamer buffers, slot segments, match buffers) fits the budget and run them one after another;
 * results are those of the undivided batch.  bytes = 0 (default): the budget is what hipMemGetInfo reports free plus
 * what the context already holds.  mtb_ctx_last_sub_batches: how many ranges the last call used.               */
mtb_status mtb_ctx_set_workspace_limit(mtb_ctx *, uint64_t bytes);
/* Placement search for the slot buffer of big batches (>= 8 GB): the join's scattered 16-byte stores run 42 - 57 ms per 10 M reads
 * depending on where that buffer landed in a device whose free memory is fragmented by the process's earlier allocations; with the
 * search on, a new buffer is picked among a few candidate allocations by a random-store probe (one-time 0.6 - 3.8 s).  Off by
 * default: a process that allocates through this library only gets the good placement from the first hipMalloc. */
mtb_status mtb_ctx_set_placement_probe(mtb_ctx *, int on);
/* The short-read join has three exact instantiations (sector-random lookups with one or two queries per thread, LDS-staged target windows);
 * by default (MTB_JOIN_AUTO) a context times them on its second to fourth batch of an (index, batch size) shape and keeps the fastest.  This
 * pins one of them (mtb_join_variant) or hands the choice back to the tuner; results are identical either way.  What ran, and the tuner's
 * timings, are in mtb_batch_stats (join_variant, join_tuned, join_tune_ms).  No reference counterpart (KmerMatcher.cpp:363-416 is one loop). */
mtb_status mtb_ctx_set_join_variant(mtb_ctx *, int variant);
/* Experiment / diagnosis switches (metabuli_amd/csrc/mtb_options.h lists them: MTB_JOIN_WIN, MTB_NO_SCORE_MANY, MTB_DIR_DEPTH, ...).  They are read
 * from the environment ONCE, by mtb_ctx_create; this sets one on a live context (value NULL = as if the variable were unset).  They select among
 * exact variants, size buffers or print -- never a result.  An index takes the switches of the context it is opened on at that moment.
 * MTB_ERR_ARG: unknown name or unparsable value. */
mtb_status mtb_ctx_set_option(mtb_ctx *, const char *name, const char *value);
uint32_t   mtb_ctx_last_sub_batches(const mtb_ctx *);
/* Bytes of the last batch's scoring temporaries (grouped overflow list, deferred reads' segments) that were placed inside buffers the join had left dead
 * -- the unsorted metamer buffer, the sort's digit arrays -- instead of allocations of their own (summed over the call's sub-batches).  Workspace
 * bookkeeping only; no reference counterpart. */
uint64_t   mtb_ctx_last_scratch_bytes(const mtb_ctx *);

/* ---- index residency ---------------------------------------------------
 * Replaces the per-call fopen/fread/mmap of diffIdx, info, split inside
 * KmerMatcher::matchKmers (KmerMatcher.cpp:127-137, 212-217) and
 * KmerMatcher::loadTaxIdList (KmerMatcher.cpp:56-120) plus loadTaxonomy /
 * loadDbParameters (common.cpp:50-133): the delta-coded index is decoded
 * once on the GPU into a flat {u64 value[T]; u32 info[T]} held in HBM.
 * Taxonomy as loadTaxonomy picks it (common.cpp:50-86): DBDIR/taxonomyDB (binary, internal ids) if present, else
 * `taxonomy_dir`, else (NULL) DBDIR/taxonomy/{names,nodes,merged}.dmp.
 * `params` is in/out: db.parameters overrides are written back.             */
mtb_status mtb_index_open(mtb_ctx *, const char *dbdir, const char *taxonomy_dir,
                          mtb_params *params, mtb_index **out);
/* How the files came in.  The target list is decoded in CHUNKS of the diffIdx stream (the reference streams the files as well,
 * KmerMatcher.cpp:212-217, 256-271): peak HBM during the open = what the index keeps + one chunk, whatever the database size; a
 * workspace limit on the context (mtb_ctx_set_workspace_limit) bounds the chunk.  Databases of >= 2^28 targets whose directory has
 * depth 7 open SEALED (packed 8-byte words, info[] folded in chunk by chunk and never resident: mtb_index_seal's state, 8.4 bytes per
 * target -- 16 G targets fit one 288 GB GPU); smaller ones open flat.
 * out4: [0] chunks decoded, [1] 16-bit words per chunk, [2] peak device bytes in use during the open beyond what was in use before
 * it (values + info + directory + chunk buffers), [3] 1 if the index was packed on load.                                          */
mtb_status mtb_index_open_stats(const mtb_index *, uint64_t *out4);
/* Same, from an already flat index resident on the device (synthetic-index
 * benchmark path).  The arrays are borrowed, not copied: they must outlive
 * the index -- and they are NOT read-only: while the index lives the fused
 * path (mtb_classify_batch*, mtb_index_seal) may rewrite d_values in place to
 * packed words (kernels_dir.h) and back, so the lender must not read them in
 * between; mtb_index_close hands d_values back in the flat state it was lent
 * in (d_info too, unless the index was sealed).  taxonomy_dir must hold the
 * *.dmp files; taxid_list (host) is the content of DBDIR/taxID_list.        */
mtb_status mtb_index_from_device(mtb_ctx *, uint64_t *d_values, uint32_t *d_info,
                                 uint64_t n_targets, const char *taxonomy_dir,
                                 const int32_t *taxid_list, size_t n_taxids,
                                 const mtb_params *params, mtb_index **out);
/* SURVEY.md 8(e) row 1 ("load once, broadcast over xGMI"): a copy of a resident index on another context's GPU, made with
 * device-to-device peer copies -- target words in the state they are in (packed / flat), info[] if resident, the directory -- plus
 * the taxonomy tables; the database files are read and decoded once per node, not once per GPU.  The copy is independent of its
 * source (own memory, own state).  `src` must not be a view.                                                          */
mtb_status mtb_index_clone(mtb_index *src, mtb_ctx *dst_ctx, mtb_index **out);
/* The same across PROCESSES (one process per GPU: torch.distributed / MPI launches): the process that holds a resident index describes
 * it -- inter-process handles (hipIpcGetMemHandle) of the target words and the directory, sizes, state; a plain-old-data record that
 * travels over any channel (a broadcast, a pipe) --, every other process of the node opens the handles and copies the arrays to ITS
 * GPU device-to-device (over xGMI between GPUs) into memory of its own, then closes them: the database is read, decoded and packed once per
 * node.  The exporter must keep the index open and unchanged (no classify call: the fused path may repack the array) until every importer
 * has returned; the import is an independent index afterwards.  taxonomy_dir / taxid_list as for mtb_index_from_device (the taxonomy
 * tables are host data: every process loads them itself).  `src` must not be a view.  No reference counterpart (one process, one address
 * space: KmerMatcher.cpp:212-217). */
typedef struct {
    uint8_t  values_handle[64], info_handle[64], dir_handle[64], dirbase_handle[64];      /* hipIpcMemHandle_t of the allocations that hold the arrays */
    uint64_t values_off, info_off, dir_off, dirbase_off;                                   /* the arrays' byte offsets inside those allocations */
    uint64_t n_targets;
    uint32_t dir_buckets, info_mask;
    int32_t  dir_depth, packed, has_info, match_last, device;                               /* device: the exporter's ordinal (peer access) */
    int64_t  exporter_pid;                                                                  /* an import inside the exporting process copies from the pointers directly */
    uint64_t values_ptr, info_ptr, dir_ptr, dirbase_ptr;
} mtb_index_share;
mtb_status mtb_index_export(mtb_index *src, mtb_index_share *out);
mtb_status mtb_index_import(mtb_ctx *, const mtb_index_share *, const char *taxonomy_dir, const int32_t *taxid_list, size_t n_taxids,
                            const mtb_params *params, mtb_index **out);
/* Dedicate an index to the fused path (mtb_classify_batch*): its target array goes to the packed state -- one 8-byte word per
 * target carrying the eighth amino-acid letter, the DNA bits and the info entry under the depth-7 amino-acid directory,
 * kernels_dir.h -- and info[] is let go of: freed if the library owns it, else the caller may free the array it lent
 * (mtb_index_from_device).  12 -> 8.4 bytes per target: a 16 G-metamer index takes 135 GB instead of 199 GB.  Entry points
 * that need the flat arrays (stage calls, download, write, slices) still work: they re-allocate info[] and unpack.
 * MTB_ERR_UNSUPPORTED if the index is too small for a depth-7 directory or is a view.                            */
mtb_status mtb_index_seal(mtb_index *);
void       mtb_index_close(mtb_index *);
uint64_t   mtb_index_num_targets(const mtb_index *);
/* How the target array is held right now: depth of the amino-acid directory (0 = none: the bisection join is used), whether the
 * array is in the packed state of the fused join, whether the index is sealed (info[] released).  Reporting only.          */
mtb_status mtb_index_state(const mtb_index *, int32_t *dir_depth, int32_t *packed, int32_t *sealed);
/* Copy the decoded flat index back to the host (parity seam for the codec). */
mtb_status mtb_index_download(mtb_index *, uint64_t *values, uint32_t *info, uint64_t cap);
/* Taxonomy services (TaxonomyWrapper / NcbiTaxonomy) for the host side.     */
int32_t    mtb_tax_lca(const mtb_index *, int32_t a, int32_t b);
int32_t    mtb_tax_species(const mtb_index *, int32_t taxid);  /* taxId2speciesId */
int32_t    mtb_tax_parent(const mtb_index *, int32_t taxid);
int32_t    mtb_tax_max_id(const mtb_index *);
/* TaxonomyWrapper::getOriginalTaxID (TaxonomyWrapper.h:70-79): databases written by `build` number their taxa
 * internally (1..maxTaxID in order of appearance) and keep the map in taxonomyDB; every id the reference prints goes
 * through it (Reporter.cpp:52,62,69,181).  Identity for dump-file taxonomies.                               */
int32_t    mtb_tax_original_id(const mtb_index *, int32_t taxid);
/* children of a node in node order (NcbiTaxonomy::getParentToChildren, used by Reporter::writeReportFile, Reporter.cpp:121) */
int32_t    mtb_tax_num_children(const mtb_index *, int32_t taxid);
int32_t    mtb_tax_child(const mtb_index *, int32_t taxid, int32_t k);
/* rank / scientific name of a node ("" if unknown); pointers stay valid while the index is open
 * (TaxonomyWrapper::getString(taxonNode(t)->rankIdx / nameIdx), used by Reporter.cpp:35-193) */
const char *mtb_tax_rank(const mtb_index *, int32_t taxid);
const char *mtb_tax_name(const mtb_index *, int32_t taxid);

/* ---- stage-level entry points (host buffers; the parity seam) ----------
 * KmerExtractor::extractQueryKmers minus the sort (KmerExtractor.cpp:52-77,
 * 83-373): reads are concatenated in `bases` with offs[n_reads+1]; mates of
 * seq_mode 2 in bases2/offs2 (else NULL).  Emits only real k-mers (no blank
 * slots).  *count receives the number produced (required size if > cap).
 * qlen/qlen2 receive Query::queryLength/queryLength2.                       */
mtb_status mtb_extract(mtb_ctx *, const mtb_params *, const char *bases, const uint64_t *offs,
                       const char *bases2, const uint64_t *offs2, uint64_t n_reads,
                       mtb_kmer *out, uint64_t cap, uint64_t *count,
                       int32_t *qlen, int32_t *qlen2);
/* SORT_PARALLEL(..., Kmer::compareQueryKmer) (KmerExtractor.cpp:79): stable
 * LSD radix sort on the 64-bit value (extraction order is by sequenceID).   */
mtb_status mtb_sort_kmers(mtb_ctx *, mtb_kmer *kmers, uint64_t n);
/* KmerMatcher::matchKmers (KmerMatcher.cpp:123-481).  Output order is
 * unspecified (as in the reference); *count = matches found.  Returns
 * MTB_ERR_CAPACITY where the reference returns false.                       */
mtb_status mtb_match_kmers(mtb_ctx *, mtb_index *, const mtb_kmer *sorted, uint64_t n,
                     mtb_match *out, uint64_t cap, uint64_t *count);
/* KmerMatcher::sortMatches (KmerMatcher.cpp:1071-1078, 1149-1166); n_reads =
 * number of reads of the batch (sequenceID is 1..n_reads).                  */
mtb_status mtb_sort_matches(mtb_ctx *, mtb_match *matches, uint64_t n, uint64_t n_reads);
/* Classifier::assignTaxonomy (Classifier.cpp:166-208) -> per read
 * Taxonomer::chooseBestTaxon.  taxcnt_* receive the concatenated
 * Query::taxCnt maps (ascending taxid per read); *n_taxcnt = entries.       */
mtb_status mtb_score(mtb_ctx *, mtb_index *, const mtb_params *, const mtb_match *sorted,
                     uint64_t n_matches, uint64_t n_reads, const int32_t *qlen,
                     const int32_t *qlen2, mtb_result *results, int32_t *taxcnt_tax,
                     uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt);

/* ---- fused batch (the product path) ------------------------------------
 * One pass of Classifier::startClassify's loop body (Classifier.cpp:81-125)
 * for one batch.  Host-buffer variant copies inputs over PCIe first.
 * taxcnt arrays of the host-buffer variants (one stream): the taxID:count lists arrive packed (taxcnt_off = running total),
 * so taxcnt_cap only has to hold what the batch's lists contain (2-3 entries per short read is typical; the device-side
 * arrays, one slot per position bucket of every read, are the library's).  MTB_ERR_CAPACITY: *n_taxcnt = entries needed,
 * nothing was copied.  With mtb_ctx_set_streams(n > 1) the lists are spread over the arrays instead and taxcnt_cap must
 * hold one entry per position bucket ((read length + 3) / dna shift + 2 per read).                                     */
mtb_status mtb_classify_batch(mtb_ctx *, mtb_index *, const mtb_params *, const char *bases,
                              const uint64_t *offs, const char *bases2, const uint64_t *offs2,
                              uint64_t n_reads, mtb_result *results, int32_t *taxcnt_tax,
                              uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt);
/* Device-buffer variant: d_bases/d_offs (and mates) are device pointers;
 * results stay on the device in d_results (n_reads entries) and the taxcnt
 * arrays d_taxcnt_* (capacity taxcnt_cap); nothing crosses PCIe except the
 * scalar counters.  This is what bench.py times.                            */
mtb_status mtb_classify_batch_device(mtb_ctx *, mtb_index *, const mtb_params *,
                                     const char *d_bases, const uint64_t *d_offs,
                                     const char *d_bases2, const uint64_t *d_offs2,
                                     uint64_t n_reads, uint64_t n_bases_total,
                                     mtb_result *d_results, int32_t *d_taxcnt_tax,
                                     uint32_t *d_taxcnt_cnt, uint64_t taxcnt_cap,
                                     uint64_t *n_taxcnt);
/* Host ingest at device rate (the reference has a single kseq producer per file, KmerExtractor.cpp:122-171).
 * mtb_host_alloc / mtb_host_free: pinned host memory for the caller's batch buffers (H2D / D2H copies of pageable memory run at
 * a fraction of the link rate).
 * mtb_classify_batch_packed: mtb_classify_batch with the bases as 2-bit codes -- packed2 holds 2 bits per base (A 0, C 1, T 2,
 * G 3 = GeneticCode's nuc2int order; IUPAC codes mapped as the reference's atcg table maps them), nmask 1 bit per base (set =
 * not a base: N, '.', ...), lens[r] the bases of read r; every read starts at a fresh group of 8 bases in both arrays (read r
 * at group sum_{q<r} ceil(lens[q] / 8)).  0.375 bytes per base cross PCIe instead of 1; a device kernel rebuilds the text
 * (invalid bases as 'N') for the extractor, so the results equal those of the text entry point.  Mates likewise (seq_mode 2). */
void      *mtb_host_alloc(size_t bytes);
void       mtb_host_free(void *);
mtb_status mtb_classify_batch_packed(mtb_ctx *, mtb_index *, const mtb_params *, const uint8_t *packed2, const uint8_t *nmask,
                                     const uint32_t *lens, const uint8_t *packed2_mate, const uint8_t *nmask_mate,
                                     const uint32_t *lens_mate, uint64_t n_reads, mtb_result *results, int32_t *taxcnt_tax,
                                     uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt);
/* Upload of the NEXT batch while the current one computes: starts the H2D copies of the arrays a following
 * mtb_classify_batch_packed call will be given (same pointers, same n_reads) on a copy stream of the context, into the second of two
 * input buffer sets, and returns at once.  Call order per context thread: prefetch(k+1), classify(k), prefetch(k+2), classify(k+1) ...
 * The arrays must be pinned (mtb_host_alloc) and stay untouched until their classify call returns.  (The reference's producer fills a
 * batch, then the batch is processed: KmerExtractor.cpp:117-173.)                                                                    */
mtb_status mtb_prefetch_batch_packed(mtb_ctx *, const mtb_params *, const uint8_t *packed2, const uint8_t *nmask, const uint32_t *lens,
                                     const uint8_t *packed2_mate, const uint8_t *nmask_mate, const uint32_t *lens_mate, uint64_t n_reads);
/* Prefetches issued on this context / prefetches whose classify call found the batch on the device (or arriving) and did not upload
 * it again.  In protocol order the two are equal; a driver that sees `used` fall behind is calling out of order.  Two prefetches may be
 * outstanding at a time (batch k waiting for its classify call while batch k+1 is issued); a third is ignored (MTB_OK, not counted). */
mtb_status mtb_ctx_prefetch_stats(mtb_ctx *, uint64_t *issued, uint64_t *used);
/* Results on their way back while the NEXT batch computes: mtb_classify_batch_packed_async is mtb_classify_batch_packed, except that it
 * returns once the copies of the rows and of the packed taxID:count lists into the caller's arrays (pinned: mtb_host_alloc) are QUEUED on a
 * download stream of the context.  *n_taxcnt is final on return; MTB_ERR_CAPACITY as before (nothing queued: call again with larger arrays).
 * The arrays of call k belong to the library until call k + 1 of the same context returns -- whatever its status -- or until
 * mtb_ctx_wait_results().  Call order per context thread: prefetch(k+1), async(k), [hand on batch k-1], prefetch(k+2), async(k+1), ...,
 * mtb_ctx_wait_results(), [hand on the last batch].  (The reference hands a batch on when it is through, Classifier.cpp:81-125; here the
 * hand-over happens one batch later and the device does not wait for PCIe between batches.) */
mtb_status mtb_classify_batch_packed_async(mtb_ctx *, mtb_index *, const mtb_params *, const uint8_t *packed2, const uint8_t *nmask,
                                           const uint32_t *lens, const uint8_t *packed2_mate, const uint8_t *nmask_mate,
                                           const uint32_t *lens_mate, uint64_t n_reads, mtb_result *results, int32_t *taxcnt_tax,
                                           uint32_t *taxcnt_cnt, uint64_t taxcnt_cap, uint64_t *n_taxcnt);
mtb_status mtb_ctx_wait_results(mtb_ctx *);
/* Host-side helpers of a driver that overlaps its start-up: mtb_db_parameters applies DBDIR/db.parameters to *p (what mtb_index_open
 * does first: common.cpp:88-133) without opening anything; mtb_ctx_reserve grows the context's big workspace buffers (metamer buffers,
 * digit arrays, slot segments) for short-read batches of n_reads reads / n_bases bases ahead of time -- it may run on ANOTHER thread while
 * the context's thread is inside mtb_index_open, so that tens of GB of hipMalloc hide behind the database load instead of delaying the
 * first batch.  Purely an optimisation: a batch that needs more grows its buffers as always.                                          */
mtb_status mtb_db_parameters(const char *dbdir, mtb_params *p);
mtb_status mtb_ctx_reserve(mtb_ctx *, const mtb_params *, uint64_t n_reads, uint64_t n_bases);
mtb_status mtb_last_batch_stats(mtb_ctx *, mtb_batch_stats *out);
/* Diagnostic, outside any timed region: the index-side working set of the LAST mtb_classify_batch* call of this context
 * when it took the directory join (short reads, index with a directory): distinct directory buckets its query metamers fall
 * into, distinct 64-byte sectors of the directory they read, distinct 64-byte sectors of the target array spanned by those
 * buckets.  target_sectors x 64 + dir_sectors x 64 + 16 x n_queries is the least the join can fetch for that batch.
 * MTB_ERR_UNSUPPORTED if the last call did not leave its sorted metamers behind (no call yet, long reads, sub-batches > 1).  */
typedef struct { uint64_t n_queries, distinct_buckets, dir_sectors, target_sectors, n_buckets, n_targets; } mtb_join_footprint;
mtb_status mtb_ctx_join_footprint(mtb_ctx *, mtb_index *, mtb_join_footprint *out);

/* Diagnostics, outside any timed region: lengths of the candidate runs (targets sharing one amino-acid part: what a query's scan in
 * KmerMatcher.cpp:363-416 walks).  hist64 receives 64 counters.
 * mtb_index_run_histogram: over the whole index (it is brought to the flat state first): [b] = runs with floor(log2(length)) = b,
 * [32 + b] = targets in them (b = 0..31).
 * mtb_ctx_join_run_histogram: over the query metamers of the LAST mtb_classify_batch* call of the context (same conditions as
 * mtb_ctx_join_footprint): [0] = queries without a candidate, [1 + b] = queries whose run has floor(log2(length)) = b (b = 0..31),
 * [40 + b] = targets in those runs (b = 0..23, longer runs in the last bin); [33] / [34] = queries, and the targets of their runs,
 * that find their own DNA part in a run of more than 8 targets -- the join selects that block by bisection without scanning the run. */
mtb_status mtb_index_run_histogram(mtb_index *, uint64_t *hist64);
mtb_status mtb_ctx_join_run_histogram(mtb_ctx *, mtb_index *, uint64_t *hist64);

/* Writes the resident index in the reference's on-disk format -- diffIdx
 * (IndexCreator::getDiffIdx, IndexCreator.cpp:874-892), info, split
 * (writeTargetFilesAndSplits, :817-872, `split_num` checkpoints; the reference
 * uses 4096), taxID_list (:329-333) and db.parameters (:1251-1272) -- into the
 * existing directory `dbdir`.  `metabuli classify` and mtb_index_open read
 * the result.  The taxonomy dump files are not written.                     */
mtb_status mtb_index_write(const mtb_index *, const char *dbdir, int split_num);

/* ---- partitioned index: databases larger than one GPU's HBM -----------
 * SURVEY.md 8(e) row 2.  The flat target array is range-partitioned by value
 * at amino-acid-part boundaries (the invariant of the reference's `split`
 * checkpoints, IndexCreator.cpp:848-857; KmerMatcher.cpp:157-205 uses them
 * the same way to let threads start mid-stream), one range per GPU.  Per
 * batch every GPU extracts + sorts its reads' metamers and cuts the sorted
 * list at the partition bounds (mtb_part_extract); the host side exchanges
 * the runs (RCCL all-to-all, metabuli_amd/parallel.py), each owner joins
 * every received run against its range (mtb_part_join: the runs stay sorted,
 * so no merge), the matches travel back and the home GPU regroups, sorts and
 * scores them (mtb_part_score).  All d_* pointers are device pointers.
 *
 * mtb_index_part_bounds: bounds[p] = lower amino-acid-part bound of range p
 * (bounds[0] = 0; an empty range repeats the next bound).
 * mtb_index_open_part: loads only range `part` (file byte ranges given by
 * the split checkpoints); the full taxonomy / taxID_list is loaded on every
 * rank.  The "last entry is never a candidate" rule (KmerMatcher.cpp:363)
 * applies to the last non-empty range only.
 * mtb_index_slice: the same as a device view [lower_bound(lo), lower_bound(hi))
 * of an index that is already resident (tests, synthetic indices); the view
 * must be closed before its parent.                                         */
mtb_status mtb_index_part_bounds(const char *dbdir, uint32_t n_parts, uint64_t *bounds);
mtb_status mtb_index_open_part(mtb_ctx *, const char *dbdir, const char *taxonomy_dir,
                               mtb_params *params, uint32_t part, uint32_t n_parts,
                               mtb_index **out);
mtb_status mtb_index_slice(mtb_index *, uint64_t lo_value, uint64_t hi_value, int is_last,
                           mtb_index **out);
/* *d_sorted (owned by the context, valid until its next call) = the batch's
 * metamers, sorted; run p = d_sorted[part_starts[p] .. + part_counts[p]) goes
 * to the owner of range p.  part_starts == NULL: the runs are consecutive and
 * the list is sorted on bits [24, 64) (five binary radix passes), matches come
 * home through regroup + segment sort.  part_starts != NULL (what the host
 * layers use): short reads take the product path of the fused batch -- ordinal
 * tags in qinfo, three letter-pair passes, runs cut at prefix granularity so
 * that NEIGHBOURING RUNS MAY OVERLAP by the metamers that share their prefix
 * with a bound (the owner that does not hold their amino-acid group emits
 * nothing for them); the owners join through their range's directory
 * (k_join_dir, matches as a dense list with the ordinal kept in qinfo and
 * pad = 1 on a query's first match) and mtb_part_score places the matches
 * into this rank's slot segments and runs the slot scorers (k_score_fast ...).
 * Query lengths / the batch's mode stay in the context for mtb_part_score.  */
mtb_status mtb_part_extract(mtb_ctx *, const mtb_params *, const char *d_bases,
                            const uint64_t *d_offs, const char *d_bases2, const uint64_t *d_offs2,
                            uint64_t n_reads, const uint64_t *bounds, uint32_t n_parts,
                            const mtb_kmer **d_sorted, uint64_t *n_kmers, uint64_t *part_counts,
                            uint64_t *part_starts);
/* one sorted run of metamers (from any rank) against this rank's range */
mtb_status mtb_part_join(mtb_ctx *, mtb_index *, const mtb_kmer *d_kmers, uint64_t n,
                         mtb_match *d_out, uint64_t cap, uint64_t *count);
/* all matches of this rank's reads, any order (d_matches is clobbered: it
 * becomes sort scratch); results / taxcnt arrays are host buffers; the
 * taxID:count lists arrive packed (taxcnt_off = running total), *n_taxcnt =
 * entries written.  taxcnt_cap must hold one entry per position bucket of
 * every read on the device side (MTB_ERR_CAPACITY + required size otherwise). */
mtb_status mtb_part_score(mtb_ctx *, mtb_index *, const mtb_params *, mtb_match *d_matches,
                          uint64_t n_matches, uint64_t n_reads, mtb_result *results,
                          int32_t *taxcnt_tax, uint32_t *taxcnt_cnt, uint64_t taxcnt_cap,
                          uint64_t *n_taxcnt);

/* The partitioned batch for a host that drives several GPUs from one process (mtb_classify --devices a,b,.. --partitioned 1):
 * ctxs[k] / parts[k] = context and range k of n (mtb_index_open_part(ctxs[k], .., k, n, &parts[k])), bounds from
 * mtb_index_part_bounds.  Host buffers in and out like mtb_classify_batch; the reads are cut into n contiguous shares, the
 * two exchanges are device-to-device peer copies (KmerMatcher.cpp:157-205 lets threads start mid-stream from `split`
 * checkpoints in one address space; here the ranges live in different HBMs).  Results in input order, every read's
 * taxID:count entries at taxcnt_off, packed.  MTB_ERR_CAPACITY with *n_taxcnt = required entries if taxcnt_cap is too small. */
mtb_status mtb_classify_batch_partitioned(mtb_ctx **ctxs, mtb_index **parts, uint32_t n, const uint64_t *bounds,
                                          const mtb_params *, const char *bases, const uint64_t *offs,
                                          const char *bases2, const uint64_t *offs2, uint64_t n_reads,
                                          mtb_result *results, int32_t *taxcnt_tax, uint32_t *taxcnt_cnt,
                                          uint64_t taxcnt_cap, uint64_t *n_taxcnt);

/* ---- synthetic data on the device (bench / tests) ------
 * Builds a flat sorted target index of `n_filler` pseudo-random valid
 * metamers (seeded) merged with `n_real` caller-provided (value,taxid)
 * entries (host, any order).  Output arrays are device buffers with room
 * for n_filler+n_real entries; *n_out = entries written (duplicates of
 * (value, taxid) removed).                                                  */
mtb_status mtb_synth_index(mtb_ctx *, uint64_t seed, uint64_t n_filler, int32_t filler_tax_lo,
                           int32_t filler_tax_hi, const uint64_t *real_values,
                           const int32_t *real_taxids, uint64_t n_real,
                           uint64_t *d_values, uint32_t *d_info, uint64_t *n_out);
/* Target-side extraction used to build synthetic indices: all six-frame
 * (sync)metamers of a genome, as (value) list on the host.                  */
mtb_status mtb_extract_targets(mtb_ctx *, const mtb_params *, const char *genome, uint64_t len,
                               uint64_t *values, uint64_t cap, uint64_t *count);

#ifdef __cplusplus
}
#endif
#endif /* MTB_H */
