/* fastx.h -- host ingest for the classify driver: FASTA / FASTQ (plain, gzip, BGZF) -> flat read batches.
 *
 * The reference parses reads with one kseq producer per file (KmerExtractor.cpp:122-171, KSeqWrapper); at GPU rates the parser is
 * the bottleneck.  This reader makes a batch in two parallel passes over a window of the input, IN PLACE:
 *   count   the window is cut at record starts into one piece per worker; every worker walks its piece with memchr and counts
 *           records, bases and name bytes (the last piece stops at its last complete record);
 *   fill    prefix sums give every piece its place in the batch's flat buffers; the workers walk their pieces again and write
 *           names, offsets and bases straight to their final positions -- optionally as 2-bit codes + an invalid-base mask
 *           (what crosses PCIe: 0.375 bytes per base instead of 1).
 * Nothing is parsed into intermediate buffers, nothing is copied twice, the batch buffers are reused from batch to batch (no
 * fresh pages) and the workers are a persistent pool (round 2 spawned threads per 64 MB block and parsed into per-piece buffers
 * that were copied again: 10 M reads/s with 128 threads, slower than with 32).
 * Sources: plain files are mapped; BGZF (blocked gzip: bgzip, many sequencers' output) is inflated block-parallel -- every
 * block's compressed and uncompressed size is in its header / trailer; an ordinary gzip file is inflated block-parallel too, from
 * guessed block starts with unknown windows (pgzip.h: 3.3 GB/s of text on 64 threads of the GPU box against 0.39 through one zlib
 * stream); files below 4 MB, or a reader with one thread, use a zlib stream on a background thread of its own.
 *
 * Formats: FASTQ with four lines per record (what sequencers and the reference's test data use; multi-line FASTQ is not
 * supported), FASTA with sequences over any number of lines.  Names end at the first blank, as kseq's do.  Lower-case bases and
 * IUPAC codes pass through (the extractor maps them).
 */
#ifndef MTB_FASTX_H
#define MTB_FASTX_H
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include "pgzip.h"

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mtbhost {

/* growable array of a trivially copyable type WITHOUT value-initialisation: batch buffers are filled by parallel writers, a
 * std::vector would zero every byte first from one thread.  An allocator pair may be given (pinned host memory for H2D copies). */
template <class T> class PodVec {
public:
    PodVec() = default;
    ~PodVec() { release(); }
    PodVec(const PodVec &) = delete; PodVec &operator=(const PodVec &) = delete;
    PodVec(PodVec &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_), alloc_(o.alloc_), free_(o.free_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    PodVec &operator=(PodVec &&o) noexcept { if (this != &o) { release(); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; alloc_ = o.alloc_; free_ = o.free_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
    /* memory from alloc(bytes) / free(ptr) instead of malloc (must be set while empty) */
    void set_allocator(void *(*a)(size_t), void (*f)(void *)) { if (!p_) { alloc_ = a; free_ = f; } }
    T *data() { return p_; } const T *data() const { return p_; }
    size_t size() const { return n_; } bool empty() const { return n_ == 0; }
    T &operator[](size_t i) { return p_[i]; } const T &operator[](size_t i) const { return p_[i]; }
    void clear() { n_ = 0; }
    void reserve(size_t c) {
        if (c <= cap_) return;
        T *q;
        if (alloc_) {
            q = (T *)alloc_(c * sizeof(T));
            if (!q) throw std::bad_alloc();
            if (n_) memcpy(q, p_, n_ * sizeof(T));
            if (p_) free_(p_);
        } else {
            q = (T *)realloc(p_, c * sizeof(T));
            if (!q) throw std::bad_alloc();
        }
        p_ = q; cap_ = c;
    }
    void resize_uninit(size_t n) { if (n > cap_) reserve(n + n / 8 + 16); n_ = n; }
    void push_back(const T &v) { if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 64); p_[n_++] = v; }
    void append(const T *a, const T *b) { const size_t k = (size_t)(b - a); if (n_ + k > cap_) reserve(std::max(cap_ * 2, n_ + k)); if (k) memcpy(p_ + n_, a, k * sizeof(T)); n_ += k; }
private:
    void release() { if (p_) { if (free_) free_(p_); else free(p_); } p_ = nullptr; n_ = cap_ = 0; }
    T *p_ = nullptr; size_t n_ = 0, cap_ = 0;
    void *(*alloc_)(size_t) = nullptr; void (*free_)(void *) = nullptr;
};

/* persistent workers: run(n, f) calls f(0) .. f(n - 1) on the pool (and the caller) and returns when all are done */
class WorkerPool {
public:
    explicit WorkerPool(int threads) {
        const int n = threads < 1 ? 1 : threads;
        for (int i = 1; i < n; i++) th_.emplace_back([this] { loop(); });
    }
    ~WorkerPool() {
        { std::lock_guard<std::mutex> l(m_); stop_ = true; }
        cv_.notify_all();
        for (auto &t : th_) t.join();
    }
    int size() const { return (int)th_.size() + 1; }
    void run(size_t n, const std::function<void(size_t)> &f) {
        if (n == 0) return;
        if (n == 1 || th_.empty()) { for (size_t i = 0; i < n; i++) f(i); return; }
        {
            std::lock_guard<std::mutex> l(m_);
            job_ = &f; n_ = n; next_ = 0; left_ = n; gen_++;
        }
        cv_.notify_all();
        work();
        std::unique_lock<std::mutex> l(m_);
        done_.wait(l, [&] { return left_ == 0; });
        job_ = nullptr;
    }
private:
    void work() {
        for (;;) {
            size_t i;
            const std::function<void(size_t)> *f;
            {
                std::lock_guard<std::mutex> l(m_);
                if (!job_ || next_ >= n_) return;
                i = next_++; f = job_;
            }
            try { (*f)(i); } catch (...) { std::lock_guard<std::mutex> l(m_); if (!err_) err_ = std::current_exception(); }
            bool last;
            { std::lock_guard<std::mutex> l(m_); last = --left_ == 0; }
            if (last) done_.notify_all();
        }
    }
    void loop() {
        uint64_t seen = 0;
        for (;;) {
            {
                std::unique_lock<std::mutex> l(m_);
                cv_.wait(l, [&] { return stop_ || gen_ != seen; });
                if (stop_) return;
                seen = gen_;
            }
            work();
        }
    }
    std::vector<std::thread> th_;
    std::mutex m_; std::condition_variable cv_, done_;
    const std::function<void(size_t)> *job_ = nullptr; size_t n_ = 0, next_ = 0, left_ = 0; uint64_t gen_ = 0; bool stop_ = false;
    std::exception_ptr err_;
public:
    void rethrow() { std::exception_ptr e; { std::lock_guard<std::mutex> l(m_); e = err_; err_ = nullptr; } if (e) std::rethrow_exception(e); }
};

/* A batch of reads in the layout the C ABI takes: concatenated bases + u64 offsets, names likewise.  With `pack` the bases are
 * ALSO written as 2-bit codes (A 0, C 1, T 2, G 3: GeneticCode's nuc2int order) + one invalid bit per base, every read starting
 * on a byte boundary of both arrays at slot[r] (units of 8 bases): what mtb_classify_batch_packed uploads instead of the text. */
struct FlatBatch {
    PodVec<char> bases; PodVec<uint64_t> offs;
    PodVec<char> names; PodVec<uint64_t> name_offs;
    PodVec<uint8_t> packed2, nmask; PodVec<uint32_t> lens;      /* only with pack */
    uint64_t slots = 0;                                         /* 8-base units used by packed2 / nmask */
    FlatBatch() { offs.push_back(0); name_offs.push_back(0); }
    FlatBatch(FlatBatch &&) = default; FlatBatch &operator=(FlatBatch &&) = default;
    size_t size() const { return offs.size() - 1; }
    void clear() { bases.clear(); offs.clear(); offs.push_back(0); names.clear(); name_offs.clear(); name_offs.push_back(0); packed2.clear(); nmask.clear(); lens.clear(); slots = 0; }
    void add(const char *name, size_t name_len, const char *seq, size_t seq_len) {
        names.append(name, name + name_len); name_offs.push_back(names.size());
        bases.append(seq, seq + seq_len); offs.push_back(bases.size());
    }
    std::string name(size_t i) const { return std::string(names.data() + name_offs[i], names.data() + name_offs[i + 1]); }
};

/* base byte -> 2-bit code or 0xFF (invalid), as the extractor's table classifies it (mtb_core.h::mtb_build_tables: canonical
 * A/C/G/T through the reference's atcg table, codes in nuc2int order A 0 C 1 T 2 G 3).  Filled by the caller (the driver takes
 * the library's table) so that this header stays free of the kernel headers. */
struct PackTable { uint8_t code[256]; };

class FastxReader {
public:
    FastxReader(const std::string &path, int threads, size_t window_bytes = 64u << 20, WorkerPool *pool = nullptr)
        : window_(window_bytes < 16 ? 16 : window_bytes) {
        if (pool) pool_ = pool; else { own_pool_.reset(new WorkerPool(threads)); pool_ = own_pool_.get(); }
        threads_ = pool ? pool->size() : (threads < 1 ? 1 : threads);
        struct stat sb;
        if (stat(path.c_str(), &sb) != 0) throw std::runtime_error("cannot open " + path);
        if (!S_ISREG(sb.st_mode)) {
            /* a pipe (process substitution, /dev/stdin): nothing to map or to look ahead in, and it can be opened only once -- zlib's
             * stream reader takes gzip and plain text alike (as the reference's KSeqWrapper over gzread does) */
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) throw std::runtime_error("cannot open " + path);
            gzbuffer(gz_, 1u << 20);
            kind_ = 'z';
            return;
        }
        int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        unsigned char magic[18]; memset(magic, 0, sizeof(magic));
        const ssize_t got = pread(fd, magic, sizeof(magic), 0);
        const bool gz = got >= 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        /* BGZF: gzip member with FEXTRA and the 'BC' subfield first (SAM specification, section 4.1) */
        const bool bgzf = gz && got >= 18 && (magic[3] & 4) && magic[12] == 'B' && magic[13] == 'C' && magic[14] == 2 && magic[15] == 0;
        /* an ordinary gzip file of some size: inflated block-parallel with unknown windows (pgzip.h); small ones, or with one thread,
         * through a zlib stream on a background thread */
        const bool pgz_ok = gz && !bgzf && sb.st_size >= (4 << 20) && threads_ >= 2 && !getenv("MTB_NO_PGZIP");
        if (gz && !bgzf && !pgz_ok) {
            close(fd);
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) throw std::runtime_error("cannot open " + path);
            gzbuffer(gz_, 1u << 20);
            kind_ = 'z';
        } else {
            map_len_ = (size_t)sb.st_size;
            if (map_len_) {
                void *m = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); }
                map_ = (const char *)m;
                madvise(m, map_len_, MADV_SEQUENTIAL);
            }
            close(fd);
            kind_ = bgzf ? 'b' : (pgz_ok ? 'g' : 'p');
            if (kind_ == 'g')
                pz_.reset(new ParallelGzip((const uint8_t *)map_, map_len_, threads_,
                                           [this](size_t n, const std::function<void(size_t)> &f) { pool_->run(n, f); pool_->rethrow(); }));
        }
    }
    ~FastxReader() {
        if (zthread_.joinable()) { { std::lock_guard<std::mutex> l(zm_); zstop_ = true; } zcv_.notify_all(); zthread_.join(); }
        if (gz_) gzclose(gz_);
        if (map_) munmap((void *)map_, map_len_);
    }
    FastxReader(const FastxReader &) = delete;

    /* 2-bit output next to the text (FlatBatch::packed2 / nmask / lens): set before the first batch */
    void set_pack(const PackTable *t, bool keep_text = true) { pack_ = t; keep_text_ = keep_text || !t; }

    /* appends up to max_reads records to `out`; returns false when the file is exhausted and nothing was added */
    bool next_batch(size_t max_reads, FlatBatch &out) {
        const size_t before = out.size();
        if (max_reads == 0) return false;
        size_t want = window_;
        if (avg_rec_ > 0.0) want = std::max<size_t>((size_t)((double)max_reads * avg_rec_ * 1.03) + 4096, 4096);
        for (;;) {
            const char *buf; size_t len; bool at_eof;
            view(want, &buf, &len, &at_eof);
            if (len == 0) break;
            if (format_ == 0) {
                size_t p = 0; while (p < len && (buf[p] == '\n' || buf[p] == '\r')) p++;
                if (p < len) {
                    format_ = buf[p] == '@' ? 'q' : (buf[p] == '>' ? 'a' : 0);
                    if (!format_) throw std::runtime_error("input is neither FASTA nor FASTQ");
                } else if (at_eof) { consume(len); break; }          /* nothing but blank lines */
                else { want *= 2; continue; }
            }
            /* pieces at record starts */
            const size_t P = (size_t)std::max(1, std::min<int>(threads_, (int)(len / 4096) + 1));
            std::vector<size_t> cut{0};
            for (size_t t = 1; t < P; t++) {
                const size_t guess = len / P * t;
                const size_t s = next_record_start(buf, guess, len);
                if (s > cut.back() && s < len) cut.push_back(s);
            }
            cut.push_back(len);
            const size_t np = cut.size() - 1;
            std::vector<Piece> pc(np);
            pool_->run(np, [&](size_t k) { count(buf, cut[k], cut[k + 1], k + 1 == np && !at_eof, (size_t)-1, &pc[k]); });
            pool_->rethrow();
            size_t total = 0;
            for (auto &x : pc) total += x.n_rec;
            if (total == 0 && !at_eof) { want = std::max(want * 2, len * 2); continue; }       /* not one complete record in the window: a longer one */
            if (total < max_reads && !at_eof) {          /* more input is there and the batch is not full: a window sized from what was seen */
                const double per_rec = (double)len / (double)total;
                want = std::max<size_t>((size_t)((double)max_reads * per_rec * 1.05) + 4096, len + len / 2);
                continue;
            }
            /* more records than asked for: the piece where the count is reached is walked again up to that record */
            size_t used = np, cum = 0; bool all_taken = true;
            for (size_t k = 0; k < np; k++) {
                if (cum + pc[k].n_rec >= max_reads) {
                    if (cum + pc[k].n_rec > max_reads) { count(buf, cut[k], cut[k + 1], false, max_reads - cum, &pc[k]); all_taken = false; }
                    used = k + 1; all_taken = all_taken && used == np; break;
                }
                cum += pc[k].n_rec;
            }
            /* places in the batch */
            std::vector<size_t> r0(used + 1), b0(used + 1), n0(used + 1), s0(used + 1);
            r0[0] = out.size(); b0[0] = out.size() ? (size_t)out.offs[out.size()] : 0; n0[0] = out.names.size(); s0[0] = out.slots;      /* (the running base count, also when the text itself is not kept) */
            for (size_t k = 0; k < used; k++) { r0[k + 1] = r0[k] + pc[k].n_rec; b0[k + 1] = b0[k] + pc[k].n_bases; n0[k + 1] = n0[k] + pc[k].n_name; s0[k + 1] = s0[k] + pc[k].n_slots; }
            out.offs.resize_uninit(r0[used] + 1); out.name_offs.resize_uninit(r0[used] + 1);
            if (text_needed()) out.bases.resize_uninit(b0[used]);
            out.names.resize_uninit(n0[used]);
            if (pack_) { out.packed2.resize_uninit(s0[used] * 2); out.nmask.resize_uninit(s0[used]); out.lens.resize_uninit(r0[used]); out.slots = s0[used]; }
            pool_->run(used, [&](size_t k) { fill(buf, cut[k], pc[k].end, pc[k].n_rec, out, r0[k], b0[k], n0[k], s0[k]); });
            pool_->rethrow();
            const size_t taken = pc[used - 1].end;
            const size_t n_new = r0[used] - r0[0];
            if (n_new) avg_rec_ = (double)taken / (double)n_new;
            consume(taken);
            if (at_eof && all_taken && taken < len) consume(len - taken);            /* trailing blank lines */
            break;
        }
        return out.size() > before;
    }

private:
    struct Piece { size_t n_rec = 0, n_bases = 0, n_name = 0, n_slots = 0, end = 0; };
    /* FASTA sequences span lines: they are packed from the assembled text */
    bool text_needed() const { return keep_text_ || format_ != 'q'; }

    /* ---- sources: `want` bytes of input from the current position (fewer only at the end of the input) ---- */
    void view(size_t want, const char **buf, size_t *len, bool *at_eof) {
        if (kind_ == 'p') {
            *buf = map_ + pos_; *len = std::min(want, map_len_ - pos_); *at_eof = pos_ + *len >= map_len_;
            return;
        }
        while (sbuf_.size() - spos_ < want && !src_eof_) refill(want - (sbuf_.size() - spos_));
        *buf = sbuf_.data() + spos_; *len = std::min(want, sbuf_.size() - spos_); *at_eof = src_eof_ && *len == sbuf_.size() - spos_;
    }
    void consume(size_t n) { if (kind_ == 'p') pos_ += n; else spos_ += n; }
    void refill(size_t more) {
        /* drop what has been consumed, then append at least `more` bytes */
        if (spos_) { const size_t keep = sbuf_.size() - spos_; if (keep) memmove(sbuf_.data(), sbuf_.data() + spos_, keep); sbuf_.resize_uninit(keep); spos_ = 0; }
        if (kind_ == 'g') {
            pz_->produce(sbuf_, std::max<size_t>(more, 16u << 20));
            if (pz_->done()) src_eof_ = true;
            return;
        }
        if (kind_ == 'z') {
            /* a plain gzip stream has one inflate thread at best (~0.4 GB/s of text); it runs in the BACKGROUND, a few chunks ahead, so
             * that it overlaps the parsing of what it delivered -- and the mate file's stream, which has a thread of its own */
            if (!zthread_.joinable()) zthread_ = std::thread([this] { inflate_loop(); });
            size_t got = 0;
            const size_t goal = std::max<size_t>(more, ZCHUNK);
            /* the thread may run a whole request ahead: while this file's batch is parsed (or the mate file's stream is waited for),
             * the next batch's text is already being inflated */
            { std::lock_guard<std::mutex> l(zm_); zcap_ = std::max(zcap_, goal / ZCHUNK + 2); }
            zcv_.notify_all();
            while (got < goal && !src_eof_) {
                std::unique_ptr<ZChunk> ch;
                {
                    std::unique_lock<std::mutex> l(zm_);
                    zcv_.wait(l, [&] { return !zq_.empty(); });
                    ch = std::move(zq_.front()); zq_.erase(zq_.begin());
                }
                zcv_.notify_all();
                if (!ch->err.empty()) throw std::runtime_error(ch->err);
                if (ch->data.size()) { sbuf_.append(ch->data.data(), ch->data.data() + ch->data.size()); got += ch->data.size(); }
                if (ch->eof) src_eof_ = true;
            }
            return;
        }
        /* BGZF: walk the block headers from the compressed position until `more` uncompressed bytes are covered, inflate the blocks
         * in parallel to their places */
        struct Blk { size_t cpos, clen, upos, ulen; };
        std::vector<Blk> blks;
        size_t cp = pos_, up = 0;
        const size_t goal = std::max<size_t>(more, 32u << 20);
        while (cp < map_len_ && up < goal) {
            if (cp + 18 > map_len_) throw std::runtime_error("truncated BGZF block header");
            const unsigned char *h = (const unsigned char *)map_ + cp;
            if (h[0] != 0x1f || h[1] != 0x8b || !(h[3] & 4)) throw std::runtime_error("not a BGZF block where one is expected");
            const size_t xlen = h[10] | ((size_t)h[11] << 8);
            size_t bsize = 0;
            for (size_t x = 12; x + 6 <= 12 + xlen && cp + x + 6 <= map_len_;) {            /* the 'BC' subfield: total block size - 1 */
                const size_t slen = h[x + 2] | ((size_t)h[x + 3] << 8);
                if (h[x] == 'B' && h[x + 1] == 'C' && slen == 2) { bsize = (h[x + 4] | ((size_t)h[x + 5] << 8)) + 1; break; }
                x += 4 + slen;
            }
            if (bsize < 12 + xlen + 8 || cp + bsize > map_len_) throw std::runtime_error("corrupt BGZF block");
            const unsigned char *tail = h + bsize - 4;
            const size_t isize = tail[0] | ((size_t)tail[1] << 8) | ((size_t)tail[2] << 16) | ((size_t)tail[3] << 24);
            if (isize > 65536) throw std::runtime_error("corrupt BGZF block (ISIZE beyond 64 KiB)");      /* the format's bound; the trailer is not trusted for the allocation */
            blks.push_back(Blk{cp + 12 + xlen, bsize - 12 - xlen - 8, up, isize});
            cp += bsize; up += isize;
        }
        const size_t at = sbuf_.size();
        sbuf_.resize_uninit(at + up);
        char *dst = sbuf_.data() + at;
        const size_t G = std::max<size_t>(1, blks.size() / ((size_t)threads_ * 4) + 1);          /* blocks per task */
        pool_->run((blks.size() + G - 1) / G, [&](size_t task) {
            z_stream zs; memset(&zs, 0, sizeof(zs));
            if (inflateInit2(&zs, -15) != Z_OK) throw std::runtime_error("zlib: inflateInit2 failed");
            for (size_t i = task * G; i < std::min(blks.size(), (task + 1) * G); i++) {
                const Blk &b = blks[i];
                if (b.ulen == 0) continue;
                inflateReset(&zs);
                zs.next_in = (Bytef *)(map_ + b.cpos); zs.avail_in = (uInt)b.clen;
                zs.next_out = (Bytef *)(dst + b.upos); zs.avail_out = (uInt)b.ulen;
                const int rc = inflate(&zs, Z_FINISH);
                if (rc != Z_STREAM_END || zs.avail_out != 0) { inflateEnd(&zs); throw std::runtime_error("corrupt BGZF block (inflate)"); }
                const unsigned char *tr = (const unsigned char *)map_ + b.cpos + b.clen;              /* CRC32 of the block's text, little endian, in front of ISIZE */
                const uint32_t want = tr[0] | ((uint32_t)tr[1] << 8) | ((uint32_t)tr[2] << 16) | ((uint32_t)tr[3] << 24);
                if ((uint32_t)crc32(crc32(0L, Z_NULL, 0), (const Bytef *)(dst + b.upos), (uInt)b.ulen) != want) { inflateEnd(&zs); throw std::runtime_error("corrupt BGZF block (CRC mismatch)"); }
            }
            inflateEnd(&zs);
        });
        pool_->rethrow();
        pos_ = cp;
        if (pos_ >= map_len_) src_eof_ = true;
    }

    /* background inflater of a plain gzip stream: chunks of inflated text through a bounded queue */
    struct ZChunk { PodVec<char> data; bool eof = false; std::string err; };
    static constexpr size_t ZCHUNK = 16u << 20;
    void inflate_loop() {
        for (;;) {
            std::unique_ptr<ZChunk> ch(new ZChunk());
            ch->data.resize_uninit(ZCHUNK);
            size_t got = 0;
            while (got < ZCHUNK) {
                const int r = gzread(gz_, ch->data.data() + got, (unsigned)(ZCHUNK - got));
                if (r < 0) { ch->err = "read error (corrupt gzip stream?)"; break; }
                if (r == 0) {
                    int en = Z_OK; (void)gzerror(gz_, &en);
                    if (en != Z_OK && en != Z_STREAM_END) ch->err = "truncated or corrupt gzip stream";      /* (zlib reports a premature end as Z_BUF_ERROR) */
                    ch->eof = true; break;
                }
                got += (size_t)r;
            }
            ch->data.resize_uninit(got);
            const bool last = ch->eof || !ch->err.empty();
            {
                std::unique_lock<std::mutex> l(zm_);
                zcv_.wait(l, [&] { return zq_.size() < zcap_ || zstop_; });
                if (zstop_) return;
                zq_.push_back(std::move(ch));
            }
            zcv_.notify_all();
            if (last) return;
        }
    }

    /* ---- record geometry ---- */
    static const char *line_end(const char *p, const char *e) { const char *q = (const char *)memchr(p, '\n', (size_t)(e - p)); return q ? q : e; }
    bool is_record_start(const char *b, size_t p, size_t end) const {
        if (p >= end) return false;
        if (p > 0 && b[p - 1] != '\n') return false;
        if (format_ == 'a') return b[p] == '>';
        if (b[p] != '@') return false;                  /* '@' may also open a quality line: check the '+' two lines below */
        const char *e = b + end;
        const char *l1 = line_end(b + p, e); if (l1 >= e) return false;
        const char *l2 = line_end(l1 + 1, e); if (l2 >= e) return false;
        return l2 + 1 < e && l2[1] == '+';
    }
    /* first record start at or behind `from` (a quality line that starts with '@' and is followed by a header line + '+' cannot
     * be told apart locally only if the header's sequence line starts with '+': not a base, so the test is safe) */
    size_t next_record_start(const char *b, size_t from, size_t end) const {
        const char *e = b + end;
        const char *p = b + from;
        while (p < e) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) return end;
            const size_t s = (size_t)(nl + 1 - b);
            if (is_record_start(b, s, end)) return s;
            p = nl + 1;
        }
        return end;
    }
    /* One record at p (p is a record start or blank lines before one): calls f(name, name_len, pieces of the sequence...) through
     * two callbacks; returns the position behind the record, or 0 if the record is not complete inside [p, e) and `partial` says
     * that more input follows. */
    template <class OnName, class OnSeq>
    const char *walk(const char *p, const char *e, bool partial, OnName on_name, OnSeq on_seq) const {
        while (p < e && (*p == '\n' || *p == '\r')) p++;
        if (p >= e) return e;
        const char *h_end = line_end(p, e);
        if (partial && h_end >= e) return nullptr;
        const char *name = p + 1, *ne = name;
        while (ne < h_end && *ne != ' ' && *ne != '\t' && *ne != '\r') ne++;
        if (format_ == 'q') {
            const char *s = h_end < e ? h_end + 1 : e, *s_end = line_end(s, e);
            if (partial && s_end >= e) return nullptr;
            const char *plus_end = s_end < e ? line_end(s_end + 1, e) : e;
            if (partial && plus_end >= e) return nullptr;
            const char *q_end = plus_end < e ? line_end(plus_end + 1, e) : e;
            if (partial && q_end >= e) return nullptr;
            const char *se = s_end; if (se > s && se[-1] == '\r') se--;
            /* four-line records only (what sequencers write); anything else -- wrapped sequence lines, a missing '+' line, a quality
             * string of another length -- is refused, not guessed at */
            const char *qs = plus_end < e ? plus_end + 1 : e, *qe = q_end; if (qe > qs && qe[-1] == '\r') qe--;
            if (*p != '@' || s_end + 1 >= e || s_end[1] != '+' || (size_t)(qe - qs) != (size_t)(se - s))
                throw std::runtime_error("malformed FASTQ record (four-line records are expected): " + std::string(p, (size_t)std::min<ptrdiff_t>(h_end - p, 60)));
            on_name(name, (size_t)(ne - name));
            on_seq(s, (size_t)(se - s));
            return q_end < e ? q_end + 1 : e;
        }
        const char *s = h_end < e ? h_end + 1 : e;
        /* FASTA: the record ends at the next '>' at a line start; with more input to come it is complete only if that '>' is in sight */
        const char *q = s;
        while (q < e && *q != '>') { const char *le = line_end(q, e); q = le < e ? le + 1 : e; }
        if (partial && q >= e) return nullptr;
        on_name(name, (size_t)(ne - name));
        while (s < q) {
            const char *le = line_end(s, q);
            const char *se = le; if (se > s && se[-1] == '\r') se--;
            on_seq(s, (size_t)(se - s));
            s = le < q ? le + 1 : q;
        }
        return q;
    }
    /* pass 1: records / bases / name bytes of [lo, hi); `partial`: the input goes on behind hi, stop at the last complete record;
     * at most max_rec records */
    void count(const char *b, size_t lo, size_t hi, bool partial, size_t max_rec, Piece *out) const {
        Piece pc; pc.end = lo;
        const char *p = b + lo, *e = b + hi;
        while (p < e && pc.n_rec < max_rec) {
            size_t nb = 0, nn = 0; bool any = false;
            const char *nx = walk(p, e, partial, [&](const char *, size_t l) { nn = l; any = true; }, [&](const char *, size_t l) { nb += l; });
            if (!nx) break;
            if (any) { pc.n_rec++; pc.n_bases += nb; pc.n_name += nn; pc.n_slots += (nb + 7) / 8; }
            p = nx; pc.end = (size_t)(p - b);
        }
        if (!partial && pc.n_rec < max_rec) pc.end = hi;
        *out = pc;
    }
    /* pass 2: the n_rec records of [lo, end) to their places */
    void fill(const char *b, size_t lo, size_t end, size_t n_rec, FlatBatch &out, size_t r0, size_t b0, size_t n0, size_t s0) const {
        const char *p = b + lo, *e = b + end;
        char *bases = out.bases.data(), *names = out.names.data();
        uint64_t *offs = out.offs.data(), *noffs = out.name_offs.data();
        size_t r = r0, bp = b0, np = n0, sp = s0;
        const bool text = text_needed();
        for (size_t i = 0; i < n_rec && p < e; i++) {
            const size_t read_start = bp;
            const uint8_t *line = nullptr;           /* FASTQ: the sequence line in the input */
            const char *nx = walk(p, e, false, [&](const char *nm, size_t l) { if (l) memcpy(names + np, nm, l); np += l; },
                                  [&](const char *s, size_t l) { if (text && l) memcpy(bases + bp, s, l); line = (const uint8_t *)s; bp += l; });
            r++;
            offs[r] = bp; noffs[r] = np;
            if (pack_) {
                const size_t L = bp - read_start;
                uint8_t *p2 = out.packed2.data() + sp * 2, *pm = out.nmask.data() + sp;
                const uint8_t *src = text ? (const uint8_t *)bases + read_start : line;
                const size_t ns = (L + 7) / 8;
                for (size_t g = 0; g < ns; g++) {
                    uint32_t w = 0, m = 0;
                    const size_t k0 = g * 8, kn = std::min<size_t>(8, L - k0);
                    for (size_t k = 0; k < kn; k++) { const uint8_t c = pack_->code[src[k0 + k]]; if (c > 3) m |= 1u << k; else w |= (uint32_t)c << (2 * k); }
                    p2[2 * g] = (uint8_t)w; p2[2 * g + 1] = (uint8_t)(w >> 8); pm[g] = (uint8_t)m;
                }
                out.lens[r - 1] = (uint32_t)L;
                sp += ns;
            }
            p = nx;
        }
    }

    WorkerPool *pool_ = nullptr; std::unique_ptr<WorkerPool> own_pool_;
    int threads_ = 1; size_t window_;
    char format_ = 0, kind_ = 'p';              /* p plain (mapped), b BGZF (mapped, inflated block-parallel), g other gzip (mapped, pgzip.h), z small gzip (zlib stream) */
    const char *map_ = nullptr; size_t map_len_ = 0, pos_ = 0;
    std::unique_ptr<ParallelGzip> pz_;
    gzFile gz_ = nullptr;
    std::thread zthread_; std::mutex zm_; std::condition_variable zcv_; std::vector<std::unique_ptr<ZChunk>> zq_; bool zstop_ = false; size_t zcap_ = 4;
    PodVec<char> sbuf_; size_t spos_ = 0; bool src_eof_ = false;         /* inflated text of the stream sources */
    double avg_rec_ = 0.0;                       /* bytes of input per record, from the previous batch: sizes the next window */
    const PackTable *pack_ = nullptr;
    bool keep_text_ = true;                      /* false (with pack, FASTQ only): FlatBatch::bases is not written, only offs / packed2 / nmask / lens */
};

} // namespace mtbhost
#endif
