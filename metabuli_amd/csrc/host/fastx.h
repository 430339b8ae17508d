/* fastx.h -- host ingest for the classify driver: FASTA / FASTQ (plain or gzip) -> flat read batches.
 *
 * The reference parses reads with one kseq producer per file (KmerExtractor.cpp:122-171, KSeqWrapper); at
 * GPU rates the parser is the bottleneck, so this reader works on large blocks:
 * a block is cut at record boundaries into one piece per worker thread, every worker scans its piece
 * with memchr and appends bases / names to its own flat buffers, and the pieces are copied into the batch
 * in parallel (uninitialised batch buffers: nothing is zero-filled or copied by one thread).  Plain files are
 * mapped and parsed in place; gzip streams are inflated one block ahead on their own thread (a single gzip
 * stream inflates serially: ~0.4 GB/s of text is the ceiling there).  Output is the layout the C ABI
 * takes: concatenated bases + u64 offsets, names likewise.
 *
 * Formats: FASTQ with four lines per record (what sequencers and the reference's test data use;
 * multi-line FASTQ is not supported), FASTA with sequences over any number of lines.  Names end at the
 * first blank, as kseq's do.  Lower-case bases and IUPAC codes pass through (the extractor maps them).
 */
#ifndef MTB_FASTX_H
#define MTB_FASTX_H
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mtbhost {

/* growable array of a trivially copyable type WITHOUT value-initialisation: batch buffers are filled by parallel copies, a
 * std::vector would zero every byte first from one thread */
template <class T> class PodVec {
public:
    PodVec() = default;
    ~PodVec() { free(p_); }
    PodVec(const PodVec &) = delete; PodVec &operator=(const PodVec &) = delete;
    PodVec(PodVec &&o) noexcept : p_(o.p_), n_(o.n_), cap_(o.cap_) { o.p_ = nullptr; o.n_ = o.cap_ = 0; }
    PodVec &operator=(PodVec &&o) noexcept { if (this != &o) { free(p_); p_ = o.p_; n_ = o.n_; cap_ = o.cap_; o.p_ = nullptr; o.n_ = o.cap_ = 0; } return *this; }
    T *data() { return p_; } const T *data() const { return p_; }
    size_t size() const { return n_; } bool empty() const { return n_ == 0; }
    T &operator[](size_t i) { return p_[i]; } const T &operator[](size_t i) const { return p_[i]; }
    void clear() { n_ = 0; }
    void reserve(size_t c) {
        if (c <= cap_) return;
        T *q = (T *)realloc(p_, c * sizeof(T));
        if (!q) throw std::bad_alloc();
        p_ = q; cap_ = c;
    }
    void resize_uninit(size_t n) { if (n > cap_) reserve(n + n / 8 + 16); n_ = n; }
    void push_back(const T &v) { if (n_ == cap_) reserve(cap_ ? cap_ * 2 : 64); p_[n_++] = v; }
    void append(const T *a, const T *b) { const size_t k = (size_t)(b - a); if (n_ + k > cap_) reserve(std::max(cap_ * 2, n_ + k)); if (k) memcpy(p_ + n_, a, k * sizeof(T)); n_ += k; }
private:
    T *p_ = nullptr; size_t n_ = 0, cap_ = 0;
};

struct FlatBatch {
    PodVec<char> bases; PodVec<uint64_t> offs;
    PodVec<char> names; PodVec<uint64_t> name_offs;
    FlatBatch() { offs.push_back(0); name_offs.push_back(0); }
    FlatBatch(FlatBatch &&) = default; FlatBatch &operator=(FlatBatch &&) = default;
    size_t size() const { return offs.size() - 1; }
    void clear() { bases.clear(); offs.clear(); offs.push_back(0); names.clear(); name_offs.clear(); name_offs.push_back(0); }
    void add(const char *name, size_t name_len, const char *seq, size_t seq_len) {
        names.append(name, name + name_len); name_offs.push_back(names.size());
        bases.append(seq, seq + seq_len); offs.push_back(bases.size());
    }
    std::string name(size_t i) const { return std::string(names.data() + name_offs[i], names.data() + name_offs[i + 1]); }
};

class FastxReader {
public:
    FastxReader(const std::string &path, int threads, size_t block_bytes = 64u << 20)
        : threads_(threads < 1 ? 1 : threads), block_(block_bytes < 16 ? 16 : block_bytes) {
        int fd = open(path.c_str(), O_RDONLY);
        if (fd < 0) throw std::runtime_error("cannot open " + path);
        unsigned char magic[2] = {0, 0};
        ssize_t got = pread(fd, magic, 2, 0);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            close(fd);
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) throw std::runtime_error("cannot open " + path);
            gzbuffer(gz_, 1u << 20);
        } else {
            struct stat sb;
            if (fstat(fd, &sb) != 0) { close(fd); throw std::runtime_error("cannot stat " + path); }
            map_len_ = (size_t)sb.st_size;
            if (map_len_) {
                void *m = mmap(nullptr, map_len_, PROT_READ, MAP_PRIVATE, fd, 0);
                if (m == MAP_FAILED) { close(fd); throw std::runtime_error("cannot map " + path); }
                map_ = (const char *)m;
                madvise(m, map_len_, MADV_SEQUENTIAL);
            }
            close(fd);
        }
    }
    ~FastxReader() {
        if (prefetch_.joinable()) prefetch_.join();
        if (gz_) gzclose(gz_);
        free(gz_buf_[0]); free(gz_buf_[1]);
        if (map_) munmap((void *)map_, map_len_);
    }
    FastxReader(const FastxReader &) = delete;

    /* appends up to max_reads records to `out`; returns false when the file is exhausted and nothing was added */
    bool next_batch(size_t max_reads, FlatBatch &out) {
        size_t before = out.size();
        while (out.size() - before < max_reads) {
            if (pending_.size() == pending_pos_ && !fill()) break;
            drain(max_reads - (out.size() - before), out);
        }
        return out.size() > before;
    }

private:
    /* gzip: inflate up to block_ bytes into buffer `k`, behind its first `keep` bytes (the record carried over) */
    void read_gz_block(int k, size_t keep) {
        try {
            gz_reserve(k, keep + block_ + 1);
            size_t got = 0;
            while (!gz_eof_ && got < block_) {                      /* gzread takes an unsigned length */
                int r = gzread(gz_, gz_buf_[k] + keep + got, (unsigned)std::min<size_t>(block_ - got, 1u << 30));
                if (r < 0) throw std::runtime_error("read error (corrupt gzip stream?)");
                if (r == 0) { gz_eof_ = true; break; }
                got += (size_t)r;
            }
            gz_len_[k] = keep + got;
        } catch (const std::exception &e) { gz_err_ = e.what(); gz_len_[k] = keep; gz_eof_ = true; }
    }
    void gz_reserve(int k, size_t n) {
        if (gz_cap_[k] >= n) return;
        char *nb = (char *)realloc(gz_buf_[k], n); if (!nb) throw std::bad_alloc();
        gz_buf_[k] = nb; gz_cap_[k] = n;
    }
    /* one parsed block waits in pending_ (FlatBatch pieces in order) until batches have consumed it */
    bool fill() {
        const char *buf; size_t len; bool at_eof; int cur = 0;
        if (!gz_) {
            if (map_pos_ >= map_len_) return false;
            buf = map_ + map_pos_;
            len = std::min(block_, map_len_ - map_pos_);
            at_eof = map_pos_ + len >= map_len_;
        } else {
            if (prefetch_.joinable()) { prefetch_.join(); cur = gz_next_; }
            else if (!gz_started_) { gz_started_ = true; read_gz_block(0, 0); cur = 0; }
            else return false;                                        /* the last block has been handed out */
            if (!gz_err_.empty()) throw std::runtime_error(gz_err_);
            buf = gz_buf_[cur]; len = gz_len_[cur];
            at_eof = gz_eof_;
        }
        if (len == 0) return false;
        if (format_ == 0) {
            size_t p = 0; while (p < len && (buf[p] == '\n' || buf[p] == '\r')) p++;
            if (p < len) {
                format_ = buf[p] == '@' ? 'q' : (buf[p] == '>' ? 'a' : 0);
                if (!format_) throw std::runtime_error("input is neither FASTA nor FASTQ");
            } else if (at_eof) return false;                          /* nothing but blank lines */
        }
        /* the block ends inside a record unless the input ended: the incomplete tail opens the next block */
        size_t end = len;
        while (!at_eof && format_) {
            end = last_record_start(buf, len);
            if (end) break;
            block_ *= 2;                                              /* one record larger than the block: take more */
            if (!gz_) { len = std::min(block_, map_len_ - map_pos_); at_eof = map_pos_ + len >= map_len_; }
            else { read_gz_block(cur, len); if (!gz_err_.empty()) throw std::runtime_error(gz_err_); buf = gz_buf_[cur]; len = gz_len_[cur]; at_eof = gz_eof_; }
            end = len;
        }
        if (!format_) end = len;                                      /* a block of blank lines */
        if (!gz_) map_pos_ += end;
        else if (!at_eof) {
            /* start inflating the next block right away (behind the carried-over tail), parse this one meanwhile */
            const int k = cur ^ 1; const size_t keep = len - end;
            gz_reserve(k, keep + block_ + 1);
            if (keep) memcpy(gz_buf_[k], buf + end, keep);
            gz_next_ = k;
            prefetch_ = std::thread([this, k, keep] { read_gz_block(k, keep); });
        }
        /* cut [0, end) into pieces at record starts, parse in parallel */
        std::vector<size_t> cut{0};
        for (int t = 1; t < threads_ && format_; t++) {
            size_t guess = end / (size_t)threads_ * (size_t)t;
            size_t s = next_record_start(buf, guess, end);
            if (s > cut.back() && s < end) cut.push_back(s);
        }
        cut.push_back(end);
        pending_.clear(); pending_.resize(cut.size() - 1);
        pending_pos_ = 0; pending_read_ = 0;
        if (format_) run_parallel(cut.size() - 1, [&](size_t k) { parse(buf + cut[k], buf + cut[k + 1], pending_[k]); });
        return true;
    }
    template <class F> void run_parallel(size_t n, F f) {
        if (n <= 1) { if (n) f(0); return; }
        std::vector<std::thread> th;
        for (size_t k = 1; k < n; k++) th.emplace_back([&f, k] { f(k); });
        f(0);
        for (auto &t : th) t.join();
    }
    /* whole pieces are copied into the batch by one thread each (offsets rebased); a piece that straddles the batch limit is
     * split record by record */
    void drain(size_t want, FlatBatch &out) {
        size_t first = pending_pos_, last = pending_pos_, take = 0;
        if (pending_read_ == 0)
            while (last < pending_.size() && take + pending_[last].size() <= want) { take += pending_[last].size(); last++; }
        if (last > first) {
            const size_t np = last - first;
            std::vector<size_t> b0(np + 1), n0(np + 1), r0(np + 1);
            b0[0] = out.bases.size(); n0[0] = out.names.size(); r0[0] = out.size();
            for (size_t k = 0; k < np; k++) { const FlatBatch &p = pending_[first + k]; b0[k + 1] = b0[k] + p.bases.size(); n0[k + 1] = n0[k] + p.names.size(); r0[k + 1] = r0[k] + p.size(); }
            out.bases.resize_uninit(b0[np]); out.names.resize_uninit(n0[np]);
            out.offs.resize_uninit(r0[np] + 1); out.name_offs.resize_uninit(r0[np] + 1);
            run_parallel(np, [&](size_t k) {
                const FlatBatch &p = pending_[first + k];
                if (p.bases.size()) memcpy(out.bases.data() + b0[k], p.bases.data(), p.bases.size());
                if (p.names.size()) memcpy(out.names.data() + n0[k], p.names.data(), p.names.size());
                uint64_t *o = out.offs.data() + r0[k], *no = out.name_offs.data() + r0[k];
                for (size_t i = 1; i <= p.size(); i++) { o[i] = b0[k] + p.offs[i]; no[i] = n0[k] + p.name_offs[i]; }
            });
            for (size_t k = first; k < last; k++) { FlatBatch e; pending_[k] = std::move(e); }
            pending_pos_ = last; want -= take;
        }
        if (want && pending_pos_ < pending_.size()) {
            FlatBatch &p = pending_[pending_pos_];
            const size_t avail = p.size() - pending_read_, n = std::min(avail, want);
            for (size_t i = pending_read_; i < pending_read_ + n; i++)
                out.add(p.names.data() + p.name_offs[i], p.name_offs[i + 1] - p.name_offs[i], p.bases.data() + p.offs[i], p.offs[i + 1] - p.offs[i]);
            pending_read_ += n;
            if (pending_read_ == p.size()) { FlatBatch e; p = std::move(e); pending_pos_++; pending_read_ = 0; }
        }
        if (pending_pos_ == pending_.size()) { pending_.clear(); pending_pos_ = 0; }
    }
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
    size_t next_record_start(const char *b, size_t from, size_t end) const {
        const char *e = b + end;
        const char *p = b + from;
        while (p < e) {
            const char *nl = (const char *)memchr(p, '\n', (size_t)(e - p));
            if (!nl) return end;
            size_t s = (size_t)(nl + 1 - b);
            if (is_record_start(b, s, end)) {
                /* a quality line that starts with '@' and is followed by a header line + '+' cannot be told apart
                 * locally only if the header's sequence line starts with '+': not a base, so the test is safe */
                return s;
            }
            p = nl + 1;
        }
        return end;
    }
    /* where the carried-over tail begins: the end of the last record that is certainly complete.  Walk the line
     * starts backwards to the last VERIFIED record start q (FASTQ: '@' with '+' two lines below); if four full lines
     * follow q the record is complete and the tail starts behind it, otherwise at q. */
    size_t last_record_start(const char *b, size_t end) const {
        size_t p = end;
        for (int lines = 0; lines < 64 && p > 0; lines++) {
            size_t q = p - 1;                            /* start of the line that ends at p (p is one past its '\n' or the buffer end) */
            while (q > 0 && b[q - 1] != '\n') q--;
            if (is_record_start(b, q, end)) {
                if (format_ == 'a') return q;
                const char *e = b + end, *l = b + q;
                int nl = 0;
                while (nl < 4) { const char *x = (const char *)memchr(l, '\n', (size_t)(e - l)); if (!x) break; l = x + 1; nl++; }
                return nl == 4 ? (size_t)(l - b) : q;
            }
            p = q;
        }
        if (format_ == 'a') {                            /* long sequence lines: look further back for the last header */
            for (size_t q = p; q > 0; q--) if (b[q - 1] == '>' && (q == 1 || b[q - 2] == '\n')) return q - 1;
        }
        return 0;
    }
    void parse(const char *p, const char *e, FlatBatch &out) const {
        out.clear();
        const size_t span = (size_t)(e - p);
        out.bases.reserve(format_ == 'q' ? span / 2 + 64 : span); out.names.reserve(span / 16 + 64);
        out.offs.reserve(span / 64 + 16); out.name_offs.reserve(span / 64 + 16);
        while (p < e) {
            while (p < e && (*p == '\n' || *p == '\r')) p++;
            if (p >= e) break;
            const char *h_end = line_end(p, e);
            const char *name = p + 1, *ne = name;
            while (ne < h_end && *ne != ' ' && *ne != '\t' && *ne != '\r') ne++;
            if (format_ == 'q') {
                const char *s = h_end < e ? h_end + 1 : e, *s_end = line_end(s, e);
                const char *se = s_end; if (se > s && se[-1] == '\r') se--;
                out.add(name, (size_t)(ne - name), s, (size_t)(se - s));
                const char *plus_end = s_end < e ? line_end(s_end + 1, e) : e;
                const char *q_end = plus_end < e ? line_end(plus_end + 1, e) : e;
                p = q_end < e ? q_end + 1 : e;
            } else {
                const char *s = h_end < e ? h_end + 1 : e;
                while (s < e && *s != '>') {
                    const char *le = line_end(s, e);
                    const char *se = le; if (se > s && se[-1] == '\r') se--;
                    out.bases.append(s, se);
                    s = le < e ? le + 1 : e;
                }
                out.names.append(name, ne); out.name_offs.push_back(out.names.size());
                out.offs.push_back(out.bases.size());
                p = s;
            }
        }
    }

    gzFile gz_ = nullptr; int threads_; size_t block_;
    char format_ = 0;
    const char *map_ = nullptr; size_t map_len_ = 0, map_pos_ = 0;                 /* plain files */
    char *gz_buf_[2] = {nullptr, nullptr}; size_t gz_cap_[2] = {0, 0}, gz_len_[2] = {0, 0};   /* gzip: two blocks */
    int gz_next_ = 0; bool gz_eof_ = false, gz_started_ = false;
    std::string gz_err_; std::thread prefetch_;
    std::vector<FlatBatch> pending_; size_t pending_pos_ = 0, pending_read_ = 0;
};

} // namespace mtbhost
#endif
