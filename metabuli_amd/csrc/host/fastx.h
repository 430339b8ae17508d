/* fastx.h -- host ingest for the classify driver: FASTA / FASTQ (plain or gzip) -> flat read batches.
 *
 * The reference parses reads with one kseq producer per file (KmerExtractor.cpp:122-171, KSeqWrapper); at
 * GPU rates the parser is the bottleneck, so this reader works on large blocks:
 * a block is cut at record boundaries into one piece per worker thread, every worker scans its piece
 * with memchr and appends bases / names to its own flat buffers, and the pieces are concatenated in
 * order.  Output is the layout the C ABI takes: concatenated bases + u64 offsets, names likewise.
 *
 * Formats: FASTQ with four lines per record (what sequencers and the reference's test data use;
 * multi-line FASTQ is not supported), FASTA with sequences over any number of lines.  Names end at the
 * first blank, as kseq's do.  Lower-case bases and IUPAC codes pass through (the extractor maps them).
 */
#ifndef MTB_FASTX_H
#define MTB_FASTX_H
#include <zlib.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

namespace mtbhost {

struct FlatBatch {
    std::vector<char> bases; std::vector<uint64_t> offs{0};
    std::vector<char> names; std::vector<uint64_t> name_offs{0};
    size_t size() const { return offs.size() - 1; }
    void clear() { bases.clear(); offs.assign(1, 0); names.clear(); name_offs.assign(1, 0); }
    void add(const char *name, size_t name_len, const char *seq, size_t seq_len) {
        names.insert(names.end(), name, name + name_len); name_offs.push_back(names.size());
        bases.insert(bases.end(), seq, seq + seq_len); offs.push_back(bases.size());
    }
    void append(const FlatBatch &o) {
        const uint64_t b0 = bases.size(), n0 = names.size();
        offs.reserve(offs.size() + o.offs.size()); name_offs.reserve(name_offs.size() + o.name_offs.size());
        bases.insert(bases.end(), o.bases.begin(), o.bases.end());
        names.insert(names.end(), o.names.begin(), o.names.end());
        for (size_t i = 1; i < o.offs.size(); i++) offs.push_back(b0 + o.offs[i]);
        for (size_t i = 1; i < o.name_offs.size(); i++) name_offs.push_back(n0 + o.name_offs[i]);
    }
    std::string name(size_t i) const { return std::string(names.data() + name_offs[i], names.data() + name_offs[i + 1]); }
};

class FastxReader {
public:
    FastxReader(const std::string &path, int threads, size_t block_bytes = 64u << 20)
        : threads_(threads < 1 ? 1 : threads), block_(block_bytes) {
        /* gzip (magic 1f 8b) goes through zlib; plain files are read directly -- zlib's transparent mode copies through its
         * own buffer at ~1.5 GB/s, which was most of the parse stage */
        FILE *f = fopen(path.c_str(), "rb");
        if (!f) throw std::runtime_error("cannot open " + path);
        unsigned char magic[2] = {0, 0};
        size_t got = fread(magic, 1, 2, f);
        if (got == 2 && magic[0] == 0x1f && magic[1] == 0x8b) {
            fclose(f);
            gz_ = gzopen(path.c_str(), "rb");
            if (!gz_) throw std::runtime_error("cannot open " + path);
            gzbuffer(gz_, 1u << 20);
        } else {
            rewind(f);
            setvbuf(f, nullptr, _IONBF, 0);
            plain_ = f;
        }
    }
    ~FastxReader() { if (gz_) gzclose(gz_); if (plain_) fclose(plain_); }
    FastxReader(const FastxReader &) = delete;

    /* appends up to max_reads records to `out`; returns false when the file is exhausted and nothing was added */
    bool next_batch(size_t max_reads, FlatBatch &out) {
        size_t before = out.size();
        while (out.size() - before < max_reads) {
            if (pending_.size() == pending_pos_) {
                if (!fill()) break;
                /* size the batch ONCE from the first block's averages (growing block by block re-copies everything each time) */
                if (out.size() == before) {
                    size_t nb = 0, nn = 0, nr = 0;
                    for (const FlatBatch &q : pending_) { nb += q.bases.size(); nn += q.names.size(); nr += q.size(); }
                    if (nr) {
                        const double f = 1.05 * (double)max_reads / (double)nr;
                        const size_t cap_b = (size_t)((double)nb * f) + 4096, cap_n = (size_t)((double)nn * f) + 4096;
                        if (cap_b < ((size_t)8 << 30)) {            /* absurd --max-reads: let the vectors grow instead */
                            out.bases.reserve(out.bases.size() + cap_b); out.names.reserve(out.names.size() + cap_n);
                            out.offs.reserve(out.offs.size() + max_reads + 1); out.name_offs.reserve(out.name_offs.size() + max_reads + 1);
                        }
                    }
                }
            }
            size_t want = max_reads - (out.size() - before);
            drain(want, out);
        }
        return out.size() > before;
    }

private:
    /* one parsed block waits in pending_ (FlatBatch pieces in order) until batches have consumed it */
    bool fill() {
        if (eof_ && carry_.empty()) return false;
        /* one persistent, never zero-filled block buffer; the file is read straight behind the carried-over tail */
        const size_t keep = carry_.size();
        if (cap_ < keep + block_ + 1) { cap_ = keep + block_ + 1; raw_.reset(new char[cap_]); }
        char *const buf = raw_.get();
        if (keep) memcpy(buf, carry_.data(), keep);
        carry_.clear();
        size_t len = keep;
        if (!eof_) {
            size_t got = 0;
            while (got < block_) {                      /* gzread takes an unsigned length */
                long r;
                if (plain_) { r = (long)fread(buf + keep + got, 1, block_ - got, plain_); if (r == 0 && ferror(plain_)) throw std::runtime_error("read error"); }
                else { r = gzread(gz_, buf + keep + got, (unsigned)std::min<size_t>(block_ - got, 1u << 30)); if (r < 0) throw std::runtime_error("read error (corrupt gzip stream?)"); }
                if (r == 0) { eof_ = true; break; }
                got += (size_t)r;
            }
            len += got;
        }
        if (len == 0) return false;
        if (format_ == 0) {
            size_t p = 0; while (p < len && (buf[p] == '\n' || buf[p] == '\r')) p++;
            if (p == len) return false;
            format_ = buf[p] == '@' ? 'q' : (buf[p] == '>' ? 'a' : 0);
            if (!format_) throw std::runtime_error("input is neither FASTA nor FASTQ");
        }
        /* the block ends inside a record unless the file ended: keep the incomplete tail for the next block */
        size_t end = len;
        if (!eof_) {
            end = last_record_start(buf, len);
            if (end == 0) {                              /* one record larger than the block: grow and retry */
                carry_.assign(buf, buf + len); block_ *= 2; return fill();
            }
            carry_.assign(buf + end, buf + len);
        }
        /* cut [0, end) into pieces at record starts, parse in parallel */
        std::vector<size_t> cut{0};
        for (int t = 1; t < threads_; t++) {
            size_t guess = end / (size_t)threads_ * (size_t)t;
            size_t s = next_record_start(buf, guess, end);
            if (s > cut.back() && s < end) cut.push_back(s);
        }
        cut.push_back(end);
        pending_.assign(cut.size() - 1, FlatBatch());
        pending_pos_ = 0; pending_read_ = 0;
        std::vector<std::thread> th;
        for (size_t k = 0; k + 1 < cut.size(); k++)
            th.emplace_back([&, k]() { parse(buf + cut[k], buf + cut[k + 1], pending_[k]); });
        for (auto &t : th) t.join();
        return true;
    }
    void drain(size_t want, FlatBatch &out) {
        while (want && pending_pos_ < pending_.size()) {
            FlatBatch &p = pending_[pending_pos_];
            size_t avail = p.size() - pending_read_;
            if (pending_read_ == 0 && avail <= want) { out.append(p); want -= avail; }
            else {
                size_t take = std::min(avail, want);
                for (size_t i = pending_read_; i < pending_read_ + take; i++)
                    out.add(p.names.data() + p.name_offs[i], p.name_offs[i + 1] - p.name_offs[i], p.bases.data() + p.offs[i], p.offs[i + 1] - p.offs[i]);
                want -= take; pending_read_ += take;
                if (pending_read_ < p.size()) return;
            }
            p.clear(); pending_pos_++; pending_read_ = 0;
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
                size_t n0 = out.bases.size();
                const char *s = h_end < e ? h_end + 1 : e;
                while (s < e && *s != '>') {
                    const char *le = line_end(s, e);
                    const char *se = le; if (se > s && se[-1] == '\r') se--;
                    out.bases.insert(out.bases.end(), s, se);
                    s = le < e ? le + 1 : e;
                }
                out.names.insert(out.names.end(), name, ne); out.name_offs.push_back(out.names.size());
                out.offs.push_back(out.bases.size());
                (void)n0;
                p = s;
            }
        }
    }

    gzFile gz_ = nullptr; FILE *plain_ = nullptr; int threads_; size_t block_;
    bool eof_ = false; char format_ = 0;
    std::vector<char> carry_;
    std::unique_ptr<char[]> raw_; size_t cap_ = 0;
    std::vector<FlatBatch> pending_; size_t pending_pos_ = 0, pending_read_ = 0;
};

} // namespace mtbhost
#endif
