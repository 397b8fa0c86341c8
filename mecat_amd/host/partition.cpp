// partition.cpp — see partition.h.
#include "partition.h"

#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>

#define PDIE(...) do { fprintf(stderr, "[partition] " __VA_ARGS__); fprintf(stderr, "\n"); exit(1); } while (0)

static const size_t kFlushRecords = 1u << 16;      // 3.3 MB per batch buffer

PartitionWriter::PartitionWriter(const std::string& can_path, long batch_size, int min_read_size, int rank)
    : can_(can_path), batch_size_(batch_size), min_read_size_(min_read_size), rank_(rank), row_(0), max_id_seen_(-1), total_(0), finished_(false) {
    if (batch_size <= 0) PDIE("batch size must be positive");
    if (rank_ >= 0) unlink(partition_meta_name(can_.c_str(), rank_).c_str());
}

std::string partition_meta_name(const char* can_path, int rank) { return std::string(can_path) + ".partmeta.rank" + std::to_string(rank); }

PartitionWriter::~PartitionWriter() {
    if (!finished_) finish();
}

std::string PartitionWriter::part_name(long batch) const {      // generate_partition_file_name, overlaps_partition.cpp:113-121
    return can_ + ".part" + std::to_string(batch) + (rank_ >= 0 ? ".rank" + std::to_string(rank_) : std::string());
}

void PartitionWriter::flush(long b) {
    Batch& B = batches_[(size_t)b];
    // append mode, one open file at a time: no descriptor limit however many batches there are
    FILE* f = fopen(part_name(b).c_str(), B.created ? "ab" : "wb");
    if (!f) PDIE("cannot open %s: %s", part_name(b).c_str(), strerror(errno));
    B.created = true;
    if (!B.buf.empty() && fwrite(B.buf.data(), sizeof(PartRecord), B.buf.size(), f) != B.buf.size()) PDIE("write error on %s", part_name(b).c_str());
    if (fclose(f) != 0) PDIE("write error on %s", part_name(b).c_str());
    if (rank_ >= 0) {
        const std::string kn = part_name(b) + ".key";
        FILE* k = fopen(kn.c_str(), B.written ? "ab" : "wb");
        if (!k) PDIE("cannot open %s: %s", kn.c_str(), strerror(errno));
        if (!B.keys.empty() && fwrite(B.keys.data(), sizeof(Key), B.keys.size(), k) != B.keys.size()) PDIE("write error on %s", kn.c_str());
        if (fclose(k) != 0) PDIE("write error on %s", kn.c_str());
        B.keys.clear();
    }
    B.written += (long)B.buf.size();
    B.buf.clear();
}

void PartitionWriter::put(long b, int32_t seq_id, const PartRecord& r, int32_t line_read) {
    if ((size_t)b >= batches_.size()) {
        const size_t old = batches_.size();
        batches_.resize((size_t)b + 1);
        for (size_t i = old; i < batches_.size(); ++i) { batches_[i].min_id = INT_MAX; batches_[i].max_id = INT_MIN; batches_[i].created = false; batches_[i].written = 0; }
    }
    Batch& B = batches_[(size_t)b];
    B.min_id = std::min(B.min_id, seq_id);
    B.max_id = std::max(B.max_id, seq_id);
    B.buf.push_back(r);
    if (rank_ >= 0) B.keys.push_back(Key{row_, line_read});
    ++total_;
    if (B.buf.size() >= kFlushRecords) flush(b);
}

static inline void normalise(const CanRec& c, bool subject_is_target, PartRecord* d) {      // overlaps_partition.cpp:141-166
    memset(d, 0, sizeof(*d));
    if (subject_is_target) {
        d->qdir = c.qdir; d->qid = c.qid; d->qext = c.qext; d->qsize = c.qsize;
        d->sdir = c.sdir; d->sid = c.sid; d->sext = c.sext; d->ssize = c.ssize;
    } else {
        d->qdir = c.sdir; d->qid = c.sid; d->qext = c.sext; d->qsize = c.ssize;
        d->sdir = c.qdir; d->sid = c.qid; d->sext = c.qext; d->ssize = c.qsize;
    }
    d->score = c.score;
    if (d->sdir == 1) { d->qdir = 1 - d->qdir; d->sdir = 1 - d->sdir; }
}

void PartitionWriter::add(const CanRec* recs, size_t n) {
    PartRecord r;
    for (size_t i = 0; i < n; ++i) {
        const CanRec& c = recs[i];
        max_id_seen_ = std::max(max_id_seen_, std::max(c.qid, c.sid));
        if (c.qsize < min_read_size_ || c.ssize < min_read_size_) continue;
        normalise(c, false, &r);
        put(c.qid / batch_size_, c.qid, r, c.qid);      // (a `.can` line belongs to the query read in its first column)
        normalise(c, true, &r);
        put(c.sid / batch_size_, c.sid, r, c.qid);
    }
}

static inline void normalise_m4(const M4Rec& m, bool subject_is_target, PartRecord* d) {   // normalize_m4record + m4_to_candidate
    if (subject_is_target) {
        d->qdir = m.qdir; d->qid = m.qid; d->qext = m.qext; d->qsize = m.qsize; d->qoff = m.qoff; d->qend = m.qend;
        d->sdir = m.sdir; d->sid = m.sid; d->sext = m.sext; d->ssize = m.ssize; d->soff = m.soff; d->send = m.send;
    } else {                                              // reverse_m4record, common/alignment.h:72-88
        d->qdir = m.sdir; d->qid = m.sid; d->qext = m.sext; d->qsize = m.ssize; d->qoff = m.soff; d->qend = m.send;
        d->sdir = m.qdir; d->sid = m.qid; d->sext = m.qext; d->ssize = m.qsize; d->soff = m.qoff; d->send = m.qend;
    }
    d->score = m.vscore;
    if (d->sdir == 1) { d->sdir = 0; d->qdir = 1 - d->qdir; }
}

void PartitionWriter::add_m4(const M4Rec* recs, size_t n, double min_cov_ratio) {
    PartRecord r;
    for (size_t i = 0; i < n; ++i) {
        const M4Rec& m = recs[i];
        max_id_seen_ = std::max(max_id_seen_, std::max(m.qid, m.sid));      // get_qualified_m4record_counts, :44-68 (all lines)
        if (m.qsize < min_read_size_ || m.ssize < min_read_size_) continue;
        const long qm = (long)m.qend - m.qoff, qs = (long)(m.qsize * min_cov_ratio);
        const long sm = (long)m.send - m.soff, ss = (long)(m.ssize * min_cov_ratio);
        if (!(qm >= qs || sm >= ss)) continue;
        normalise_m4(m, false, &r);
        put(m.qid / batch_size_, m.qid, r, m.sid);      // (an `.m4` line carries its query read in the second column, pw_impl.cpp:467-506)
        normalise_m4(m, true, &r);
        put(m.sid / batch_size_, m.sid, r, m.sid);
    }
}

void PartitionWriter::finish() {
    if (finished_) return;
    finished_ = true;
    if (rank_ >= 0) {
        // a rank's streams: what is buffered, then the meta file — its appearance (a rename) tells rank 0 that the streams are complete
        for (long b = 0; b < (long)batches_.size(); ++b)
            if (!batches_[(size_t)b].buf.empty()) flush(b);
        const std::string mn = partition_meta_name(can_.c_str(), rank_), tmp = mn + ".tmp";
        FILE* f = fopen(tmp.c_str(), "w");
        if (!f) PDIE("cannot open %s: %s", tmp.c_str(), strerror(errno));
        fprintf(f, "%d %ld\n", max_id_seen_, (long)batches_.size());
        for (long b = 0; b < (long)batches_.size(); ++b)
            if (batches_[(size_t)b].written) fprintf(f, "%ld %ld\n", b, batches_[(size_t)b].written);
        if (fclose(f) != 0 || rename(tmp.c_str(), mn.c_str()) != 0) PDIE("write error on %s", mn.c_str());
        return;
    }
    // the reference opens (creates) a file for every batch below num_batches, also the ones that stay empty (overlaps_store.h:41-58)
    const long num_reads = (long)max_id_seen_ + 1;
    const long num_batches = (num_reads + batch_size_ - 1) / batch_size_;
    if ((long)batches_.size() < num_batches) {
        const size_t old = batches_.size();
        batches_.resize((size_t)num_batches);
        for (size_t i = old; i < batches_.size(); ++i) { batches_[i].min_id = INT_MAX; batches_[i].max_id = INT_MIN; batches_[i].created = false; batches_[i].written = 0; }
    }
    for (long b = 0; b < (long)batches_.size(); ++b)
        if (!batches_[(size_t)b].buf.empty() || !batches_[(size_t)b].created) flush(b);
    const std::string idx = can_ + ".partition_files";      // generate_partition_index_file_name, :106-111
    FILE* f = fopen(idx.c_str(), "w");
    if (!f) PDIE("cannot open %s: %s", idx.c_str(), strerror(errno));
    for (long b = 0; b < (long)batches_.size(); ++b) {
        const Batch& B = batches_[(size_t)b];
        if (B.max_id == INT_MIN) continue;                    // :212
        fprintf(f, "%s\t%d\t%d\n", part_name(b).c_str(), B.min_id, B.max_id);
        fprintf(stderr, "%s contains reads %d --- %d\n", part_name(b).c_str(), B.min_id, B.max_id);
    }
    if (fclose(f) != 0) PDIE("write error on %s", idx.c_str());
}

template <class T>
static std::vector<T> read_all(const std::string& path, long count) {
    std::vector<T> v((size_t)count);
    FILE* f = fopen(path.c_str(), "rb");
    if (!f) PDIE("cannot open %s: %s", path.c_str(), strerror(errno));
    if (count && fread(v.data(), sizeof(T), (size_t)count, f) != (size_t)count) PDIE("%s is shorter than its rank's meta file says", path.c_str());
    fclose(f);
    return v;
}

long partition_merge_ranks(const char* can_path, int world, long batch_size) {
    struct KeyT { int32_t row, read; };
    const std::string can(can_path);
    int32_t max_id = -1;
    std::vector<std::vector<long>> count((size_t)world);       // [rank][batch] records
    for (int r = 0; r < world; ++r) {
        const std::string mn = partition_meta_name(can_path, r);
        FILE* f = fopen(mn.c_str(), "r");
        if (!f) PDIE("cannot open %s: %s", mn.c_str(), strerror(errno));
        int mid;
        long nb, b, c;
        if (fscanf(f, "%d %ld", &mid, &nb) != 2) PDIE("%s: malformed", mn.c_str());
        max_id = std::max(max_id, (int32_t)mid);
        count[(size_t)r].assign((size_t)std::max(0L, nb), 0L);
        while (fscanf(f, "%ld %ld", &b, &c) == 2) {
            if (b < 0 || b >= nb) PDIE("%s: malformed", mn.c_str());
            count[(size_t)r][(size_t)b] = c;
        }
        fclose(f);
    }
    const long num_reads = (long)max_id + 1, num_batches = (num_reads + batch_size - 1) / batch_size;
    long nfiles = num_batches;
    for (int r = 0; r < world; ++r) nfiles = std::max(nfiles, (long)count[(size_t)r].size());
    const std::string idx = can + ".partition_files";
    FILE* fi = fopen(idx.c_str(), "w");
    if (!fi) PDIE("cannot open %s: %s", idx.c_str(), strerror(errno));
    long total = 0;
    for (long b = 0; b < nfiles; ++b) {
        std::vector<std::vector<PartRecord>> recs((size_t)world);
        std::vector<std::vector<KeyT>> keys((size_t)world);
        long n = 0;
        for (int r = 0; r < world; ++r) {
            const long c = b < (long)count[(size_t)r].size() ? count[(size_t)r][(size_t)b] : 0;
            if (!c) continue;
            const std::string pn = can + ".part" + std::to_string(b) + ".rank" + std::to_string(r);
            recs[(size_t)r] = read_all<PartRecord>(pn, c);
            keys[(size_t)r] = read_all<KeyT>(pn + ".key", c);
            unlink(pn.c_str());
            unlink((pn + ".key").c_str());
            n += c;
        }
        // a rank's stream is in (row, read) order, and the lines of one (row, read) are all on one rank: merge by the smallest head
        std::vector<PartRecord> out;
        out.reserve((size_t)n);
        std::vector<size_t> head((size_t)world, 0);
        int32_t mn = INT_MAX, mx = INT_MIN;
        for (long k = 0; k < n;) {
            int best = -1;
            for (int r = 0; r < world; ++r) {
                if (head[(size_t)r] >= keys[(size_t)r].size()) continue;
                if (best < 0) { best = r; continue; }
                const KeyT &a = keys[(size_t)r][head[(size_t)r]], &c = keys[(size_t)best][head[(size_t)best]];
                if (a.row < c.row || (a.row == c.row && a.read < c.read)) best = r;
            }
            const KeyT key = keys[(size_t)best][head[(size_t)best]];
            size_t& h = head[(size_t)best];
            while (h < keys[(size_t)best].size() && keys[(size_t)best][h].row == key.row && keys[(size_t)best][h].read == key.read) {
                const PartRecord& pr = recs[(size_t)best][h];
                mn = std::min(mn, pr.sid);                  // (a record sits in the batch of its template read, PartitionWriter::add)
                mx = std::max(mx, pr.sid);
                out.push_back(pr);
                ++h;
                ++k;
            }
        }
        const std::string pn = can + ".part" + std::to_string(b);
        FILE* f = fopen(pn.c_str(), "wb");
        if (!f) PDIE("cannot open %s: %s", pn.c_str(), strerror(errno));
        if (!out.empty() && fwrite(out.data(), sizeof(PartRecord), out.size(), f) != out.size()) PDIE("write error on %s", pn.c_str());
        if (fclose(f) != 0) PDIE("write error on %s", pn.c_str());
        total += n;
        if (mx == INT_MIN) continue;
        fprintf(fi, "%s\t%d\t%d\n", pn.c_str(), mn, mx);
        fprintf(stderr, "%s contains reads %d --- %d\n", pn.c_str(), mn, mx);
    }
    if (fclose(fi) != 0) PDIE("write error on %s", idx.c_str());
    for (int r = 0; r < world; ++r) unlink(partition_meta_name(can_path, r).c_str());
    return total;
}

// whitespace separated numbers; column `skip` (the m4 identity, a real) is stepped over.  Returns the number of integers read.
static const char* parse_ints(const char* p, const char* end, int32_t* v, int want, int skip, int* got) {
    int k = 0;
    for (int col = 0; k < want; ++col) {
        while (p < end && (*p == ' ' || *p == '\t')) ++p;
        if (p >= end || *p == '\n') break;
        if (col == skip) {
            while (p < end && *p != ' ' && *p != '\t' && *p != '\n') ++p;
            continue;
        }
        bool neg = false;
        if (*p == '-' || *p == '+') { neg = *p == '-'; ++p; }
        if (p >= end || *p < '0' || *p > '9') break;
        long x = 0;
        while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; }
        v[k++] = (int32_t)(neg ? -x : x);
    }
    while (p < end && *p != '\n') ++p;
    if (p < end) ++p;
    *got = k;
    return p;
}

// REC = CanRec (9 ints per line) or M4Rec (13 ints + the identity column); FEED(writer, recs, n)
template <class REC, class FEED>
static long partition_text(const char* path, long batch_size, int min_read_size, int num_threads, int ints, int skip, const char* what,
                           FEED feed) {
    const int fd = open(path, O_RDONLY);
    if (fd < 0) PDIE("cannot open %s: %s", path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0) PDIE("cannot stat %s", path);
    const size_t size = (size_t)sb.st_size;
    PartitionWriter w(path, batch_size, min_read_size);
    if (size == 0) { close(fd); w.finish(); return 0; }
    const char* base = (const char*)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) PDIE("cannot map %s: %s", path, strerror(errno));
    const int nt = std::max(1, std::min(num_threads, 64));
    const size_t slab = (size_t)64 << 20;                     // text bytes per thread and round
    std::vector<std::vector<REC>> recs((size_t)nt);
    std::vector<int> bad((size_t)nt, 0);
    size_t pos = 0;
    while (pos < size) {
        // cut [pos, pos + nt * slab) into nt pieces at line ends
        std::vector<size_t> cut((size_t)nt + 1);
        cut[0] = pos;
        for (int t = 1; t <= nt; ++t) {
            size_t c = std::min(size, pos + (size_t)t * slab);
            while (c < size && c > cut[(size_t)t - 1] && base[c - 1] != '\n') ++c;
            cut[(size_t)t] = std::max(c, cut[(size_t)t - 1]);
        }
        auto work = [&](int t) {
            std::vector<REC>& o = recs[(size_t)t];
            o.clear();
            const char* p = base + cut[(size_t)t];
            const char* e = base + cut[(size_t)t + 1];
            REC r;
            while (p < e) {
                int got;
                p = parse_ints(p, e, (int32_t*)&r, ints, skip, &got);      // (an empty line repeats the previous record in the reference: not supported)
                if (got != ints) { bad[(size_t)t] = got + 1; break; }
                o.push_back(r);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (int t = 0; t < nt; ++t) {
            if (bad[(size_t)t] == 12 && ints == 13) {         // get_qualified_m4record_counts, overlaps_partition.cpp:56-59
                w.abandon();
                PDIE("no gapped start position is provided, please make sure that you have run 'mecat2pw' with option '-g 1'");
            }
            if (bad[(size_t)t]) { w.abandon(); PDIE("%s: malformed line (%s expected)", path, what); }
            feed(w, recs[(size_t)t].data(), recs[(size_t)t].size());
        }
        pos = cut[(size_t)nt];
    }
    munmap((void*)base, size);
    close(fd);
    w.finish();
    return w.records_written();
}

long partition_candidates_text(const char* can_path, long batch_size, int min_read_size, int num_threads) {
    return partition_text<CanRec>(can_path, batch_size, min_read_size, num_threads, 9, -1, "nine integers per line",
                                  [](PartitionWriter& w, const CanRec* r, size_t n) { w.add(r, n); });
}

long partition_m4_text(const char* m4_path, double min_cov_ratio, long batch_size, int min_read_size, int num_threads) {
    return partition_text<M4Rec>(m4_path, batch_size, min_read_size, num_threads, 13, 2, "14 columns per line",
                                 [min_cov_ratio](PartitionWriter& w, const M4Rec* r, size_t n) { w.add_m4(r, n, min_cov_ratio); });
}
