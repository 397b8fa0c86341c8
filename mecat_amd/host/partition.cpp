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

PartitionWriter::PartitionWriter(const std::string& can_path, long batch_size, int min_read_size)
    : can_(can_path), batch_size_(batch_size), min_read_size_(min_read_size), max_id_seen_(-1), total_(0), finished_(false) {
    if (batch_size <= 0) PDIE("batch size must be positive");
}

PartitionWriter::~PartitionWriter() {
    if (!finished_) finish();
}

std::string PartitionWriter::part_name(long batch) const {      // generate_partition_file_name, overlaps_partition.cpp:113-121
    return can_ + ".part" + std::to_string(batch);
}

void PartitionWriter::flush(long b) {
    Batch& B = batches_[(size_t)b];
    // append mode, one open file at a time: no descriptor limit however many batches there are
    FILE* f = fopen(part_name(b).c_str(), B.created ? "ab" : "wb");
    if (!f) PDIE("cannot open %s: %s", part_name(b).c_str(), strerror(errno));
    B.created = true;
    if (!B.buf.empty() && fwrite(B.buf.data(), sizeof(PartRecord), B.buf.size(), f) != B.buf.size()) PDIE("write error on %s", part_name(b).c_str());
    if (fclose(f) != 0) PDIE("write error on %s", part_name(b).c_str());
    B.buf.clear();
}

void PartitionWriter::put(long b, int32_t seq_id, const PartRecord& r) {
    if ((size_t)b >= batches_.size()) {
        const size_t old = batches_.size();
        batches_.resize((size_t)b + 1);
        for (size_t i = old; i < batches_.size(); ++i) { batches_[i].min_id = INT_MAX; batches_[i].max_id = INT_MIN; batches_[i].created = false; }
    }
    Batch& B = batches_[(size_t)b];
    B.min_id = std::min(B.min_id, seq_id);
    B.max_id = std::max(B.max_id, seq_id);
    B.buf.push_back(r);
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
        put(c.qid / batch_size_, c.qid, r);
        normalise(c, true, &r);
        put(c.sid / batch_size_, c.sid, r);
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
        put(m.qid / batch_size_, m.qid, r);
        normalise_m4(m, true, &r);
        put(m.sid / batch_size_, m.sid, r);
    }
}

void PartitionWriter::finish() {
    if (finished_) return;
    finished_ = true;
    // the reference opens (creates) a file for every batch below num_batches, also the ones that stay empty (overlaps_store.h:41-58)
    const long num_reads = (long)max_id_seen_ + 1;
    const long num_batches = (num_reads + batch_size_ - 1) / batch_size_;
    if ((long)batches_.size() < num_batches) {
        const size_t old = batches_.size();
        batches_.resize((size_t)num_batches);
        for (size_t i = old; i < batches_.size(); ++i) { batches_[i].min_id = INT_MAX; batches_[i].max_id = INT_MIN; batches_[i].created = false; }
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
