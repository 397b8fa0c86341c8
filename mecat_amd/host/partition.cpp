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

// nine whitespace separated integers per line (operator>>, common/alignment.cpp:8-16)
static const char* parse_line(const char* p, const char* end, CanRec* r, bool* ok) {
    int32_t v[9];
    for (int k = 0; k < 9; ++k) {
        while (p < end && (*p == ' ' || *p == '\t')) ++p;
        bool neg = false;
        if (p < end && (*p == '-' || *p == '+')) { neg = *p == '-'; ++p; }
        if (p >= end || *p < '0' || *p > '9') { *ok = false; return p; }
        long x = 0;
        while (p < end && *p >= '0' && *p <= '9') { x = x * 10 + (*p - '0'); ++p; }
        v[k] = (int32_t)(neg ? -x : x);
    }
    while (p < end && *p != '\n') ++p;
    if (p < end) ++p;
    r->qid = v[0]; r->sid = v[1]; r->qdir = v[2]; r->sdir = v[3]; r->qext = v[4]; r->sext = v[5]; r->score = v[6]; r->qsize = v[7]; r->ssize = v[8];
    *ok = true;
    return p;
}

long partition_candidates_text(const char* can_path, long batch_size, int min_read_size, int num_threads) {
    const int fd = open(can_path, O_RDONLY);
    if (fd < 0) PDIE("cannot open %s: %s", can_path, strerror(errno));
    struct stat sb;
    if (fstat(fd, &sb) != 0) PDIE("cannot stat %s", can_path);
    const size_t size = (size_t)sb.st_size;
    PartitionWriter w(can_path, batch_size, min_read_size);
    if (size == 0) { close(fd); w.finish(); return 0; }
    const char* base = (const char*)mmap(NULL, size, PROT_READ, MAP_PRIVATE, fd, 0);
    if (base == MAP_FAILED) PDIE("cannot map %s: %s", can_path, strerror(errno));
    const int nt = std::max(1, std::min(num_threads, 64));
    const size_t slab = (size_t)64 << 20;                     // text bytes per thread and round
    std::vector<std::vector<CanRec>> recs((size_t)nt);
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
            std::vector<CanRec>& o = recs[(size_t)t];
            o.clear();
            const char* p = base + cut[(size_t)t];
            const char* e = base + cut[(size_t)t + 1];
            CanRec r;
            while (p < e) {
                if (*p == '\n') { ++p; bad[(size_t)t] = 1; continue; }      // an empty line repeats the previous record in the reference: not supported
                bool ok;
                p = parse_line(p, e, &r, &ok);
                if (!ok) { bad[(size_t)t] = 1; break; }
                o.push_back(r);
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto& x : th) x.join();
        for (int t = 0; t < nt; ++t) {
            if (bad[(size_t)t]) PDIE("%s: malformed candidate line (nine integers per line expected)", can_path);
            w.add(recs[(size_t)t].data(), recs[(size_t)t].size());
        }
        pos = cut[(size_t)nt];
    }
    munmap((void*)base, size);
    close(fd);
    w.finish();
    return w.records_written();
}
