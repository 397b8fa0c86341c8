// partition.h — the on-disk step between mecat2pw -j 0 and mecat2cns: the candidate file split
// into per-read-batch binary files, as mecat2cns' partition_candidates does at start-up
// (reference src/mecat2cns/overlaps_partition.cpp:175-224, record = ExtensionCandidate of common/alignment.h:8-13,
// writer = PartitionResultsWriter of overlaps_store.h:10-104).  Same file names, same record order, same index file.
#pragma once

#include <stdint.h>

#include <string>
#include <vector>

struct CanRec {            // one `.can` line in column order (common/alignment.cpp:18-32)
    int32_t qid, sid, qdir, sdir, qext, sext, score, qsize, ssize;
};

struct PartRecord {        // ExtensionCandidate, 13 ints = 52 bytes (common/alignment.h:8-13)
    int32_t qdir, qid, qext, qsize, qoff, qend;
    int32_t sdir, sid, sext, ssize, soff, send;
    int32_t score;
};

// Fed with the candidate lines in the order they stand in `can_path`, writes `<can_path>.part<k>` for read batch k
// (k = id / batch_size) and, in finish(), `<can_path>.partition_files`.  For every line whose two reads are at least
// min_read_size long: the line seen from the query's side goes to the query's batch, then the line as it stands to the
// subject's batch (overlaps_partition.cpp:199-208), both normalised so that the template (sid) is forward (:141-166).
// The reference leaves qoff/qend/soff/send of the records uninitialised (the `.can` parser fills nine fields); they are
// written as 0 here.  The reference opens at most num_files partition files at a time and re-reads the text once per
// group; the bytes written do not depend on that, so this writer has no such parameter.
class PartitionWriter {
public:
    PartitionWriter(const std::string& can_path, long batch_size, int min_read_size);
    ~PartitionWriter();
    void add(const CanRec* recs, size_t n);
    void finish();
    void abandon() { finished_ = true; }      // stop without an index file (the caller will partition the text instead)
    long records_written() const { return total_; }

private:
    struct Batch {
        std::vector<PartRecord> buf;
        int32_t min_id, max_id;
        bool created;
    };
    void put(long batch, int32_t seq_id, const PartRecord& r);
    void flush(long batch);
    std::string part_name(long batch) const;

    std::string can_;
    long batch_size_;
    int min_read_size_;
    std::vector<Batch> batches_;
    int32_t max_id_seen_;      // over ALL lines, as get_num_reads (overlaps_partition.cpp:125-139)
    long total_;
    bool finished_;
};

// The same from the text file itself (what mecat2cns would parse), scanned by `num_threads` threads over the mapped file.
// Returns the number of records written.
long partition_candidates_text(const char* can_path, long batch_size, int min_read_size, int num_threads);
