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

struct M4Rec {             // one `.m4` line written with -g 1 (common/alignment.cpp:34-56; ident is not used here)
    int32_t qid, sid, vscore, qdir, qoff, qend, qsize, sdir, soff, send, ssize, qext, sext;
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
//
// Multi-process runs (rank >= 0): every rank writes the records of ITS lines as streams of its own, `<can_path>.part<k>.rank<r>`, with a
// key per record beside them (`.key`: grid row, query read of the line — the order the lines have in a one-process run's output) and,
// in finish(), `<can_path>.partmeta.rank<r>`; partition_merge_ranks() on rank 0 merges the streams of a batch by key into
// `<can_path>.part<k>` and writes the index file: the same bytes a one-process run writes, and no text is parsed.
class PartitionWriter {
public:
    PartitionWriter(const std::string& can_path, long batch_size, int min_read_size, int rank = -1);
    void set_row(int row) { row_ = row; }      // the grid row (reference volume) the lines added from now on belong to
    ~PartitionWriter();
    void add(const CanRec* recs, size_t n);
    // the `-j 1 -g 1` flavour (partition_m4records, overlaps_partition.cpp:344-412): lines whose reads are long enough and of
    // which min_cov_ratio of the query or of the subject is covered (check_m4record_mapping_range, :17-25); all 13 ints of a
    // record are defined here (m4_to_candidate, common/alignment.h:170-186).  mecat2cns passes its -r value minus 0.02
    // (reads_correction_m4.cpp:79-80).
    void add_m4(const M4Rec* recs, size_t n, double min_cov_ratio);
    void finish();
    void abandon() { finished_ = true; }      // stop without an index file (the caller will partition the text instead)
    long records_written() const { return total_; }

private:
    struct Key { int32_t row, read; };
    struct Batch {
        std::vector<PartRecord> buf;
        std::vector<Key> keys;     // rank streams only
        int32_t min_id, max_id;
        bool created;
        long written;
    };
    void put(long batch, int32_t seq_id, const PartRecord& r, int32_t line_read);
    void flush(long batch);
    std::string part_name(long batch) const;

    std::string can_;
    long batch_size_;
    int min_read_size_;
    std::vector<Batch> batches_;
    int rank_, row_;
    int32_t max_id_seen_;      // over ALL lines, as get_num_reads (overlaps_partition.cpp:125-139)
    long total_;
    bool finished_;
};

// rank 0 of a multi-process run, once every rank's `<can_path>.partmeta.rank<r>` exists: merges the ranks' streams (see PartitionWriter)
// into the partition files and the index file of a one-process run and removes the streams.  Returns the number of records written.
long partition_merge_ranks(const char* can_path, int world, long batch_size);
std::string partition_meta_name(const char* can_path, int rank);

// The same from the text file itself (what mecat2cns would parse), scanned by `num_threads` threads over the mapped file.
// Returns the number of records written.
long partition_candidates_text(const char* can_path, long batch_size, int min_read_size, int num_threads);
// `.m4` text written with -g 1 (14 columns; 12 columns is the reference's "run with -g 1" error)
long partition_m4_text(const char* m4_path, double min_cov_ratio, long batch_size, int min_read_size, int num_threads);
