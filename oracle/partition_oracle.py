"""partition_oracle.py — TEST INFRASTRUCTURE.  Plain restatement of mecat2cns' partition_candidates (reference
src/mecat2cns/overlaps_partition.cpp:175-224) for small inputs: which 13-int records land in which `<can>.part<k>` file, in which
order, and the lines of `<can>.partition_files`.  Pinned against the compiled reference (oracle/_ref/libref_part.so) by
tests/test_partition_cpu.py.  Only tests import this.

The reference leaves qoff/qend/soff/send (record ints 4, 5, 10, 11) uninitialised; DEFINED lists the ints that carry data."""

DEFINED = (0, 1, 2, 3, 6, 7, 8, 9, 12)


def parse_can(text):
    """operator>>(istream&, ExtensionCandidate&), common/alignment.cpp:8-16: qid sid qdir sdir qext sext score qsize ssize"""
    return [tuple(int(x) for x in ln.split()) for ln in text.splitlines()]


def normalise(c, subject_is_target):
    """normalise_candidate, overlaps_partition.cpp:141-166 -> (qdir,qid,qext,qsize,qoff,qend,sdir,sid,sext,ssize,soff,send,score)"""
    qid, sid, qdir, sdir, qext, sext, score, qsize, ssize = c
    if subject_is_target:
        d = [qdir, qid, qext, qsize, 0, 0, sdir, sid, sext, ssize, 0, 0, score]
    else:
        d = [sdir, sid, sext, ssize, 0, 0, qdir, qid, qext, qsize, 0, 0, score]
    if d[6] == 1:                      # template reversed: flip both strands (REVERSE_STRAND, defs.h:201)
        d[0] = 1 - d[0]
        d[6] = 1 - d[6]
    return tuple(d)


def partition(cands, batch_size, min_read_size):
    """-> (files: {k: [record, ...]} for every batch below num_batches, index: [(k, min_id, max_id), ...])"""
    num_reads = max([max(c[0], c[1]) for c in cands], default=-1) + 1          # get_num_reads, :125-139 (all lines)
    num_batches = (num_reads + batch_size - 1) // batch_size
    files = {k: [] for k in range(num_batches)}
    lo, hi = {}, {}
    for c in cands:                                                              # :199-208
        if c[7] < min_read_size or c[8] < min_read_size:
            continue
        for rid, subject_is_target in ((c[0], False), (c[1], True)):
            k = rid // batch_size
            files[k].append(normalise(c, subject_is_target))
            lo[k] = min(lo.get(k, rid), rid)
            hi[k] = max(hi.get(k, rid), rid)
    return files, [(k, lo[k], hi[k]) for k in sorted(lo)]                        # :210-215: batches without records are skipped


def parse_m4(text):
    """operator>>(istream&, M4Record&), common/alignment.cpp:34-56 with -g 1: qid sid ident vscore qdir qoff qend qsize sdir soff send
    ssize qext sext -> (qid, sid, vscore, qdir, qoff, qend, qsize, sdir, soff, send, ssize, qext, sext)"""
    out = []
    for ln in text.splitlines():
        f = ln.split()
        out.append(tuple(int(x) for x in f[:2] + f[3:14]))
    return out


def normalise_m4(m, subject_is_target):
    """normalize_m4record (common/alignment.h:90-102) + m4_to_candidate (:170-186)"""
    qid, sid, vscore, qdir, qoff, qend, qsize, sdir, soff, send, ssize, qext, sext = m
    if subject_is_target:
        d = [qdir, qid, qext, qsize, qoff, qend, sdir, sid, sext, ssize, soff, send, vscore]
    else:
        d = [sdir, sid, sext, ssize, soff, send, qdir, qid, qext, qsize, qoff, qend, vscore]
    if d[6] == 1:
        d[6] = 0
        d[0] = 1 - d[0]
    return tuple(d)


def partition_m4(recs, min_cov_ratio, batch_size, min_read_size):
    """partition_m4records, overlaps_partition.cpp:344-412 (the repeat-read set is empty there)"""
    num_reads = max([max(m[0], m[1]) for m in recs], default=-1) + 1
    num_batches = (num_reads + batch_size - 1) // batch_size
    files = {k: [] for k in range(num_batches)}
    lo, hi = {}, {}
    for m in recs:
        if m[6] < min_read_size or m[10] < min_read_size:
            continue
        if not (m[5] - m[4] >= int(m[6] * min_cov_ratio) or m[9] - m[8] >= int(m[10] * min_cov_ratio)):      # :17-25
            continue
        for rid, subject_is_target in ((m[0], False), (m[1], True)):
            k = rid // batch_size
            files[k].append(normalise_m4(m, subject_is_target))
            lo[k] = min(lo.get(k, rid), rid)
            hi[k] = max(hi.get(k, rid), rid)
    return files, [(k, lo[k], hi[k]) for k in sorted(lo)]
