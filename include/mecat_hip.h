/*
 * mecat_hip.h — C ABI of libmecat_hip.so: the MI355X (gfx950) implementation of mecat2pw's hot path.
 *
 * This is the drop-in boundary.  MECAT has no plugin/FFI layer; the hot path sits behind three
 * function seams inside the mecat2pw binary, and each entry point below replaces one of them (reference file:line,
 * relative to /root/reference/src/):
 *
 *   mhip_index_build      <- create_ref_index / destroy_ref_index      common/lookup_table.h:13-17, lookup_table.cpp:63
 *   mhip_seed_reads       <- seeding + get_candidates, both strands     mecat2pw/pw_impl.cpp:241, :288, loop at :752-765
 *   mhip_align_candidates <- GapAligner::go + accessors (DiffAligner)   common/gapalign.h:4-25, diff_gapalign.cpp:294,
 *                                                                       called at mecat2pw/pw_impl.cpp:688-697
 *   mhip_xalign_candidates<- GapAligner::go + accessors (XdropAligner)  common/xdrop_gapalign.cpp:359 (nanopore, -x 1)
 *   mhip_volume_upload    <- load_volume's in-memory volume_t           common/split_database.h:18-24, split_database.cpp:155
 *   mhip_params           <- the file-static tuning values              mecat2pw/pw_impl.cpp:18-26, set at :838-851
 *
 * Conventions (mirroring the reference where it has any): plain C types only; opaque handles with explicit *_free;
 * every call returns 0 on success and non-zero on failure with the message in mhip_last_error() (the reference
 * abort()s instead — the host driver maps non-zero to abort-with-message).  Handles are immutable after creation and
 * may be shared by host threads; a context (one HIP stream + scratch) must not be used by two threads at once.
 * There is NO CPU fallback: every entry point fails if no gfx950 device is usable.
 *
 * INTEGRATION.md shows the binding a MECAT maintainer would add to call these from pw_impl.cpp.
 */
#ifndef MECAT_HIP_H
#define MECAT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MHIP_ABI_VERSION 1
#define MHIP_KMER_SIZE 13          /* pw_impl.cpp:20 */
#define MHIP_MAX_SEQ_SIZE 500000   /* common/defs.h:202 */

typedef struct mhip_ctx mhip_ctx;        /* device + stream + scratch arena */
typedef struct mhip_volume mhip_volume;  /* device-resident 2-bit volume (volume_t) */
typedef struct mhip_index mhip_index;    /* device-resident k-mer index (ref_index) */

typedef struct { int32_t offset, size; } mhip_offset_t;      /* offset_t, split_database.h:9-11 */

/* candidate_save, mecat2pw/pw_impl.h:21-25 — same field order, 48 bytes; `chain` is 0/1 (FWD/REV, common/defs.h:196-197),
   the 'F'/'R' spelling of pairwise_mapping is derived by the caller */
typedef struct {
    int32_t loc1, loc2, left1, left2, right1, right2, score, num1, num2, readno, readstart;
    int32_t chain;
} mhip_candidate;

typedef struct {
    int32_t maxc;            /* -n  MAXC, default 100 (pw_options.cpp:9) */
    int32_t min_align_size;  /* -a  2000 pacbio / 500 nanopore */
    int32_t min_kmer_match;  /* -k  4 / 2 */
    int32_t min_kmer_dist;   /* 1800 / 400 (pw_impl.cpp:845,848) */
    int32_t tech;            /* 0 pacbio (DiffAligner), 1 nanopore */
    double  ddfs_cutoff;     /* 0.25 (pw_impl.cpp:21-23) */
} mhip_params;

/* what mecat2pw consumes of an alignment: go()'s return value, the four coordinates and the identity counts
   (ident = 100.0 * matches / columns is formed by the caller in double, diff_gapalign.h:90-97) */
typedef struct {
    int32_t ok, query_start, query_end, target_start, target_end, matches, columns;
    int32_t blocks;          /* number of 500-bp blocks aligned (work counter) */
} mhip_aln_result;

/* one alignment job: candidate `cand` of query read `qid_local` (index inside the query volume) */
typedef struct {
    int32_t qid_local;       /* read index in the query volume */
    int32_t sid_local;       /* read index in the reference volume (candidate.readno - ref.start_read_id) */
    int32_t chain;           /* 0 forward query, 1 reverse-complemented query */
    int32_t qstart, sstart;  /* extension start points AFTER the +kmer_size/2 adjustment of pw_impl.cpp:681-685 */
} mhip_aln_job;

int  mhip_abi_version(void);
const char* mhip_last_error(void);
int  mhip_device_count(void);

/* Page-locked host memory for the buffers handed to the host-pointer entry points (mhip_seed_reads,
 * mhip_align_candidates ...): optional, any host pointer works, but pageable memory moves at a fraction of the link rate.
 * Additive (the reference has no counterpart: its buffers never leave the host). */
int  mhip_host_alloc(size_t bytes, void** out);
void mhip_host_free(void* p);
/* page-lock / release a caller-owned host buffer around the calls that copy from or into it (0 = locked; a failure leaves it pageable,
 * which only costs copy speed) */
int  mhip_host_register(void* p, size_t bytes);
void mhip_host_unregister(void* p);

/* `stream` may be NULL (the context creates its own) or a hipStream_t the caller owns (e.g. torch's current stream). */
int  mhip_ctx_create(int device, void* stream, mhip_ctx** out);
void mhip_ctx_destroy(mhip_ctx* ctx);
int  mhip_ctx_sync(mhip_ctx* ctx);
/* Optional, additive: allocates the scratch the first mhip_index_build of a volume of up to `bases` bases would allocate
 * (two arrays of 8 bytes per base; tens of GB take hundreds of milliseconds to map).  May run on another thread while the
 * caller still parses its input; every other call on the context that needs scratch waits for it. */
int  mhip_ctx_reserve_index(mhip_ctx* ctx, int64_t bases);
/* for callers that keep tables in HBM between calls (the mecat2pw driver keeps the candidate lists of a whole grid cell there):
   a named device buffer of the context, grown on demand (contents are lost when it grows), freed with the context; and a copy
   from device memory to the host on the context's stream that returns when the bytes have arrived */
int  mhip_ctx_buffer(mhip_ctx* ctx, const char* name, size_t bytes, void** d_ptr);
int  mhip_download(mhip_ctx* ctx, void* host_dst, const void* d_src, size_t bytes);
/* free / total device memory as the driver sees it now (callers size the tables they keep resident from it) */
int  mhip_ctx_mem_info(mhip_ctx* ctx, size_t* free_bytes, size_t* total_bytes);
void mhip_params_default(mhip_params* p, int tech);           /* pw_options.cpp:30-50 + pw_impl.cpp:843-851 */

/* per-kernel timing with HIP events on the context's stream (for bench.py's roofline block).
   set_profiling(1) makes every launch record start/stop events; kernel_stats returns, for the kernel whose name
   matches `name`, the number of launches and the summed duration in ms since the last reset. */
int  mhip_ctx_set_profiling(mhip_ctx* ctx, int on);
int  mhip_ctx_kernel_stats(mhip_ctx* ctx, const char* name, int64_t* launches, double* total_ms);
int  mhip_ctx_kernel_names(mhip_ctx* ctx, char* buf, int buflen);   /* '\n'-separated */
int  mhip_ctx_reset_stats(mhip_ctx* ctx);
/* device work counters accumulated by seed/align calls since the last reset:
   [0] query k-mer lookups  [1] bucket hits H  [2] candidates emitted  [3] dw blocks  [4] d-path cells  [5] snake bases
   [6] aligned query bases  [7] alignments ok */
int  mhip_ctx_counters(mhip_ctx* ctx, int64_t out[8]);

/* volume: host arrays exactly as load_volume() leaves them (pac = (num_bases+3)/4 bytes) */
int  mhip_volume_upload(mhip_ctx* ctx, const uint8_t* pac, const mhip_offset_t* offs, int num_reads, int num_bases,
                        int start_read_id, mhip_volume** out);
/* the same volume made on the device from the residue LETTERS (replaces the packing step of split_raw_dataset,
   common/split_database.cpp:221-266 + PackedDB::set_char, packed_db.h:98-107, unmasked OR of the 4-bit IUPAC value included).
   text = the input file's bytes (host), seq_start[r] = index of read r's first residue, line_width[r] = residues per line of read r (0: one
   line; otherwise every line but the last holds exactly that many and ends in one byte), offs = the volume layout (offset, size; one pad
   base behind every read).  pac_out, when not NULL, receives the (num_bases + 3) / 4 packed bytes (what dump_volume writes). */
int  mhip_volume_pack(mhip_ctx* ctx, const uint8_t* text, int64_t text_bytes, const int64_t* seq_start, const int32_t* line_width,
                      const mhip_offset_t* offs, int num_reads, int num_bases, int start_read_id, mhip_volume** out, uint8_t* pac_out);
void mhip_volume_free(mhip_volume* v);
int  mhip_volume_num_reads(const mhip_volume* v);
int  mhip_volume_num_bases(const mhip_volume* v);

/* index of one volume (k = 13).  Buckets with more than 128 occurrences are empty; bucket contents ascend. */
int  mhip_index_build(mhip_ctx* ctx, const mhip_volume* v, mhip_index** out);
/* the same table with another bucket cap (1..256): mecat2canu's overlappers drop buckets of more than 256 occurrences
   (mecat2asmpw.c:307-314 sumvalue_x) where mecat2pw drops those above 128; everything else about the table is the same */
int  mhip_index_build_ex(mhip_ctx* ctx, const mhip_volume* v, int max_bucket, mhip_index** out);
void mhip_index_free(mhip_index* idx);                       /* waits for the device: work queued through *_dev calls may still read it */
int64_t mhip_index_num_kmers(const mhip_index* idx);
/* parity/debug: counts[4^13] (kept occurrences) and/or offsets[num_kmers]; either may be NULL */
int  mhip_index_download(mhip_ctx* ctx, const mhip_index* idx, int32_t* counts, int32_t* offsets);
/* test hook: the two side arrays the seeding stage reads — slots[num_kmers] ((position / 2000) mod 2^15) and the 16-byte bucket
 * records recs[4^13][4] (start, occurrences below seven position cuts at multiples of *cut_step, occurrences); returns 1 when the
 * table has no records (bucket cap above 255, or not built yet) */
int  mhip_index_download_aux(mhip_ctx* ctx, const mhip_index* idx, uint16_t* slots, uint32_t* recs, int* cut_step);

/* candidates of reads [rid_begin, rid_end) of `reads` against (`ref`, `idx`): for read r the list is written to
   out[(r - rid_begin) * maxc ...] in the reference's list order (score descending, stable), out_counts[r - rid_begin]
   entries.  `out`/`out_counts` are HOST pointers here and DEVICE pointers in the _dev variant (results stay in HBM,
   e.g. for the multi-GPU all-gather). */
int  mhip_seed_reads(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads,
                     int rid_begin, int rid_end, const mhip_params* p, mhip_candidate* out, int32_t* out_counts);
int  mhip_seed_reads_dev(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads,
                         int rid_begin, int rid_end, const mhip_params* p, void* d_out, void* d_out_counts);

/* same for the reads rid_begin + i * rid_stride, i in [0, n): the static multi-GPU shard of a (ref volume, query volume)
   grid cell — rank r of P seeds reads r, r + P, r + 2P, ... (work per read grows with the read id inside a diagonal cell
   because candidates with sid > qid are dropped, pw_impl.cpp:370, so the shard is cyclic, not contiguous) */
int  mhip_seed_reads_strided_dev(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads,
                                 int rid_begin, int rid_stride, int n, const mhip_params* p, void* d_out, void* d_out_counts);

/* candidate lists -> alignment jobs, on the device (the loop head of pairwise_mapping, pw_impl.cpp:674-686: the
   +kmer_size/2 shift of both start points only when both are non-zero).  d_cands[n_reads][maxc], d_counts[n_reads];
   read i of the table is query read rid_begin + i * rid_stride.  Only jobs with (job index % part_count) == part_index
   are emitted (multi-GPU split of the extension stage; 0/1 = all).  d_jobs must hold n_reads * maxc entries;
   *num_jobs receives the number written. */
int  mhip_jobs_from_candidates_dev(mhip_ctx* ctx, const void* d_cands, const void* d_counts, int n_reads, int maxc,
                                   int rid_begin, int rid_stride, int ref_start_read_id, int part_index, int part_count,
                                   void* d_jobs, int* num_jobs);

/* the occupied entries of a candidate table, dense and read-major: d_pack[first(i) + k] = d_cands[i][k] with first = exclusive prefix
   sum of d_counts; *total = number of entries.  d_pack must hold them (n_reads * maxc entries always suffice). */
int  mhip_pack_candidates_dev(mhip_ctx* ctx, const void* d_cands, const void* d_counts, int n_reads, int maxc, void* d_pack, int64_t* total);

/* dw extension of n jobs (PacBio / DiffAligner semantics); jobs/out are HOST pointers (_dev: DEVICE pointers) */
int  mhip_align_candidates(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs,
                           int n, int min_align_size, mhip_aln_result* out);
int  mhip_align_candidates_dev(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs,
                               int n, int min_align_size, void* d_out);

/* the same contract with XdropAligner semantics (nanopore mode, common/xdrop_gapalign.cpp:359-439: reward 1, penalty -1,
   gap_open 0, gap_extend 1, X = 30; ok = query_end - query_start >= min_align_size; `columns`/`matches` count the
   reference's aligned strings, whose left half is emitted without its last column) */
int  mhip_xalign_candidates(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs,
                            int n, int min_align_size, mhip_aln_result* out);
int  mhip_xalign_candidates_dev(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs,
                                int n, int min_align_size, void* d_out);

/* ---- multi-GPU: one process per GPU, the all-vs-all grid sharded statically, candidate lists exchanged over
 * RCCL.  The reference parallelises a grid cell (reference volume i x query volume j) over pthreads pulling chunks of 500 query
 * reads from a shared cursor (mecat2pw/pw_impl.cpp:612-621, 866-876); here chunk c of query volume j belongs to rank
 * (c + cell_shift) mod nranks (the driver passes cell_shift = j).  Every rank builds the index of volume i itself
 * (mhip_index_build: recompute beats moving it), seeds the reads of its own chunks, and ONE count-then-payload all-gather per
 * slab of reads leaves the complete candidate table on every rank: one int32 per read first, then only the occupied 48-byte
 * records, sent directly to every peer (xGMI is point to point).  Nothing here has a counterpart in the reference.
 *
 * mhip_comm_unique_id / mhip_comm_init wrap ncclGetUniqueId / ncclCommInitRank (RCCL is opened lazily, single-GPU runs never
 * load it): rank 0 creates the id and hands it to the other ranks by any means (the driver uses a file in wrk_dir, bench.py a
 * torch.distributed broadcast).  Collectives run on the context's stream.  mhip_comm_init_hostfile is a TEST HOOK for boxes
 * with fewer GPUs than ranks (RCCL refuses two ranks on one device): same calls, transport through files in `dir`. */
typedef struct mhip_comm mhip_comm;
#define MHIP_COMM_ID_BYTES 128
#define MHIP_SHARD_CHUNK 500       /* CHUNK_SIZE, mecat2pw/pw_impl.h:15 */
int  mhip_comm_unique_id(uint8_t id[MHIP_COMM_ID_BYTES]);
int  mhip_comm_init(mhip_ctx* ctx, int nranks, int rank, const uint8_t id[MHIP_COMM_ID_BYTES], mhip_comm** out);
int  mhip_comm_init_hostfile(mhip_ctx* ctx, int nranks, int rank, const char* dir, const char* run_id, mhip_comm** out);
void mhip_comm_destroy(mhip_comm* comm);
/* transport: 0 = RCCL, 1 = host files (test hook); rccl_ranks = ncclCommCount of the communicator (0 with host files).  With the
 * context profiling (mhip_ctx_set_profiling) every exchange is timed on the stream under the kernel-stat names "xg_exchange"
 * (candidate / result all-gathers) and "xg_exchange_index" (mhip_index_build_sharded). */
int  mhip_comm_info(const mhip_comm* comm, int* transport, int* rccl_ranks);      /* transport 0 RCCL, 1 host files, 2 none (solo) */
/* BENCH HOOK (bench.py --simulate-ranks): rank `rank` of `nranks` with no transport.  The sharded calls below then do exactly this rank's
 * share of the work on the device — its key range of mhip_index_build_sharded, its chunks in mhip_seed_reads_sharded / mhip_align_sharded,
 * the pack and scatter kernels of the exchanges — and its peers' contributions read as zero counts: what one rank of P computes, timed
 * alone.  mhip_comm_bytes_sent = the bytes this rank's contributions put on the links (its bytes x (P - 1) peers), in any transport. */
int  mhip_comm_init_solo(mhip_ctx* ctx, int nranks, int rank, mhip_comm** out);
int64_t mhip_comm_bytes_sent(const mhip_comm* comm);
int64_t mhip_comm_local_jobs(const mhip_comm* comm);         /* candidates of this rank's own reads in the last mhip_seed_reads_sharded slab */
int  mhip_comm_rank(const mhip_comm* comm);
int  mhip_comm_nranks(const mhip_comm* comm);
int  mhip_comm_barrier(mhip_comm* comm);
int  mhip_comm_selftest(mhip_ctx* ctx);                       /* RCCL loads and moves bytes on this device (one-rank communicator) */
int64_t mhip_comm_bytes_received(const mhip_comm* comm);      /* payload + counts received from peers so far */

/* shard arithmetic for the reads [rid_begin, rid_end) of a query volume (rid_begin must be a multiple of chunk): how many of
 * them `rank` owns and the first one; local index i is read first + (i / chunk) * chunk * nranks + i % chunk */
int  mhip_shard_local_count(int rid_begin, int rid_end, int chunk, int cell_shift, int rank, int nranks);
int  mhip_shard_first_read(int rid_begin, int rid_end, int chunk, int cell_shift, int rank, int nranks);
/* rows mode (#volumes >= ranks: a grid row per GPU, no data moves): the static deal of the rows still to do.  Row i of the upper
 * triangular volume grid holds the cells (i, i) .. (i, num_vols - 1) (mecat2pw/pw.cpp:65-81, pw_impl.cpp:859-879), so a row costs
 * num_vols - i cell visits (+ one index build, priced at a quarter of a cell); rows are taken heaviest first and each goes to the rank
 * with the least work so far (ties: the lowest rank) — 19 rows over 8 ranks: heaviest rank 26 cells against a mean of 23.75, where the
 * cyclic deal i mod P gave 33.  Every rank derives the same deal from the split marker alone.  owner[i] = the rank of row i, -1 for a
 * row that is not in todo[].  Returns the heaviest rank's cell count, -1 on bad arguments. */
int  mhip_shard_deal_rows(int num_vols, const int* todo, int ntodo, int nranks, int* owner /*[num_vols]*/);
/* mhip_seed_reads_dev for the local reads of such a shard (first = mhip_shard_first_read, n = mhip_shard_local_count) */
int  mhip_seed_reads_chunked_dev(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int first,
                                 int chunk, int nranks, int n, const mhip_params* p, void* d_out, void* d_out_counts);

/* The exchange step: d_cands[n_local][maxc] / d_counts[n_local] are this rank's lists (DEVICE pointers); on return
 * d_all_cands[rid_end - rid_begin][maxc] / d_all_counts[rid_end - rid_begin] hold the complete table of the slab, read-major,
 * on every rank — the layout mhip_seed_reads produces. */
int  mhip_allgather_candidates(mhip_comm* comm, const void* d_cands, const void* d_counts, int rid_begin, int rid_end, int chunk,
                               int cell_shift, int maxc, void* d_all_cands, void* d_all_counts);

/* mhip_seed_reads over all ranks: seeds this rank's chunks of [rid_begin, rid_end) and exchanges.  out / out_counts are HOST
 * pointers in mhip_seed_reads' layout and may be NULL (the table stays on the device, see mhip_sharded_tables). */
int  mhip_seed_reads_sharded(mhip_comm* comm, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int rid_begin,
                             int rid_end, int chunk, int cell_shift, const mhip_params* p, mhip_candidate* out, int32_t* out_counts);
/* mhip_align_candidates / mhip_xalign_candidates (tech 0 / 1) over all ranks, for the slab of the preceding
 * mhip_seed_reads_sharded call: every rank extends the candidates of its own reads, the 32-byte results are exchanged like the
 * candidates.  out (HOST, may be NULL) receives one result per candidate, read-major: read rid_begin's candidates in list order,
 * then read rid_begin + 1's ... — the order of the job list the driver builds from the table.  *num_jobs = their number. */
int  mhip_align_sharded(mhip_comm* comm, const mhip_volume* ref, const mhip_volume* reads, int tech, int min_align_size,
                        mhip_aln_result* out, int64_t* num_jobs);
/* device views of the tables the last two calls left on this rank (complete on every rank) */
int  mhip_sharded_tables(mhip_comm* comm, void** d_cands, void** d_counts, void** d_results, int64_t* num_jobs);
/* ---- mecat2canu's overlapper for corrected reads (mecat2asmpw / mecat2trimpw), candidate stage ----
 * Replaces the part of pairwise_mapping in front of its extension loop (mecat2canu/src/mecat2asmpw/mecat2asmpw.c:580-718): seeding of
 * both strands of query reads [rid_begin, rid_end) of `reads` against one block of reads (`block`: the tool's STRMEM as a volume — one
 * pad base after every read is the tool's one NUL after every read, so offsets agree; start_read_id = the block's first read number)
 * whose table was built with mhip_index_build_ex(ctx, block, 256, ..) (creat_ref_index + sumvalue_x: cap 256), and candidate
 * selection.  out[(rid - rid_begin) * 100 .. ] = the read's candidates in list order (canidate_save, :55-58; positions 1-based as in
 * the tool, chain 0 = 'F', 1 = 'R'), out_counts[] their number (<= 100).  Every read starts from an all-zero segment array (the
 * tool's worker threads keep stale seeds of the reads they mapped before, which can move a score by a few votes: INTEGRATION.md). */
typedef struct { int32_t loc1, loc2, left1, left2, right1, right2, score, num1, num2, readno, readstart, chain; } mhip_asm_candidate;
/* Bases other than A, C, G, T (an N of a corrected read): the tools restart their k-mer at such a base in table and query
 * (mecat2asmpw.c:445, 486, 316-335) and compare it as a character in the extension (N equals N only).  A volume may carry a second
 * plane in its own 2-bit layout — 3 at every such base (stored as code 0 in the volume), 0 elsewhere; nplane = (num_bases + 3) / 4 bytes,
 * NULL removes it.  mhip_asm_seed_reads skips the query k-mers that touch one, mhip_asm_extend compares with both planes; the table of a
 * block with such bases is built from a volume whose read table ends a "read" at every one of them (the index never starts a k-mer inside
 * the 13 positions in front of a read end: exactly the positions whose 13-mer would hold the base). */
int  mhip_volume_set_nplane(mhip_ctx* ctx, mhip_volume* vol, const uint8_t* nplane);
int  mhip_asm_seed_reads(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* block, const mhip_volume* reads, int rid_begin,
                         int rid_end, mhip_asm_candidate* out, int32_t* out_counts);
int  mhip_asm_seed_reads_ex(mhip_ctx* ctx, const mhip_index* idx, const mhip_volume* block, const mhip_volume* reads, int rid_begin,
                            int rid_end, int gate /*10: mecat2asmpw, 8: mecat2trimpw*/, int maxc /*100; 50: the *50 variants*/,
                            mhip_asm_candidate* out /*[n][100]*/, int32_t* out_counts);
/* The tool's extension loop (mecat2asmpw.c:723-841: blocks of 500 bases while more than 600 remain, O(ND) `align` at error rate 0.10,
 * a block's tail cut in front of its last run of four matches) for a batch of (candidate, both directions) jobs.  A job names the two
 * reads (xid in `block`, yid in `reads`), the strand of the mapped read, and per direction the start positions (local to the read /
 * to the mapped strand) and the bases available: from a candidate c of query read r, with x0 = c.loc1 - 1 - c.readstart,
 *   left   lx = x0 + 12, ly = c.loc2 + 12, lnx = c.left1, lny = c.left2      (backwards from the LAST base of the seed 13-mer)
 *   right  rx = x0,      ry = c.loc2,      rnx = c.right1, rny = c.right2    (forwards from its first base)
 * dirs[2 i + d] = {columns, x bases, y bases, y-only columns, x-only columns, 0} of direction d (0 left, 1 right);
 * ops[(2 i + d) * dir_cols_cap / 16 ...] the columns, 2 bits each, in extension order: 0 = both bases (always equal: O(ND) paths have
 * no mismatch columns), 1 = y base only, 2 = x base only. */
typedef struct { int32_t xid, yid, chain, lx, ly, lnx, lny, rx, ry, rnx, rny, pad; } mhip_asm_job;
int  mhip_asm_extend(mhip_ctx* ctx, const mhip_volume* block, const mhip_volume* reads, const mhip_asm_job* jobs, int n, int dir_cols_cap,
                     int32_t* dirs /*[2 n][6]*/, uint32_t* ops /*[2 n][dir_cols_cap / 16]*/);
/* The same extension with the columns handed over densely, in two calls (what the tools use: a direction fills a fraction of
 * dir_cols_cap, so the fixed-stride form moves 5-10x the bytes over the PCIe link).  mhip_asm_extend_run extends the batch, leaves the
 * results on the device and returns the number of 32-bit words the columns of all directions take (ceil(columns / 16) per direction);
 * mhip_asm_extend_fetch — same context, same n, before the next _run — copies them out: direction k = 2 i + d occupies words
 * [word_offs[k], word_offs[k + 1]) of ops_dense (total_words words).  Host buffers from mhip_host_alloc make the copies DMA. */
int  mhip_asm_extend_run(mhip_ctx* ctx, const mhip_volume* block, const mhip_volume* reads, const mhip_asm_job* jobs, int n, int dir_cols_cap,
                         int64_t* total_words);
int  mhip_asm_extend_fetch(mhip_ctx* ctx, int n, int32_t* dirs /*[2 n][6]*/, uint64_t* word_offs /*[2 n + 1]*/, uint32_t* ops_dense /*[total_words]*/);

/* mhip_index_build by all ranks of the communicator together (replaces create_ref_index, common/lookup_table.cpp:63-160, in a
 * multi-GPU cell): the 4^13 key space is cut into P contiguous ranges of equal occupancy, each rank builds the buckets of its range,
 * positions and table slices are all-gathered (counts first, then payload), and every rank returns the complete table — equal, array
 * for array, to the one mhip_index_build makes.  Collective: every rank of `comm` calls it with the same volume. */
int  mhip_index_build_sharded(mhip_comm* comm, const mhip_volume* v, mhip_index** out);
/* The faster of mhip_index_build on every rank and mhip_index_build_sharded, decided by measurement: the first call on a communicator runs
 * both (once to warm up, once timed barrier to barrier; the slowest rank's time counts) and keeps the faster one for this and every later
 * call; MECAT_HIP_INDEX_SHARD=0 / 1 decides without measuring.  ms[0] / ms[1] (may be NULL): the replicated / sharded build times the
 * decision rests on, 0 when nothing was measured; *sharded (may be NULL): what was used.  No counterpart in the reference. */
int  mhip_index_build_auto(mhip_comm* comm, const mhip_volume* v, mhip_index** out, double ms[2], int* sharded);

/* ---- mecat2cns re-aligner: ns_banded_sw::GetAlignment of src/mecat2cns/dw.cpp:482-553, which
 * mecat2cns calls for up to 200 candidates per template read (mecat_correction.cpp:286, 347, 431, 494) -------------------
 * Jobs are mhip_aln_job records: qid_local = the candidate read in `reads`, taken reverse-complemented when chain != 0;
 * sid_local = the template read in `ref`; qstart / sstart = the extension start (qext, sext).  error_rate is 0.15 (PacBio)
 * or 0.20 (nanopore), <= 0.20.  An O(ND) alignment has no mismatch columns, so a column is 0 = both bases (equal),
 * 1 = target base only (gap in the query string), 2 = query base only (gap in the target string): 2 bits per column,
 * 16 columns per uint32 from the least significant bits, dir_cols_cap columns (a multiple of 16) per direction:
 *   ops[(2 * i    ) * dir_cols_cap / 16 ...]  left  direction of job i, in extension order (outwards from the start point)
 *   ops[(2 * i + 1) * dir_cols_cap / 16 ...]  right direction
 * The reference's merged strings are reverse(left) + right; GetAlignment keeps merged columns [first_col, last_col). */
typedef struct mhip_cns_result {
    int32_t ok;                                  /* GetAlignment's return value */
    int32_t qoff, qend, soff, send;              /* m5qoff / m5qend / m5soff / m5send */
    int32_t left_cols, right_cols;               /* columns stored per direction */
    int32_t first_col, last_col;                 /* the aligned strings are merged columns [first_col, last_col) */
    int32_t mat, ins, del;                       /* OutputStore counts over the untrimmed string (dw.cpp:441-470) */
    int32_t query_start, query_end, target_start, target_end;    /* OutputStore coordinates before GetAlignment's trimming */
} mhip_cns_result;

int mhip_cns_align_candidates(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs, int n,
                              double error_rate, int min_align_size, int dir_cols_cap, mhip_cns_result* results, uint32_t* ops);
int mhip_cns_align_candidates_dev(mhip_ctx* ctx, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n,
                                  double error_rate, int min_align_size, int dir_cols_cap, void* d_results, void* d_ops);

/* ---- mecat2cns' candidate accept loop (BASELINE config 4): the front half of
 * consensus_one_read_can_pacbio / _nanopore (mecat2cns/mecat_correction.cpp:388-450, 452-515) for a batch of template reads.
 * cands: ExtensionCandidate records (common/alignment.h:8-19, the 13 ints of a .can line as mecat2cns normalises them: sdir == 0,
 * sid = the template), grouped by template: template t owns cands[tmpl_begin[t] .. tmpl_begin[t + 1]); they are sorted IN PLACE by
 * (score desc, qid, qext) as the reference does.  All reads live in `vol` (host_pac, the packed bytes given to mhip_volume_upload, is no
 * longer read and may be NULL: since round 6 the aligned strings are built on the device and copied into the result buffer).
 * tech 0: error rate 0.15, at most 60 accepted; tech 1: 0.20 / 100.  min_mapping_ratio is the option value (0.9 / 0.4), the
 * function subtracts the reference's 0.02.  Output (malloc'ed, release with mhip_cns_free): one record per accepted alignment
 * in the order the reference would add them, template by template, and the gap-normalised aligned strings
 * (normalize_gaps(.., push = true), reads_correction_aux.cpp:3-81) that meap_add_one_aln / CnsAlns::add_aln consume:
 * strings + str_offset = qaln (aln_size chars + NUL), then saln (aln_size chars + NUL).
 * Ties: candidates equal in (score, qid, qext) keep the order the reference's own std::sort gives them when cands[] arrives in the
 * reference's order (the partition file's); any other input order may accept tied records in another order.  Every record's
 * qsize / ssize must equal the volume's read lengths (refused otherwise). */
typedef struct { int32_t qdir, qid, qext, qsize, qoff, qend, sdir, sid, sext, ssize, soff, send, score; } mhip_ext_candidate;
typedef struct {
    int32_t template_index;      /* index into tmpl_begin */
    int32_t qid, sid;
    int32_t qoff, qend, soff, send;          /* m5qoff / m5qend / m5soff / m5send */
    int32_t aln_size;
    int64_t cand_index;          /* position of the candidate in cands[] (after the in-place sort) */
    int64_t str_offset;
} mhip_cns_accepted;
int  mhip_cns_accept_templates(mhip_ctx* ctx, const mhip_volume* vol, const uint8_t* host_pac, mhip_ext_candidate* cands,
                               const int64_t* tmpl_begin, int num_templates, int tech, int min_align_size, double min_mapping_ratio,
                               int num_threads, mhip_cns_accepted** out_accepted, int64_t* out_count, char** out_strings,
                               int64_t* out_strings_bytes, int64_t* out_jobs /* alignments computed, may be NULL */);
void mhip_cns_free(void* p);
/* test hook: normalize_gaps' gap pushing (reads_correction_aux.cpp:36-67) by the device kernel of the accept stage, in place, on n_pairs
 * pairs of NUL-terminated host strings: pair p = buf + off[p] (len[p] characters + NUL) and its partner right behind it */
int  mhip_debug_push_gaps(mhip_ctx* ctx, char* buf, int64_t bytes, const int64_t* off, const int32_t* len, int n_pairs);
/* mhip_cns_free does not return a string buffer to the system at once: the library keeps the LARGEST released one (gigabytes — about
 * 14 GB for a config-2-sized batch) and hands it out again to the next batch that fits, because first-touching fresh pages costs
 * more than the batch's GPU time.  The parked buffer belongs to the process, not to a context (mhip_ctx_destroy leaves it).  This call
 * frees it; a process that is done with mecat2cns batches but lives on should call it. */
void mhip_cns_release_parked(void);

#ifdef __cplusplus
}
#endif
#endif
