/*
 * mecat_oracle.h — CPU restatement of the reference's mecat2pw hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is the parity oracle (task rule 3): a plain-C restatement of the algorithm in
 * /root/reference/src/{mecat2pw,common} for the path named in BASELINE.json.  Only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product path
 * (mecat_amd/, include/) never links, imports or calls anything in oracle/.
 *
 * Parity status: PINNED.  Every function below is checked against the unmodified reference compiled into
 * oracle/_ref/ (tests/golden/make_golden.py generated the committed vectors in tests/golden/ from it; when
 * oracle/_ref is present tests/test_oracle_vs_ref.py additionally compares on random inputs).
 *
 * All file:line citations are relative to /root/reference/src/.
 */
#ifndef MECAT_ORACLE_H
#define MECAT_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* constants: mecat2pw/pw_impl.h:10-19, common/defs.h:202, common/split_database.h:6 */
#define ORC_BC 10
#define ORC_SM 40
#define ORC_SI 41
#define ORC_ZV 2000
#define ORC_KMER 13
#define ORC_MAX_SEQ_SIZE 500000
#define ORC_MCS 2140000000L
#define ORC_GAP_CODE 4

typedef struct { int offset, size; } orc_offset_t;          /* split_database.h:9-11 */

typedef struct {                                             /* volume_t, split_database.h:18-24 */
    int num_reads;
    int num_bases;          /* `curr`: bases incl. one pad base after every read */
    int start_read_id;
    orc_offset_t* offs;
    uint8_t* pac;           /* (num_bases+3)/4 bytes, first base in the two MSBs (packed_db.h:98-107) */
} orc_volume;

typedef struct {                                             /* ref_index, lookup_table.h:6-11 */
    int* counts;            /* [4^13] occurrences kept (0 when > 128) */
    int64_t* starts;        /* [4^13] offset of the bucket in `offsets`, -1 when empty (NULL pointer in the reference) */
    int* offsets;           /* [num_kmers] volume coordinates of the k-mer starts, ascending inside a bucket */
    int64_t num_kmers;
} orc_index;

typedef struct {                                             /* candidate_save, pw_impl.h:21-25 (48 bytes) */
    int loc1, loc2, left1, left2, right1, right2, score, num1, num2, readno, readstart;
    char chain;
} orc_candidate;

typedef struct {                                             /* Back_List, pw_impl.h:29-33 (168 bytes) */
    int16_t score, loczhi[ORC_SM], seedno[ORC_SM], seednum;
    int index;
} orc_back_list;

typedef struct {                                             /* file-statics pw_impl.cpp:18-26, set at :838-851 */
    int maxc;               /* -n, default 100 */
    int min_align_size;     /* -a, 2000 / 500 */
    int min_kmer_match;     /* -k, 4 / 2 */
    int min_kmer_dist;      /* 1800 pacbio / 400 nanopore */
    double ddfs_cutoff;     /* 0.25 */
    int tech;               /* 0 pacbio, 1 nanopore */
    int output_gapped_start_point; /* -g */
} orc_params;

typedef struct {                                             /* ExtensionCandidate, alignment.h:8-13 */
    int qdir, qid, qext, qsize, qoff, qend;
    int sdir, sid, sext, ssize, soff, send;
    int score;
} orc_ext_candidate;

typedef struct {                                             /* M4Record, alignment.h:21-37 (104 bytes) */
    int64_t qid, sid;
    double ident;
    int vscore, qdir;
    int64_t qoff, qend, qsize;
    int sdir;
    int64_t soff, send, ssize, qext, sext;
} orc_m4;

typedef struct {                                             /* what mecat2pw consumes of DiffAligner::go */
    int ok, query_start, query_end, target_start, target_end, matches, columns;
} orc_aln_result;

void orc_params_default(orc_params* p, int tech);            /* pw_options.cpp:30-50 + pw_impl.cpp:843-851 */

/* ---- A1/A3: 2-bit volumes ---- */
uint8_t orc_encode_base(int c);                              /* defs.cpp:3-36 (16 = invalid) */
orc_volume* orc_volume_pack(const uint8_t* codes, const int* lens, int nreads, int start_read_id);
orc_volume* orc_volume_load(const char* path);               /* split_database.cpp:155-181 */
int orc_volume_dump(const orc_volume* v, const char* path);  /* split_database.cpp:135-153 */
void orc_volume_free(orc_volume* v);
void orc_extract_one_seq(const orc_volume* v, int id, char* s);          /* split_database.cpp:121-133 */
void orc_reverse_complement(char* dst, const char* src, int size);       /* pw_impl.cpp:69-81 */
int orc_read_id_from_offset(const orc_volume* v, int offset);            /* split_database.cpp:15-35 */

/* ---- A2: index ---- */
orc_index* orc_index_build(const orc_volume* v);             /* lookup_table.cpp:63-160 */
void orc_index_free(orc_index* idx);

/* ---- A4-A8: seeding + candidates ---- */
typedef struct orc_seeding_bk orc_seeding_bk;                /* SeedingBK, pw_impl.h:55-64 */
orc_seeding_bk* orc_bk_new(int ref_size);
void orc_bk_free(orc_seeding_bk* bk);
int orc_extract_kmers(const char* s, int ssize, int* kmer_ids);                          /* pw_impl.cpp:83-97 */
void orc_insert_loc(orc_back_list* spr, int loc, int seedn, float len, double cutoff);   /* pw_impl.cpp:121-159 */
int orc_find_location(int* t_loc, int* t_seedn, int* t_score, int* loc, int k, int* rep_loc,
                      float len, int read_len1, double cutoff);                          /* pw_impl.cpp:161-239 */
int orc_seeding(const char* read, int read_size, const orc_index* ridx, orc_seeding_bk* bk); /* pw_impl.cpp:241-286 */
/* sweep statistics of the 41st-seed rule (see mecat_oracle.c) */
void orc_stats_reset(void);
void orc_stats_get(int64_t* out8);

int orc_get_candidates(const orc_volume* ref, orc_seeding_bk* bk, int num_segs, int read_id, int read_size,
                       char chain, orc_candidate* cands, int candidatenum, const orc_params* p); /* :288-465 */
/* both strands of one read (loop body of candidate_detect, pw_impl.cpp:742-765); chain stored as 0/1 (FWD/REV) when
   chain_as_char == 0, 'F'/'R' otherwise (pairwise_mapping :659-672) */
int orc_seed_read(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                  int rid, int chain_as_char, const orc_params* p, orc_candidate* out);

int64_t orc_bench_candidates(const uint8_t* codes, const int* lens, int nreads, int tech);   /* bench.py cpu_baseline "port" leg */

/* test/debug: run orc_seeding for one strand, copy the state get_candidates would see, then reset the touched segments.
   seg_ids/idx_score in index_list (first-touch) order; scores[i], loczhi[i*40..], seedno[i*40..] for segment seg_ids[i]. */
int orc_seeding_state(const char* read, int read_size, const orc_index* ridx, orc_seeding_bk* bk, int cap,
                      int* seg_ids, int16_t* idx_score, int16_t* scores, int16_t* loczhi, int16_t* seedno);

/* ---- A9: .can records ---- */
void orc_can_record(const orc_candidate* c, int qid, int qsize, int ssize, orc_ext_candidate* ec); /* :767-792 */
int orc_can_line(const orc_ext_candidate* ec, char* buf);    /* alignment.cpp:18-32 ; returns length */

/* ---- A10-A12: dw aligner ---- */
typedef struct orc_aligner orc_aligner;                      /* DiffAligner, diff_gapalign.h:137-199 */
orc_aligner* orc_aligner_new(void);
void orc_aligner_free(orc_aligner* a);
/* Align (diff_gapalign.cpp:107-219) on forward-ordered code arrays; for right_extend==0 the arrays are read from the
   last element backwards like dw_in_one_direction does.  res = {aln_str_size, dist, aln_q_s, aln_q_e, aln_t_s, aln_t_e} */
int orc_align(orc_aligner* a, const char* q, int qlen, const char* t, int tlen, int band_tol, int get_aln,
              int right_extend, int* res, char* q_aln, char* t_aln);
int orc_dw_go(orc_aligner* a, const char* query, int qstart, int qsize, const char* target, int tstart, int tsize,
              int min_aln_size, orc_aln_result* out);        /* diff_gapalign.cpp:294-349 */
/* work counters of the last orc_dw_go call: blocks, d-path cells, snake bases */
void orc_dw_counters(const orc_aligner* a, int64_t* blocks, int64_t* cells, int64_t* snake);

/* ---- A13: X-drop aligner (nanopore mode, -x 1 -j 1) ---- */
typedef struct orc_xaligner orc_xaligner;                    /* XdropAligner, xdrop_gapalign.h:117-212 */
orc_xaligner* orc_xaligner_new(void);
void orc_xaligner_free(orc_xaligner* a);
/* xdrop_align (xdrop_gapalign.cpp:10-213) on forward-ordered code arrays (left extension reads them backwards from the
   last element like align_ex does); returns the best score; res = {ae, be, num_ops}; ops receives num_ops pairs
   (op_type, count) in edit_block order (i.e. from the alignment end towards its start) */
int orc_xdrop_align(orc_xaligner* a, const char* A, int M, const char* B, int N, int forward, int* res, int* ops);
int orc_xdrop_go(orc_xaligner* a, const char* query, int qstart, int qsize, const char* target, int tstart, int tsize,
                 int min_aln_size, orc_aln_result* out);     /* xdrop_gapalign.cpp:359-439 */

/* ---- A14: m4 records ---- */
void orc_m4_fill(const orc_aln_result* r, int qid, int sid, char qchain, int qsize, int ssize,
                 int qstart, int sstart, int vscore, orc_m4* m);                          /* pw_impl.cpp:467-506 */
int orc_m4_postfilter(orc_m4* list, int n, orc_m4* out);     /* append_m4v :576-610 ; returns kept count */
int orc_m4_line(const orc_m4* m, int gapped, char* buf);     /* output_m4record :509-531 */
/* whole `-j 1` body for one read (pairwise_mapping :651-700): candidates -> dw -> m4 -> post-filter */
int orc_map_read(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                 orc_aligner* al, int rid, const orc_params* p, orc_m4* out);
/* same with the X-drop aligner when p->tech == 1 (xal may be NULL for tech 0) */
int orc_map_read_x(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                   orc_aligner* al, orc_xaligner* xal, int rid, const orc_params* p, orc_m4* out);

/* ---- N1: the mecat2cns re-aligner, src/mecat2cns/dw.cpp.  Sequences are code arrays (0..3). ----
   Columns are reported as ops: 0 = both bases (always equal: O(ND) paths have no mismatch columns), 1 = gap in the
   query (target base only), 2 = gap in the target (query base only). */
typedef struct orc_cns orc_cns;
orc_cns* orc_cns_new(void);
void orc_cns_free(orc_cns* a);
/* Align (dw.cpp:146-305) on a square block of `len` bases per side; q / t point at the first base of the block and are
   read forwards (right != 0) or backwards.  Returns 1 when an end of either sequence was reached; then *qe, *te are the
   consumed bases, *dist the edit distance and ops[0 .. *ncols) the columns. */
int orc_cns_align_block(orc_cns* a, const char* q, const char* t, int len, int right, double error_rate,
                        int* qe, int* te, int* dist, uint8_t* ops, int* ncols);
/* dw_in_one_direction (dw.cpp:307-376): ops of one direction in extension order, bases covered by them */
int orc_cns_one_direction(orc_cns* a, const char* q, int qsize, const char* t, int tsize, int right, double error_rate,
                          uint8_t* ops, int* qbases, int* tbases);
/* dw (dw.cpp:378-480): res = {query_start, query_end, target_start, target_end, columns, mat, mis, ins, del};
   out1 / out2 (may be NULL) receive the merged "ACGT-" strings, NUL terminated.  Returns the reference's flag. */
int orc_cns_dw(orc_cns* a, const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, double error_rate,
               int min_aln_size, int* res, char* out1, char* out2);
/* GetAlignment (dw.cpp:482-553): res = {qoff, qend, soff, send, aligned length}; qaln / saln may be NULL */
int orc_cns_get_alignment(orc_cns* a, const char* q, int qstart, int qsize, const char* t, int tstart, int tsize,
                          double error_rate, int min_aln_size, int* res, char* qaln, char* saln);

#ifdef __cplusplus
}
#endif
#endif
