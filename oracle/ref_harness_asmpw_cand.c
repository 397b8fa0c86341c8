/* ref_harness_asmpw_cand.c — TEST INFRASTRUCTURE: the candidate stage of the UNMODIFIED mecat2asmpw.c.
 *
 * pairwise_mapping (mecat2asmpw.c:514-976) is one function: seeding, candidate selection, extension and output follow each other
 * inline, and its candidate list is a local array.  To observe the list without touching the source, the reference file is compiled
 * as it lies into its own object (oracle/Makefile: main renamed on the command line, nothing else), the symbol `align` of that object
 * is made weak (objcopy), and this file supplies the `align` the object then calls: it records its arguments and reports "no
 * alignment".  With that answer the reference makes exactly two calls per candidate, in list order — the first block of the left
 * extension and the first block of the right extension (:741-749, :800-809) — whose arguments are slices of the two sequences that
 * start at the candidate's (loc1, loc2) and are min(num, 500) long, and it prints nothing (:903).  So the recorded calls ARE the
 * candidate list: subject position, query position, strand (the query slice is cut from the forward or from the reverse-complemented
 * read), num1 and num2.  tests/test_asmpw_ref_cpu.py compares them with the restatement (oracle/asmpw_oracle.c).
 * Never linked by the product path. */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

/* the reference's file-scope state (mecat2asmpw.c:71-81), external linkage there */
typedef struct { int readno, readlen; char* seqloc; } ReadFasta;
typedef struct { int readno, length; char* mem; } readmemory;
extern pthread_mutex_t mutilock;
extern int runnumber, runthreadnum, readcount, terminalnum;
extern int *countin, **databaseindex, *allloc, sumcount;
extern int seed_len, *llocation, seqcount, curreadcount;
extern char* STRMEM;
extern readmemory* indexread;
extern ReadFasta* readinfo;
extern FILE** outfile;
void creat_ref_index(char* seq, int seqcount);
void pairwise_mapping(int threadint);

static char* g_rec = NULL;
static long g_cap = 0, g_used = 0;
static int g_calls = 0, g_overflow = 0;

/* the `align` the reference object calls: record (query length, target length, band, both strings), report failure */
int align(char* query_seq, char* target_seq, int band_tolerance, int get_aln_str, void* align_rtn, int* V, int* U, void* d_path, void* aln_path) {
    const int ql = (int)strlen(query_seq), tl = (int)strlen(target_seq);
    const long need = 12 + (long)ql + tl;
    (void)get_aln_str; (void)align_rtn; (void)V; (void)U; (void)d_path; (void)aln_path;
    if (g_used + need > g_cap) { g_overflow = 1; return 0; }
    memcpy(g_rec + g_used, &ql, 4);
    memcpy(g_rec + g_used + 4, &tl, 4);
    memcpy(g_rec + g_used + 8, &band_tolerance, 4);
    memcpy(g_rec + g_used + 12, query_seq, (size_t)ql);
    memcpy(g_rec + g_used + 12 + ql, target_seq, (size_t)tl);
    g_used += need;
    ++g_calls;
    return 0;
}

static ReadFasta g_query;
static FILE* g_null = NULL;

/* The block the tool indexes, as load_read leaves it (:388-409): text = the reads in upper case, one NUL after each; starts[i] = offset
 * of read i; n = bytes.  llocation has one more entry than reads, which load_read never writes: the tool reads it for the last read
 * of the block (:640) and finds what malloc returned, zero on a fresh heap — zero here. */
int refasmc_setup(char* text, int n, const int* starts, const int* lens, int nreads, int first_readno) {
    int i;
    seed_len = 13;
    STRMEM = text;
    seqcount = n;
    curreadcount = nreads;
    free(llocation);
    free(indexread);
    llocation = (int*)calloc((size_t)nreads + 1, sizeof(int));
    indexread = (readmemory*)calloc((size_t)nreads + 1, sizeof(readmemory));
    for (i = 0; i < nreads; ++i) {
        llocation[i] = starts[i];
        indexread[i].mem = text + starts[i];
        indexread[i].readno = first_readno + i;
        indexread[i].length = lens[i];
    }
    free(countin); free(allloc); free(databaseindex);
    creat_ref_index(STRMEM, seqcount);
    if (!g_null) g_null = fopen("/dev/null", "w");
    if (!outfile) outfile = (FILE**)malloc(sizeof(FILE*));
    outfile[0] = g_null;
    if (!readinfo) readinfo = &g_query;
    pthread_mutex_init(&mutilock, NULL);
    return sumcount;
}

/* candidates of one query read (upper-case text, NUL-terminated): the recorded align calls, two per candidate.  Returns their number,
 * or -1 when `cap` bytes were not enough. */
int refasmc_candidates(char* query, int read_name, char* rec, long cap, long* used) {
    g_rec = rec; g_cap = cap; g_used = 0; g_calls = 0; g_overflow = 0;
    readinfo = &g_query;
    g_query.readno = read_name;
    g_query.readlen = (int)strlen(query);
    g_query.seqloc = query;
    readcount = 1;
    terminalnum = 1;
    runnumber = 0;
    runthreadnum = 0;
    pairwise_mapping(0);
    *used = g_used;
    return g_overflow ? -1 : g_calls;
}
