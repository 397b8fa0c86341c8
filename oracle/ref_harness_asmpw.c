/* ref_harness_asmpw.c — TEST INFRASTRUCTURE: the UNMODIFIED mecat2asmpw.c of mecat2canu, compiled where it
 * lies (oracle/Makefile target `ref`; nothing of it is copied here), with its main() renamed, so that its file-scope functions can be
 * called one at a time.  First use: creat_ref_index (mecat2asmpw.c:422-512) — the look-up table of one block of reads — to pin
 * mhip_index_build_ex(.., 256, ..).  Never linked by the product path. */
#define main mecat2asmpw_reference_main
#include "mecat2asmpw.c"
#undef main

/* seq: the block's reads as the tool holds them (load_read, :388-409): upper-case text, one NUL after every read; n = bytes.
 * Returns the number of kept positions; counts[] (4^k ints: occurrences per k-mer id, buckets of more than 256 emptied by sumvalue_x,
 * :307-314) and index[] (per k-mer id a pointer to its 1-based positions, or NULL) stay owned by the reference's globals until
 * refasm_index_free(). */
int refasm_index(char* seq, int n, int k, int** counts, int*** index) {
    seed_len = k;
    creat_ref_index(seq, n);
    *counts = countin;
    *index = databaseindex;
    return sumcount;
}

void refasm_index_free(void) {
    free(countin);
    free(allloc);
    free(databaseindex);
    countin = NULL;
    allloc = NULL;
    databaseindex = NULL;
}
