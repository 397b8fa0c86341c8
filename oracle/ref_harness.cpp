/*
 * ref_harness — TEST INFRASTRUCTURE, this container only.
 *
 * Function-level access to the *unmodified* reference implementation so that the oracle restatement
 * (oracle/mecat_oracle.c) can be pinned against it and golden vectors can be generated
 * (tests/golden/make_golden.py).  The reference sources are compiled from where they lie under
 * $(REF)/src (default /root/reference/src); nothing is copied.  pw_impl.cpp is #included so the file-static
 * tuning parameters (MAXC, min_kmer_match, ...) that process_one_volume() sets (pw_impl.cpp:838-851) are reachable.
 *
 * Output: oracle/_ref/libref_harness.so (git-ignored).  Never linked or loaded by the product path.
 */
#include "mecat2pw/pw_impl.cpp"

#include <cstring>
#include <sstream>

/* defined (non-static) in common/xdrop_gapalign.cpp:10 */
int xdrop_align(const char* A, const int M, const char* B, const int N, int matrix[][4], int gap_open, int gap_extend,
                int x_dropoff, u8* state_array, BlastGapDP* score_array, u8** edit_script, int* edit_start_offset,
                GapPrelimEditBlock* edit_block, const bool forward, int& ae, int& be);
/* defined (non-static) in common/diff_gapalign.cpp:107 */
int Align(const char* query, const int q_len, const char* target, const int t_len,
          const int band_tolerance, const int get_aln_str, Alignment* align,
          int* V, int* U, DPathData2* d_path, PathPoint* aln_path, const int right_extend);

extern "C" {

void refh_set_params(int maxc, int min_aln, int min_kmer_match_, int tech)
{
    /* mirrors process_one_volume, pw_impl.cpp:838-851 */
    MAXC = maxc;
    min_align_size = min_aln;
    min_kmer_match = min_kmer_match_;
    if (tech == TECH_PACBIO) { ddfs_cutoff = ddfs_cutoff_pacbio; min_kmer_dist = 1800; }
    else { ddfs_cutoff = ddfs_cutoff_nanopore; min_kmer_dist = 400; }
}

int refh_split(const char* fasta, const char* wrk) { return split_raw_dataset(fasta, wrk); }

void* refh_load_volume(const char* path) { return load_volume(path); }
void  refh_free_volume(void* v) { delete_volume_t((volume_t*)v); }
int   refh_vol_num_reads(void* v) { return ((volume_t*)v)->num_reads; }
int   refh_vol_num_bases(void* v) { return ((volume_t*)v)->curr; }
int   refh_vol_start_id(void* v) { return ((volume_t*)v)->start_read_id; }
void  refh_vol_offsets(void* v, int* out) { volume_t* vv = (volume_t*)v; memcpy(out, vv->offset_list->offset_list, sizeof(offset_t) * vv->num_reads); }
void  refh_vol_pac(void* v, uint8_t* out) { volume_t* vv = (volume_t*)v; memcpy(out, vv->data, (vv->curr + 3) / 4); }

void* refh_build_index(void* v, int threads) { return create_ref_index((volume_t*)v, 13, threads); }
void  refh_free_index(void* i) { destroy_ref_index((ref_index*)i); }
/* counts[4^13]; returns number of kept k-mer positions; offsets_out may be NULL (size query) */
long  refh_index_dump(void* idx, int* counts_out, int* offsets_out)
{
    ref_index* r = (ref_index*)idx;
    long n = 0;
    const uint32_t nk = 1u << 26;
    for (uint32_t i = 0; i < nk; ++i) {
        int c = r->kmer_starts[i] ? r->kmer_counts[i] : 0;
        if (counts_out) counts_out[i] = c;
        if (offsets_out && c) memcpy(offsets_out + n, r->kmer_starts[i], sizeof(int) * c);
        n += c;
    }
    return n;
}

/* both strands of read `rid` of volume `reads` against (ref, idx); out must hold MAXC entries (12 ints each:
   loc1,loc2,left1,left2,right1,right2,score,num1,num2,readno,readstart,chain) -- pw_impl.cpp:740-765 */
static SeedingBK* g_sbk = NULL; static int g_sbk_size = -1;
int refh_seed_read(void* refv, void* readsv, void* idxv, int rid, int chain_as_char, int* out)
{
    volume_t* ref = (volume_t*)refv; volume_t* reads = (volume_t*)readsv; ref_index* ridx = (ref_index*)idxv;
    if (g_sbk_size != ref->curr) { delete g_sbk; g_sbk = new SeedingBK(ref->curr); g_sbk_size = ref->curr; }
    static char* read1 = (char*)malloc(MAX_SEQ_SIZE); static char* read2 = (char*)malloc(MAX_SEQ_SIZE);
    Candidate cands[MAXC];
    int rsize = reads->offset_list->offset_list[rid].size;
    extract_one_seq(reads, rid, read1);
    reverse_complement(read2, read1, rsize);
    int n = 0;
    for (int s = 0; s < 2; ++s) {
        const char* read = s ? read2 : read1;
        int chain = chain_as_char ? (s ? 'R' : 'F') : (s ? REV : FWD);
        int num_segs = seeding(read, rsize, ridx, g_sbk);
        n = get_candidates(ref, g_sbk, num_segs, rid + reads->start_read_id, rsize, chain, cands, n);
    }
    for (int i = 0; i < n; ++i) {
        int* o = out + 12 * i;
        o[0] = cands[i].loc1; o[1] = cands[i].loc2; o[2] = cands[i].left1; o[3] = cands[i].left2;
        o[4] = cands[i].right1; o[5] = cands[i].right2; o[6] = cands[i].score; o[7] = cands[i].num1;
        o[8] = cands[i].num2; o[9] = cands[i].readno; o[10] = cands[i].readstart; o[11] = cands[i].chain;
    }
    return n;
}

/* insert_loc on an explicit Back_List image: score, loczhi[40], seedno[40] (pw_impl.cpp:121-159) */
void refh_insert_loc(short* score, short* loczhi, short* seedno, int loc, int seedn, float len)
{
    Back_List b; b.score = *score; memcpy(b.loczhi, loczhi, sizeof(b.loczhi)); memcpy(b.seedno, seedno, sizeof(b.seedno));
    b.seednum = 0; b.index = 0;
    insert_loc(&b, loc, seedn, len);
    *score = b.score; memcpy(loczhi, b.loczhi, sizeof(b.loczhi)); memcpy(seedno, b.seedno, sizeof(b.seedno));
}

int refh_find_location(int* t_loc, int* t_seedn, int* t_score, int* loc, int k, int* rep_loc, float len, int read_len)
{
    *rep_loc = -1;
    return find_location(t_loc, t_seedn, t_score, loc, k, rep_loc, len, read_len);
}

/* Align (diff_gapalign.cpp:107): sequences as codes 0..3; left extension passes pointers to the LAST element and
   reads backwards, exactly as dw_in_one_direction does.  Returns Align()'s value; res = {aln_str_size, dist,
   aln_q_s, aln_q_e, aln_t_s, aln_t_e}; q_aln/t_aln receive aln_str_size codes (0..4). */
static DiffAligner* g_da = NULL;
int refh_align(const char* q, int qlen, const char* t, int tlen, int band_tol, int get_aln, int right_extend,
               int* res, char* q_aln, char* t_aln)
{
    if (!g_da) g_da = new DiffAligner(0);
    std::fill(g_da->dynq, g_da->dynq + 4096, 0);
    std::fill(g_da->dynt, g_da->dynt + 4096, 0);
    const char* qq = right_extend ? q : q + qlen - 1;
    const char* tt = right_extend ? t : t + tlen - 1;
    /* note the reference passes (U=dynq, V=dynt) into parameters named (V, U): diff_gapalign.cpp:257-262 */
    int r = Align(qq, qlen, tt, tlen, band_tol, get_aln, g_da->align, g_da->dynq, g_da->dynt, g_da->d_path, g_da->aln_path, right_extend);
    Alignment* a = g_da->align;
    res[0] = a->aln_str_size; res[1] = a->dist; res[2] = a->aln_q_s; res[3] = a->aln_q_e; res[4] = a->aln_t_s; res[5] = a->aln_t_e;
    if (get_aln && q_aln) { memcpy(q_aln, a->q_aln_str, a->aln_str_size); memcpy(t_aln, a->t_aln_str, a->aln_str_size); }
    return r;
}

/* DiffAligner::go (diff_gapalign.cpp:294). res = {ok, query_start, query_end, target_start, target_end, matches, columns} */
int refh_dw_go(const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, int min_aln, int* res, double* ident)
{
    if (!g_da) g_da = new DiffAligner(0);
    bool ok = g_da->go(q, qstart, qsize, t, tstart, tsize, min_aln);
    OutputStore* r = g_da->result;
    int n = 0;
    for (int i = 0; i < r->out_store_size; ++i) if (r->out_store1[i] == r->out_store2[i]) ++n;
    res[0] = ok; res[1] = r->query_start; res[2] = r->query_end; res[3] = r->target_start; res[4] = r->target_end;
    res[5] = n; res[6] = r->out_store_size;
    *ident = r->calc_ident();
    return ok;
}

/* XdropAligner::go (xdrop_gapalign.cpp:359). res = {ok, qoff, qend, toff, tend, matches, columns} */
static XdropAligner* g_xa = NULL;
int refh_xdrop_go(const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, int min_aln, int* res, double* ident)
{
    if (!g_xa) g_xa = new XdropAligner(0);
    bool ok = g_xa->go(q, qstart, qsize, t, tstart, tsize, min_aln);
    int n = 0;
    for (int i = 0; i < g_xa->aln_size; ++i) if (g_xa->qaln[i] == g_xa->taln[i]) ++n;
    res[0] = ok; res[1] = g_xa->qoff; res[2] = g_xa->qend; res[3] = g_xa->toff; res[4] = g_xa->tend; res[5] = n; res[6] = g_xa->aln_size;
    *ident = g_xa->calc_ident();
    return ok;
}

/* xdrop_align on one block (xdrop_gapalign.cpp:10); res = {ae, be, num_ops}, ops = (type, count) pairs in edit_block order */
int refh_xdrop_align(const char* A, int M, const char* B, int N, int forward, int* res, int* ops)
{
    if (!g_xa) g_xa = new XdropAligner(0);
    const char* AA = forward ? A : A + M - 1;
    const char* BB = forward ? B : B + N - 1;
    int ae, be;
    int sc = xdrop_align(AA, M, BB, N, g_xa->score_matrix, g_xa->param.gap_open, g_xa->param.gap_extend, g_xa->param.x_dropoff,
                         g_xa->state_array, g_xa->score_array, g_xa->edit_script, g_xa->edit_start_offset, &g_xa->edit_block,
                         forward, ae, be);
    res[0] = ae; res[1] = be; res[2] = g_xa->edit_block.num_ops;
    for (int i = 0; i < g_xa->edit_block.num_ops; ++i) { ops[2 * i] = g_xa->edit_block.edit_ops[i].op_type; ops[2 * i + 1] = g_xa->edit_block.edit_ops[i].num; }
    return sc;
}

/* append_m4v (pw_impl.cpp:576-610) on a caller-made list of n M4Records (104 bytes each, alignment.h:21-37): the per-read sort
 * (std::sort in this build's mode: libstdc++ parallel mode, compiled the way the reference's release build compiles it) and the
 * containment filter.  out receives the records append_m4v keeps, in its order; returns their number.  (n <= kResultListSize.) */
int refh_append_m4v(const void* recs, int n, void* out)
{
    M4Record* llist = new M4Record[n > 0 ? n : 1];
    M4Record* glist = new M4Record[PWThreadData::kResultListSize + (n > 0 ? n : 1)];
    memcpy((void*)llist, recs, sizeof(M4Record) * (size_t)n);
    int gsz = 0, lsz = n;
    std::ostringstream sink;
    pthread_mutex_t mu = PTHREAD_MUTEX_INITIALIZER;
    append_m4v(glist, &gsz, llist, &lsz, &sink, &mu);
    memcpy(out, (const void*)glist, sizeof(M4Record) * (size_t)gsz);
    delete[] llist;
    delete[] glist;
    return gsz;
}

int refh_sizeof(int what)
{
    switch (what) {
    case 0: return sizeof(Back_List); case 1: return sizeof(candidate_save); case 2: return sizeof(M4Record);
    case 3: return sizeof(ExtensionCandidate); case 4: return sizeof(DPathData2); case 5: return sizeof(offset_t);
    case 6: return sizeof(volume_t);
    }
    return -1;
}

} /* extern "C" */
