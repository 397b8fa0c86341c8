/*
 * mecat_oracle.c — plain-C restatement of the reference mecat2pw hot path.  TEST INFRASTRUCTURE ONLY.
 * See mecat_oracle.h for the rules; every function cites the reference file:line it follows
 * (paths relative to /root/reference/src/).  Quirks are reproduced literally, not fixed.
 *
 * Compile with -ffp-contract=off on x86-64 (SSE2, FLT_EVAL_METHOD 0) so the three floating-point precisions used
 * by the reference (f32 in find_location, f32 divide + f64 compare in insert_loc, f64 in the neighbour sweeps)
 * round exactly as g++ -O3 rounds them.
 */
#include "mecat_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MUL_ZV(a) ((a) * ORC_ZV)
#define DIV_ZV(a) ((a) / ORC_ZV)
#define MOD_ZV(a) ((a) % ORC_ZV)

static void* xmalloc(size_t n) { void* p = malloc(n ? n : 1); if (!p) { fprintf(stderr, "oracle: malloc fail\n"); abort(); } return p; }
static void* xcalloc(size_t n) { void* p = calloc(1, n ? n : 1); if (!p) { fprintf(stderr, "oracle: calloc fail\n"); abort(); } return p; }

/* pw_options.cpp:8-13,30-50 and pw_impl.cpp:843-851 */
void orc_params_default(orc_params* p, int tech)
{
    p->maxc = 100;
    p->tech = tech;
    p->output_gapped_start_point = 0;
    p->ddfs_cutoff = 0.25;
    if (tech == 0) { p->min_align_size = 2000; p->min_kmer_match = 4; p->min_kmer_dist = 1800; }
    else           { p->min_align_size = 500;  p->min_kmer_match = 2; p->min_kmer_dist = 400; }
}

/* ------------------------------------------------------------------ A1 / A3: volumes */

/* defs.cpp:3-36 */
uint8_t orc_encode_base(int c)
{
    switch (c) {
    case '-': return 15;
    case 'a': case 'A': return 0;  case 'c': case 'C': return 1;  case 'm': case 'M': return 6;
    case 'g': case 'G': return 2;  case 'r': case 'R': return 4;  case 's': case 'S': return 9;
    case 'v': case 'V': return 13; case 't': case 'T': return 3;  case 'w': case 'W': return 8;
    case 'y': case 'Y': return 5;  case 'h': case 'H': return 12; case 'k': case 'K': return 7;
    case 'd': case 'D': return 11; case 'b': case 'B': return 10; case 'n': case 'N': return 14;
    }
    return 16;
}

/* packed_db.h:98-107 */
static inline void set_char(uint8_t* p, int64_t idx, uint8_t c) { p[idx >> 2] |= (uint8_t)(c << ((~idx & 3) << 1)); }
static inline uint8_t get_char(const uint8_t* p, int64_t idx) { return (uint8_t)((p[idx >> 2] >> ((~idx & 3) << 1)) & 3); }

/* add_one_seq (split_database.cpp:103-119) + the `++v->curr` pad of split_raw_dataset (:249-250), for one volume.
   `codes` are encode-table values (0..15), OR-ed unmasked exactly like the reference. */
orc_volume* orc_volume_pack(const uint8_t* codes, const int* lens, int nreads, int start_read_id)
{
    orc_volume* v = (orc_volume*)xcalloc(sizeof(orc_volume));
    int64_t tot = 0;
    for (int i = 0; i < nreads; ++i) tot += lens[i] + 1;
    if (tot > ORC_MCS) { fprintf(stderr, "oracle: volume too large\n"); abort(); }
    v->num_reads = nreads;
    v->start_read_id = start_read_id;
    v->offs = (orc_offset_t*)xmalloc(sizeof(orc_offset_t) * (size_t)nreads);
    v->pac = (uint8_t*)xcalloc((size_t)((tot + 3) / 4));
    int curr = 0;
    int64_t src = 0;
    for (int i = 0; i < nreads; ++i) {
        v->offs[i].offset = curr;
        v->offs[i].size = lens[i];
        for (int k = 0; k < lens[i]; ++k) { set_char(v->pac, curr, codes[src + k]); ++curr; }
        src += lens[i];
        ++curr; /* pad */
    }
    v->num_bases = curr;
    return v;
}

/* split_database.cpp:155-181 */
orc_volume* orc_volume_load(const char* path)
{
    FILE* in = fopen(path, "rb");
    if (!in) return NULL;
    orc_volume* v = (orc_volume*)xcalloc(sizeof(orc_volume));
    int ok = fread(&v->num_reads, sizeof(int), 1, in) == 1 && fread(&v->num_bases, sizeof(int), 1, in) == 1 &&
             fread(&v->start_read_id, sizeof(int), 1, in) == 1;
    if (!ok) { fclose(in); free(v); return NULL; }
    v->offs = (orc_offset_t*)xmalloc(sizeof(orc_offset_t) * (size_t)v->num_reads);
    size_t nb = (size_t)((v->num_bases + 3) / 4);
    v->pac = (uint8_t*)xmalloc(nb);
    ok = fread(v->offs, sizeof(orc_offset_t), (size_t)v->num_reads, in) == (size_t)v->num_reads &&
         fread(v->pac, 1, nb, in) == nb;
    fclose(in);
    if (!ok) { orc_volume_free(v); return NULL; }
    return v;
}

/* split_database.cpp:135-153 */
int orc_volume_dump(const orc_volume* v, const char* path)
{
    FILE* out = fopen(path, "wb");
    if (!out) return -1;
    fwrite(&v->num_reads, sizeof(int), 1, out);
    fwrite(&v->num_bases, sizeof(int), 1, out);
    fwrite(&v->start_read_id, sizeof(int), 1, out);
    fwrite(v->offs, sizeof(orc_offset_t), (size_t)v->num_reads, out);
    fwrite(v->pac, 1, (size_t)((v->num_bases + 3) / 4), out);
    return fclose(out);
}

void orc_volume_free(orc_volume* v)
{
    if (!v) return;
    free(v->offs); free(v->pac); free(v);
}

/* split_database.cpp:121-133 */
void orc_extract_one_seq(const orc_volume* v, int id, char* s)
{
    int offset = v->offs[id].offset, size = v->offs[id].size;
    for (int i = 0; i < size; ++i) s[i] = (char)get_char(v->pac, offset + i);
}

/* pw_impl.cpp:69-81 */
void orc_reverse_complement(char* dst, const char* src, int size)
{
    for (int i = 0; i < size; ++i) dst[size - 1 - i] = (char)(3 - (uint8_t)src[i]);
}

/* split_database.cpp:15-35 */
int orc_read_id_from_offset(const orc_volume* v, int offset)
{
    int n = v->num_reads;
    const orc_offset_t* a = v->offs;
    int left = 0, right = n - 1, mid = (left + right) / 2;
    if (a[right].offset < offset) return right;
    while (left <= right) {
        if (a[mid].offset <= offset && a[mid].offset + a[mid].size > offset) return mid;
        if (a[mid].offset + a[mid].size <= offset) left = mid + 1;
        else if (a[mid].offset > offset) right = mid - 1;
        else { fprintf(stderr, "oracle: offset search error\n"); exit(1); }
        mid = (left + right) / 2;
    }
    return mid;
}

/* ------------------------------------------------------------------ A2: index (lookup_table.cpp:63-160) */

orc_index* orc_index_build(const orc_volume* v)
{
    const int kmer_size = ORC_KMER;
    const uint32_t index_count = 1u << (kmer_size * 2);
    const uint32_t leftnum = 34 - 2 * kmer_size;
    orc_index* index = (orc_index*)xcalloc(sizeof(orc_index));
    index->counts = (int*)xcalloc(sizeof(int) * (size_t)index_count);
    /* pass 1, :73-92 */
    for (int i = 0; i != v->num_reads; ++i) {
        int read_start = v->offs[i].offset, read_size = v->offs[i].size;
        uint32_t eit = 0;
        for (int j = 0; j < read_size; ++j) {
            uint8_t c = get_char(v->pac, read_start + j);
            eit = (eit << 2) | c;
            if (j >= kmer_size - 1) {
                ++index->counts[eit];
                eit = eit << leftnum;
                eit = eit >> leftnum;
            }
        }
    }
    /* :94-99 */
    int64_t num_kmers = 0;
    for (uint32_t i = 0; i != index_count; ++i) {
        if (index->counts[i] > 128) index->counts[i] = 0;
        num_kmers += index->counts[i];
    }
    index->num_kmers = num_kmers;
    index->offsets = (int*)xmalloc(sizeof(int) * (size_t)num_kmers);
    index->starts = (int64_t*)xmalloc(sizeof(int64_t) * (size_t)index_count);
    /* :111-141 (thread partition by key range does not change bucket order; see fill func :25-61) */
    num_kmers = 0;
    for (uint32_t i = 0; i != index_count; ++i) {
        if (index->counts[i]) { index->starts[i] = num_kmers; num_kmers += index->counts[i]; index->counts[i] = 0; }
        else index->starts[i] = -1;
    }
    /* fill_ref_index_offsets_func :25-61, single key range */
    for (int i = 0; i < v->num_reads; ++i) {
        int read_start = v->offs[i].offset, read_size = v->offs[i].size;
        uint32_t eit = 0;
        for (int j = 0; j < read_size; ++j) {
            int k = read_start + j;
            uint8_t c = get_char(v->pac, k);
            eit = (eit << 2) | c;
            if (j >= kmer_size - 1) {
                if (index->starts[eit] >= 0) {
                    index->offsets[index->starts[eit] + index->counts[eit]] = k + 1 - kmer_size;
                    ++index->counts[eit];
                }
                eit <<= leftnum;
                eit >>= leftnum;
            }
        }
    }
    return index;
}

void orc_index_free(orc_index* idx)
{
    if (!idx) return;
    free(idx->counts); free(idx->starts); free(idx->offsets); free(idx);
}

/* ------------------------------------------------------------------ A4-A8: seeding */

struct orc_seeding_bk {             /* pw_impl.h:55-64, pw_impl.cpp:99-119 */
    int* index_list;
    int16_t* index_score;
    orc_back_list* database;
    int* kmer_ids;
    int num_segs;
};

orc_seeding_bk* orc_bk_new(int ref_size)
{
    orc_seeding_bk* bk = (orc_seeding_bk*)xcalloc(sizeof(*bk));
    const int num_segs = ref_size / ORC_ZV + 5;
    bk->num_segs = num_segs;
    bk->index_list = (int*)xmalloc(sizeof(int) * (size_t)num_segs);
    bk->index_score = (int16_t*)xmalloc(sizeof(int16_t) * (size_t)num_segs);
    bk->database = (orc_back_list*)xmalloc(sizeof(orc_back_list) * (size_t)num_segs);
    bk->kmer_ids = (int*)xmalloc(sizeof(int) * ORC_MAX_SEQ_SIZE);
    for (int i = 0; i < num_segs; ++i) {
        memset(&bk->database[i], 0, sizeof(orc_back_list)); /* the reference leaves the lists uninitialised */
        bk->database[i].score = 0;
        bk->database[i].index = -1;
    }
    return bk;
}

void orc_bk_free(orc_seeding_bk* bk)
{
    if (!bk) return;
    free(bk->index_list); free(bk->index_score); free(bk->database); free(bk->kmer_ids); free(bk);
}

/* pw_impl.cpp:83-97 */
int orc_extract_kmers(const char* s, int ssize, int* kmer_ids)
{
    int num_kmers = (ssize - ORC_KMER) / ORC_BC + 1;
    for (int i = 0; i < num_kmers; ++i) {
        int eit = 0, start = i * ORC_BC;
        for (int j = 0; j < ORC_KMER; ++j) eit = (eit << 2) | s[start + j];
        kmer_ids[i] = eit;
    }
    return num_kmers;
}

/* sweep statistics (tools/dev/parity_sweep.py; not part of the restated algorithm): how often the 41st-seed rule ran, how often on
   a hit outside the query read's own copy in the ref volume ([self_lo, self_hi), set by orc_seed_read), and its three outcomes
   [0] calls  [1] calls on non-self hits  [2] tail replaced (all 41 agree)  [3] one entry dropped  [4] new seed ignored
   [5] non-self calls that dropped an entry.  Plain globals: the sweeps call the oracle from one thread. */
static int64_t g_stat[8];
static int64_t g_self_lo = -1, g_self_hi = -1;
static int g_stat_nonself = 0;
void orc_stats_reset(void) { memset(g_stat, 0, sizeof(g_stat)); }
void orc_stats_get(int64_t* out) { memcpy(out, g_stat, sizeof(g_stat)); }

/* pw_impl.cpp:121-159.  FP: int/(int*float) is an f32 divide; "- 1.0" and the compare are f64. */
void orc_insert_loc(orc_back_list* spr, int loc, int seedn, float len, double cutoff)
{
    int list_loc[ORC_SI], list_score[ORC_SI], list_seed[ORC_SI], i, j, minval, mini;
    for (i = 0; i < ORC_SM; i++) {
        list_loc[i] = spr->loczhi[i];
        list_seed[i] = spr->seedno[i];
        list_score[i] = 0;
    }
    list_loc[ORC_SM] = loc;
    list_seed[ORC_SM] = seedn;
    list_score[ORC_SM] = 0;
    mini = -1;
    minval = 10000;
    for (i = 0; i < ORC_SM; i++)
        for (j = i + 1; j < ORC_SI; j++)
            if (list_seed[j] - list_seed[i] > 0 && list_loc[j] - list_loc[i] > 0 &&
                fabs((list_loc[j] - list_loc[i]) / ((list_seed[j] - list_seed[i]) * len) - 1.0) < cutoff) {
                list_score[i]++;
                list_score[j]++;
            }
    for (i = 0; i < ORC_SI; i++)
        if (minval > list_score[i]) { minval = list_score[i]; mini = i; }
    g_stat[0]++; g_stat[1] += g_stat_nonself;
    if (minval == ORC_SM) {
        spr->loczhi[ORC_SM - 1] = (int16_t)loc;
        spr->seedno[ORC_SM - 1] = (int16_t)seedn;
        g_stat[2]++;
    } else if (minval < ORC_SM && mini < ORC_SM) {
        for (i = mini; i < ORC_SM; i++) {
            spr->loczhi[i] = (int16_t)list_loc[i + 1];
            spr->seedno[i] = (int16_t)list_seed[i + 1];
        }
        spr->score--;
        g_stat[3]++; g_stat[5] += g_stat_nonself;
    } else g_stat[4]++;
}

/* pw_impl.cpp:161-239.  FP: everything before the compare is f32 ("- 1" converts the int). */
static inline int ddf_f32(int dloc, int dseed, float len, double cutoff)
{
    float r = dloc / (dseed * len) - 1;
    return fabsf(r) < cutoff;
}

int orc_find_location(int* t_loc, int* t_seedn, int* t_score, int* loc, int k, int* rep_loc,
                      float len, int read_len1, double cutoff)
{
    int i, j, maxval = 0, maxi = 0 /* uninitialised in the reference; only read when maxval >= 5 */, rep = 0, lasti = 0, tempi;
    for (i = 0; i < k; i++) t_score[i] = 0;
    for (i = 0; i < k - 1; i++)
        for (j = i + 1, tempi = t_seedn[i]; j < k; j++)
            if (tempi != t_seedn[j] && t_seedn[j] - t_seedn[i] > 0 && t_loc[j] - t_loc[i] > 0 &&
                t_loc[j] - t_loc[i] < read_len1 && ddf_f32(t_loc[j] - t_loc[i], t_seedn[j] - t_seedn[i], len, cutoff)) {
                t_score[i]++;
                t_score[j]++;
                tempi = t_seedn[j];
            }
    for (i = 0; i < k; i++) {
        if (maxval < t_score[i]) { maxval = t_score[i]; maxi = i; rep = 0; }
        else if (maxval == t_score[i]) { rep++; lasti = i; }
    }
    for (i = 0; i < 4; i++) loc[i] = 0;
    if (maxval >= 5 && rep == maxval) {
        loc[0] = t_loc[maxi], loc[1] = t_seedn[maxi];
        *rep_loc = maxi;
        loc[2] = t_loc[lasti], loc[3] = t_seedn[lasti];
        return 1;
    } else if (maxval >= 5 && rep != maxval) {
        for (j = 0; j < maxi; j++)
            if (t_seedn[maxi] - t_seedn[j] > 0 && t_loc[maxi] - t_loc[j] > 0 && t_loc[maxi] - t_loc[j] < read_len1 &&
                ddf_f32(t_loc[maxi] - t_loc[j], t_seedn[maxi] - t_seedn[j], len, cutoff)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        j = maxi;
        if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
        else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
        for (j = maxi + 1; j < k; j++)
            if (t_seedn[j] - t_seedn[maxi] > 0 && t_loc[j] - t_loc[maxi] > 0 && t_loc[j] - t_loc[maxi] <= read_len1 &&
                ddf_f32(t_loc[j] - t_loc[maxi], t_seedn[j] - t_seedn[maxi], len, cutoff)) {
                if (loc[0] == 0) { loc[0] = t_loc[j]; loc[1] = t_seedn[j]; *rep_loc = j; }
                else { loc[2] = t_loc[j]; loc[3] = t_seedn[j]; }
            }
        return 1;
    }
    return 0;
}

/* pw_impl.cpp:241-286 */
int orc_seeding(const char* read, int read_size, const orc_index* ridx, orc_seeding_bk* sbk)
{
    int* kmer_ids = sbk->kmer_ids;
    int* index_spr = sbk->index_list;
    int16_t* index_score = sbk->index_score;
    int16_t* index_ss = index_score;
    orc_back_list* database = sbk->database;
    int num_kmers = orc_extract_kmers(read, read_size, kmer_ids);
    int used_segs = 0;
    for (int km = 0; km < num_kmers; ++km) {
        int num_seeds = ridx->counts[kmer_ids[km]];
        const int* seed_arr = ridx->starts[kmer_ids[km]] >= 0 ? ridx->offsets + ridx->starts[kmer_ids[km]] : NULL;
        for (int sid = 0; sid < num_seeds; ++sid) {
            int seg_id = seed_arr[sid] / ORC_ZV;
            int seg_off = seed_arr[sid] % ORC_ZV;
            orc_back_list* spr = database + seg_id;
            if (spr->score == 0 || spr->seednum < km + 1) {
                int loc = ++spr->score;
                if (loc <= ORC_SM) { spr->loczhi[loc - 1] = (int16_t)seg_off; spr->seedno[loc - 1] = (int16_t)(km + 1); }
                else {
                    g_stat_nonself = !(seed_arr[sid] >= g_self_lo && seed_arr[sid] < g_self_hi);      /* statistics only */
                    orc_insert_loc(spr, seg_off, km + 1, ORC_BC, 0.25 /* ddfs_cutoff: 0.25 for both techs, pw_impl.cpp:21-23 */);
                }
                int s_k;
                if (seg_id > 0) s_k = spr->score + (spr - 1)->score;
                else s_k = spr->score;
                if (spr->index == -1) {
                    *(index_spr++) = seg_id;
                    *(index_ss++) = (int16_t)s_k;
                    spr->index = used_segs++;
                } else index_score[spr->index] = (int16_t)s_k;
            }
            spr->seednum = (int16_t)(km + 1);
        }
    }
    return used_segs;
}

/* pw_impl.cpp:288-465 */
int orc_get_candidates(const orc_volume* ref, orc_seeding_bk* sbk, int num_segs, int read_id, int read_size,
                       char chain, orc_candidate* candidates, int candidatenum, const orc_params* P)
{
    const int MAXC = P->maxc, min_kmer_match = P->min_kmer_match, min_kmer_dist = P->min_kmer_dist;
    const double ddfs_cutoff = P->ddfs_cutoff;
    int* index_list = sbk->index_list;
    int* index_spr = index_list;
    int16_t* index_ss = sbk->index_score;
    orc_back_list* database = sbk->database;
    enum { temp_arr_size = 2 * ORC_SM + 10 };
    int temp_list[temp_arr_size], temp_seedn[temp_arr_size], temp_score[temp_arr_size];
    orc_candidate *candidate_loc = candidates, candidate_temp;
    int location_loc[4], repeat_loc = 0;
    int i, j, k, u_k;
    memset(&candidate_temp, 0, sizeof(candidate_temp));
    for (i = 0; i < num_segs; ++i, ++index_spr, ++index_ss)
        if (*index_ss >= 2 * min_kmer_match) {
            orc_back_list *spr = database + (*index_spr), *spr1;
            if (spr->score == 0) continue;
            int s_k = spr->score;
            int start_loc = *index_spr;
            start_loc = MUL_ZV(start_loc);
            int loc;
            if ((*index_spr) > 0) {
                loc = (spr - 1)->score;
                if (loc > 0) { start_loc = (*index_spr - 1); start_loc = MUL_ZV(start_loc); }
            } else loc = 0;

            if (!loc)
                for (j = 0, u_k = 0; j < s_k && j < ORC_SM; ++j) {
                    temp_list[u_k] = spr->loczhi[j];
                    temp_seedn[u_k] = spr->seedno[j];
                    ++u_k;
                }
            else {
                k = loc;
                u_k = 0;
                spr1 = spr - 1;
                for (j = 0; j < k && j < ORC_SM; ++j) {
                    temp_list[u_k] = spr1->loczhi[j];
                    temp_seedn[u_k] = spr1->seedno[j];
                    ++u_k;
                }
                for (j = 0; j < s_k && j < ORC_SM; ++j) {
                    temp_list[u_k] = spr->loczhi[j] + ORC_ZV;
                    temp_seedn[u_k] = spr->seedno[j];
                    ++u_k;
                }
            }
            {
                int f = orc_find_location(temp_list, temp_seedn, temp_score, location_loc, u_k, &repeat_loc, ORC_BC, read_size, ddfs_cutoff);
                if (!f) continue;
                if (temp_score[repeat_loc] < 2 * min_kmer_match + 2) continue;
            }
            candidate_temp.score = temp_score[repeat_loc];
            candidate_temp.chain = chain;
            int loc_seed = temp_seedn[repeat_loc];
            location_loc[0] = start_loc + location_loc[0];
            int loc_list = location_loc[0];
            int sid = orc_read_id_from_offset(ref, location_loc[0]);
            int sstart = ref->offs[sid].offset;
            int ssize = ref->offs[sid].size;
            int send = sstart + ssize + 1;
            sid += ref->start_read_id;
            if (sid > read_id) continue;
            if (sid == read_id) {
                u_k = DIV_ZV(sstart);
                spr = database + u_k;
                s_k = MOD_ZV(sstart);
                for (j = 0, k = 0; j < spr->score && j < ORC_SM; ++j)
                    if (spr->loczhi[j] < s_k) { spr->loczhi[k] = spr->loczhi[j]; ++k; }
                spr->score = (int16_t)k;
                for (++spr, ++u_k, k = DIV_ZV(send); u_k < k; ++u_k, ++spr) spr->score = 0;
                for (j = 0, k = 0, s_k = MOD_ZV(send); j < spr->score && j < ORC_SM; ++j)
                    if (spr->loczhi[j] > s_k) { spr->loczhi[k] = spr->loczhi[j]; ++k; }
                spr->score = (int16_t)k;
            } else {
                candidate_temp.readno = sid;
                candidate_temp.readstart = sstart;
                location_loc[1] = (location_loc[1] - 1) * ORC_BC;
                int left_length1 = location_loc[0] - sstart + ORC_KMER - 1;
                int right_length1 = send - location_loc[0];
                int left_length2 = location_loc[1] + ORC_KMER - 1;
                int right_length2 = read_size - location_loc[1];
                int num1 = (left_length1 > left_length2) ? left_length2 : left_length1;
                int num2 = (right_length1 > right_length2) ? right_length2 : right_length1;
                if (num1 + num2 < min_kmer_dist) continue;
                candidate_temp.loc1 = location_loc[0] - sstart;
                candidate_temp.num1 = num1;
                candidate_temp.loc2 = location_loc[1];
                candidate_temp.num2 = num2;
                candidate_temp.left1 = left_length1;
                candidate_temp.left2 = left_length2;
                candidate_temp.right1 = right_length1;
                candidate_temp.right2 = right_length2;

                int seedcount = 0;
                int nlb = (num1 + ORC_ZV - 1);
                nlb = DIV_ZV(nlb);
                for (u_k = *index_spr - 1, spr1 = spr - 1; u_k >= 0 && nlb > 0; spr1--, --nlb, u_k--)
                    if (spr1->score > 0) {
                        start_loc = MUL_ZV(u_k);
                        int scnt = spr1->score < ORC_SM ? spr1->score : ORC_SM;
                        for (j = 0, s_k = 0; j < scnt; j++)
                            if (fabs((loc_list - start_loc - spr1->loczhi[j]) / ((loc_seed - spr1->seedno[j]) * ORC_BC * 1.0) - 1.0) < ddfs_cutoff) {
                                seedcount++;
                                s_k++;
                            }
                        if (s_k * 1.0 / scnt > 0.4) spr1->score = 0;
                    }
                int nrb = (num2 + ORC_ZV - 1);
                nrb = DIV_ZV(nrb);
                for (u_k = *index_spr + 1, spr1 = spr + 1; nrb; spr1++, --nrb, u_k++)
                    if (spr1->score > 0) {
                        start_loc = MUL_ZV(u_k);
                        int scnt = spr1->score < ORC_SM ? spr1->score : ORC_SM;
                        for (j = 0, s_k = 0; j < scnt; j++)
                            if (fabs((start_loc + spr1->loczhi[j] - loc_list) / ((spr1->seedno[j] - loc_seed) * ORC_BC * 1.0) - 1.0) < ddfs_cutoff) {
                                seedcount++;
                                s_k++;
                            }
                        if (s_k * 1.0 / scnt > 0.4) spr1->score = 0;
                    }

                candidate_temp.score = candidate_temp.score + seedcount;
                int low = 0;
                int high = candidatenum - 1;
                int mid;
                while (low <= high) {
                    mid = (low + high) / 2;
                    if (mid >= candidatenum || candidate_loc[mid].score < candidate_temp.score) high = mid - 1;
                    else low = mid + 1;
                }
                if (candidatenum < MAXC) for (u_k = candidatenum - 1; u_k > high; u_k--) candidate_loc[u_k + 1] = candidate_loc[u_k];
                else for (u_k = candidatenum - 2; u_k > high; u_k--) candidate_loc[u_k + 1] = candidate_loc[u_k];
                if (high + 1 < MAXC) candidate_loc[high + 1] = candidate_temp;
                if (candidatenum < MAXC) candidatenum++;
                else candidatenum = MAXC;
            }
        }
    for (i = 0, index_spr = index_list; i < num_segs; i++, index_spr++) {
        database[*index_spr].score = 0;
        database[*index_spr].index = -1;
    }
    return candidatenum;
}

/* candidate_detect pw_impl.cpp:742-765 / pairwise_mapping :653-672 */
int orc_seed_read(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                  int rid, int chain_as_char, const orc_params* p, orc_candidate* out)
{
    int rsize = reads->offs[rid].size;
    /* a read of 4 .. 12 bases gets ONE k-mer from extract_kmers ((rsize - 13) / 10 + 1 with C's truncation) that runs past the read:
       in the reference those bytes are whatever an earlier read left in the thread's MAX_SEQ_SIZE buffer (undefined, and not
       reproducible); here they are zeros, so that the restatement at least stays inside its buffers.  The product defines such a
       read to have no k-mers (known divergences); tests leave these reads out of the comparison. */
    char* read1 = (char*)xmalloc((size_t)rsize + ORC_KMER + 1);
    char* read2 = (char*)xmalloc((size_t)rsize + ORC_KMER + 1);
    memset(read1, 0, (size_t)rsize + ORC_KMER + 1);
    memset(read2, 0, (size_t)rsize + ORC_KMER + 1);
    orc_extract_one_seq(reads, rid, read1);
    orc_reverse_complement(read2, read1, rsize);
    int n = 0;
    {   /* statistics only: where the query read's own copy lies in the ref volume, if it is there */
        int lr = rid + reads->start_read_id - ref->start_read_id;
        g_self_lo = g_self_hi = -1;
        if (lr >= 0 && lr < ref->num_reads) { g_self_lo = ref->offs[lr].offset; g_self_hi = g_self_lo + ref->offs[lr].size; }
    }
    for (int s = 0; s < 2; ++s) {
        const char* read = s ? read2 : read1;
        char chain = chain_as_char ? (s ? 'R' : 'F') : (char)(s ? 1 : 0);
        int num_segs = orc_seeding(read, rsize, ridx, bk);
        n = orc_get_candidates(ref, bk, num_segs, rid + reads->start_read_id, rsize, chain, out, n, p);
    }
    free(read1); free(read2);
    return n;
}

/* bench.py's cpu_baseline "port" leg (only when oracle/_ref was not built): index + candidates of every read of one
   volume built from code arrays; returns the number of candidates */
int64_t orc_bench_candidates(const uint8_t* codes, const int* lens, int nreads, int tech)
{
    orc_params p;
    orc_params_default(&p, tech);
    orc_volume* v = orc_volume_pack(codes, lens, nreads, 0);
    orc_index* idx = orc_index_build(v);
    orc_seeding_bk* bk = orc_bk_new(v->num_bases);
    orc_candidate* out = (orc_candidate*)xmalloc(sizeof(orc_candidate) * (size_t)p.maxc);
    int64_t total = 0;
    for (int r = 0; r < nreads; ++r) total += orc_seed_read(v, v, idx, bk, r, 0, &p, out);
    free(out);
    orc_bk_free(bk);
    orc_index_free(idx);
    orc_volume_free(v);
    return total;
}

int orc_seeding_state(const char* read, int read_size, const orc_index* ridx, orc_seeding_bk* bk, int cap,
                      int* seg_ids, int16_t* idx_score, int16_t* scores, int16_t* loczhi, int16_t* seedno)
{
    int n = orc_seeding(read, read_size, ridx, bk);
    for (int i = 0; i < n; ++i) {
        int sgi = bk->index_list[i];
        if (i < cap) {
            seg_ids[i] = sgi;
            idx_score[i] = bk->index_score[i];
            scores[i] = bk->database[sgi].score;
            memcpy(loczhi + (size_t)i * ORC_SM, bk->database[sgi].loczhi, sizeof(int16_t) * ORC_SM);
            memcpy(seedno + (size_t)i * ORC_SM, bk->database[sgi].seedno, sizeof(int16_t) * ORC_SM);
        }
        bk->database[sgi].score = 0;
        bk->database[sgi].index = -1;
    }
    return n;
}

/* ------------------------------------------------------------------ A9: .can */

/* candidate_detect pw_impl.cpp:767-792 */
void orc_can_record(const orc_candidate* c, int qid, int qsize, int ssize, orc_ext_candidate* ec)
{
    int qstart = c->loc2, sstart = c->loc1;
    if (qstart && sstart) { qstart += ORC_KMER / 2; sstart += ORC_KMER / 2; }
    memset(ec, 0, sizeof(*ec));
    ec->qid = qid; ec->qdir = c->chain; ec->qext = qstart;
    ec->sid = c->readno; ec->sdir = 0; ec->sext = sstart;
    ec->score = c->score; ec->qsize = qsize; ec->ssize = ssize;
    if (ec->qdir == 1) ec->qext = ec->qsize - 1 - ec->qext;
    if (ec->sdir == 1) ec->sext = ec->ssize - 1 - ec->sext;
}

/* alignment.cpp:18-32 */
int orc_can_line(const orc_ext_candidate* ec, char* buf)
{
    return sprintf(buf, "%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\t%d\n", ec->qid, ec->sid, ec->qdir, ec->sdir,
                   ec->qext, ec->sext, ec->score, ec->qsize, ec->ssize);
}

/* ------------------------------------------------------------------ A10-A12: dw */

typedef struct { int d, k, pre_k, x1, y1, x2, y2; } dpath2;      /* DPathData2, diff_gapalign.h:126-129 */
typedef struct { int x, y; } path_point;                        /* diff_gapalign.h:131-134 */
typedef struct { int aln_str_size, dist, aln_q_s, aln_q_e, aln_t_s, aln_t_e; char *q_aln_str, *t_aln_str; } alignment_t;

struct orc_aligner {                                            /* DiffAligner + OutputStore, diff_gapalign.h:76-199 */
    int segment_size, row_size, column_size;
    int *dynq, *dynt;
    alignment_t align;
    dpath2* d_path;
    path_point* aln_path;
    char *left1, *left2, *right1, *right2;
    int left_size, right_size;
    int64_t n_blocks, n_cells, n_snake;
};

orc_aligner* orc_aligner_new(void)
{
    orc_aligner* a = (orc_aligner*)xcalloc(sizeof(*a));
    a->segment_size = 500; a->row_size = 4096; a->column_size = 4096;           /* DiffAlignParameters::init(0) */
    a->dynq = (int*)xmalloc(sizeof(int) * 4096);
    a->dynt = (int*)xmalloc(sizeof(int) * 4096);
    a->align.q_aln_str = (char*)xmalloc(4096);
    a->align.t_aln_str = (char*)xmalloc(4096);
    a->d_path = (dpath2*)xmalloc(sizeof(dpath2) * 5000000);
    a->aln_path = (path_point*)xmalloc(sizeof(path_point) * 5000000);
    a->left1 = (char*)xmalloc(ORC_MAX_SEQ_SIZE); a->left2 = (char*)xmalloc(ORC_MAX_SEQ_SIZE);
    a->right1 = (char*)xmalloc(ORC_MAX_SEQ_SIZE); a->right2 = (char*)xmalloc(ORC_MAX_SEQ_SIZE);
    return a;
}

void orc_aligner_free(orc_aligner* a)
{
    if (!a) return;
    free(a->dynq); free(a->dynt); free(a->align.q_aln_str); free(a->align.t_aln_str); free(a->d_path);
    free(a->aln_path); free(a->left1); free(a->left2); free(a->right1); free(a->right2); free(a);
}

void orc_dw_counters(const orc_aligner* a, int64_t* blocks, int64_t* cells, int64_t* snake)
{
    *blocks = a->n_blocks; *cells = a->n_cells; *snake = a->n_snake;
}

static inline char extract_char(const char* A, int i, int forward) { return forward ? A[i] : A[-i]; } /* gapalign.h:27-35 */

/* std::lower_bound over (d,k) — DPathDataExtractor, diff_gapalign.cpp:12-37 */
static const dpath2* dpath_lower_bound(const dpath2* list, int n, int d, int k)
{
    int lo = 0, cnt = n;
    while (cnt > 0) {
        int step = cnt / 2, mid = lo + step;
        int less = (list[mid].d == d) ? (list[mid].k < k) : (list[mid].d < d);
        if (less) { lo = mid + 1; cnt -= step + 1; } else cnt = step;
    }
    return list + lo;
}

/* diff_gapalign.cpp:39-104 */
static void get_align_string(const char* query, int q_len, const char* target, int t_len, const dpath2* d_path,
                             int d_path_size, path_point* aln_path, alignment_t* align, int d, int k, int right_extend)
{
    int cd = d, ck = k, aln_path_idx = 0, i;
    while (cd >= 0 && aln_path_idx < q_len + t_len + 1) {
        const dpath2* aux = dpath_lower_bound(d_path, d_path_size, cd, ck);
        aln_path[aln_path_idx].x = aux->x2; aln_path[aln_path_idx].y = aux->y2; ++aln_path_idx;
        aln_path[aln_path_idx].x = aux->x1; aln_path[aln_path_idx].y = aux->y1; ++aln_path_idx;
        ck = aux->pre_k;
        cd -= 1;
    }
    --aln_path_idx;
    int cx = aln_path[aln_path_idx].x, cy = aln_path[aln_path_idx].y;
    align->aln_q_s = cx;
    align->aln_t_s = cy;
    int aln_pos = 0;
    while (aln_path_idx > 0) {
        --aln_path_idx;
        int nx = aln_path[aln_path_idx].x, ny = aln_path[aln_path_idx].y;
        if (cx == nx && cy == ny) continue;
        if (cx == nx && cy != ny) {
            for (i = 0; i < ny - cy; ++i) {
                align->q_aln_str[aln_pos + i] = ORC_GAP_CODE;
                align->t_aln_str[aln_pos + i] = extract_char(target, cy + i, right_extend);
            }
            aln_pos += ny - cy;
        } else if (cx != nx && cy == ny) {
            for (i = 0; i < nx - cx; ++i) {
                align->q_aln_str[aln_pos + i] = extract_char(query, cx + i, right_extend);
                align->t_aln_str[aln_pos + i] = ORC_GAP_CODE;
            }
            aln_pos += nx - cx;
        } else {
            for (i = 0; i < nx - cx; ++i) align->q_aln_str[aln_pos + i] = extract_char(query, cx + i, right_extend);
            for (i = 0; i < ny - cy; ++i) align->t_aln_str[aln_pos + i] = extract_char(target, cy + i, right_extend);
            aln_pos += ny - cy;
        }
        cx = nx; cy = ny;
    }
    align->aln_str_size = aln_pos;
}

/* diff_gapalign.cpp:107-219 */
static int align_core(orc_aligner* A, const char* query, int q_len, const char* target, int t_len, int band_tolerance,
                      int get_aln_str, alignment_t* align, int* V, int* U, dpath2* d_path, path_point* aln_path, int right_extend)
{
    int k_offset, d, k, k2, best_m, min_k, new_min_k, max_k, new_max_k, pre_k;
    int x = -1, y = -1, max_d, band_size;
    unsigned long d_path_idx = 0, max_idx = 0;
    int aligned = 0;
    int best_x = -1, best_y = -1, best_d = q_len + t_len + 100, best_k = 0, best_d_path_idx = -1;

    max_d = (int)(.3 * (q_len + t_len));
    k_offset = max_d;
    band_size = band_tolerance * 2;
    align->aln_str_size = 0; align->aln_q_s = align->aln_q_e = 0; align->aln_t_s = align->aln_t_e = 0; /* Alignment::init */
    best_m = -1;
    min_k = 0;
    max_k = 0;
    for (d = 0; d < max_d; ++d) {
        if (max_k - min_k > band_size) break;
        for (k = min_k; k <= max_k; k += 2) {
            if (k == min_k || (k != max_k && V[k - 1 + k_offset] < V[k + 1 + k_offset])) { pre_k = k + 1; x = V[k + 1 + k_offset]; }
            else { pre_k = k - 1; x = V[k - 1 + k_offset] + 1; }
            y = x - k;
            d_path[d_path_idx].d = d; d_path[d_path_idx].k = k; d_path[d_path_idx].x1 = x; d_path[d_path_idx].y1 = y;
            int x0 = x;
            if (right_extend) while (x < q_len && y < t_len && query[x] == target[y]) { ++x; ++y; }
            else while (x < q_len && y < t_len && query[-x] == target[-y]) { ++x; ++y; }
            A->n_snake += x - x0; A->n_cells += 1;
            d_path[d_path_idx].x2 = x; d_path[d_path_idx].y2 = y; d_path[d_path_idx].pre_k = pre_k;
            ++d_path_idx;
            V[k + k_offset] = x;
            U[k + k_offset] = x + y;
            if (x + y > best_m) { best_m = x + y; best_x = x; best_y = y; best_d = d; best_k = k; best_d_path_idx = (int)d_path_idx; }
            if (x >= q_len || y >= t_len) { aligned = 1; max_idx = d_path_idx; break; }
        }
        new_min_k = max_k;
        new_max_k = min_k;
        for (k2 = min_k; k2 <= max_k; k2 += 2)
            if (U[k2 + k_offset] >= best_m - band_tolerance) {
                if (k2 < new_min_k) new_min_k = k2;
                if (k2 > new_max_k) new_max_k = k2;
            }
        max_k = new_max_k + 1;
        min_k = new_min_k - 1;
        if (aligned) {
            align->aln_q_e = x; align->aln_t_e = y; align->dist = d;
            align->aln_str_size = (x + y + d) / 2;
            align->aln_q_s = 0; align->aln_t_s = 0;
            if (get_aln_str) get_align_string(query, q_len, target, t_len, d_path, (int)max_idx, aln_path, align, d, k, right_extend);
            break;
        }
    }
    if (!aligned) {
        if (best_x > 0) {
            align->aln_q_e = best_x; align->aln_t_e = best_y; align->dist = best_d;
            align->aln_str_size = (best_x + best_y + best_d) / 2;
            align->aln_q_s = 0; align->aln_t_s = 0;
            if (get_aln_str) get_align_string(query, q_len, target, t_len, d_path, best_d_path_idx, aln_path, align, best_d, best_k, right_extend);
        } else {
            align->aln_q_e = 0; align->aln_t_e = 0; align->dist = 0; align->aln_str_size = 0;
            align->aln_q_s = 0; align->aln_t_s = 0;
        }
    }
    return (align->aln_q_e == q_len || align->aln_t_e == t_len);
}

int orc_align(orc_aligner* a, const char* q, int qlen, const char* t, int tlen, int band_tol, int get_aln,
              int right_extend, int* res, char* q_aln, char* t_aln)
{
    memset(a->dynq, 0, sizeof(int) * 4096);
    memset(a->dynt, 0, sizeof(int) * 4096);
    const char* qq = right_extend ? q : q + qlen - 1;
    const char* tt = right_extend ? t : t + tlen - 1;
    int r = align_core(a, qq, qlen, tt, tlen, band_tol, get_aln, &a->align, a->dynq, a->dynt, a->d_path, a->aln_path, right_extend);
    res[0] = a->align.aln_str_size; res[1] = a->align.dist; res[2] = a->align.aln_q_s; res[3] = a->align.aln_q_e;
    res[4] = a->align.aln_t_s; res[5] = a->align.aln_t_e;
    if (get_aln && q_aln) { memcpy(q_aln, a->align.q_aln_str, (size_t)a->align.aln_str_size); memcpy(t_aln, a->align.t_aln_str, (size_t)a->align.aln_str_size); }
    return r;
}

/* gapalign.cpp:9-45 */
static int retrieve_next_aln_block(const char* query, int qidx, int qsize, const char* target, int tidx, int tsize,
                                   int desired_block_size, int forward, const char** Q, const char** T, int* qblk, int* tblk)
{
    int last_block;
    int qleft = qsize - qidx, tleft = tsize - tidx;
    if (qleft < desired_block_size + 100 || tleft < desired_block_size + 100) {
        int a = (int)(tleft + tleft * 0.2), b = (int)(qleft + qleft * 0.2);
        *qblk = qleft < a ? qleft : a;
        *tblk = tleft < b ? tleft : b;
        last_block = 1;
    } else { *qblk = desired_block_size; *tblk = desired_block_size; last_block = 0; }
    if (forward) { *Q = query + qidx; *T = target + tidx; }
    else { *Q = query - qidx; *T = target - tidx; }
    return last_block;
}

/* gapalign.cpp:47-68 */
static int trim_mismatch_end(const char* qaln, const char* taln, int aln_size, int mat_cnt, int* qcnt, int* tcnt, int* aln_cnt)
{
    int m = 0, k;
    for (k = aln_size - 1, *qcnt = 0, *tcnt = 0, *aln_cnt = 0; k >= 0 && m < mat_cnt; --k) {
        ++*aln_cnt;
        if (qaln[k] != ORC_GAP_CODE) ++*qcnt;
        if (taln[k] != ORC_GAP_CODE) ++*tcnt;
        if (qaln[k] == taln[k]) ++m; else m = 0;
    }
    return m == mat_cnt && k > 0;
}

/* diff_gapalign.cpp:221-292 */
static void dw_in_one_direction(orc_aligner* A, const char* query, int query_size, const char* target, int target_size, int right_extend)
{
    const int kTailMatchBP = 4;
    const int kBlkSize = A->segment_size;
    int qidx = 0, tidx = 0, qblk, tblk;
    const char *seq1, *seq2;
    alignment_t* align = &A->align;
    while (1) {
        memset(A->dynq, 0, sizeof(int) * (size_t)A->row_size);
        memset(A->dynt, 0, sizeof(int) * (size_t)A->column_size);
        int last_block = retrieve_next_aln_block(query, qidx, query_size, target, tidx, target_size, kBlkSize, right_extend, &seq1, &seq2, &qblk, &tblk);
        int mx = qblk > tblk ? qblk : tblk;
        A->n_blocks += 1;
        align_core(A, seq1, qblk, seq2, tblk, (int)(0.3 * mx), 400, align, A->dynq, A->dynt, A->d_path, A->aln_path, right_extend);
        int qcnt = 0, tcnt = 0, acnt = 0;
        int trim = trim_mismatch_end(align->q_aln_str, align->t_aln_str, align->aln_str_size, kTailMatchBP, &qcnt, &tcnt, &acnt);
        if (!trim) break;
        int full_map = 0;
        if (qblk - align->aln_q_e <= 20 || tblk - align->aln_t_e <= 20) full_map = 1;
        if (last_block || (!full_map)) { qcnt -= kTailMatchBP; tcnt -= kTailMatchBP; acnt -= kTailMatchBP; }
        align->aln_str_size -= acnt;
        if (right_extend) {
            memcpy(A->right1 + A->right_size, align->q_aln_str, (size_t)align->aln_str_size);
            memcpy(A->right2 + A->right_size, align->t_aln_str, (size_t)align->aln_str_size);
            A->right_size += align->aln_str_size;
        } else {
            memcpy(A->left1 + A->left_size, align->q_aln_str, (size_t)align->aln_str_size);
            memcpy(A->left2 + A->left_size, align->t_aln_str, (size_t)align->aln_str_size);
            A->left_size += align->aln_str_size;
        }
        if (last_block || (!full_map)) break;
        qidx += (align->aln_q_e - qcnt);
        tidx += (align->aln_t_e - tcnt);
    }
}

/* diff_gapalign.cpp:294-349 ; identity counting = OutputStore::calc_ident diff_gapalign.h:90-97 */
int orc_dw_go(orc_aligner* A, const char* query, int qstart, int qsize, const char* target, int tstart, int tsize,
              int min_aln_size, orc_aln_result* out)
{
    A->left_size = A->right_size = 0;
    A->n_blocks = A->n_cells = A->n_snake = 0;
    A->align.aln_str_size = 0; A->align.aln_q_s = A->align.aln_q_e = 0; A->align.aln_t_s = A->align.aln_t_e = 0;
    dw_in_one_direction(A, query + qstart - 1, qstart, target + tstart - 1, tstart, 0);
    dw_in_one_direction(A, query + qstart, qsize - qstart, target + tstart, tsize - tstart, 1);
    int i = 0, j = 0, k, n = 0, idx = 0;
    for (k = A->left_size - 1; k >= 0; --k, ++idx) {
        if (A->left1[k] != ORC_GAP_CODE) ++i;
        if (A->left2[k] != ORC_GAP_CODE) ++j;
        if (A->left1[k] == A->left2[k]) ++n;
    }
    out->query_start = qstart - i;
    out->target_start = tstart - j;
    for (k = 0, i = 0, j = 0; k < A->right_size; ++k, ++idx) {
        if (A->right1[k] != ORC_GAP_CODE) ++i;
        if (A->right2[k] != ORC_GAP_CODE) ++j;
        if (A->right1[k] == A->right2[k]) ++n;
    }
    out->query_end = qstart + i;
    out->target_end = tstart + j;
    out->matches = n;
    out->columns = idx;
    out->ok = idx >= min_aln_size;
    return out->ok;
}

/* ------------------------------------------------------------------ A13: X-drop aligner (xdrop_gapalign.cpp) */

#define X_MIN_SCORE (-100000000)
enum { X_SUB = 3, X_GAP_IN_A = 0, X_GAP_IN_B = 6, X_OP_MASK = 0x07, X_EXT_A = 0x10, X_EXT_B = 0x40 };   /* xdrop_gapalign.h:13-34 */
typedef struct { int best, best_gap; } gapdp;                 /* BlastGapDP, xdrop_gapalign.h:37-40 */

struct orc_xaligner {                                         /* xdrop_gapalign.h:117-212, parameters :85-115 */
    int reward, penalty, gap_open, gap_extend, x_dropoff, block_size;
    uint8_t* state_array;
    gapdp* score_array;
    uint8_t** edit_script;
    int* edit_start_offset;
    int* op_type; int* op_num; int num_ops, last_op;           /* GapPrelimEditBlock */
    int matrix[4][4];
    char *lq, *lt, *rq, *rt, *tq, *tt;                         /* left/right/tmp aligned strings (codes 0..4) */
    int lsize, rsize;
};

orc_xaligner* orc_xaligner_new(void)
{
    orc_xaligner* a = (orc_xaligner*)xcalloc(sizeof(*a));
    a->reward = 1; a->penalty = -1; a->gap_open = 0; a->gap_extend = 1; a->x_dropoff = 30; a->block_size = 500;
    a->state_array = (uint8_t*)xmalloc(5000000);
    a->score_array = (gapdp*)xmalloc(sizeof(gapdp) * 4096);
    a->edit_script = (uint8_t**)xmalloc(sizeof(uint8_t*) * 4096);
    a->edit_start_offset = (int*)xmalloc(sizeof(int) * 4096);
    a->op_type = (int*)xmalloc(sizeof(int) * ORC_MAX_SEQ_SIZE);
    a->op_num = (int*)xmalloc(sizeof(int) * ORC_MAX_SEQ_SIZE);
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) a->matrix[i][j] = i == j ? a->reward : a->penalty;
    a->lq = (char*)xmalloc(ORC_MAX_SEQ_SIZE); a->lt = (char*)xmalloc(ORC_MAX_SEQ_SIZE);
    a->rq = (char*)xmalloc(ORC_MAX_SEQ_SIZE); a->rt = (char*)xmalloc(ORC_MAX_SEQ_SIZE);
    a->tq = (char*)xmalloc(8192); a->tt = (char*)xmalloc(8192);
    return a;
}

void orc_xaligner_free(orc_xaligner* a)
{
    if (!a) return;
    free(a->state_array); free(a->score_array); free(a->edit_script); free(a->edit_start_offset); free(a->op_type);
    free(a->op_num); free(a->lq); free(a->lt); free(a->rq); free(a->rt); free(a->tq); free(a->tt); free(a);
}

static void xblock_add(orc_xaligner* a, int op, int n)        /* GapPrelimEditBlock::add, xdrop_gapalign.h:70-81 */
{
    if (n == 0) return;
    if (a->last_op == op) a->op_num[a->num_ops - 1] += n;
    else { a->last_op = op; a->op_type[a->num_ops] = op; a->op_num[a->num_ops] = n; ++a->num_ops; }
}

/* xdrop_gapalign.cpp:10-213 */
static int xdrop_align(orc_xaligner* X, const char* A, int M, const char* B, int N, int forward, int* ae, int* be)
{
    const int gap_open = X->gap_open, gap_extend = X->gap_extend;
    int x_dropoff = X->x_dropoff;
    uint8_t* state_array = X->state_array;
    gapdp* score_array = X->score_array;
    uint8_t** edit_script = X->edit_script;
    int* edit_start_offset = X->edit_start_offset;
    *ae = 0; *be = 0;
    X->last_op = 8; X->num_ops = 0;                            /* edit_block->clear() */
    if (M <= 0 || N <= 0) return 0;
    int i, a_index, b_index, b_size, first_b_index, last_b_index;
    const int gap_open_extend = gap_open + gap_extend;
    int best_score, score, score_gap_row, score_gap_col, next_score, orig_b_index, states_used = 0;
    const int* matrix_row;
    uint8_t* edit_script_row;
    uint8_t script, next_script, script_row, script_col;
    if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
    edit_script[0] = state_array;
    edit_start_offset[0] = 0;
    edit_script_row = state_array;
    score = -gap_open_extend;
    score_array[0].best = 0;
    score_array[0].best_gap = -gap_open_extend;
    for (i = 1; i <= N; ++i) {
        if (score < -x_dropoff) break;
        score_array[i].best = score;
        score_array[i].best_gap = score - gap_open_extend;
        score -= gap_extend;
        edit_script_row[i] = X_GAP_IN_A;
    }
    states_used = N < i + 1 ? N : i + 1;
    b_size = i;
    best_score = 0;
    first_b_index = 0;
    for (a_index = 1; a_index <= M; ++a_index) {
        const uint8_t AC = (uint8_t)extract_char(A, a_index - 1, forward);
        edit_script[a_index] = state_array + states_used + 1;
        edit_start_offset[a_index] = first_b_index;
        edit_script_row = edit_script[a_index] - first_b_index;
        orig_b_index = first_b_index;
        matrix_row = X->matrix[AC];
        score = X_MIN_SCORE;
        score_gap_row = X_MIN_SCORE;
        last_b_index = first_b_index;
        for (b_index = first_b_index; b_index < b_size; ++b_index) {
            const uint8_t BC = (uint8_t)extract_char(B, b_index, forward);
            score_gap_col = score_array[b_index].best_gap;
            next_score = score_array[b_index].best + matrix_row[BC];
            script = X_SUB;
            script_row = X_EXT_B;
            script_col = X_EXT_A;
            if (score < score_gap_col) { script = X_GAP_IN_B; score = score_gap_col; }
            if (score < score_gap_row) { script = X_GAP_IN_A; score = score_gap_row; }
            if (best_score - score > x_dropoff) {
                if (first_b_index == b_index) ++first_b_index;
                else score_array[b_index].best = X_MIN_SCORE;
            } else {
                last_b_index = b_index;
                if (score > best_score) { best_score = score; *ae = a_index; *be = b_index; }
                score_gap_col -= gap_extend;
                if (score_gap_col < (score - gap_open_extend)) score_array[b_index].best_gap = score - gap_open_extend;
                else { score_array[b_index].best_gap = score_gap_col; script += script_col; }
                score_gap_row -= gap_extend;
                if (score_gap_row < (score - gap_open_extend)) score_gap_row = score - gap_open_extend;
                else script += script_row;
                score_array[b_index].best = score;
            }
            score = next_score;
            edit_script_row[b_index] = script;
        }
        if (first_b_index == b_size) break;
        if (last_b_index < b_size - 1) b_size = last_b_index + 1;
        else {
            while (score_gap_row >= (best_score - x_dropoff) && b_size < N) {
                score_array[b_size].best = score_gap_row;
                score_array[b_size].best_gap = score_gap_row - gap_open_extend;
                score_gap_row -= gap_extend;
                edit_script_row[b_size] = X_GAP_IN_A;
                ++b_size;
            }
        }
        states_used += (b_index > b_size ? b_index : b_size) - orig_b_index + 1;
        if (b_size < N) {
            score_array[b_size].best = X_MIN_SCORE;
            score_array[b_size].best_gap = X_MIN_SCORE;
            ++b_size;
        }
    }
    a_index = *ae;
    b_index = *be;
    script = X_SUB;
    while (a_index > 0 || b_index > 0) {
        next_script = edit_script[a_index][b_index - edit_start_offset[a_index]];
        switch (script) {
        case X_GAP_IN_A:
            script = next_script & X_OP_MASK;
            if (next_script & X_EXT_A) script = X_GAP_IN_A;
            break;
        case X_GAP_IN_B:
            script = next_script & X_OP_MASK;
            if (next_script & X_EXT_B) script = X_GAP_IN_B;
            break;
        default:
            script = next_script & X_OP_MASK;
            break;
        }
        if (script == X_GAP_IN_A) --b_index;
        else if (script == X_GAP_IN_B) --a_index;
        else { --a_index; --b_index; }
        xblock_add(X, script, 1);
    }
    return best_score;
}

int orc_xdrop_align(orc_xaligner* a, const char* A, int M, const char* B, int N, int forward, int* res, int* ops)
{
    const char* AA = forward ? A : A + M - 1;
    const char* BB = forward ? B : B + N - 1;
    int ae, be;
    int sc = xdrop_align(a, AA, M, BB, N, forward, &ae, &be);
    res[0] = ae; res[1] = be; res[2] = a->num_ops;
    for (int i = 0; i < a->num_ops; ++i) { ops[2 * i] = a->op_type[i]; ops[2 * i + 1] = a->op_num[i]; }
    return sc;
}

/* script_to_aligned_string, xdrop_gapalign.cpp:215-261 ; returns the string length */
static int script_to_aligned_string(orc_xaligner* X, const char* query, const char* target, int forward, char* qaln, char* taln)
{
    const char *q = query, *t = target;
    const int inc = forward ? 1 : -1;
    int n = 0;
    for (int i = X->num_ops - 1; i >= 0; --i) {
        const int op = X->op_type[i], cnt = X->op_num[i];
        for (int j = 0; j < cnt; ++j) {
            if (op == X_SUB) { qaln[n] = *q; taln[n] = *t; q += inc; t += inc; }
            else if (op == X_GAP_IN_A) { qaln[n] = ORC_GAP_CODE; taln[n] = *t; t += inc; }
            else { qaln[n] = *q; taln[n] = ORC_GAP_CODE; q += inc; }
            ++n;
        }
    }
    return n;
}

/* align_ex, xdrop_gapalign.cpp:263-357 ; appends to (qaln, taln), returns the new length */
static int align_ex(orc_xaligner* X, const char* query, int query_size, const char* target, int target_size, int forward,
                    char* qaln, char* taln)
{
    const int kTailMatchBP = 4;
    int qidx = 0, tidx = 0, qblk, tblk, aln_qe, aln_te, len = 0;
    const char *Q, *T;
    while (1) {
        int last_block = retrieve_next_aln_block(query, qidx, query_size, target, tidx, target_size, X->block_size, forward, &Q, &T, &qblk, &tblk);
        xdrop_align(X, Q, qblk, T, tblk, forward, &aln_qe, &aln_te);
        int n = script_to_aligned_string(X, Q, T, forward, X->tq, X->tt);
        int full_map = 0;
        if (qblk - aln_qe <= 20 || tblk - aln_te <= 20) full_map = 1;
        if ((!full_map) || last_block) {
            memcpy(qaln + len, X->tq, (size_t)n); memcpy(taln + len, X->tt, (size_t)n);
            len += n;
            break;
        }
        int qcnt = 0, tcnt = 0, acnt = 0;
        int trim = trim_mismatch_end(X->tq, X->tt, n, kTailMatchBP, &qcnt, &tcnt, &acnt);
        if (!trim) break;
        n -= acnt;
        memcpy(qaln + len, X->tq, (size_t)n); memcpy(taln + len, X->tt, (size_t)n);
        len += n;
        qidx += (aln_qe - qcnt);
        tidx += (aln_te - tcnt);
    }
    return len;
}

/* XdropAligner::go, xdrop_gapalign.cpp:359-439 (the left half skips its last column: `n = size() - 1; k = n - 1`) */
int orc_xdrop_go(orc_xaligner* X, const char* query, int qstart, int qsize, const char* target, int tstart, int tsize,
                 int min_aln_size, orc_aln_result* out)
{
    X->lsize = align_ex(X, query + qstart - 1, qstart, target + tstart - 1, tstart, 0, X->lq, X->lt);
    X->rsize = align_ex(X, query + qstart, qsize - qstart, target + tstart, tsize - tstart, 1, X->rq, X->rt);
    int i = 0, j = 0, k, n, idx = 0, ident = 0;
    n = X->lsize - 1;
    for (k = n - 1; k >= 0; --k, ++idx) {
        if (X->lq[k] != ORC_GAP_CODE) ++i;
        if (X->lt[k] != ORC_GAP_CODE) ++j;
        if (X->lq[k] == X->lt[k]) ++ident;
    }
    out->query_start = qstart - i;
    out->target_start = tstart - j;
    n = X->rsize;
    for (i = 0, j = 0, k = 0; k < n; ++k, ++idx) {
        if (X->rq[k] != ORC_GAP_CODE) ++i;
        if (X->rt[k] != ORC_GAP_CODE) ++j;
        if (X->rq[k] == X->rt[k]) ++ident;
    }
    out->query_end = qstart + i;
    out->target_end = tstart + j;
    out->matches = ident;
    out->columns = idx;
    out->ok = out->query_end - out->query_start >= min_aln_size;
    return out->ok;
}

/* ------------------------------------------------------------------ A14: m4 */

/* pw_impl.cpp:467-506 ; ident = 100.0 * n / out_store_size, 0.0 when empty (diff_gapalign.h:90-97) */
void orc_m4_fill(const orc_aln_result* r, int qid, int sid, char qchain, int qsize, int ssize,
                 int qstart, int sstart, int vscore, orc_m4* m)
{
    m->qid = sid;
    m->sid = qid;
    m->ident = r->columns == 0 ? 0.0 : 100.0 * r->matches / r->columns;
    m->vscore = vscore;
    m->qdir = 0;
    m->qoff = r->target_start;
    m->qend = r->target_end;
    m->qsize = ssize;
    m->ssize = qsize;
    m->qext = sstart;
    if (qchain == 'F') {
        m->sdir = 0;
        m->soff = r->query_start;
        m->send = r->query_end;
        m->sext = qstart;
    } else {
        m->sdir = 1;
        m->soff = qsize - r->query_end;
        m->send = qsize - r->query_start;
        m->sext = qsize - 1 - qstart;
    }
}

void orc_m4_std_sort(orc_m4* list, int n);  /* m4sort.cpp: std::sort with CmpM4RecordByQidAndOvlpSize (pw_impl.cpp:539-548, 581) */

/* check_records_containment pw_impl.cpp:550-574 */
static void check_records_containment(const orc_m4* m4v, int s, int e, int* valid)
{
    const int soft = 100;
    for (int i = s; i < e; ++i) {
        if (!valid[i]) continue;
        int qb1 = (int)m4v[i].qoff, qe1 = (int)m4v[i].qend, sb1 = (int)m4v[i].soff, se1 = (int)m4v[i].send;
        for (int j = i + 1; j < e; ++j) {
            if (!valid[j]) continue;
            if (m4v[i].sdir != m4v[j].sdir) continue;
            int qb2 = (int)m4v[j].qoff, qe2 = (int)m4v[j].qend, sb2 = (int)m4v[j].soff, se2 = (int)m4v[j].send;
            if (qb2 + soft >= qb1 && qe2 - soft <= qe1 && sb2 + soft >= sb1 && se2 - soft <= se1) valid[j] = 0;
        }
    }
}

/* append_m4v pw_impl.cpp:576-610 (without the buffered print) */
int orc_m4_postfilter(orc_m4* llist, int n, orc_m4* out)
{
    orc_m4_std_sort(llist, n);
    int* valid = (int*)xmalloc(sizeof(int) * (size_t)(n + 1));
    for (int i = 0; i < n; ++i) valid[i] = 1;
    int i = 0, j;
    while (i < n) {
        int64_t qid = llist[i].qid;
        j = i + 1;
        while (j < n && llist[j].qid == qid) ++j;
        if (j - i > 1) check_records_containment(llist, i, j, valid);
        i = j;
    }
    int m = 0;
    for (i = 0; i < n; ++i) if (valid[i]) out[m++] = llist[i];
    free(valid);
    return m;
}

/* output_m4record pw_impl.cpp:509-531 ; ostream<<double default == "%g" with precision 6 */
int orc_m4_line(const orc_m4* m, int gapped, char* buf)
{
    int n = sprintf(buf, "%lld\t%lld\t%g\t%d\t%d\t%lld\t%lld\t%lld\t%d\t%lld\t%lld\t%lld", (long long)m->qid, (long long)m->sid,
                    m->ident, m->vscore, m->qdir, (long long)m->qoff, (long long)m->qend, (long long)m->qsize, m->sdir,
                    (long long)m->soff, (long long)m->send, (long long)m->ssize);
    if (gapped) n += sprintf(buf + n, "\t%lld\t%lld", (long long)m->qext, (long long)m->sext);
    buf[n++] = '\n'; buf[n] = 0;
    return n;
}

int orc_map_read(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                 orc_aligner* al, int rid, const orc_params* p, orc_m4* out)
{
    return orc_map_read_x(ref, reads, ridx, bk, al, NULL, rid, p, out);
}

/* pairwise_mapping pw_impl.cpp:651-700 for one read; aligner by tech (:638-644) */
int orc_map_read_x(const orc_volume* ref, const orc_volume* reads, const orc_index* ridx, orc_seeding_bk* bk,
                   orc_aligner* al, orc_xaligner* xal, int rid, const orc_params* p, orc_m4* out)
{
    orc_candidate* cands = (orc_candidate*)xmalloc(sizeof(orc_candidate) * (size_t)p->maxc);
    orc_m4* m4v = (orc_m4*)xmalloc(sizeof(orc_m4) * (size_t)p->maxc);
    int rsize = reads->offs[rid].size;
    char* read1 = (char*)xmalloc((size_t)rsize + 1);
    char* read2 = (char*)xmalloc((size_t)rsize + 1);
    char* subject = (char*)xmalloc(ORC_MAX_SEQ_SIZE);
    orc_extract_one_seq(reads, rid, read1);
    orc_reverse_complement(read2, read1, rsize);
    int nc = orc_seed_read(ref, reads, ridx, bk, rid, 1, p, cands);
    int num_m4 = 0;
    for (int s = 0; s < nc; ++s) {
        const char* read = cands[s].chain == 'F' ? read1 : read2;
        int lid = cands[s].readno - ref->start_read_id;
        orc_extract_one_seq(ref, lid, subject);
        int sstart = cands[s].loc1, qstart = cands[s].loc2;
        if (qstart && sstart) { qstart += ORC_KMER / 2; sstart += ORC_KMER / 2; }
        int ssize = ref->offs[lid].size;
        orc_aln_result r;
        int okk = (p->tech == 1 && xal) ? orc_xdrop_go(xal, read, qstart, rsize, subject, sstart, ssize, p->min_align_size, &r)
                                       : orc_dw_go(al, read, qstart, rsize, subject, sstart, ssize, p->min_align_size, &r);
        if (okk) {
            orc_m4_fill(&r, rid + reads->start_read_id, cands[s].readno, cands[s].chain, rsize, ssize, qstart, sstart,
                        cands[s].score, m4v + num_m4);
            ++num_m4;
        }
    }
    int kept = orc_m4_postfilter(m4v, num_m4, out);
    free(cands); free(m4v); free(read1); free(read2); free(subject);
    return kept;
}


/* ================================================================================================================
 * N1: the mecat2cns re-aligner (reference src/mecat2cns/dw.cpp).  Same O(ND) core as A9-A12 with different rules:
 * max_d from an error rate, square blocks, no partial-best fallback, full alignment strings, the 4-match anchor of a
 * block is re-aligned by the next block, extra end trimming in GetAlignment.
 * ================================================================================================================ */
#define CNS_SEG 500            /* get_sw_parameters_small, dw.cpp:8-22 */
#define CNS_VSIZE 4096         /* row_size / column_size */
#define CNS_MAX_ALN 100000     /* max_aln_size */

struct orc_cns {
    int* V;                 /* furthest x per diagonal (index k + k_offset) */
    int* U;                 /* x + y per diagonal */
    int* row_first;         /* per d: index of the row's first cell in cell_x2 / cell_k */
    int* cell_k;            /* per computed cell: its diagonal */
    int* cell_x1;           /* snake start */
    int* cell_x2;           /* snake end */
    int* cell_pre;          /* diagonal the cell was entered from */
    int cap_cells, cap_rows;
    uint8_t *lops, *rops, *bops;   /* left / right direction ops, one block's ops */
};

orc_cns* orc_cns_new(void) {
    orc_cns* a = (orc_cns*)xcalloc(sizeof(orc_cns));
    a->V = (int*)xmalloc(sizeof(int) * CNS_VSIZE);
    a->U = (int*)xmalloc(sizeof(int) * CNS_VSIZE);
    a->cap_rows = 2048;
    a->cap_cells = 1 << 20;
    a->row_first = (int*)xmalloc(sizeof(int) * (size_t)(a->cap_rows + 1));
    a->cell_k = (int*)xmalloc(sizeof(int) * (size_t)a->cap_cells);
    a->cell_x1 = (int*)xmalloc(sizeof(int) * (size_t)a->cap_cells);
    a->cell_x2 = (int*)xmalloc(sizeof(int) * (size_t)a->cap_cells);
    a->cell_pre = (int*)xmalloc(sizeof(int) * (size_t)a->cap_cells);
    a->lops = (uint8_t*)xmalloc(CNS_MAX_ALN);
    a->rops = (uint8_t*)xmalloc(CNS_MAX_ALN);
    a->bops = (uint8_t*)xmalloc(8192);
    return a;
}

void orc_cns_free(orc_cns* a) {
    if (!a) return;
    free(a->V); free(a->U); free(a->row_first); free(a->cell_k); free(a->cell_x1); free(a->cell_x2); free(a->cell_pre);
    free(a->lops); free(a->rops); free(a->bops);
    free(a);
}

int orc_cns_align_block(orc_cns* a, const char* q, const char* t, int len, int right, double error_rate,
                        int* qe, int* te, int* dist, uint8_t* ops, int* ncols) {
    const int q_len = len, t_len = len;
    const int band_tol = (int)(0.3 * len);                       /* the call site passes 0.3 * seg_size to an int, dw.cpp:332 */
    const int max_d = (int)(2.0 * error_rate * (q_len + t_len));  /* :163 */
    const int koff = max_d, band_size = band_tol * 2;
    int best_m = -1, min_k = 0, max_k = 0, ncell = 0;
    *qe = *te = *dist = 0; *ncols = 0;
    memset(a->U, 0, sizeof(int) * CNS_VSIZE);                     /* the caller zeroes both per block, :329-330 */
    memset(a->V, 0, sizeof(int) * CNS_VSIZE);
    for (int d = 0; d < max_d; ++d) {
        if (max_k - min_k > band_size) break;
        if (d >= a->cap_rows) { fprintf(stderr, "oracle: cns row capacity\n"); abort(); }
        a->row_first[d] = ncell;
        int hit = 0, hit_k = 0, hx = 0, hy = 0;
        for (int k = min_k; k <= max_k; k += 2) {
            int pre_k, x;
            if (k == min_k || (k != max_k && a->V[k - 1 + koff] < a->V[k + 1 + koff])) { pre_k = k + 1; x = a->V[k + 1 + koff]; }
            else { pre_k = k - 1; x = a->V[k - 1 + koff] + 1; }
            int y = x - k;
            const int x1 = x;
            if (right) while (x < q_len && y < t_len && q[x] == t[y]) { ++x; ++y; }
            else while (x < q_len && y < t_len && q[-x] == t[-y]) { ++x; ++y; }
            if (ncell >= a->cap_cells) { fprintf(stderr, "oracle: cns cell capacity\n"); abort(); }
            a->cell_k[ncell] = k; a->cell_x1[ncell] = x1; a->cell_x2[ncell] = x; a->cell_pre[ncell] = pre_k;
            ++ncell;
            a->V[k + koff] = x;
            a->U[k + koff] = x + y;
            if (x + y > best_m) best_m = x + y;
            if (x >= q_len || y >= t_len) { hit = 1; hit_k = k; hx = x; hy = y; break; }   /* first diagonal in k order */
        }
        a->row_first[d + 1] = ncell;
        /* band for the next row (:202-209); after a hit the diagonals above hit_k still hold the values of row d - 2 */
        int nmin = max_k, nmax = min_k;
        for (int k2 = min_k; k2 <= max_k; k2 += 2)
            if (a->U[k2 + koff] >= best_m - band_tol) { if (k2 < nmin) nmin = k2; if (k2 > nmax) nmax = k2; }
        max_k = nmax + 1;
        min_k = nmin - 1;
        if (hit) {
            /* walk the path back: cell (cd, ck) -> its predecessor diagonal in row cd - 1 (:222-240) */
            static int px1[4100], px2[4100], pk[4100];
            int ck = hit_k;
            for (int cd = d; cd >= 0; --cd) {
                int lo = a->row_first[cd], hi = a->row_first[cd + 1], at = -1;
                for (int c = lo; c < hi; ++c) if (a->cell_k[c] == ck) { at = c; break; }
                if (at < 0) { fprintf(stderr, "oracle: cns path lost\n"); abort(); }
                px1[cd] = a->cell_x1[at]; px2[cd] = a->cell_x2[at]; pk[cd] = ck;
                ck = a->cell_pre[at];
            }
            /* forward: one indel per row after the first, then the row's snake (:241-297) */
            int n = 0;
            for (int cd = 0; cd <= d; ++cd) {
                if (cd > 0) ops[n++] = (uint8_t)(pk[cd] < pk[cd - 1] ? 1 : 2);   /* from k + 1: target base only; from k - 1: query base only */
                for (int i = px1[cd]; i < px2[cd]; ++i) ops[n++] = 0;
            }
            *ncols = n; *qe = hx; *te = hy; *dist = d;
            return (hx == q_len || hy == t_len) ? 1 : 0;
        }
    }
    return (0 == q_len || 0 == t_len) ? 1 : 0;       /* Alignment::init leaves aln_q_e = aln_t_e = 0 (:302-303) */
}

int orc_cns_one_direction(orc_cns* a, const char* q, int qsize, const char* t, int tsize, int right, double error_rate,
                          uint8_t* ops, int* qbases, int* tbases) {
    int extend1 = 0, extend2 = 0, total = 0;
    int extend_size = qsize < tsize ? qsize : tsize;
    int more = 1;
    while (more) {
        int seg;
        if (extend_size > CNS_SEG + 100) seg = CNS_SEG;
        else { seg = extend_size; more = 0; }
        const char* s1 = right ? q + extend1 : q - extend1;
        const char* s2 = right ? t + extend2 : t - extend2;
        int qe, te, dist, n;
        int ok = orc_cns_align_block(a, s1, s2, seg, right, error_rate, &qe, &te, &dist, a->bops, &n);
        if (!ok) break;
        /* bases and columns of the tail through the last run of four matches (:336-343) */
        int k, i = 0, j = 0, nm = 0;
        for (k = n - 1; k > -1 && nm < 4; --k) {
            if (a->bops[k] != 1) ++i;
            if (a->bops[k] != 2) ++j;
            if (a->bops[k] == 0) ++nm; else nm = 0;
        }
        if (more) {
            i = CNS_SEG - qe + i;
            j = CNS_SEG - te + j;
            if (i == CNS_SEG) ok = 0;
            extend1 += CNS_SEG - i; extend2 += CNS_SEG - j;
        } else {
            i = extend_size - qe;
            j = extend_size - te;
            if (i == extend_size) ok = 0;
            extend1 += extend_size - i; extend2 += extend_size - j;
            k = n - 1;
        }
        if (!ok) break;
        memcpy(ops + total, a->bops, (size_t)(k + 1));
        total += k + 1;
        extend_size = (qsize - extend1) < (tsize - extend2) ? (qsize - extend1) : (tsize - extend2);
    }
    int qb = 0, tb = 0;
    for (int c = 0; c < total; ++c) { if (ops[c] != 1) ++qb; if (ops[c] != 2) ++tb; }
    *qbases = qb; *tbases = tb;
    return total;
}

int orc_cns_dw(orc_cns* a, const char* q, int qstart, int qsize, const char* t, int tstart, int tsize, double error_rate,
               int min_aln_size, int* res, char* out1, char* out2) {
    static const char dec[] = "ACGT-";
    int lq, lt, rq, rt;
    const int nl = orc_cns_one_direction(a, q + qstart - 1, qstart, t + tstart - 1, tstart, 0, error_rate, a->lops, &lq, &lt);
    const int nr = orc_cns_one_direction(a, q + qstart, qsize - qstart, t + tstart, tsize - tstart, 1, error_rate, a->rops, &rq, &rt);
    const int n = nl + nr;
    res[0] = qstart - lq; res[1] = qstart + rq; res[2] = tstart - lt; res[3] = tstart + rt; res[4] = n;
    res[5] = res[6] = res[7] = res[8] = 0;
    /* merged strings: the left part reversed, then the right part (:397-437) */
    int qi = res[0], ti = res[2], mat = 0, ins = 0, del = 0;
    for (int c = 0; c < n; ++c) {
        const int op = c < nl ? a->lops[nl - 1 - c] : a->rops[c - nl];
        const char c1 = op == 1 ? '-' : dec[(int)q[qi]], c2 = op == 2 ? '-' : dec[(int)t[ti]];
        if (op != 1) ++qi;
        if (op != 2) ++ti;
        if (out1) out1[c] = c1;
        if (out2) out2[c] = c2;
        if (op == 0) ++mat; else if (op == 1) ++ins; else ++del;
    }
    if (out1) out1[n] = 0;
    if (out2) out2[n] = 0;
    if (n >= min_aln_size) { res[5] = mat; res[7] = ins; res[8] = del; return 1; }
    return 0;
}

int orc_cns_get_alignment(orc_cns* a, const char* q, int qstart, int qsize, const char* t, int tstart, int tsize,
                          double error_rate, int min_aln_size, int* res, char* qaln, char* saln) {
    static char s1[CNS_MAX_ALN + 1], s2[CNS_MAX_ALN + 1];
    int r[9];
    res[0] = res[1] = res[2] = res[3] = res[4] = 0;
    if (!orc_cns_dw(a, q, qstart, qsize, t, tstart, tsize, error_rate, min_aln_size, r, s1, s2)) return 0;
    const int n = r[4], run = 4;
    int qrb = 0, trb = 0, qre = 0, tre = 0, eit = 0, k;
    for (k = 0; k < n && eit < run; ++k) {           /* first run of four matches (:495-505) */
        if (s1[k] != '-') ++qrb;
        if (s2[k] != '-') ++trb;
        if (s1[k] == s2[k]) ++eit; else eit = 0;
    }
    if (eit < run) return 0;
    k -= run; qrb -= run; trb -= run;
    const int first = k;
    for (k = n - 1, eit = 0; k >= 0 && eit < run; --k) {   /* last run of four matches (:513-523) */
        if (s1[k] != '-') ++qre;
        if (s2[k] != '-') ++tre;
        if (s1[k] == s2[k]) ++eit; else eit = 0;
    }
    if (eit < run) return 0;
    k += run; qre -= run; tre -= run;
    const int last = k + 1;
    res[0] = r[0] + qrb; res[1] = r[1] - qre; res[2] = r[2] + trb; res[3] = r[3] - tre; res[4] = last - first;
    if (qaln) { memcpy(qaln, s1 + first, (size_t)(last - first)); qaln[last - first] = 0; }
    if (saln) { memcpy(saln, s2 + first, (size_t)(last - first)); saln[last - first] = 0; }
    return 1;
}
