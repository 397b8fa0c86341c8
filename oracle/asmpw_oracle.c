/*
 * asmpw_oracle.c — CPU restatement of the seeding + candidate stage of mecat2canu's mecat2asmpw.
 * TEST INFRASTRUCTURE ONLY: built into oracle/liboracle.so; the product path never links it.
 *
 * Restates, for one query read against one indexed block of reads, what pairwise_mapping does before its extension loop
 * (/root/reference/mecat2canu/src/mecat2asmpw/mecat2asmpw.c:580-718; all line numbers below are of that file):
 *   creat_ref_index :422-512 + sumvalue_x :307-314    13-mer table of the block text, buckets of more than 256 occurrences emptied,
 *                                                     positions 1-BASED (:503: i + 2 - seed_len)
 *   transnum_buchang :316-335                         query 13-mers at stride 10
 *   the seeding loop :601-641                         segments of ZV = 1000 text positions, SM = 60 stored seeds, the overflow
 *                                                     replacement (insert_loc) is commented out there (:621): seeds past 60 only count
 *   find_location :355-386                            = mecat2pw's with a 0.10 cutoff (orc_find_location, pinned to pw_impl.cpp)
 *   the candidate loop :643-716                       gates `> 10` (:644) and `< 6` (:662), subject read by binary() :283-297,
 *                                                     self-hit scrub, geometry, the two sweeps, top-MAXC insertion
 * Parity status: PINNED through oracle/_ref/libref_asmpw_cand.so (ref_harness_asmpw_cand.c: the unmodified file with a recording
 * `align`), tests/test_asmpw_ref_cpu.py::test_candidate_stage_equals_reference: every candidate of every read of the golden set
 * (subject position, query position, strand, num1, num2, list order) for both start blocks.
 *
 * What is literal on purpose:
 *   - struct Back_List has the reference's layout (:65-68).  The sweeps (:689-703) read loczhi[j] / seedno[j] for j < score although
 *     only 60 entries exist: past 60 they read the seedno[] array, seednum, the index word and the following segments.  The segment
 *     array is one allocation here too, and the sweeps index it through the same short pointers.
 *   - llocation[] has one entry per read plus one that load_read (:388-409) never writes; the tool reads it as the end of the last
 *     read of the block (:640).  A fresh heap gives 0; the block below stores 0 there.
 *   - positions are 1-based, read starts (llocation) 0-based: left_length1 (:673) is one larger than the bases it names.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "mecat_oracle.h"

#define AZV 1000        /* :17 */
#define ABC 10          /* :19 */
#define ASM 60          /* :20 */
#define AMAXC 100       /* :22 */
#define AK 13           /* seed_len, :1084 */
#define ACAP 256        /* sumvalue_x :310-311 */

typedef struct {        /* struct Back_List, :65-68 (248 bytes) */
    short score, loczhi[ASM], seedno[ASM], seednum;
    int index;
} asm_seg;

typedef struct {        /* canidate_save, :55-58 */
    int loc1, loc2, left1, left2, right1, right2, score, num1, num2, readno, readstart;
    char chain;
} asm_candidate;

typedef struct {
    const char* text;   /* the block: reads in upper case, one NUL after each (load_read :388-409) */
    int n;              /* bytes of text (seqcount) */
    int nreads;
    int* lloc;          /* [nreads + 1] read starts; [nreads] = 0 (never written by the tool) */
    int* readno;        /* [nreads] read numbers (indexread[].readno) */
    int* counts;        /* [4^13] kept occurrences */
    int64_t* first;     /* [4^13] offset into pos[], -1 when empty */
    int* pos;           /* 1-based k-mer start positions, ascending inside a bucket */
    asm_seg* db;        /* [n / ZV + 5] (+ slack behind for the sweeps' reads past the last segment) */
    int* index_list;
    short* index_score;
    int fresh;          /* asm_block_fresh(): every read starts from an all-zero segment array (see there) */
    int* dirty;         /* segments touched by the current read */
    int ndirty;
} asm_block;

static int base_code(char c) {      /* atcttrans :299-305 (the digit order only names the buckets: any bijection gives the same buckets) */
    switch (c) {
    case 'A': case 'a': return 0;
    case 'T': return 1;             /* (the reference tests 'T' twice: a lower-case t is "other"; reads are upper-cased on load) */
    case 'C': case 'c': return 2;
    case 'G': case 'g': return 3;
    default: return 4;
    }
}

void asm_block_free(asm_block* B) {
    if (!B) return;
    free(B->lloc); free(B->readno); free(B->counts); free(B->first); free(B->pos); free(B->db); free(B->index_list); free(B->index_score);
    free(B->dirty);
    free(B);
}

asm_block* asm_block_new(const char* text, int n, const int* starts, int nreads, int first_readno) {
    asm_block* B = (asm_block*)calloc(1, sizeof(asm_block));
    const int64_t nk = (int64_t)1 << (2 * AK);
    int64_t i, sum;
    unsigned int eit;
    int run, pass;
    B->text = text; B->n = n; B->nreads = nreads;
    B->lloc = (int*)calloc((size_t)nreads + 1, sizeof(int));
    B->readno = (int*)calloc((size_t)nreads + 1, sizeof(int));
    for (i = 0; i < nreads; ++i) { B->lloc[i] = starts[i]; B->readno[i] = first_readno + (int)i; }
    B->counts = (int*)calloc((size_t)nk, sizeof(int));
    B->first = (int64_t*)malloc((size_t)nk * sizeof(int64_t));
    /* creat_ref_index: a rolling word that restarts at every character that is not A/C/G/T; the k-mer that ends at text index i
       starts at i + 1 - k and is stored as i + 2 - k */
    for (pass = 0; pass < 2; ++pass) {
        eit = 0; run = 0;
        for (i = 0; i < n; ++i) {
            const int t = base_code(text[i]);
            if (text[i] == 'N' || t == 4) { eit = 0; run = 0; continue; }
            eit = ((eit << 2) | (unsigned)t) & (unsigned)(nk - 1);
            if (++run >= AK) {
                if (pass == 0) B->counts[eit]++;
                else if (B->first[eit] >= 0) B->pos[B->first[eit] + B->counts[eit]++] = (int)(i + 2 - AK);
            }
        }
        if (pass == 0) {
            for (i = 0, sum = 0; i < nk; ++i) {
                if (B->counts[i] > ACAP) B->counts[i] = 0;      /* sumvalue_x: more than 256 -> dropped */
                B->first[i] = B->counts[i] > 0 ? sum : -1;
                sum += B->counts[i];
            }
            B->pos = (int*)malloc((size_t)(sum > 0 ? sum : 1) * sizeof(int));
            for (i = 0; i < nk; ++i) B->counts[i] = 0;           /* refilled as cursors by the second pass */
        }
    }
    {
        const int nseg = n / AZV + 5;                             /* :546 */
        B->db = (asm_seg*)calloc((size_t)nseg + 8, sizeof(asm_seg));
        for (i = 0; i < nseg + 8; ++i) B->db[i].index = -1;
        B->index_list = (int*)malloc((size_t)nseg * sizeof(int));
        B->index_score = (short*)malloc((size_t)nseg * sizeof(short));
        B->dirty = (int*)malloc((size_t)nseg * 2 * sizeof(int));
    }
    return B;
}

/* The reference resets only `score` and `index` of the segments a strand touched (:717): the stored seeds, `seednum` and whatever a
 * worker thread's earlier reads left in a segment stay, and the sweeps' reads past the 60 stored entries can land on them — so its
 * candidates depend, in principle, on which reads the same thread mapped before.  With on != 0 every READ starts from an all-zero
 * segment array (the forward strand's leftovers are still there for the reverse strand, as in the reference, where the two strands
 * always follow each other on one thread): the deterministic definition the device path implements.  The golden sets give the same
 * candidates either way (tests/test_asmpw_ref_cpu.py). */
void asm_block_fresh(asm_block* B, int on) { B->fresh = on; }

static int find_read(const int* a, int key, int n) {             /* binary :283-297, statement for statement (its quirks decide results) */
    int left = 0, right = n - 1, mid = (left + right) / 2;
    if (a[right] < key) return right;
    while (left <= right && a[mid] != key) {
        if (a[mid] < key && a[mid + 1] > key) return mid;
        if (a[mid] < key && a[mid + 1] == key) return mid + 1;
        else if (a[mid] < key && a[mid + 1] < key) left = mid + 1;
        else if (a[mid] > key && a[mid - 1] <= key) return mid - 1;
        else if (a[mid] > key && a[mid - 1] > key) right = mid - 1;
        mid = (left + right) / 2;
    }
    if (a[mid] == key) return mid;
    return mid;     /* (the reference falls off the end here: not reached with ascending read starts) */
}

/* one strand: seeding, then the candidate loop; cands[0 .. *ncand) is the list shared by both strands */
static void strand(asm_block* B, const char* s, int len, int read_name, char chain, asm_candidate* cands, int* ncand) {
    asm_seg* db = B->db;
    const int K = (len - AK) / ABC + 1;                           /* transnum_buchang :319 */
    int k, i, j, touched = 0;
    if (len < AK) return;                                          /* (K <= 0: no k-mers) */
    for (k = 0; k < K; ++k) {
        int id = 0, bad = 0;
        for (j = 0; j < AK; ++j) {
            const int t = base_code(s[k * ABC + j]);
            if (t == 4) { bad = 1; break; }
            id = (id << 2) + t;
        }
        if (bad || B->first[id] < 0) continue;
        {
            const int* p = B->pos + B->first[id];
            const int c = B->counts[id];
            for (i = 0; i < c; ++i) {                              /* :608-640 */
                const int seg = p[i] / AZV, off = p[i] % AZV;
                asm_seg* g = db + seg;
                if (g->score == 0 || g->seednum < k + 1) {
                    const int at = ++g->score;
                    int sk;
                    if (at <= ASM) { g->loczhi[at - 1] = (short)off; g->seedno[at - 1] = (short)(k + 1); }
                    sk = seg > 0 ? g->score + (g - 1)->score : g->score;
                    if (g->index == -1) {
                        B->index_list[touched] = seg;
                        B->index_score[touched] = (short)sk;
                        g->index = touched++;
                    } else B->index_score[g->index] = (short)sk;
                }
                g->seednum = (short)(k + 1);
            }
        }
    }
    for (i = 0; i < touched; ++i) {                                /* :643-716 */
        const int seg = B->index_list[i];
        asm_seg* g = db + seg;
        int tl[150], ts[150], tsc[150], loc[4], rep = 0, n = 0, start_loc, prev = 0;
        int loc_seed, loc_list, rd, rstart, rend, vote;
        if (!(B->index_score[i] > 10)) continue;
        if (g->score == 0) continue;
        start_loc = seg * AZV;
        if (seg > 0) { prev = (g - 1)->score; if (prev > 0) start_loc = (seg - 1) * AZV; }
        if (prev > 0)
            for (j = 0; j < prev && j < ASM; ++j) { tl[n] = (g - 1)->loczhi[j]; ts[n] = (g - 1)->seedno[j]; ++n; }
        for (j = 0; j < g->score && j < ASM; ++j) { tl[n] = g->loczhi[j] + (prev > 0 ? AZV : 0); ts[n] = g->seedno[j]; ++n; }
        if (!orc_find_location(tl, ts, tsc, loc, n, &rep, (float)ABC, len, 0.10)) continue;
        if (tsc[rep] < 6) continue;
        vote = tsc[rep];
        loc_seed = ts[rep];
        loc[0] += start_loc;
        loc_list = loc[0];
        rd = find_read(B->lloc, loc[0], B->nreads);
        rstart = B->lloc[rd];
        rend = B->lloc[rd + 1];
        if (B->readno[rd] > read_name) continue;
        if (B->readno[rd] == read_name) {                          /* the read meets itself: its own stretch of the block is scrubbed (:651-658) */
            int u = rstart / AZV, cut = rstart % AZV, kk, last;
            asm_seg* h = db + u;
            for (j = 0, kk = 0; j < h->score && j < ASM; ++j) if (h->loczhi[j] < cut) h->loczhi[kk++] = h->loczhi[j];
            h->score = (short)kk;
            for (++h, ++u, last = rend / AZV; u < last; ++u, ++h) h->score = 0;
            for (j = 0, kk = 0, cut = rend % AZV; j < h->score && j < ASM; ++j) if (h->loczhi[j] > cut) h->loczhi[kk++] = h->loczhi[j];
            h->score = (short)kk;
            continue;
        }
        {
            asm_candidate c;
            int seedcount = 0, u, kk, lo, hi, mid;
            const asm_seg* o;
            loc[1] = (loc[1] - 1) * ABC;
            c.readno = rd; c.readstart = rstart;
            c.left1 = loc[0] - rstart + AK - 1; c.right1 = rend - loc[0];
            c.left2 = loc[1] + AK - 1; c.right2 = len - loc[1];
            c.num1 = c.left1 >= c.left2 ? c.left2 : c.left1;
            c.num2 = c.right1 >= c.right2 ? c.right2 : c.right1;
            if (c.num1 + c.num2 < 400) continue;
            c.loc1 = loc[0]; c.loc2 = loc[1];
            /* sweeps: from two segments to the left (:689-695) and one to the right (:697-703); entries j < score, not j < 60 */
            /* (entry j of a segment through the flat view of the segment array: LOC(o, j) is loczhi[j] and SEED(o, j) is seedno[j]
               for j < 60, and whatever lies behind them for larger j — the same bytes the reference's out-of-range indices reach) */
#define LOC(o, j) (((const short*)(o))[1 + (j)])
#define SEED(o, j) (((const short*)(o))[1 + ASM + (j)])
            for (u = seg - 2, kk = c.num1 / AZV, o = g - 2; u >= 0 && kk >= 0; --o, --kk, --u)
                if (o->score > 0) {
                    const int sl = u * AZV;
                    int agree = 0;
                    for (j = 0; j < o->score; ++j)
                        if (fabs((loc_list - sl - LOC(o, j)) / ((loc_seed - SEED(o, j)) * ABC * 1.0) - 1.0) < 0.10) { ++seedcount; ++agree; }
                    if (agree * 1.0 / o->score > 0.4) ((asm_seg*)o)->score = 0;
                }
            for (u = seg + 1, kk = c.num2 / AZV, o = g + 1; kk > 0; ++o, --kk, ++u)
                if (o->score > 0) {
                    const int sl = u * AZV;
                    int agree = 0;
                    for (j = 0; j < o->score; ++j)
                        if (fabs((sl + LOC(o, j) - loc_list) / ((SEED(o, j) - loc_seed) * ABC * 1.0) - 1.0) < 0.10) { ++seedcount; ++agree; }
                    if (agree * 1.0 / o->score > 0.4) ((asm_seg*)o)->score = 0;
                }
#undef LOC
#undef SEED
            c.score = vote + seedcount;
            c.chain = chain;
            /* stable insertion into the descending top-MAXC list (:707-716) */
            lo = 0; hi = *ncand - 1;
            while (lo <= hi) {
                mid = (lo + hi) / 2;
                if (mid >= *ncand || cands[mid].score < c.score) hi = mid - 1; else lo = mid + 1;
            }
            if (*ncand < AMAXC) for (u = *ncand - 1; u > hi; --u) cands[u + 1] = cands[u];
            else for (u = *ncand - 2; u > hi; --u) cands[u + 1] = cands[u];
            if (hi + 1 < AMAXC) cands[hi + 1] = c;
            if (*ncand < AMAXC) ++*ncand;
        }
    }
    for (i = 0; i < touched; ++i) {                                /* :717 */
        db[B->index_list[i]].score = 0;
        db[B->index_list[i]].index = -1;
        if (B->fresh) B->dirty[B->ndirty++] = B->index_list[i];
    }
}

/* candidates of one query read (upper-case text) in list order; out[AMAXC].  Returns their number. */
int asm_candidates(asm_block* B, const char* query, int qlen, int read_name, asm_candidate* out) {
    int n = 0, i;
    char* rc = (char*)malloc((size_t)qlen + 1);
    strand(B, query, qlen, read_name, 'F', out, &n);
    for (i = 0; i < qlen; ++i) {                                   /* reverse, then complement A<->T, C<->G; anything else stays (:571-584) */
        const char ch = query[qlen - 1 - i];
        rc[i] = ch == 'A' ? 'T' : ch == 'T' ? 'A' : ch == 'C' ? 'G' : ch == 'G' ? 'C' : ch;
    }
    rc[qlen] = 0;
    strand(B, rc, qlen, read_name, 'R', out, &n);
    free(rc);
    for (i = 0; i < B->ndirty; ++i) { memset(&B->db[B->dirty[i]], 0, sizeof(asm_seg)); B->db[B->dirty[i]].index = -1; }
    B->ndirty = 0;
    return n;
}
