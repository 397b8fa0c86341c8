/* xdrop_rowpar.c — TEST INFRASTRUCTURE.  The X-drop row (reference src/common/xdrop_gapalign.cpp:85-158) written as
 * per-cell independent work + prefix scans, checked row by row against the literal sequential form.  This is the
 * formulation a wave-parallel kernel can use; the self-check below is how it was validated.
 *
 *   int orc_xdrop_rowpar_selfcheck(const char* A, int M, const char* B, int N)   -> 0 ok, else 1000 * row + what
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define XMIN (-100000000)
enum { S_SUB = 3, S_GA = 0, S_GB = 6, F_EXT_A = 0x10, F_EXT_B = 0x40 };

typedef struct { int h, f; } cell_t;

static int imax(int a, int b) { return a > b ? a : b; }

int orc_xdrop_rowpar_selfcheck(const char* A, int M, const char* B, int N) {
    const int X = 30;
    cell_t* s1 = (cell_t*)malloc(sizeof(cell_t) * (size_t)(N + 2));     /* sequential */
    cell_t* s2 = (cell_t*)malloc(sizeof(cell_t) * (size_t)(N + 2));     /* scan form */
    uint8_t* r1 = (uint8_t*)malloc((size_t)N + 2);
    uint8_t* r2 = (uint8_t*)malloc((size_t)N + 2);
    int *Mv = (int*)malloc(sizeof(int) * (size_t)(N + 2)), *Ec = (int*)malloc(sizeof(int) * (size_t)(N + 2)),
        *Hc = (int*)malloc(sizeof(int) * (size_t)(N + 2)), *dg = (int*)malloc(sizeof(int) * (size_t)(N + 2));
    int rc = 0;
    /* row 0 (xdrop_gapalign.cpp:45-57) */
    int score = -1, i;
    s1[0].h = 0; s1[0].f = -1;
    for (i = 1; i <= N; ++i) {
        if (score < -X) break;
        s1[i].h = score; s1[i].f = score - 1;
        score -= 1;
    }
    memcpy(s2, s1, sizeof(cell_t) * (size_t)(N + 2));
    int bsz1 = i, best1 = 0, first1 = 0, ae1 = 0, be1 = 0;
    int bsz2 = i, best2 = 0, first2 = 0, ae2 = 0, be2 = 0;
    for (int a = 1; a <= M && !rc; ++a) {
        const int AC = A[a - 1];
        const int f_start = first1, n_start = bsz1;
        int ext1 = bsz1, ext2 = bsz2;               /* end of the cells that received a script byte */
        /* ---- sequential (:85-158) */
        int done1 = 0;
        {
            int sc = XMIN, gr = XMIN, last = first1, b;
            const int orig = first1;
            (void)orig;
            for (b = first1; b < bsz1; ++b) {
                int gc = s1[b].f;
                const int next = s1[b].h + (AC == B[b] ? 1 : -1);
                int script = S_SUB;
                if (sc < gc) { script = S_GB; sc = gc; }
                if (sc < gr) { script = S_GA; sc = gr; }
                if (best1 - sc > X) {
                    if (first1 == b) ++first1; else s1[b].h = XMIN;
                } else {
                    last = b;
                    if (sc > best1) { best1 = sc; ae1 = a; be1 = b; }
                    gc -= 1;
                    if (gc < sc - 1) s1[b].f = sc - 1; else { s1[b].f = gc; script += F_EXT_A; }
                    gr -= 1;
                    if (gr < sc - 1) gr = sc - 1; else script += F_EXT_B;
                    s1[b].h = sc;
                }
                sc = next;
                r1[b] = (uint8_t)script;
            }
            if (first1 == bsz1) done1 = 1;
            else {
                if (last < bsz1 - 1) bsz1 = last + 1;
                else while (gr >= best1 - X && bsz1 < N) { s1[bsz1].h = gr; s1[bsz1].f = gr - 1; gr -= 1; r1[bsz1] = S_GA; ++bsz1; }
                ext1 = bsz1 > n_start ? bsz1 : n_start;
                if (bsz1 < N) { s1[bsz1].h = XMIN; s1[bsz1].f = XMIN; ++bsz1; }
            }
        }
        /* ---- scan form */
        int done2 = 0;
        {
            const int f0 = first2, n0 = bsz2;
            /* per cell, independent */
            for (int b = f0; b < n0; ++b) {
                dg[b] = b == f0 ? XMIN : s2[b - 1].h + (AC == B[b - 1] ? 1 : -1);
                Mv[b] = imax(dg[b], s2[b].f);
            }
            /* exclusive prefix max of M + j  ->  E = that - b (distance decay, no barriers); first cell: XMIN */
            {
                long long run = 0;
                int have = 0;
                for (int b = f0; b < n0; ++b) {
                    Ec[b] = have ? (int)(run - b) : XMIN;
                    const long long v = (long long)Mv[b] + b;
                    if (!have || v > run) { run = v; have = 1; }
                    Hc[b] = imax(Mv[b], Ec[b]);
                }
            }
            /* exclusive prefix max of H -> drop decision; kept mask */
            int bb = best2, rowmax = XMIN, rowarg = -1, lastkept = -1, firstkept = -1;
            for (int b = f0; b < n0; ++b) {
                const int kept = !(bb - Hc[b] > X);
                if (Hc[b] > bb) bb = Hc[b];
                /* NOTE: bb includes dropped cells' Hc: harmless, they are below bb - X */
                int script = S_SUB, sc = dg[b];
                if (sc < s2[b].f) { script = S_GB; sc = s2[b].f; }
                if (kept) {
                    if (sc < Ec[b]) { script = S_GA; sc = Ec[b]; }
                    if (firstkept < 0) firstkept = b;
                    lastkept = b;
                    if (Hc[b] > rowmax) { rowmax = Hc[b]; rowarg = b; }
                    if (s2[b].f - 1 < Hc[b] - 1) s2[b].f = Hc[b] - 1; else { s2[b].f = s2[b].f - 1; script += F_EXT_A; }
                    if (!(Ec[b] - 1 < Hc[b] - 1)) script += F_EXT_B;
                    s2[b].h = Hc[b];
                } else {
                    /* true E here: after the nearest kept cell to the left it is H - 1 and stays (no decay over dropped cells) */
                    const int et = lastkept >= 0 ? s2[lastkept].h - 1 : XMIN;
                    if (sc < et) script = S_GA;
                    if (firstkept >= 0) s2[b].h = XMIN;       /* interior; a leading one just moves first_b */
                }
                r2[b] = (uint8_t)script;
            }
            if (rowmax > best2) { best2 = rowmax; ae2 = a; be2 = rowarg; }
            if (firstkept < 0) { first2 = n0; done2 = 1; }
            else {
                first2 = firstkept;
                if (lastkept < n0 - 1) bsz2 = lastkept + 1;
                else {
                    int gr = imax(Ec[lastkept], Hc[lastkept]) - 1;
                    while (gr >= best2 - X && bsz2 < N) { s2[bsz2].h = gr; s2[bsz2].f = gr - 1; gr -= 1; r2[bsz2] = S_GA; ++bsz2; }
                }
                ext2 = bsz2 > n_start ? bsz2 : n_start;
                if (bsz2 < N) { s2[bsz2].h = XMIN; s2[bsz2].f = XMIN; ++bsz2; }
            }
        }
        /* ---- compare */
        if (done1 != done2) { rc = 1000 * a + 1; break; }
        if (first1 != first2 || bsz1 != bsz2) { rc = 1000 * a + 2; break; }
        if (best1 != best2 || ae1 != ae2 || be1 != be2) { rc = 1000 * a + 3; break; }
        if (done1) break;
        for (int b = first1; b < bsz1; ++b)
            if (s1[b].h != s2[b].h || s1[b].f != s2[b].f) { rc = 1000 * a + 4; break; }
        if (rc) break;
        /* script bytes of every cell the row visited: the window at the row's start plus the appended gap cells */
        if (ext1 != ext2) { rc = 1000 * a + 5; break; }
        for (int b = f_start; b < ext1; ++b)
            if (r1[b] != r2[b]) { rc = 1000 * a + 6; break; }
    }
    free(s1); free(s2); free(r1); free(r2); free(Mv); free(Ec); free(Hc); free(dg);
    return rc;
}
