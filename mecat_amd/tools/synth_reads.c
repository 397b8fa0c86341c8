/*
 * synth_reads — deterministic synthetic long-read generator (test/bench input).
 *
 * The reference ships no data; every workload in BASELINE.json is
 * "synthetic PacBio-/ONT-style reads".  Model:
 *   genome: uniform random ACGT of length G
 *   read i: uniform start, template length L (clipped to the genome end), strand by coin flip,
 *           per template base: delete w.p. pdel, substitute by a uniform base w.p. psub, else copy;
 *           after each template base insert a uniform base w.p. pins.
 *   PacBio-style  e: pdel=0.25e psub=0.15e pins=0.60e ; ONT-style e: 0.35e / 0.35e / 0.30e.
 * RNG: xorshift64*, the genome from stream `seed`, read i from stream splitmix64(seed, i) so reads can be
 * generated in any order / in parallel with identical output.
 *
 * Built both as a CLI (FASTA to stdout/file) and as a tiny shared library used by bench.py and the tests.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

static inline uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
typedef struct { uint64_t s; } rng_t;
static inline void rng_seed(rng_t* r, uint64_t seed) { r->s = splitmix64(seed); if (!r->s) r->s = 1; }
static inline uint64_t rng_next(rng_t* r) {
    uint64_t x = r->s;
    x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
    r->s = x;
    return x * 0x2545F4914F6CDD1Dull;
}
static inline double rng_unit(rng_t* r) { return (double)(rng_next(r) >> 11) * (1.0 / 9007199254740992.0); }

/* genome of G codes 0..3 */
void synth_genome(uint8_t* g, int64_t G, uint64_t seed) {
    rng_t r; rng_seed(&r, seed ^ 0x47454E4F4D45ull);
    int64_t i = 0;
    while (i < G) {
        uint64_t w = rng_next(&r);
        for (int k = 0; k < 32 && i < G; ++k, ++i) { g[i] = (uint8_t)(w & 3); w >>= 2; }
    }
}

/*
 * Repeat structure laid over a genome from synth_genome (seeds recorded by the caller: everything derives from `seed`):
 *   nfam interspersed families   one element of 300 .. 5 000 bases each, 2 .. max_copies copies pasted at uniform positions, every copy
 *                                with its own divergence of 0 .. 5 % (substitutions and single-base indels) and on either strand —
 *                                rRNA operons / IS elements / transposons: identical k-mers from many loci, which is what fills
 *                                k-mer buckets up to and beyond the index's cap of 128, fires the 41st-seed replacement rule on hits
 *                                that are not self hits and produces tied scores in the top-MAXC list
 *   nsat microsatellites         a motif of 1 .. 6 bases (1 = a homopolymer run) repeated over 40 .. 600 bases
 * stats[0] = bases covered by family copies, stats[1] = bases covered by microsatellites, stats[2] = copies pasted.
 */
void synth_genome_repeats(uint8_t* g, int64_t G, uint64_t seed, int nfam, int max_copies, int nsat, int64_t* stats) {
    rng_t r; rng_seed(&r, splitmix64(seed) ^ 0x5245504541545321ull);
    int64_t fam_bases = 0, sat_bases = 0, copies_total = 0;
    uint8_t* el = (uint8_t*)malloc(5000);
    uint8_t* cp = (uint8_t*)malloc(5600);
    for (int f = 0; f < nfam && el && cp; ++f) {
        int len = 300 + (int)(rng_unit(&r) * 4701.0);
        if (len > G / 4) len = (int)(G / 4);
        if (len < 20) break;
        for (int i = 0; i < len; ++i) el[i] = (uint8_t)(rng_next(&r) >> 62);
        int copies = 2 + (int)(rng_unit(&r) * (double)(max_copies > 2 ? max_copies - 1 : 1));
        for (int c = 0; c < copies; ++c) {
            const double div = rng_unit(&r) * 0.05;
            const int rev = (int)(rng_next(&r) >> 63);
            int n = 0;
            for (int i = 0; i < len && n < 5590; ++i) {
                uint8_t b = rev ? (uint8_t)(3 - el[len - 1 - i]) : el[i];
                const double u = rng_unit(&r);
                if (u < div * 0.6) cp[n++] = (uint8_t)(rng_next(&r) >> 62);            /* substitution (may redraw the same base) */
                else if (u < div * 0.8) { /* deletion */ }
                else if (u < div) { cp[n++] = b; cp[n++] = (uint8_t)(rng_next(&r) >> 62); } /* insertion */
                else cp[n++] = b;
            }
            if (n > G) n = (int)G;
            int64_t at = (int64_t)(rng_unit(&r) * (double)(G - n + 1));
            if (at > G - n) at = G - n;
            memcpy(g + at, cp, (size_t)n);
            fam_bases += n; ++copies_total;
        }
    }
    for (int s = 0; s < nsat; ++s) {
        const int m = 1 + (int)(rng_unit(&r) * 6.0);
        uint8_t motif[8];
        for (int i = 0; i < m; ++i) motif[i] = (uint8_t)(rng_next(&r) >> 62);
        int64_t len = 40 + (int64_t)(rng_unit(&r) * 561.0);
        if (len > G) len = G;
        int64_t at = (int64_t)(rng_unit(&r) * (double)(G - len + 1));
        if (at > G - len) at = G - len;
        for (int64_t i = 0; i < len; ++i) g[at + i] = motif[i % m];
        sat_bases += len;
    }
    free(el); free(cp);
    if (stats) { stats[0] = fam_bases; stats[1] = sat_bases; stats[2] = copies_total; }
}

/* one read -> codes 0..3 in out (capacity cap); returns length */
int synth_one_read(const uint8_t* g, int64_t G, int64_t idx, int L, double pdel, double psub, double pins,
                   uint64_t seed, uint8_t* out, int cap) {
    rng_t r; rng_seed(&r, splitmix64(seed) ^ splitmix64((uint64_t)idx * 2 + 1));
    int64_t tl = L < G ? L : G;
    int64_t start = (int64_t)(rng_unit(&r) * (double)(G - tl + 1));
    if (start > G - tl) start = G - tl;
    int rev = (int)(rng_next(&r) >> 63);
    int n = 0;
    for (int64_t t = 0; t < tl && n < cap; ++t) {
        uint8_t b = rev ? (uint8_t)(3 - g[start + tl - 1 - t]) : g[start + t];
        double u = rng_unit(&r);
        if (u < pdel) { /* deleted */ }
        else if (u < pdel + psub) out[n++] = (uint8_t)(rng_next(&r) >> 62);
        else out[n++] = b;
        if (n < cap && rng_unit(&r) < pins) out[n++] = (uint8_t)(rng_next(&r) >> 62);
    }
    return n;
}

void synth_rates(double e, int ont, double* pdel, double* psub, double* pins) {
    if (ont) { *pdel = 0.35 * e; *psub = 0.35 * e; *pins = 0.30 * e; }
    else     { *pdel = 0.25 * e; *psub = 0.15 * e; *pins = 0.60 * e; }
}

/*
 * Generate `nreads` reads into one contiguous code buffer (0..3), lengths in lens[].
 * bases must hold nreads * cap bytes where cap = (int)(L*1.25)+64; reads are packed back to back afterwards.
 * Returns total bases.
 */
int64_t synth_reads(int64_t G, int64_t nreads, int L, double e, int ont, uint64_t seed,
                    uint8_t* bases, int64_t bases_cap, int32_t* lens) {
    double pdel, psub, pins; synth_rates(e, ont, &pdel, &psub, &pins);
    uint8_t* g = (uint8_t*)malloc((size_t)G);
    if (!g) return -1;
    synth_genome(g, G, seed);
    int cap = (int)(L * 1.25) + 64;
    if (bases_cap < nreads * (int64_t)cap) { free(g); return -2; }
    /* every read has its own RNG stream: generate in place at slot i * cap in parallel, then compact in order */
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < nreads; ++i)
        lens[i] = synth_one_read(g, G, i, L, pdel, psub, pins, seed, bases + i * cap, cap);
    int64_t tot = 0;
    for (int64_t i = 0; i < nreads; ++i) {
        if (tot != i * cap) memmove(bases + tot, bases + i * cap, (size_t)lens[i]);
        tot += lens[i];
    }
    free(g);
    return tot;
}

/* 2-bit volume image of reads given as codes 0..3 (layout of common/split_database.cpp:103-119,249-250 in the reference:
   first base in the two MSBs of a byte, one zero pad base after every read).  pac must hold (total + nreads + 3) / 4
   zeroed bytes, offs 2 * nreads ints (offset, size).  Returns num_bases (incl. pads). */
int64_t synth_pack_volume(const uint8_t* codes, const int32_t* lens, int64_t nreads, uint8_t* pac, int32_t* offs) {
    int64_t curr = 0, src = 0;
    for (int64_t i = 0; i < nreads; ++i) {
        offs[2 * i] = (int32_t)curr; offs[2 * i + 1] = lens[i];
        for (int k = 0; k < lens[i]; ++k, ++curr)
            pac[curr >> 2] |= (uint8_t)((codes[src + k] & 3) << ((~curr & 3) << 1));
        src += lens[i];
        ++curr;
    }
    return curr;
}

/* the same for reads [first, first + nreads) of the set (read i has its own RNG stream, so any range of a set can be made alone:
   the first two volumes of config 5 without the other seventeen).  `g` = the genome from synth_genome, or NULL to make it here. */
int64_t synth_reads_range(const uint8_t* g_in, int64_t G, int64_t first, int64_t nreads, int L, double e, int ont, uint64_t seed,
                          uint8_t* bases, int64_t bases_cap, int32_t* lens) {
    double pdel, psub, pins; synth_rates(e, ont, &pdel, &psub, &pins);
    uint8_t* g = (uint8_t*)g_in;
    if (!g) {
        g = (uint8_t*)malloc((size_t)G);
        if (!g) return -1;
        synth_genome(g, G, seed);
    }
    int cap = (int)(L * 1.25) + 64;
    if (bases_cap < nreads * (int64_t)cap) { if (!g_in) free(g); return -2; }
#pragma omp parallel for schedule(dynamic, 64)
    for (int64_t i = 0; i < nreads; ++i)
        lens[i] = synth_one_read(g, G, first + i, L, pdel, psub, pins, seed, bases + i * cap, cap);
    int64_t tot = 0;
    for (int64_t i = 0; i < nreads; ++i) {
        if (tot != i * cap) memmove(bases + tot, bases + i * cap, (size_t)lens[i]);
        tot += lens[i];
    }
    if (!g_in) free(g);
    return tot;
}

/* synth_pack_volume on all cores: offsets first, then every read packs its own bases — whole bytes with plain stores, the bytes it
   shares with its neighbours (head and tail) with atomic ORs.  Same bytes as synth_pack_volume. */
int64_t synth_pack_volume_mt(const uint8_t* codes, const int32_t* lens, int64_t nreads, uint8_t* pac, int32_t* offs) {
    int64_t* src0 = (int64_t*)malloc((size_t)(nreads + 1) * 8);
    if (!src0) return -1;
    int64_t curr = 0, src = 0;
    for (int64_t i = 0; i < nreads; ++i) {
        offs[2 * i] = (int32_t)curr; offs[2 * i + 1] = lens[i];
        src0[i] = src;
        src += lens[i];
        curr += lens[i] + 1;
    }
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < nreads; ++i) {
        const uint8_t* c = codes + src0[i];
        int64_t p = offs[2 * i];
        const int64_t end = p + lens[i];
        for (; p < end && (p & 3); ++p, ++c) __atomic_fetch_or(&pac[p >> 2], (uint8_t)((*c & 3) << ((~p & 3) << 1)), __ATOMIC_RELAXED);
        for (; p + 4 <= end; p += 4, c += 4) pac[p >> 2] = (uint8_t)(((c[0] & 3) << 6) | ((c[1] & 3) << 4) | ((c[2] & 3) << 2) | (c[3] & 3));
        for (; p < end; ++p, ++c) __atomic_fetch_or(&pac[p >> 2], (uint8_t)((*c & 3) << ((~p & 3) << 1)), __ATOMIC_RELAXED);
    }
    free(src0);
    return curr;
}

int synth_write_fasta(const char* path, const uint8_t* bases, const int32_t* lens, int64_t nreads) {
    FILE* f = fopen(path, "w");
    if (!f) return -1;
    int64_t off = 0;
    char* line = NULL; int lcap = 0;
    for (int64_t i = 0; i < nreads; ++i) {
        int n = lens[i];
        if (n + 2 > lcap) { lcap = n + 2; line = (char*)realloc(line, (size_t)lcap); }
        for (int k = 0; k < n; ++k) line[k] = "ACGT"[bases[off + k] & 3];
        line[n] = '\n';
        fprintf(f, ">r%lld\n", (long long)i);
        fwrite(line, 1, (size_t)n + 1, f);
        off += n;
    }
    free(line);
    return fclose(f);
}

#ifdef SYNTH_MAIN
/* CLI: streams the reads in batches (generation and FASTA text assembly in parallel, one write per batch), so that a 40 Gbase
   set (config 5) needs a few GB of memory, not the whole read set.  Output is byte-identical to synth_reads + synth_write_fasta. */
int main(int argc, char** argv) {
    if (argc < 7) {
        fprintf(stderr, "usage: %s out.fa nreads L err genome_len seed [ont=0 [nfam max_copies nsat]]\n", argv[0]);
        return 1;
    }
    const char* out = argv[1];
    int64_t n = atoll(argv[2]); int L = atoi(argv[3]); double e = atof(argv[4]);
    int64_t G = atoll(argv[5]); uint64_t seed = strtoull(argv[6], NULL, 10);
    int ont = argc > 7 ? atoi(argv[7]) : 0;
    double pdel, psub, pins; synth_rates(e, ont, &pdel, &psub, &pins);
    uint8_t* g = (uint8_t*)malloc((size_t)G);
    if (!g) { fprintf(stderr, "out of memory (genome)\n"); return 2; }
    synth_genome(g, G, seed);
    if (argc > 10) {       /* repeat-structured genome */
        int64_t st[3];
        synth_genome_repeats(g, G, seed, atoi(argv[8]), atoi(argv[9]), atoi(argv[10]), st);
        fprintf(stderr, "synth_reads: repeats: %lld bases in %lld family copies, %lld bases of microsatellites\n", (long long)st[0], (long long)st[2], (long long)st[1]);
    }
    const int cap = (int)(L * 1.25) + 64;
    const int64_t B = 65536;
    uint8_t* bases = (uint8_t*)malloc((size_t)(B * cap));
    int32_t* lens = (int32_t*)malloc((size_t)B * 4);
    int64_t* tpos = (int64_t*)malloc((size_t)(B + 1) * 8);
    char* text = (char*)malloc((size_t)(B * (cap + 32)));
    if (!bases || !lens || !tpos || !text) { fprintf(stderr, "out of memory (batch)\n"); return 2; }
    FILE* f = fopen(out, "w");
    if (!f) { perror(out); return 3; }
    int64_t tot = 0;
    for (int64_t b0 = 0; b0 < n; b0 += B) {
        const int64_t nb = n - b0 < B ? n - b0 : B;
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < nb; ++i)
            lens[i] = synth_one_read(g, G, b0 + i, L, pdel, psub, pins, seed, bases + i * cap, cap);
        tpos[0] = 0;
        for (int64_t i = 0; i < nb; ++i) {
            char hdr[32];
            tpos[i + 1] = tpos[i] + snprintf(hdr, sizeof(hdr), ">r%lld\n", (long long)(b0 + i)) + lens[i] + 1;
            tot += lens[i];
        }
#pragma omp parallel for schedule(dynamic, 64)
        for (int64_t i = 0; i < nb; ++i) {
            char* o = text + tpos[i];
            char hdr[32];
            const int h = snprintf(hdr, sizeof(hdr), ">r%lld\n", (long long)(b0 + i));
            memcpy(o, hdr, (size_t)h);
            o += h;
            const uint8_t* src = bases + i * cap;
            for (int k = 0; k < lens[i]; ++k) o[k] = "ACGT"[src[k] & 3];
            o[lens[i]] = '\n';
        }
        if (fwrite(text, 1, (size_t)tpos[nb], f) != (size_t)tpos[nb]) { perror(out); return 3; }
    }
    if (fclose(f)) { perror(out); return 3; }
    fprintf(stderr, "synth_reads: %lld reads, %lld bases, L=%d e=%.3f G=%lld seed=%llu ont=%d\n",
            (long long)n, (long long)tot, L, e, (long long)G, (unsigned long long)seed, ont);
    return 0;
}
#endif
