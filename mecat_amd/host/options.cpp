// options.cpp — see options.h.  Behaviour follows mecat2pw/pw_options.cpp of the reference: getopt string
// "j:d:o:w:t:n:g:x:a:k:" (:92), defaults (:8-13, :30-50), validation order and messages (:153-193), working directory
// creation (:197-208).  GPU selection is additive and environment-only (MECAT_HIP_DEVICE), so existing invocations
// run unchanged.
#include "options.h"

#include <dirent.h>
#include <errno.h>
#include <getopt.h>
#include <stdio.h>
#include <stdlib.h>
#include <sys/stat.h>
#include <unistd.h>

static void logmsg(const char* func, int line, const char* msg) { fprintf(stderr, "[%s, %d] %s\n", func, line, msg); }
#define LOGF(...)                                         \
    do {                                                  \
        char _b[1024];                                    \
        snprintf(_b, sizeof(_b), __VA_ARGS__);            \
        logmsg(__func__, __LINE__, _b);                   \
    } while (0)

void print_options(const Options* o) {
    LOGF("task\t\t%d", o->task);
    LOGF("reads\t\t%s", o->reads);
    LOGF("output\t\t%s", o->output);
    LOGF("working folder\t%s", o->wrk_dir);
    LOGF("# of threads\t\t%d", o->num_threads);
    LOGF("# of candidates\t%d", o->num_candidates);
    LOGF("min align size\t%d", o->min_align_size);
    LOGF("min block score\t%d", o->min_kmer_match);
    LOGF("output gapped start\t%c", o->output_gapped_start_point ? 'Y' : 'N');
    LOGF("tech\t%d", o->tech);
}

void print_usage(const char* prog) {
    fprintf(stderr, "\n\n");
    fprintf(stderr, "usage:\n");
    fprintf(stderr, "%s [-j task] [-d dataset] [-o output] [-w working dir] [-t threads] [-n candidates] [-g 0/1]", prog);
    fprintf(stderr, "\n\n");
    fprintf(stderr, "options:\n");
    fprintf(stderr, "-j <integer>\tjob: %d = seeding, %d = align\n\t\tdefault: %d\n", TASK_SEED, TASK_ALN, TASK_ALN);
    fprintf(stderr, "-d <string>\treads file name\n");
    fprintf(stderr, "-o <string>\toutput file name\n");
    fprintf(stderr, "-w <string>\tworking folder name, will be created if not exist\n");
    fprintf(stderr, "-t <integer>\tnumber of cput threads\n\t\tdefault: 1\n");
    fprintf(stderr, "-n <integer>\tnumber of candidates for gapped extension\n\t\tDefault: 100\n");
    fprintf(stderr, "-a <integer>\tminimum size of overlaps\n\t\t");
    fprintf(stderr, "Default: %d if x = %d, %d if x = %d\n", 2000, TECH_PACBIO, 500, TECH_NANOPORE);
    fprintf(stderr, "-k <integer>\tminimum number of kmer match a matched block has\n\t\t");
    fprintf(stderr, "Default: %d if x = %d, %d if x = %d\n", 4, TECH_PACBIO, 2, TECH_NANOPORE);
    fprintf(stderr, "-g <0/1>\twhether print gapped extension start point, 0 = no, 1 = yes\n\t\tDefault: 0\n");
    fprintf(stderr, "-x <0/x>\tsequencing technology: 0 = pacbio, 1 = nanopore\n\t\tDefault: 0\n");
}

int parse_arguments(int argc, char* argv[], Options* o) {
    int c;
    opterr = 0;
    int task = -1, num_threads = -1, num_candidates = -1, min_align_size = -1, min_kmer_match = -1, gapped = -1;
    const char *reads = NULL, *output = NULL, *wrk_dir = NULL;
    int tech = TECH_PACBIO;
    while ((c = getopt(argc, argv, "j:d:o:w:t:n:g:x:a:k:")) != -1) {
        switch (c) {
        case 'j': task = atoi(optarg); break;
        case 'd': reads = optarg; break;
        case 'o': output = optarg; break;
        case 'w': wrk_dir = optarg; break;
        case 't': num_threads = atoi(optarg); break;
        case 'n': num_candidates = atoi(optarg); break;
        case 'a': min_align_size = atoi(optarg); break;
        case 'k': min_kmer_match = atoi(optarg); break;
        case 'g':
            if (optarg[0] == '0') gapped = 0;
            else if (optarg[0] == '1') gapped = 1;
            else { LOGF("argument to option '-g' must be either '0' or '1'"); return 1; }
            break;
        case 'x':
            if (optarg[0] == '0') tech = TECH_PACBIO;
            else if (optarg[0] == '1') tech = TECH_NANOPORE;
            else { LOGF("invalid argument to option 'x': %s", optarg); abort(); }
            break;
        case '?': LOGF("unrecognised option '%c'", (char)optopt); return 1;
        case ':': LOGF("argument to option '%c' is not provided!", (char)optopt); return 1;
        }
    }
    o->task = TASK_ALN;
    o->num_threads = 1;
    o->num_candidates = 100;
    o->output_gapped_start_point = 0;
    o->tech = tech;
    if (tech == TECH_PACBIO) { o->min_align_size = 2000; o->min_kmer_match = 4; }
    else { o->min_align_size = 500; o->min_kmer_match = 2; }
    if (task != -1) o->task = task;
    o->reads = reads;
    o->output = output;
    o->wrk_dir = wrk_dir;
    if (num_threads != -1) o->num_threads = num_threads;
    if (num_candidates != -1) o->num_candidates = num_candidates;
    if (min_align_size != -1) o->min_align_size = min_align_size;
    if (min_kmer_match != -1) o->min_kmer_match = min_kmer_match;
    if (gapped != -1) o->output_gapped_start_point = gapped;

    int ret = 0;
    if (o->task != TASK_SEED && o->task != TASK_ALN) { LOGF("task (-j) must be %d or %d, not %d.", TASK_SEED, TASK_ALN, o->task); ret = 1; }
    if (!o->reads) { LOGF("dataset must be specified."); ret = 1; }
    else if (!o->output) { LOGF("output must be specified."); ret = 1; }
    else if (!o->wrk_dir) { LOGF("working directory must be specified."); ret = 1; }
    else if (o->num_threads < 1) { LOGF("number of cpu threads must be > 0."); ret = 1; }
    else if (o->num_candidates < 1) { LOGF("number of candidates must be > 0."); ret = 1; }
    // (any positive -n, as in the reference, pw_options.cpp:9: lists above 1024 entries are built in HBM instead of LDS; the candidate
    // table of a slab — reads x n x 48 bytes — has to fit the device)
    else if (o->num_candidates > (1 << 20)) { LOGF("number of candidates (-n) must be <= %d (got %d).", 1 << 20, o->num_candidates); ret = 1; }
    if (ret) return ret;

    DIR* dir = opendir(o->wrk_dir);
    if (dir == NULL) {
        // (EEXIST: another process of a multi-process run created it between the two calls)
        if (mkdir(o->wrk_dir, S_IRWXU) == -1 && errno != EEXIST) { LOGF("fail to create folder '%s'!", o->wrk_dir); exit(1); }
    } else closedir(dir);
    return 0;
}
