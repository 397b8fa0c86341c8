// options.h — command line of mecat2pw (drop-in: same getopt string, defaults and validation as the reference's
// mecat2pw/pw_options.cpp:73-211; usage text :52-71).
#pragma once

#define TASK_SEED 0
#define TASK_ALN 1
#define TECH_PACBIO 0
#define TECH_NANOPORE 1

struct Options {
    int task;
    const char* reads;
    const char* output;
    const char* wrk_dir;
    int num_threads;
    int num_candidates;
    int min_align_size;
    int min_kmer_match;
    int output_gapped_start_point;
    int tech;
};

// returns 0 on success, 1 when the usage text should be printed (the caller exits with status 1)
int parse_arguments(int argc, char* argv[], Options* opt);
void print_usage(const char* prog);
void print_options(const Options* opt);
