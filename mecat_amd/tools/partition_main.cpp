// mecat2cns_partition — writes the candidate partition files mecat2cns builds at start-up (reference
// src/mecat2cns/overlaps_partition.cpp:175-224) from a `.can` file, with threads.  See mecat_amd/host/partition.h.
//   mecat2cns_partition <candidates.can> <batch_size> <min_read_size> [threads]
#include <stdio.h>
#include <stdlib.h>

#include "../host/partition.h"

int main(int argc, char* argv[]) {
    if (argc < 4) {
        fprintf(stderr, "usage: %s <candidates.can> <batch_size> <min_read_size> [threads]\n", argv[0]);
        return 1;
    }
    const long n = partition_candidates_text(argv[1], atol(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 1);
    fprintf(stderr, "%ld records\n", n);
    return 0;
}
