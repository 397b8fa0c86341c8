// mecat2cns_partition — writes the candidate partition files mecat2cns builds at start-up (reference
// src/mecat2cns/overlaps_partition.cpp:175-224) from a `.can` file, with threads.  See mecat_amd/host/partition.h.
//   mecat2cns_partition <candidates.can> <batch_size> <min_read_size> [threads]
//   mecat2cns_partition -m <min_cov_ratio> <overlaps.m4> <batch_size> <min_read_size> [threads]     (partition_m4records, :344-412)
#include <stdio.h>
#include <stdlib.h>

#include "../host/partition.h"

int main(int argc, char* argv[]) {
    if (argc >= 6 && argv[1][0] == '-' && argv[1][1] == 'm') {
        const long n = partition_m4_text(argv[3], atof(argv[2]), atol(argv[4]), atoi(argv[5]), argc > 6 ? atoi(argv[6]) : 1);
        fprintf(stderr, "%ld records\n", n);
        return 0;
    }
    if (argc < 4) {
        fprintf(stderr, "usage: %s [-m <min_cov_ratio>] <candidates.can | overlaps.m4> <batch_size> <min_read_size> [threads]\n", argv[0]);
        return 1;
    }
    const long n = partition_candidates_text(argv[1], atol(argv[2]), atoi(argv[3]), argc > 4 ? atoi(argv[4]) : 1);
    fprintf(stderr, "%ld records\n", n);
    return 0;
}
