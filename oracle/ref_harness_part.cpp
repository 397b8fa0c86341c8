// ref_harness_part.cpp — TEST INFRASTRUCTURE: function-level harness around the UNMODIFIED partition_candidates of mecat2cns
// (reference src/mecat2cns/overlaps_partition.cpp:175-224, compiled where it lies; see oracle/Makefile target `ref`).  Used to
// pin mecat_amd/host/partition.cpp.  Never linked by the product path.
#include "mecat2cns/overlaps_partition.h"

extern "C" {

// writes <input>.part<k> and <input>.partition_files next to `input`
void refp_partition_candidates(const char* input, long batch_size, int min_read_size, int num_files) {
    partition_candidates(input, (idx_t)batch_size, min_read_size, num_files);
}

// partition_m4records (overlaps_partition.cpp:344-412), the `-j 1 -g 1` flavour
void refp_partition_m4records(const char* input, double min_cov_ratio, long batch_size, int min_read_size, int num_files) {
    partition_m4records(input, min_cov_ratio, (index_t)batch_size, min_read_size, num_files);
}

}
