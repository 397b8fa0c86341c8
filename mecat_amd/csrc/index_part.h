// index_part.h — the stable two-level partition build of the look-up table (index_part.hip), shared with index.hip (whole table)
// and comm.hip (key-range shards of a multi-GPU build).
#pragma once
#include <vector>

#include "common.h"

#define IXP_L1_BITS 9      // level-1 bins = the top 9 bits of a 13-mer's 26-bit id
#define IXP_L2_BITS 9      // sub-bins of a level-1 bin
#define IXP_ID_BITS 8      // k-mer ids of a sub-bin
#define IXP_NB1 (1 << IXP_L1_BITS)
#define IXP_IDS_PER_BIN (1u << (26 - IXP_L1_BITS))

struct IxpSlice {
    // in
    bool count_only = false;                 // stop after the first volume walk: only bin_total is produced
    int cut_step = 0;                        // position cuts of the bucket records (0 with d_recs == nullptr)
    int (*alloc)(IxpSlice*, size_t kept) = nullptr;      // called once the number of kept positions is known: sets d_offsets, d_slots (and d_recs)
    void* user = nullptr;
    uint32_t* d_starts = nullptr;            // [((bin_hi - bin_lo) << 17) + 1], allocated by the caller; values relative to the slice
    // set by alloc
    int32_t* d_offsets = nullptr;
    uint16_t* d_slots = nullptr;
    uint4* d_recs = nullptr;
    // out
    int64_t num_kept = 0;
    std::vector<uint32_t> bin_total;         // [IXP_NB1 + 1] exclusive prefix of the level-1 bin occupancies (every k-mer start of the volume)
};

int index_build_partitioned(mhip_ctx* c, const mhip_volume* v, int max_bucket, int bin_lo, int bin_hi, IxpSlice* out);

// index.hip: slots[] of a finished table ((position / 2000) mod 2^15 of every kept position), and a bare table handle for comm.hip
int index_add_slots(mhip_ctx* c, mhip_index* idx);
