// cns_strings.h — device-built aligned strings of accepted mecat2cns alignments (cns_strings.hip), used by cns_accept.hip
#pragma once
#include "common.h"

struct CnsStrItem {
    int32_t job;                 // index into the jobs / results / column rows the launch is given
    int32_t aln_size;            // kept columns = characters per string
    unsigned long long off;      // byte offset of the query string in the output; the template string follows at off + aln_size + 1
};

// d_out: 32-byte aligned device buffer; strings may be read in whole 32-byte blocks, so the allocation must reach 64 bytes beyond the
// last string.  Launches on c->stream, waits for nothing.
int cns_strings_launch(mhip_ctx* c, const mhip_volume* vol, const mhip_aln_job* d_jobs, const mhip_cns_result* d_res, const uint32_t* d_ops, int row_words,
                       const CnsStrItem* d_items, int n_items, char* d_out);
