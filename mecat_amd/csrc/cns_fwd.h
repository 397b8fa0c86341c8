// cns_fwd.h — what the mecat2cns re-aligner's two kernels share (round 5).
//
// The re-aligner needs the whole edit path of every block (the consensus stage consumes the aligned strings), and rounds 1-4 kept it the
// way the reference does: every d-row's furthest-x values, 2 bytes per cell, written to a per-wave scratch and read back by a traceback
// that one whole wave walked step by step — 113 GB written per launch for 442 k jobs, one unit per wave, a fifth of dw_extend2's rate.
// Now the forward rows ARE dw_extend2 (align.hip, instantiated with mecat2cns' block rules: two units per wave, d-rows in a 1 K-entry LDS
// ring, tail walk for the cut in front of the last four matches), and what it keeps of a row is what a traceback cannot recompute: ONE
// BIT per cell — "came from diagonal k - 1" — plus the row's width and the cut of the band update behind it: a 16-byte record per row
// (CnsRowRec) instead of ~54 bytes.  The path itself is found afterwards, for all blocks of all units at once, by cns_trace
// (cns_align.hip): one LANE per block walks the bits back from the block's end point, then walks the path forward again, re-measuring
// every snake on the packed reads (an O(ND) path has no mismatch columns: a row is one indel and a run of equal bases), and writes the
// 2-bit columns.  13 M independent blocks per config-2 pass keep every lane of the chip busy; nothing waits for a scalar walk.
#pragma once
#include "common.h"

struct CnsDir {                    // one direction of one candidate
    int32_t cols, qbases, tbases, ins, del, pad;      // pad < 0: the unit was handed over to cns_extend (its blocks here are void)
};

struct CnsRowRec {                 // one d-row of a block
    uint32_t lo, hi;               // bit t: diagonal min_k + 2 t of the row took the step from k - 1 (x = V[k - 1] + 1; dw.cpp:176-179)
    uint32_t meta;                 // slots of the row (<= 96) | cut of the band update behind it << 8 (min_k of row d + 1 = min_k of row d + 2 cut - 1)
    uint32_t top;                  // bits 64 .. 95 of the same
};

struct CnsBlockRec {               // one block that contributes columns (dw_in_one_direction, dw.cpp:319-375)
    uint32_t unit;                 // 2 * job + direction
    int32_t qidx, tidx;            // where the block starts in the unit's two sequences
    int32_t seg;                   // its (square) size
    int32_t end_d, end_k, end_mk, end_x;      // the cell that reached an end of the block, and the first diagonal of its row
    int32_t col0, kept_cols;       // columns of the unit in front of this block; columns of this block that are kept
    uint32_t log0;                 // record of the block's row 0 in the row log
    int32_t pad;
};

struct CnsFwdArgs {
    double error_rate;
    const uint32_t* logbase;       // [units + 1] first row record of every unit (a unit that outgrows its share is handed over)
    CnsRowRec* rowlog;
    const uint32_t* blockbase;     // [units + 1] first block record of every unit.  (One counter for all records was the first version: 9 M
                                   // atomic adds on one word per 100 ms is all a word takes — the forward pass ran at the counter's pace.)
    CnsBlockRec* blocks;           // unit == 0xffffffff: a record nobody wrote (the array is filled with 0xff before the pass)
    unsigned int* hand_units;      // units for cns_extend: [0] their number, then the units
    CnsDir* dres;                  // [units]
    int dir_cols_cap;
    int* err_flag;                 // a direction needed more than dir_cols_cap columns
};

// align.hip: the forward pass over jobs [0, n) (both directions of each), on the context's stream
int cns_forward_launch(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n, const CnsFwdArgs& args);
