// xalign.hip — BLAST-style affine X-drop gapped extension in 500-bp blocks, the nanopore-mode aligner
// (used by mecat2pw -x 1 -j 1).
//
// Replaces XdropAligner::go (common/xdrop_gapalign.cpp:359-439), align_ex (:263-357), xdrop_align (:10-213) and
// script_to_aligned_string (:215-261) with reward 1, penalty -1, gap_open 0, gap_extend 1, X = 30, block 500
// (common/xdrop_gapalign.h:98-114).
//
// First correct version: ONE LANE per (candidate, direction) unit replays the reference's row-by-row dynamic program
// literally — the recurrence carries the running best score, the gap-in-row score and the first/last window index along
// each row, keeps stale best_gap values for dropped cells and lets the traceback walk through them, so a wave-parallel
// reformulation has to reproduce all of that (next round).  All DP state lives in a per-lane slice of a
// global scratch buffer (score pairs for the current window, one script byte per cell, per-row offsets).  As in dw, only
// what mecat2pw consumes is produced: end coordinates, identity counts; the traceback walks the script bytes once and
// does trim_mismatch_end and the match/column counting on the fly.
#include <stdlib.h>

#include <algorithm>

#include "common.h"

#define XB_BLOCK 64
#define X_SEG 500
#define X_MAXN 736                       // last block: one side < 600, the other <= 718 (gapalign.cpp:24-30)
#define X_STATE_CAP (192 * 1024)         // script bytes per block and lane (rows x (window + 2)); overflow -> error flag
#define X_MIN_SCORE (-100000000)

enum { XS_SUB = 3, XS_GAP_IN_A = 0, XS_GAP_IN_B = 6, XS_OP_MASK = 0x07, XS_EXT_A = 0x10, XS_EXT_B = 0x40 };   // xdrop_gapalign.h:13-34

struct XDir {
    int32_t qbases, tbases, matches, columns;     // kept alignment of one direction
    int32_t last_q, last_t, last_m;               // type of its LAST column (the left half drops it, xdrop_gapalign.cpp:401-402)
    int32_t blocks;
};

struct XLane {
    int2* score;        // [X_MAXN + 2] (best, best_gap)
    uint8_t* state;     // [X_STATE_CAP]
    int32_t* row_off;   // [X_MAXN + 2] offset of edit_script[a] inside state
    int32_t* row_start; // [X_MAXN + 2] edit_start_offset[a]
};
#define X_LANE_BYTES ((size_t)(X_MAXN + 2) * 8 + X_STATE_CAP + (size_t)(X_MAXN + 2) * 8)

struct XView {
    const uint32_t* pac;
    int64_t off;
    int A, B, comp;     // logical position i -> volume base off + A + B * i, complemented if comp
};
__device__ __forceinline__ int xv_at(const XView& v, int i) {
    const uint32_t c = pac_base(v.pac, v.off + v.A + (int64_t)v.B * i);
    return (int)(v.comp ? 3u - c : c);
}

struct XBlockOut {
    int ae, be;                 // aln_qe, aln_te
    int n, nmatch;              // alignment columns / equal columns of the block
    int qcnt, tcnt, acnt, mtail;// trim_mismatch_end: bases/columns/equal columns from the tail through the 4-match run
    int trim_ok;
    int l0q, l0t, l0m;          // type of the last column of the whole block string
    int l1q, l1t, l1m;          // type of the column just before the trimmed tail
    int overflow;
};

// xdrop_align (xdrop_gapalign.cpp:10-213) for block [qidx, qidx + M) x [tidx, tidx + N), then one pass over the path
__device__ void xdrop_block(const XView& q, int qidx, int M, const XView& t, int tidx, int N, XLane& L, XBlockOut& o) {
    o.ae = o.be = 0; o.n = o.nmatch = 0; o.qcnt = o.tcnt = o.acnt = o.mtail = 0; o.trim_ok = 0; o.overflow = 0;
    o.l0q = o.l0t = o.l0m = o.l1q = o.l1t = o.l1m = 0;
    if (M <= 0 || N <= 0) return;
    const int gap_open = 0, gap_extend = 1, gap_open_extend = 1;
    int x_dropoff = 30;
    if (x_dropoff < gap_open_extend) x_dropoff = gap_open_extend;
    int2* score_array = L.score;
    uint8_t* state_array = L.state;
    int states_used = 0;
    L.row_off[0] = 0;
    L.row_start[0] = 0;
    int score = -gap_open_extend;
    score_array[0] = make_int2(0, -gap_open_extend);
    int i;
    for (i = 1; i <= N; ++i) {
        if (score < -x_dropoff) break;
        score_array[i] = make_int2(score, score - gap_open_extend);
        score -= gap_extend;
        state_array[i] = XS_GAP_IN_A;
    }
    states_used = N < i + 1 ? N : i + 1;
    int b_size = i, best_score = 0, first_b_index = 0;
    int ae = 0, be = 0;
    (void)gap_open;
    for (int a_index = 1; a_index <= M; ++a_index) {
        const int AC = xv_at(q, qidx + a_index - 1);
        const int roff = states_used + 1;
        if (roff + (N - first_b_index) + 4 >= X_STATE_CAP) { o.overflow = 1; return; }
        L.row_off[a_index] = roff;
        L.row_start[a_index] = first_b_index;
        uint8_t* edit_script_row = state_array + roff - first_b_index;
        const int orig_b_index = first_b_index;
        score = X_MIN_SCORE;
        int score_gap_row = X_MIN_SCORE;
        int last_b_index = first_b_index;
        int b_index;
        for (b_index = first_b_index; b_index < b_size; ++b_index) {
            const int bch = xv_at(t, tidx + b_index);
            const int2 sa = score_array[b_index];
            int score_gap_col = sa.y;
            const int next_score = sa.x + (AC == bch ? 1 : -1);
            int script = XS_SUB;
            if (score < score_gap_col) { script = XS_GAP_IN_B; score = score_gap_col; }
            if (score < score_gap_row) { script = XS_GAP_IN_A; score = score_gap_row; }
            if (best_score - score > x_dropoff) {
                if (first_b_index == b_index) ++first_b_index;
                else score_array[b_index].x = X_MIN_SCORE;
            } else {
                last_b_index = b_index;
                if (score > best_score) { best_score = score; ae = a_index; be = b_index; }
                int2 ns;
                score_gap_col -= gap_extend;
                if (score_gap_col < (score - gap_open_extend)) ns.y = score - gap_open_extend;
                else { ns.y = score_gap_col; script += XS_EXT_A; }
                score_gap_row -= gap_extend;
                if (score_gap_row < (score - gap_open_extend)) score_gap_row = score - gap_open_extend;
                else script += XS_EXT_B;
                ns.x = score;
                score_array[b_index] = ns;
            }
            score = next_score;
            edit_script_row[b_index] = (uint8_t)script;
        }
        if (first_b_index == b_size) break;
        if (last_b_index < b_size - 1) b_size = last_b_index + 1;
        else {
            while (score_gap_row >= (best_score - x_dropoff) && b_size < N) {
                score_array[b_size] = make_int2(score_gap_row, score_gap_row - gap_open_extend);
                score_gap_row -= gap_extend;
                edit_script_row[b_size] = XS_GAP_IN_A;
                ++b_size;
            }
        }
        states_used += (b_index > b_size ? b_index : b_size) - orig_b_index + 1;
        if (b_size < N) {
            score_array[b_size] = make_int2(X_MIN_SCORE, X_MIN_SCORE);
            ++b_size;
        }
    }
    o.ae = ae; o.be = be;
    // traceback (:165-210) fused with script_to_aligned_string + trim_mismatch_end: columns are visited tail first
    int a_index = ae, b_index = be;
    int script = XS_SUB;
    int n = 0, nmatch = 0, m = 0, found = 0;
    int qcnt = 0, tcnt = 0, acnt = 0, mtail = 0, want_l1 = 0;
    while (a_index > 0 || b_index > 0) {
        const int next_script = state_array[L.row_off[a_index] + b_index - L.row_start[a_index]];
        switch (script) {
        case XS_GAP_IN_A:
            script = next_script & XS_OP_MASK;
            if (next_script & XS_EXT_A) script = XS_GAP_IN_A;
            break;
        case XS_GAP_IN_B:
            script = next_script & XS_OP_MASK;
            if (next_script & XS_EXT_B) script = XS_GAP_IN_B;
            break;
        default:
            script = next_script & XS_OP_MASK;
            break;
        }
        int cq, ct, cm;
        if (script == XS_GAP_IN_A) { --b_index; cq = 0; ct = 1; cm = 0; }
        else if (script == XS_GAP_IN_B) { --a_index; cq = 1; ct = 0; cm = 0; }
        else {
            --a_index; --b_index;
            cq = 1; ct = 1;
            cm = xv_at(q, qidx + a_index) == xv_at(t, tidx + b_index);
        }
        if (n == 0) { o.l0q = cq; o.l0t = ct; o.l0m = cm; }
        if (want_l1) { o.l1q = cq; o.l1t = ct; o.l1m = cm; want_l1 = 0; }
        if (!found) {
            ++acnt; qcnt += cq; tcnt += ct; mtail += cm;
            if (cm) ++m; else m = 0;
            if (m == 4) { found = 1; want_l1 = 1; }
        }
        ++n;
        nmatch += cm;
    }
    o.n = n; o.nmatch = nmatch;
    o.qcnt = qcnt; o.tcnt = tcnt; o.acnt = acnt; o.mtail = mtail;
    o.trim_ok = found && (n - acnt >= 2);      // "m == mat_cnt && k > 0" (gapalign.cpp:67)
}

__global__ __launch_bounds__(XB_BLOCK) void xd_extend(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                      const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                      const mhip_aln_job* __restrict__ jobs, int n, XDir* __restrict__ dres,
                                                      uint8_t* __restrict__ scratch, unsigned int* __restrict__ cursor,
                                                      int* __restrict__ err_flag, unsigned long long* __restrict__ counters) {
    const size_t tid = (size_t)blockIdx.x * XB_BLOCK + threadIdx.x;
    uint8_t* base = scratch + tid * X_LANE_BYTES;
    XLane L;
    L.score = (int2*)base;
    L.state = base + (size_t)(X_MAXN + 2) * 8;
    L.row_off = (int32_t*)(L.state + X_STATE_CAP);
    L.row_start = L.row_off + (X_MAXN + 2);
    unsigned long long nblocks = 0;
    while (true) {
        const unsigned int unit = atomicAdd(cursor, 1u);
        if (unit >= 2u * (unsigned)n) break;
        const mhip_aln_job jb = jobs[unit >> 1];
        const int right = unit & 1;
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        XView q, t;
        q.pac = qpac; q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
        t.pac = rpac; t.off = roffs[jb.sid_local].offset; t.comp = 0;
        const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
        if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
        t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
        const int query_size = right ? qsize - jb.qstart : jb.qstart;
        const int target_size = right ? tsize - jb.sstart : jb.sstart;
        int qidx = 0, tidx = 0;
        XDir R = {0, 0, 0, 0, 0, 0, 0, 0};
        while (true) {      // align_ex (xdrop_gapalign.cpp:263-357)
            const int qleft = query_size - qidx, tleft = target_size - tidx;
            int qblk, tblk, last_block;
            if (qleft < X_SEG + 100 || tleft < X_SEG + 100) {
                qblk = min(qleft, (int)(tleft + tleft * 0.2));
                tblk = min(tleft, (int)(qleft + qleft * 0.2));
                last_block = 1;
            } else { qblk = X_SEG; tblk = X_SEG; last_block = 0; }
            XBlockOut o;
            xdrop_block(q, qidx, qblk, t, tidx, tblk, L, o);
            ++nblocks;
            R.blocks += 1;
            if (o.overflow) { atomicExch(err_flag, 1); break; }
            const int full_map = (qblk - o.ae <= 20 || tblk - o.be <= 20);
            if (!full_map || last_block) {      // the whole block string is appended
                if (o.n > 0) {
                    R.columns += o.n; R.matches += o.nmatch; R.qbases += o.ae; R.tbases += o.be;
                    R.last_q = o.l0q; R.last_t = o.l0t; R.last_m = o.l0m;
                }
                break;
            }
            if (!o.trim_ok) break;
            const int kept = o.n - o.acnt;
            if (kept > 0) {
                R.columns += kept; R.matches += o.nmatch - o.mtail; R.qbases += o.ae - o.qcnt; R.tbases += o.be - o.tcnt;
                R.last_q = o.l1q; R.last_t = o.l1t; R.last_m = o.l1m;
            }
            qidx += o.ae - o.qcnt;
            tidx += o.be - o.tcnt;
        }
        dres[unit] = R;
    }
    atomicAdd(&counters[3], nblocks);
}

// XdropAligner::go tail (xdrop_gapalign.cpp:396-438): the left half is emitted without its last column
__global__ void xd_stitch(const mhip_aln_job* __restrict__ jobs, const XDir* __restrict__ dres, int n, int min_aln,
                          mhip_aln_result* __restrict__ out, unsigned long long* __restrict__ counters) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    XDir L = dres[2 * i];
    const XDir R = dres[2 * i + 1];
    if (L.columns > 0) { L.columns -= 1; L.qbases -= L.last_q; L.tbases -= L.last_t; L.matches -= L.last_m; }
    mhip_aln_result r;
    r.query_start = jobs[i].qstart - L.qbases;
    r.target_start = jobs[i].sstart - L.tbases;
    r.query_end = jobs[i].qstart + R.qbases;
    r.target_end = jobs[i].sstart + R.tbases;
    r.matches = L.matches + R.matches;
    r.columns = L.columns + R.columns;
    r.blocks = L.blocks + R.blocks;
    r.ok = (r.query_end - r.query_start) >= min_aln;      // :438
    out[i] = r;
    if (r.ok) {
        atomicAdd(&counters[6], (unsigned long long)(r.query_end - r.query_start));
        atomicAdd(&counters[7], 1ull);
    }
}

extern "C" {

int mhip_xalign_candidates_dev(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n,
                               int min_align_size, void* d_out) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    int nthreads = c->num_cus * 4 * XB_BLOCK;
    nthreads = std::min(nthreads, ((2 * n + XB_BLOCK - 1) / XB_BLOCK) * XB_BLOCK);
    XDir* d_dres;
    uint8_t* d_s;
    unsigned int* d_cur;
    if (c->scratch("xa_dres", sizeof(XDir) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("xa_lanes", X_LANE_BYTES * (size_t)nthreads, (void**)&d_s)) return -1;
    if (c->scratch("xa_cursor", 64, (void**)&d_cur)) return -1;
    HIPCHK(hipMemsetAsync(d_cur, 0, 8, c->stream));
    LAUNCH(c, "xd_extend", xd_extend, nthreads / XB_BLOCK, XB_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
           (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_s, d_cur,
           (int*)(d_cur + 1), (unsigned long long*)c->d_counters);
    LAUNCH(c, "xd_stitch", xd_stitch, (n + 255) / 256, 256, 0, (const mhip_aln_job*)d_jobs, (const XDir*)d_dres, n, min_align_size,
           (mhip_aln_result*)d_out, (unsigned long long*)c->d_counters);
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, d_cur + 1, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (err) { mhip_set_error("X-drop aligner: traceback scratch overflow (a block needed more than %d script bytes)", X_STATE_CAP); return -1; }
    return 0;
}

int mhip_xalign_candidates(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs, int n,
                           int min_align_size, mhip_aln_result* out) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    for (int i = 0; i < n; ++i) {
        const mhip_aln_job& j = jobs[i];
        if (j.qid_local < 0 || j.qid_local >= reads->num_reads || j.sid_local < 0 || j.sid_local >= ref->num_reads) {
            mhip_set_error("alignment job %d: read index out of range", i);
            return -1;
        }
        const int qs = reads->h_offs[(size_t)j.qid_local].size, ts = ref->h_offs[(size_t)j.sid_local].size;
        if (j.qstart < 0 || j.qstart > qs || j.sstart < 0 || j.sstart > ts) {
            mhip_set_error("alignment job %d: start point outside the reads", i);
            return -1;
        }
    }
    mhip_aln_job* d_jobs;
    mhip_aln_result* d_out;
    if (c->scratch("al_jobs", sizeof(mhip_aln_job) * (size_t)n, (void**)&d_jobs)) return -1;
    if (c->scratch("al_out", sizeof(mhip_aln_result) * (size_t)n, (void**)&d_out)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, sizeof(mhip_aln_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (mhip_xalign_candidates_dev(c, ref, reads, d_jobs, n, min_align_size, d_out)) return -1;
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_aln_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
