// align.hip — "dw": banded O(ND) furthest-reaching d-path local aligner in 500-bp blocks.
//
// Replaces, for PacBio-mode candidates (pairwise_mapping, mecat2pw/pw_impl.cpp:674-698):
//   DiffAligner::go          common/diff_gapalign.cpp:294-349   left + right extension from the seed point, stitching
//   dw_in_one_direction      common/diff_gapalign.cpp:221-292   block chaining with the 4-match tail anchor
//   retrieve_next_aln_block  common/gapalign.cpp:9-45           500-bp blocks, last-block sizing
//   Align                    common/diff_gapalign.cpp:107-219   O(ND) wavefront with the adaptive band
//   GetAlignString           common/diff_gapalign.cpp:39-104    traceback
//   trim_mismatch_end        common/gapalign.cpp:47-68          tail anchor
//
// mecat2pw consumes only the four end coordinates and the identity (matches / columns) of an alignment, never the
// aligned strings, so the traceback is reduced to what those need.  An O(ND) path is a chain of d+1 match runs
// ("snakes") separated by single-base indels, hence per block
//     columns = (x + y + d) / 2,  matches = (x + y - d) / 2                       (Align: aln_str_size)
// and trim_mismatch_end (4 consecutive equal columns from the tail) is "walk the path back to the first snake of
// length >= 4".  Only that tail of the path is traced.
//
// Mapping: one wave per (candidate, direction); a direction is a sequential chain of blocks; inside a block the
// d-loop is sequential and the <= 217 diagonals of the band are the lanes (up to 4 per lane).  Per-wave LDS (9.2 KB):
//   V[]      furthest x per diagonal (int16; x + y is recomputed as 2x - k, no U array)
//   Qp/Tp    the two block sequences, 2 bits per base MSB-first, staged straight from the packed volume with two
//            word loads per lane (reverse / reverse-complement views by a 2-bit-group reversal and a bit complement)
//   ring     the most recent 16 d-rows (u16 x per diagonal) + their band limits, for the tail traceback
// Snakes compare 16 bases per step: XOR of two unaligned 32-bit windows + count-leading-zeros.  A path whose last
// >= 4 snake is more than 16 rows back re-runs the block with all rows spilled to a per-wave global scratch.
// Waves are persistent and pull (candidate, direction) units from an atomic cursor; a second tiny kernel stitches the
// two directions (query_start = qstart - left bases, ...).
//
// Roofline: integer compare + LDS; HBM traffic is the two block sequences (2 bits per base) and a 32-byte result, so the
// HBM fraction is small by construction; cells and snake bases are counted for the VALU/LDS view.
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <type_traits>

#include "common.h"
#include "cns_fwd.h"

#define AL_BLOCK 256
#define AL_WAVES (AL_BLOCK / WAVE)
#define SEG_BLK 500            // DiffAlignParameters::segment_size, diff_gapalign.h:35
#define MAX_BLK 736            // last block (gapalign.cpp:24-30): one side < 600, the other <= int(599 * 1.2) = 718
#define SEQ_WORDS 52           // 736 / 16 = 46 words + window slack
#define MAX_D 400              // int(0.3 * (599 + 718)) = 395
#define VU_LEN (2 * MAX_D + 8)
#define ROW_W 224              // diagonals per d-row: (2 * int(0.3 * 718)) / 2 + 1 = 216
#define RING 16
#define GROW_STRIDE ((size_t)MAX_D * ROW_W + 2 * (MAX_D + 8))   // u16 per wave: rows + rmin + rmax

struct DirResult {
    int32_t qbases, tbases, matches, columns, blocks, pad;
};

// a unit dw_extend2 could not finish (a block whose tail traceback needs a row that left the ring, or that never reached an end
// of either sequence): where it stands, for dw_extend to take over
struct DwHandover {
    uint32_t unit;
    int32_t qidx, tidx;
    DirResult R;
};

struct AlnWaveLds {
    uint32_t Qp[SEQ_WORDS];     // staged block sequences first: their byte offsets fit the ds_read2 offset field
    uint32_t Tp[SEQ_WORDS];
    int16_t V[VU_LEN];
    uint16_t ring[RING * ROW_W];
    int16_t rmin[RING];
    int16_t rmax[RING];
};

#include "dw_helpers.h"

// One d-row for NJ diagonals per lane (straight-line: lanes past the last diagonal recompute the last one -- same
// addresses, same values -- so nothing is predicated per lane).  Returns the wave-reduced keys
//   bkey = (x+y) << 10 | (1023 - (k + k_offset))   max  -> first maximum of x + y in k order (diff_gapalign.cpp:160-167)
//   hkey = (k + k_offset) << 10 | x                min  -> lowest diagonal that reached an end (:168-169), or INT_MAX
// and leaves x + y of the lane's diagonals in S-independent registers via xy[] for the band update.
template <int NJ, bool SPILL, bool LE>
__device__ __forceinline__ void row_body(AlnWaveLds& S, uint16_t* __restrict__ grow, const int lane, const int d, const int nslot,
                                         const int min_k, const int max_k, const int k_offset, const int q_len, const int t_len,
                                         const int best_m, const int band_tol, unsigned int& snake, int& bkey_out, int& hkey_out,
                                         int& x0_out) {
    int xs[NJ], ys[NJ], kks[NJ];
    // start points (:138-142)
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = min(lane + 64 * j, nslot - 1);
        const int k = min_k + 2 * t;
        kks[j] = k + k_offset;
        const int vl = S.V[k - 1 + k_offset], vr = S.V[k + 1 + k_offset];
        const int x = (k == min_k || (k != max_k && vl < vr)) ? vr : vl + 1;
        xs[j] = x; ys[j] = x - k;
    }
    // snakes, 32 bases per round; a round is idempotent for a diagonal that already stopped
    bool more;
    do {
        more = false;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int x = xs[j], y = ys[j];
            const int lim = min(q_len - x, t_len - y);
            const int n0 = LE ? match32_le(S.Qp, x, S.Tp, min(y, MAX_BLK)) : match32(S.Qp, x, S.Tp, min(y, MAX_BLK));
            const int n = max(0, min(n0, lim));
            xs[j] = x + n; ys[j] = y + n;
            snake += (unsigned int)n;
            more |= (n == 32) & (lim > 32);
        }
    } while (__ballot(more));
    __builtin_amdgcn_wave_barrier();
    int bkey = -1, hkey = 0x7fffffff;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int t = min(lane + 64 * j, nslot - 1);
        S.V[kks[j]] = (int16_t)xs[j];
        if (SPILL) grow[(size_t)d * ROW_W + t] = (uint16_t)xs[j];
        else S.ring[(d % RING) * ROW_W + t] = (uint16_t)xs[j];
        bkey = max(bkey, ((xs[j] + ys[j]) << 10) | (1023 - kks[j]));
        hkey = min(hkey, (xs[j] >= q_len || ys[j] >= t_len) ? ((kks[j] << 10) | xs[j]) : 0x7fffffff);
    }
    bkey_out = wave_max(bkey);
    hkey_out = wave_min(hkey);
    x0_out = xs[0];
}

// Fast d-row: at most 64 diagonals and the band's left edge moved by -1 (no pruning on the left, SHL = false) or by +1
// (one diagonal pruned on the left, SHL = true).  The previous row is still in a register: diagonal k of this row sits
// in the lane that held k+1 (SHL = false) resp. k-1 (SHL = true) in the previous row, and the other neighbour is one
// lane away: one wave_shr:1 / wave_shl:1 DPP move replaces the two LDS reads of V.  The lowest end-reaching diagonal
// and the first maximum are found with ballots (lane order == diagonal order) instead of key reductions.
template <bool SPILL, bool SHL, bool LE>
__device__ __forceinline__ void row_fast(AlnWaveLds& S, uint16_t* __restrict__ grow, const int lane, const int d, const int nslot,
                                         const int min_k, const int max_k, const int k_offset, const int q_len, const int t_len,
                                         unsigned int& snake, int& xreg, int& m_out, int& bkey_out, int& hkey_out) {
    const bool act = lane < nslot;
    const int k = min_k + 2 * lane;
    int vl, vr;
    if (SHL) { vl = xreg; vr = __builtin_amdgcn_update_dpp(0, xreg, 0x130 /* wave_shl:1 */, 0xF, 0xF, false); }
    else { vr = xreg; vl = __builtin_amdgcn_update_dpp(0, xreg, 0x138 /* wave_shr:1 */, 0xF, 0xF, false); }
    int x = (lane == 0 || (k != max_k && vl < vr)) ? vr : vl + 1;       // :138-142
    x = act ? x : 0;
    int y = act ? x - k : 0;
    bool more;
    do {
        const int lim = min(q_len - x, t_len - y);
        const int n0 = LE ? match32_le(S.Qp, x, S.Tp, min(y, MAX_BLK)) : match32(S.Qp, x, S.Tp, min(y, MAX_BLK));
        const int n = max(0, min(n0, lim));
        x += n; y += n;
        snake += (unsigned int)n;           // inactive lanes sit at (0, 0) of both sequences; corrected below
        more = (n == 32) & (lim > 32);
    } while (__ballot(more) & (nslot >= 64 ? ~0ull : ((1ull << nslot) - 1ull)));
    __builtin_amdgcn_wave_barrier();
    if (act) {
        S.V[k + k_offset] = (int16_t)x;
        if (SPILL) grow[(size_t)d * ROW_W + lane] = (uint16_t)x;
        else S.ring[(d % RING) * ROW_W + lane] = (uint16_t)x;
    } else snake -= (unsigned int)x;        // what the idle lane "matched" from (0, 0)
    const unsigned long long amask = nslot >= 64 ? ~0ull : ((1ull << nslot) - 1ull);
    const int m = act ? x + y : -1;
    const int rm = wave_max(m);
    const int lb = __ffsll((unsigned long long)(__ballot(m == rm) & amask)) - 1;          // first maximum in k order
    bkey_out = (rm << 10) | (1023 - (min_k + 2 * lb + k_offset));
    const unsigned long long hb = __ballot(x >= q_len || y >= t_len) & amask;              // lowest diagonal that reached an end
    if (hb) {
        const int lh = __ffsll(hb) - 1;
        hkey_out = ((min_k + 2 * lh + k_offset) << 10) | __builtin_amdgcn_readlane(x, lh);
    } else hkey_out = 0x7fffffff;
    xreg = x;
    m_out = m;
}

// band update (:172-179) on U[k] = x + y = 2 V[k] - k of the row just written.  Neutral elements are the reference's
// initial values (new_min_k = max_k, new_max_k = min_k).
template <bool SPILL>
__device__ __forceinline__ void band_keys(AlnWaveLds& S, const int lane, const int nj, const int nslot, const int min_k, const int max_k,
                                          const int k_offset, const int best_m, const int band_tol, int& nmin, int& nmax) {
    int lo = max_k, hi = min_k;
    for (int j = 0; j < nj; ++j) {
        const int t = min(lane + 64 * j, nslot - 1);
        const int k = min_k + 2 * t;
        const bool q = 2 * (int)S.V[k + k_offset] - k >= best_m - band_tol;
        lo = min(lo, q ? k : max_k);
        hi = max(hi, q ? k : min_k);
    }
    nmin = wave_min(lo);
    nmax = wave_max(hi);
}

struct DwStats { unsigned int rows, fast_rows, wide_rows, unaligned, spills; };

struct BlockOut {
    int aligned_or_best;   // 1 when an alignment exists (aln_str_size > 0 possible)
    int qe, te, dist;      // aln_q_e, aln_t_e, dist
    int qcnt, tcnt, acnt;  // trim_mismatch_end outputs
    int trim_ok;
    int fallback;          // ring too short for the tail traceback
};

// One block: Align + tail traceback + trim_mismatch_end.  SPILL = false: d-rows in the LDS ring; true: in global rows.
// LE: the staged sequences are in dw_extend2's layout (view_word_le), otherwise in dw_extend's (view_word + pad word).
template <bool SPILL, bool LE = false>
__device__ void align_block(AlnWaveLds& S, const int q_len, const int t_len, uint16_t* __restrict__ grow, BlockOut& o,
                            unsigned int& cells, unsigned int& snake, DwStats& st) {
    const int lane = lane_id();
    const int band_tol = (int)(0.3 * (q_len > t_len ? q_len : t_len));      // dw_in_one_direction passes 0.3 * max(qblk, tblk)
    const int max_d = (int)(.3 * (q_len + t_len));
    const int k_offset = max_d;
    const int band_size = band_tol * 2;
    int16_t* g_rmin = (int16_t*)(grow + (size_t)MAX_D * ROW_W);
    int16_t* g_rmax = g_rmin + (MAX_D + 8);
    for (int i = lane; i < 2 * max_d + 4 && i < VU_LEN; i += 64) S.V[i] = 0;     // V is zero-filled per block (:232-233)
    __builtin_amdgcn_wave_barrier();
    int best_m = -1, best_x = -1, best_y = -1, best_d = 0, best_k = 0;
    int min_k = 0, max_k = 0;
    int aligned = 0, end_x = 0, end_y = 0, end_d = 0, end_k = 0;
    o.fallback = 0;
    int d, last_row = -1;
    int xreg = 0, reg_min_k = 0;      // previous row's x per lane (valid when that row had <= 64 diagonals)
    bool reg_ok = false;
    for (d = 0; d < max_d; ++d) {
        if (max_k - min_k > band_size) break;
        last_row = d;
        const int nslot = (max_k - min_k) / 2 + 1;
        const int nj = (nslot + 63) >> 6;            // diagonals per lane this row (uniform)
        if (lane == 0) {
            if (SPILL) { g_rmin[d] = (int16_t)min_k; g_rmax[d] = (int16_t)max_k; }
            else { S.rmin[d % RING] = (int16_t)min_k; S.rmax[d % RING] = (int16_t)max_k; }
        }
        int bkey, hkey, x0 = 0, mreg = 0;
        const bool fast_r = nj == 1 && reg_ok && min_k == reg_min_k - 1;
        const bool fast_l = nj == 1 && reg_ok && min_k == reg_min_k + 1 && nslot <= 63;
        const bool fast = fast_r || fast_l;
        if (fast_r) {
            row_fast<SPILL, false, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, snake, xreg, mreg, bkey, hkey);
        } else if (fast_l) {
            row_fast<SPILL, true, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, snake, xreg, mreg, bkey, hkey);
        } else {
            switch (nj) {
            case 1: row_body<1, SPILL, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, best_m, band_tol, snake, bkey, hkey, x0); break;
            case 2: row_body<2, SPILL, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, best_m, band_tol, snake, bkey, hkey, x0); break;
            case 3: row_body<3, SPILL, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, best_m, band_tol, snake, bkey, hkey, x0); break;
            default: row_body<4, SPILL, LE>(S, grow, lane, d, nslot, min_k, max_k, k_offset, q_len, t_len, best_m, band_tol, snake, bkey, hkey, x0); break;
            }
            xreg = x0;
        }
        bkey = __builtin_amdgcn_readfirstlane(bkey);      // wave-uniform: keep the bookkeeping on the scalar unit
        hkey = __builtin_amdgcn_readfirstlane(hkey);
        reg_ok = nj == 1;
        reg_min_k = min_k;
        st.rows += 1; st.fast_rows += fast ? 1u : 0u; st.wide_rows += nj > 1 ? 1u : 0u;
        cells += (unsigned int)nslot;
        {
            const int rm = bkey >> 10, rk = 1023 - (bkey & 1023) - k_offset;
            if (rm > best_m) { best_m = rm; best_x = (rm + rk) / 2; best_y = best_x - rk; best_d = d; best_k = rk; }
        }
        // band update (:172-179): needs the new best_m.  When both outermost diagonals qualify, so does the whole range.
        int nmin, nmax;
        bool whole = false;
        if (fast) {
            const int m0 = __builtin_amdgcn_readlane(mreg, 0), ml = __builtin_amdgcn_readlane(mreg, nslot - 1);
            whole = m0 >= best_m - band_tol && ml >= best_m - band_tol;
        }
        if (whole) { nmin = min_k; nmax = max_k; }
        else band_keys<SPILL>(S, lane, nj, nslot, min_k, max_k, k_offset, best_m, band_tol, nmin, nmax);
        max_k = __builtin_amdgcn_readfirstlane(nmax) + 1;
        min_k = __builtin_amdgcn_readfirstlane(nmin) - 1;
        __builtin_amdgcn_wave_barrier();
        if (hkey != 0x7fffffff) {
            aligned = 1; end_k = (hkey >> 10) - k_offset; end_x = hkey & 1023; end_y = end_x - end_k; end_d = d;
            break;
        }
    }
    o.trim_ok = 0; o.qcnt = o.tcnt = o.acnt = 0;
    if (!aligned) {
        st.unaligned += 1;
        if (best_x > 0) { end_x = best_x; end_y = best_y; end_d = best_d; end_k = best_k; }
        else { o.aligned_or_best = 0; o.qe = o.te = o.dist = 0; return; }
    }
    o.aligned_or_best = 1; o.qe = end_x; o.te = end_y; o.dist = end_d;
    // ---- tail traceback == trim_mismatch_end(.., 4, ..) on the alignment string (gapalign.cpp:47-68)
    const int aln_size = (end_x + end_y + end_d) / 2;
    int cd = end_d, ck = end_k, cx2 = end_x;
    int qcnt = 0, tcnt = 0, acnt = 0, found = 0;
    while (true) {
        int x1, pre_k = 0, takes_q = 0;
        if (cd == 0) x1 = 0;   // V is zero-filled: the d = 0 point starts at (0, 0)
        else {
            if (!SPILL && last_row - (cd - 1) >= RING) { o.fallback = 1; return; }
            int pmin, pmax, cmin, cmax, vl = 0, vr = 0;
            const int kl = ck - 1, kr = ck + 1;
            // values the forward pass read: row d-1 inside its band (always the case), else the zero fill
            if (SPILL) {
                pmin = g_rmin[cd - 1]; pmax = g_rmax[cd - 1]; cmin = g_rmin[cd]; cmax = g_rmax[cd];
                const uint16_t* prow = grow + (size_t)(cd - 1) * ROW_W;
                if (kl >= pmin && kl <= pmax) vl = prow[(kl - pmin) >> 1];
                if (kr >= pmin && kr <= pmax) vr = prow[(kr - pmin) >> 1];
            } else {
                pmin = S.rmin[(cd - 1) % RING]; pmax = S.rmax[(cd - 1) % RING]; cmin = S.rmin[cd % RING]; cmax = S.rmax[cd % RING];
                const uint16_t* prow = S.ring + ((cd - 1) % RING) * ROW_W;
                if (kl >= pmin && kl <= pmax) vl = prow[(kl - pmin) >> 1];
                if (kr >= pmin && kr <= pmax) vr = prow[(kr - pmin) >> 1];
            }
            if (ck == cmin || (ck != cmax && vl < vr)) { x1 = vr; pre_k = kr; takes_q = 0; }
            else { x1 = vl + 1; pre_k = kl; takes_q = 1; }
        }
        const int s = cx2 - x1;     // snake length
        if (s >= 4) { acnt += 4; qcnt += 4; tcnt += 4; found = 1; break; }
        acnt += s; qcnt += s; tcnt += s;
        if (cd == 0) break;
        acnt += 1;                  // the indel column
        if (takes_q) { qcnt += 1; cx2 = x1 - 1; } else { tcnt += 1; cx2 = x1; }
        ck = pre_k;
        --cd;
    }
    o.qcnt = qcnt; o.tcnt = tcnt; o.acnt = acnt;
    o.trim_ok = found && (aln_size - acnt >= 2);     // "m == mat_cnt && k > 0"
}

__global__ __launch_bounds__(AL_BLOCK) void dw_extend(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                      const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                      const mhip_aln_job* __restrict__ jobs, int n, DirResult* __restrict__ dres,
                                                      uint16_t* __restrict__ gscratch, unsigned int* __restrict__ cursor,
                                                      unsigned long long* __restrict__ counters, const DwHandover* __restrict__ hand,
                                                      const unsigned int* __restrict__ hand_count) {
    __shared__ AlnWaveLds lds[AL_WAVES];
    AlnWaveLds& S = lds[threadIdx.x >> 6];
    const int lane = lane_id();
    const int gw = blockIdx.x * AL_WAVES + (threadIdx.x >> 6);
    uint16_t* grow = gscratch + (size_t)gw * GROW_STRIDE;
    unsigned long long cells = 0, snake = 0, nblocks = 0, nfallback = 0;
    unsigned int ucells = 0, usnake = 0;
    DwStats st = {0, 0, 0, 0, 0};
    while (true) {
        unsigned int unit = 0;
        if (lane == 0) unit = atomicAdd(cursor, 1u);
        unit = __shfl(unit, 0);
        int qidx = 0, tidx = 0;
        DirResult R = {0, 0, 0, 0, 0, 0};
        if (hand) {                                   // second launch: the units dw_extend2 handed over, from where they stand
            if (unit >= *hand_count) break;
            const DwHandover h = hand[unit];
            unit = h.unit; qidx = h.qidx; tidx = h.tidx; R = h.R;
        } else if (unit >= 2u * (unsigned)n) break;
        const mhip_aln_job jb = jobs[unit >> 1];
        const int right = unit & 1;
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        // extension views: logical position i of the query is strand position qstart + i (right) or qstart - 1 - i (left);
        // strand position p of a reverse-complemented query is original index qsize - 1 - p, complemented
        SeqView q, t;
        q.pac = qpac; q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
        t.pac = rpac; t.off = roffs[jb.sid_local].offset; t.comp = 0;
        int query_size, target_size;
        const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
        if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
        t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
        if (right) { query_size = qsize - jb.qstart; target_size = tsize - jb.sstart; }
        else { query_size = jb.qstart; target_size = jb.sstart; }
        // a start point in front of a read (the reference's wrapped `short` seed numbers give a read beyond 327 670 bases negative query
        // starts, pw_impl.cpp:388; there the reference extends from in front of its buffer): nothing to extend here
        if (jb.qstart < 0 || jb.sstart < 0) { query_size = 0; target_size = 0; }
        while (true) {
            // retrieve_next_aln_block (gapalign.cpp:9-45)
            const int qleft = query_size - qidx, tleft = target_size - tidx;
            int qblk, tblk, last_block;
            if (qleft < SEG_BLK + 100 || tleft < SEG_BLK + 100) {
                qblk = min(qleft, (int)(tleft + tleft * 0.2));
                tblk = min(tleft, (int)(qleft + qleft * 0.2));
                last_block = 1;
            } else { qblk = SEG_BLK; tblk = SEG_BLK; last_block = 0; }
            if (qblk < 0) qblk = 0;
            if (tblk < 0) tblk = 0;
            __builtin_amdgcn_wave_barrier();
            if (lane < SEQ_WORDS) {     // word 0 is the pad word, word 1 + j holds logical bases 16j .. 16j+15
                S.Qp[lane] = (lane > 0 && (lane - 1) * 16 < qblk + 32) ? view_word(q, qidx + (lane - 1) * 16) : 0u;
                S.Tp[lane] = (lane > 0 && (lane - 1) * 16 < tblk + 32) ? view_word(t, tidx + (lane - 1) * 16) : 0u;
            }
            __builtin_amdgcn_wave_barrier();
            BlockOut o;
            align_block<false>(S, qblk, tblk, grow, o, ucells, usnake, st);
            if (o.fallback) { ++nfallback; align_block<true>(S, qblk, tblk, grow, o, ucells, usnake, st); }
            ++nblocks;
            R.blocks += 1;
            if (!o.aligned_or_best || !o.trim_ok) break;
            const int full_map = (qblk - o.qe <= 20 || tblk - o.te <= 20);
            int qcnt = o.qcnt, tcnt = o.tcnt, acnt = o.acnt;
            if (last_block || !full_map) { qcnt -= 4; tcnt -= 4; acnt -= 4; }
            const int cols = (o.qe + o.te + o.dist) / 2 - acnt;
            const int mcut = qcnt + tcnt - acnt;
            R.columns += cols;
            R.matches += (o.qe + o.te - o.dist) / 2 - mcut;
            R.qbases += o.qe - qcnt;
            R.tbases += o.te - tcnt;
            if (last_block || !full_map) break;
            qidx += o.qe - qcnt;
            tidx += o.te - tcnt;
        }
        if (lane == 0) dres[unit] = R;
        cells += ucells; snake += usnake;
        ucells = 0; usnake = 0;
    }
    // snake bases are counted per lane
    for (int off = 32; off > 0; off >>= 1) snake += __shfl_xor(snake, off);
    if (lane == 0) {
        atomicAdd(&counters[3], nblocks);
        atomicAdd(&counters[4], cells);
        atomicAdd(&counters[5], snake);
        atomicAdd(&counters[8], nfallback);       // debug slots: blocks re-run with spilled rows, rows, fast rows, wide rows, unaligned blocks
        atomicAdd(&counters[9], (unsigned long long)st.rows);
        atomicAdd(&counters[10], (unsigned long long)st.fast_rows);
        atomicAdd(&counters[11], (unsigned long long)st.wide_rows);
        atomicAdd(&counters[12], (unsigned long long)st.unaligned);
    }
}

// ------------------------------------------------------------------------------------------------------------------
// dw_extend2 — two (candidate, direction) units per wave, one per 32-lane half.  (-DMECAT_DW_STATS: per-row debug counters.)
//
// The adaptive band keeps ~26 diagonals alive on average (config 2: 6.0e9 rows, 1.5e11 cells), so a whole wave per
// unit leaves 60 % of the lanes idle and the kernel is VALU-issue bound.  Here each half-wave runs its own unit with its
// own block and row counter: one pass over the row code advances both halves by one d-row of their respective blocks; a
// half whose rows ended does its tail traceback, accounting and next block setup (or pulls a new unit) while the other
// half is masked off, and rejoins the row code at its row 0.  Everything that is a
// scalar in the one-unit kernel (band limits, best point, block sizes ...) is a per-half-uniform VGPR value here (the band's
// first diagonal and the two ring positions with the lane's own offset already added); what steers the row loops — rows left, slot
// counts, lane masks of a pass — lives on the scalar unit.  Per-half row maxima: four mirrored DPP steps + v_permlane16_swap;
// first / last qualifying diagonal of the band update from the two 32-bit halves of a ballot.  d-rows live in a 1024-entry
// circular buffer per half, packed back to back with one entry between two rows that is the row record (see row_passes): ~37 rows
// stay traceable; the rare block whose tail traceback needs an overwritten row, or that never reached an end of either sequence,
// is handed over to the one-unit kernel (dw_extend), which keeps every row.
#ifndef RCAP
#define RCAP 1024
#endif
#define SEQ_WORDS2 48          // 736 bases + one 16-base window, 16 bases per word, no pad word
// Per-half LDS, 2.4 KB incl. the ring below (19 KB per workgroup of four waves: eight workgroups per CU).  There is no V[] array: row d
// reads the furthest x of diagonals k - 1 and k + 1 of row d - 1 straight from that row's entries in the ring, at pbase + tt and
// pbase + tt + 1 (inside the band they are entries of that row; at its two edges the entry between rows or a dropped diagonal: see
// the start point in row_passes).
struct HalfLds {
    uint32_t Qp[SEQ_WORDS2];
    uint32_t Tp[SEQ_WORDS2];
};
// The ring of d-rows of a half: 16-bit rows packed back to back, wrapping (in front of row 0: -1, 0, -1, see the block set-up).
// It is its own 2 KB-aligned LDS array so that the address of ring byte position p is `base | (p & 0x7fe)`: one v_and_or_b32.
// Positions (lin, pbase, rlin) are kept in bytes.
typedef __attribute__((address_space(3))) uint16_t lds_u16_t;
typedef __attribute__((address_space(3))) int16_t lds_i16_t;
__device__ __forceinline__ int ring_ld(uint32_t base, uint32_t pos) { return (int)*(lds_i16_t*)(uintptr_t)(base | (pos & (2 * RCAP - 2))); }
__device__ __forceinline__ void ring_st(uint32_t base, uint32_t pos, int x) { *(lds_u16_t*)(uintptr_t)(base | (pos & (2 * RCAP - 2))) = (uint16_t)x; }

// mask of the lanes where p holds, without the bool -> int -> compare round trip of __ballot
#define BALLOT(p) __builtin_amdgcn_ballot_w64(p)

// max over each 32-lane half, returned in every lane of the half: four mirrored DPP steps leave each 16-lane row with its
// maximum in every lane; v_permlane16_swap (gfx950) then exchanges rows 1 <-> 0 and 3 <-> 2 between two copies.
// Written with the DPP builtin (old = the identity of max, so the DPP combiner fuses mov_dpp + max into one v_max_i32_dpp) rather
// than as one asm block: the compiler then fills the two wait states each step needs with independent instructions of the row
// instead of s_nops (the first step is asm, three-address: its input stays live for the band update without a copy).
template <int CTRL> __device__ __forceinline__ int dpp_max_step(int v) {
    return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ int half_max(int v0) {
    int v;                               // quad_perm:[1,0,3,2], into a register of its own: the caller keeps v0 (no copy in front of the chain)
    asm("s_nop 1\n\tv_max_i32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(v) : "v"(v0));
    v = dpp_max_step<0x4E>(v);           // quad_perm:[2,3,0,1]
    v = dpp_max_step<0x141>(v);          // row_half_mirror
    v = dpp_max_step<0x140>(v);          // row_mirror: every lane of a 16-lane row holds the row's maximum
    int r = v, w = v;
    // rows 1 <-> 0 and 3 <-> 2 between the two copies (the wait states around it are not known to the compiler's hazard recogniser)
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(r), "+v"(w));
    return max(r, w);
}
__device__ __forceinline__ int half_min(int v) { return -half_max(-v); }

// the low min(max(n, 0), 32) bits set; n wave-uniform (from v_readlane): three scalar instructions (written out: the compiler turns
// the clamp into a vector v_med3)
__device__ __forceinline__ unsigned int lowbits32(int n) {
    unsigned long long r;
    int c;
    asm("s_min_i32 %1, %2, 32\n\ts_max_i32 %1, %1, 0\n\ts_bfm_b64 %0, %1, 0" : "=s"(r), "=&s"(c) : "s"(n) : "scc");
    return (unsigned int)r;
}

// n bits set from bit `off` up (s_bfm_b64; 0 <= n <= 32, off 0 or 32: a half's lane mask already in its half of the wave's mask)
template <int OFF> __device__ __forceinline__ unsigned long long bits_at(int n) {
    unsigned long long r;
    asm("s_bfm_b64 %0, %1, %2" : "=s"(r) : "s"(n), "n"(OFF));
    return r;
}

// the same for any n: min(max(n, 0), 32) bits (the clamp written out on the scalar unit, like lowbits32)
template <int OFF> __device__ __forceinline__ unsigned long long clamped_bits_at(int n) {
    unsigned long long r;
    int c;
    asm("s_min_i32 %1, %2, 32\n\ts_max_i32 %1, %1, 0\n\ts_bfm_b64 %0, %1, %3" : "=s"(r), "=&s"(c) : "s"(n), "n"(OFF) : "scc");
    return r;
}

#ifndef DW2_WAVES_PER_SIMD
#define DW2_WAVES_PER_SIMD 8
#endif
// CNS = false: mecat2pw's aligner (DiffAligner::go).  CNS = true: the forward rows of mecat2cns' re-aligner (dw.cpp:146-375; cns_fwd.h) —
// the same d-rows under its block rules (square blocks of 500, or the rest when it is <= 600; max_d from the error rate; a block that
// reaches no end ends the direction: no best-point fallback; every block but the last is cut in front of its last four matches), plus
// a 16-byte record per row in the row log and a record per block that contributes columns, for cns_trace to find the paths.
#ifndef CNS_FWD_WAVES_PER_SIMD
#define CNS_FWD_WAVES_PER_SIMD 8
#endif
template <bool CNS>
__global__ __launch_bounds__(AL_BLOCK, CNS ? CNS_FWD_WAVES_PER_SIMD : DW2_WAVES_PER_SIMD) void dw_extend2(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                       const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                       const mhip_aln_job* __restrict__ jobs, int n, DirResult* __restrict__ dres,
                                                       DwHandover* __restrict__ hand, unsigned int* __restrict__ hand_count,
                                                       unsigned int* __restrict__ cursor, unsigned long long* __restrict__ counters,
                                                       const CnsFwdArgs ca) {
    __shared__ HalfLds lds[AL_WAVES][2];
    __shared__ __attribute__((aligned(2 * RCAP))) uint16_t rings[AL_WAVES * 2][RCAP];
    const int lane = lane_id(), hh = lane >> 5, sl = lane & 31;
    HalfLds& S = lds[threadIdx.x >> 6][hh];
    const uint32_t rbase = (uint32_t)(uintptr_t)(lds_u16_t*)&rings[(threadIdx.x >> 6) * 2 + hh][0];
    unsigned int cells = 0, nblocks = 0, nhand = 0;
#ifdef MECAT_DW_STATS
    // development build only (tools/dev/dw_breakdown.sh -> profiles/rNN_dw_row_breakdown.md): where the wave's time and rows go.
    // Clocks are s_memrealtime ticks (100 MHz), summed per section over the wave's life; counts are per dual row.
    unsigned long long nrows = 0, nidle = 0, nwide = 0;
    unsigned long long tk_rows = 0, tk_setup = 0, tk_ended = 0, tk_trace = 0, tk_acct = 0, n_outer = 0, n_setup = 0, n_ended = 0, n_pass3 = 0;
    unsigned long long nh[8] = {0, 0, 0, 0, 0, 0, 0, 0};     // unit rows by band width: <= 8, 16, 24, 32, 48, 64, 96, more
    unsigned long long q16[6] = {0, 0, 0, 0, 0, 0};           // unit rows by ceil(nslot / 16): 1, 2, 3, 4, 5-6, more
    unsigned long long snake2 = 0;                            // snake steps beyond the first (per pass)
    const unsigned long long tk_start = wall_clock64();
#define DWS_T(var) const unsigned long long var = wall_clock64()
#define DWS_ADD(acc, a, b) acc += (b) - (a)
#else
#define DWS_T(var)
#define DWS_ADD(acc, a, b)
#endif

    // per-half unit state (uniform inside a half)
    bool need_unit = true, exhausted = false;
    bool setup = true;          // the half is between blocks
    bool inblock = false;       // the half has a block whose rows are running or have just ended
    unsigned int unit = 0;
    SeqView q, t;
    q.pac = qpac; t.pac = rpac; q.off = 0; t.off = 0; q.A = q.B = t.A = t.B = 0; q.comp = t.comp = 0;
    int query_size = 0, target_size = 0, qidx = 0, tidx = 0;
    int Rq = 0, Rt = 0, Rm = 0, Rc = 0, Rb = 0;
    // per-half block state
    int qblk = 0, tblk = 0, last_block = 0, band_tol = 0, max_d = 0;
    // band = nslot diagonals from min_k; d = rows done.  The three values every lane adds its own 2 * sl to on every row are kept with
    // it added (`_l`: per lane): the row code then uses them as they are, and the band update moves all lanes by the same amount.
    int best_m = -1, mk_l = 0, nslot = 0, aligned = 0, end_x = 0, end_k = 0, end_d = 0, end_mk = 0, end_ns = 0, d = 0;
    int sepv = -1;              // what idle lanes store behind the row: ~(slots of the row | left cut of the row before << 8), see row_passes
    int cut = 0;                // diagonals the last band update dropped on the left (`first`)
    unsigned int lin_l = 0;     // ring position behind the last row (bytes, like the next two) + 2 sl
    unsigned int pb_l = 0;      // ring position of the previous row's entry for diagonal (this row's min_k) - 1, + 2 sl
    unsigned int rlin_l = 0;    // ring position of the row that ran last, + 2 sl
    int dlim = 0;               // rows run while d < dlim: max_d of the block, 0 once an end was reached / without a block
    // CNS: the unit's share of the row log [logpos, logend), the record of the current block's row 0, and whether the unit met something
    // the log cannot hold (a row of more than 64 diagonals, a cut of 127 or more, more rows than its share): such a unit is handed over
    unsigned int logpos = 0, logend = 0, blk_log0 = 0, blkpos = 0, blkend = 0;
    bool cns_bad = false;
    unsigned long long fl0 = 0, fl1 = 0, fl2 = 0;      // lanes whose diagonal came from k - 1, first / second / third pass of the row that ran last
    int last_ns = 0;                           // its slots

    while (true) {
#ifdef MECAT_DW_STATS
        n_outer += 1;
        n_setup += BALLOT(setup) ? 1u : 0u;
#endif
        DWS_T(t_a);
        if (BALLOT(setup)) {
            // ---- 1. a half without a unit pulls the next one
            if (setup && need_unit) {
                unsigned int u = 0;
                if (sl == 0) u = atomicAdd(cursor, 1u);
                u = __shfl(u, hh << 5);
                if (u >= 2u * (unsigned)n) { exhausted = true; setup = false; mk_l = 2 * sl; nslot = 0; dlim = 0; }      // never rowing
                else {
                    unit = u;
                    const mhip_aln_job jb = jobs[u >> 1];
                    const int right = u & 1;
                    const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
                    q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
                    t.off = roffs[jb.sid_local].offset; t.comp = 0;
                    const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
                    if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
                    t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
                    if (right) { query_size = qsize - jb.qstart; target_size = tsize - jb.sstart; }
                    else { query_size = jb.qstart; target_size = jb.sstart; }
                    if (jb.qstart < 0 || jb.sstart < 0) { query_size = 0; target_size = 0; }      // (see dw_extend)
                    qidx = tidx = 0;
                    Rq = Rt = Rm = Rc = Rb = 0;
                    need_unit = false;
                    if (CNS) { logpos = ca.logbase[u]; logend = ca.logbase[u + 1]; blkpos = ca.blockbase[u]; blkend = ca.blockbase[u + 1]; cns_bad = false; }
                }
            }
            // ---- 2. block setup: retrieve_next_aln_block (gapalign.cpp:9-45) + staging
            if (setup) {
                const int qleft = query_size - qidx, tleft = target_size - tidx;
                if (CNS) {      // dw_in_one_direction, dw.cpp:319-332
                    const int ext = min(qleft, tleft);
                    if (ext > SEG_BLK + 100) { qblk = SEG_BLK; last_block = 0; } else { qblk = max(ext, 0); last_block = 1; }
                    tblk = qblk;
                    band_tol = (int)(0.3 * qblk);
                    max_d = (int)(2.0 * ca.error_rate * (qblk + qblk));
                    blk_log0 = logpos;
                    if (logend - logpos < (unsigned)max_d) { cns_bad = true; max_d = 0; }      // the rows of this block may not fit the unit's share of the log
                } else {
                if (qleft < SEG_BLK + 100 || tleft < SEG_BLK + 100) {
                    qblk = min(qleft, (int)(tleft + tleft * 0.2));
                    tblk = min(tleft, (int)(qleft + qleft * 0.2));
                    last_block = 1;
                } else { qblk = SEG_BLK; tblk = SEG_BLK; last_block = 0; }
                qblk = max(qblk, 0);
                tblk = max(tblk, 0);
                band_tol = (int)(0.3 * (qblk > tblk ? qblk : tblk));
                max_d = (int)(.3 * (qblk + tblk));
                }
                for (int w = sl; w < SEQ_WORDS2; w += 32) {      // word w = logical bases 16w .. 16w+15, first base in the low bits
                    S.Qp[w] = (w * 16 < qblk + 32) ? view_word_le(q, qidx + w * 16) : 0u;
                    S.Tp[w] = (w * 16 < tblk + 32) ? view_word_le(t, tidx + w * 16) : 0u;
                }
                // The reference zero-fills V per block (:232-233); row d only reads diagonals written by row d - 1, except row 0,
                // which reads V[k_offset - 1] and V[k_offset + 1]: the two zeros in front of row 0.
                // Between two rows of the ring sits one entry of -1 (see the start point below); in front of row 0: -1, 0, -1 — row 0 reads
                // the first two (x = max(-1 + 1, 0) = 0), row 1 reads the third as the entry "left of row 0".
                if (sl < 3) ring_st(rbase, 2u * sl, sl == 1 ? 0 : -1);
                best_m = -1; mk_l = 2 * sl; nslot = 1; sepv = ~1; cut = 0;
                aligned = 0; end_x = 0; end_k = 0; end_d = 0; d = 0;
                lin_l = 6u + 2u * sl; pb_l = 2u * sl; rlin_l = 2u * sl;
                dlim = max_d; inblock = true;
                setup = false;
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (!BALLOT(!exhausted)) break;
        const int q_len = qblk, t_len = tblk, k_offset = max_d;
        DWS_T(t_b);
        DWS_ADD(tk_setup, t_a, t_b);

        // ---- 3. one row per half (Align, diff_gapalign.cpp:107-219); the halves' row counters are independent.  Only the
        // running maximum of x + y is tracked here; a block that ends without reaching an end of either sequence (0.06 %
        // of blocks) needs the position of that maximum and is handed over to dw_extend, like a block whose traceback outran
        // the ring.  The inner loop runs while every half that has a block is still rowing.
        const unsigned long long inmask = BALLOT(inblock);
        bool row_ok;
        unsigned long long ended = 0;        // lanes whose diagonal reached an end of its block in the row that was just run
        int NJ = 1;
        // band update (:172-179) of the row at ring position rlin; m0 / mp = x + y of the lane's diagonal in the last / previous
        // pass (NJ <= 2), otherwise recomputed from the ring
        // sa / sb: the new slot counts of the two halves for the scalar unit — with both halves in a block and NJ <= 2 straight from
        // the ballots (scalar find-first / find-last), otherwise read back from the vector state
        auto band_update = [&](const int NJ, const int m0, const int mp, auto both_tag, int& sa, int& sb) __attribute__((always_inline)) {
            constexpr bool BOTH = decltype(both_tag)::value;      // both halves are in a block: no per-half guard on the band state
            // qualifying lanes first .. last of the half.  The mask of a half that is in a block is never empty: the lane that holds
            // the row maximum qualifies, and the row maximum is the running maximum (x + y grows by at least one per row along
            // the best path, which the band never prunes).
            int first, last;
            if (NJ == 1) {
                // 32 diagonals per half: one ballot; first / last set bit of the half's 32 bits with the raw instructions
                // (v_ffbl / v_ffbh return -1 for an empty mask: a half without a block, whose values are not used)
                const unsigned long long q0 = BALLOT(m0 >= best_m - band_tol);
                const unsigned int m32 = (unsigned int)(q0 >> (hh << 5));
                int lz;
                asm("v_ffbl_b32 %0, %1" : "=v"(first) : "v"(m32));
                asm("v_ffbh_u32 %0, %1" : "=v"(lz) : "v"(m32));
                last = 31 - lz;
                if (BOTH) {
                    const unsigned int wa = (unsigned int)q0, wb = (unsigned int)(q0 >> 32);
                    sa = 33 - (__builtin_ctz(wa) + __builtin_clz(wa));
                    sb = 33 - (__builtin_ctz(wb) + __builtin_clz(wb));
                }
            } else if (NJ == 2) {
                // up to 64 diagonals per half: two ballots, the half's 2 x 32 qualification bits.  Branch-free: v_ffbl / v_ffbh return
                // -1 for an empty word, which `| 32` leaves at 0xffffffff (first: unsigned min) and `^ 31` turns into -32 (last:
                // signed max; x ^ 31 == 31 - x for x in 0..31)
                const int thr = best_m - band_tol;
                const unsigned long long q0 = BALLOT(mp >= thr), q1 = BALLOT(m0 >= thr);
                if (BOTH) {
                    // first / last qualifying diagonal of each half on the scalar unit (its 64 bits: the half's word of both ballots),
                    // handed to the lanes as one packed value per half: two moves and a select instead of two 64-bit shifts, four
                    // find-first / find-last and their merge
                    const unsigned long long wa = (q0 & 0xffffffffull) | (q1 << 32), wb = (q0 >> 32) | (q1 & 0xffffffff00000000ull);
                    const int fa = __builtin_ctzll(wa), fb = __builtin_ctzll(wb);
                    const int la = 63 - __builtin_clzll(wa), lb = 63 - __builtin_clzll(wb);
                    sa = la - fa + 2;
                    sb = lb - fb + 2;
                    int pka = fa | (sa << 8), pkb = fb | (sb << 8);          // (first, new slot count: last = first + slots - 2)
                    asm("" : "+s"(pka), "+s"(pkb));                          // (kept packed: the compiler would select the four values one by one)
                    const int pk = hh ? pkb : pka;
                    first = pk & 0xff;
                    last = first + (pk >> 8) - 2;
                } else {
                    const unsigned int lo32 = (unsigned int)(q0 >> (hh << 5)), hi32 = (unsigned int)(q1 >> (hh << 5));
                    unsigned int fl, fh, ll, lh;
                    asm("v_ffbl_b32 %0, %1" : "=v"(fl) : "v"(lo32));
                    asm("v_ffbl_b32 %0, %1" : "=v"(fh) : "v"(hi32));
                    asm("v_ffbh_u32 %0, %1" : "=v"(ll) : "v"(lo32));
                    asm("v_ffbh_u32 %0, %1" : "=v"(lh) : "v"(hi32));
                    first = (int)min(fl, fh | 32u);
                    last = max((int)(ll ^ 31u), (int)((lh ^ 31u) | 32u));
                }
            } else {
                int lo = 0x7fffffff, hi = -0x7fffffff;
                for (int j = 0; j < NJ; ++j) {
                    const int tt = sl + 32 * j;
                    const bool act = inblock && tt < nslot;
                    const int k = mk_l + 64 * j;
                    const int u = act ? 2 * ring_ld(rbase, rlin_l + 64u * (unsigned)j) - k : -0x40000000;
                    if (act && u >= best_m - band_tol) { lo = min(lo, tt); hi = max(hi, tt); }
                }
                first = half_min(lo); last = half_max(hi);
            }
            if (BOTH || inblock) {
                // new band [min_k + 2 first - 1, min_k + 2 last + 1] = last - first + 2 diagonals; the previous-row entry of diagonal
                // (new min_k) - 1 sits at ring position rlin + first - 1
                pb_l = rlin_l + 2u * (unsigned)(first - 1);
                nslot = last - first + 2;
                mk_l = mk_l + 2 * first - 1;
                cut = first;
                // what the idle lanes of the next row store behind it: its slot count and this cut (7 bits, saturating: the tail
                // traceback hands a block over when it meets 127), as a negative 16-bit number — which is all the start point needs
                // of the entry between two rows
                int c7;
                asm("v_min_u16 %0, 0x7f, %1" : "=v"(c7) : "v"(first));
                sepv = ~(nslot | (c7 << 8));
            }
            if (!(BOTH && NJ <= 2)) { sa = __builtin_amdgcn_readlane(nslot, 0); sb = __builtin_amdgcn_readlane(nslot, 32); }
        };
        int last_m0 = 0, last_mp = 0;
        unsigned int row_bytes = 0;
        // One d-row of both halves in NJ passes of 32 diagonals per half (NJ a constant for 1 and 2: the pass loop and the previous-pass
        // bookkeeping fold away).  ns_a / ns_b: the two halves' slot counts on the scalar unit.  A row is followed in the ring by one
        // entry of -1, which the first idle lane of the last pass writes: every lane of a pass stores (its x, or -1 when idle) at its
        // own position — no exec juggling around the store, no store of its own for the -1 (unless a half fills its last pass to the
        // last lane: the caller's business, see `full` below).  What idle lanes write beyond the -1 is overwritten by the rows that
        // follow, except for up to 64 entries behind the last row of a block with NJ <= 2: the tail traceback's window is shortened
        // by as much (rows with more passes guard their stores).  FAST 1: 1 <= slots <= 32 in both halves, one pass; FAST 2: 1 <= slots
        // <= 63, two passes; FAST 0: anything.
        auto row_passes = [&](const int NJ, const int ns_a, const int ns_b, auto fast_tag) __attribute__((always_inline)) {
            constexpr int FAST = decltype(fast_tag)::value;

            row_bytes = 2u * (unsigned)nslot + 2u;
            last_ns = nslot;
            int mmax = -0x40000000, m0 = -0x40000000, mp = -0x40000000;
            unsigned long long e = 0;
            int j = 0;
            do {                                 // at least one pass (a pass over an empty band only moves idle lanes)
                const int tt = sl + 32 * j;
                // the active lanes of this pass: the low (slots - 32 j) bits of each half, made on the scalar unit from the two
                // slot counts and used as the lane predicate as it is
                unsigned long long amask;
                if (FAST == 2) {
                    amask = j == 0 ? (bits_at<0>(min(ns_a, 32)) | bits_at<32>(min(ns_b, 32)))
                                   : (bits_at<0>(max(ns_a, 32) - 32) | bits_at<32>(max(ns_b, 32) - 32));
                } else if (FAST == 1) {
                    amask = bits_at<0>(ns_a) | bits_at<32>(ns_b);
                } else {
                    amask = clamped_bits_at<0>(ns_a - 32 * j) | clamped_bits_at<32>(ns_b - 32 * j);
                }
                const bool act = __builtin_amdgcn_inverse_ballot_w64(amask);
                const int k = mk_l + 64 * j;
                const unsigned int rp = pb_l + 64u * (unsigned)j;       // idle lanes read (and ignore) whatever the ring holds there
                const int vl = ring_ld(rbase, rp), vr = ring_ld(rbase, rp + 2u);
                // :138-142 `if (k == min_k || (k != max_k && V[k-1] < V[k+1])) x = V[k+1]; else x = V[k-1] + 1;` as max(vl + 1, vr):
                // inside the band the two are the same thing (vl < vr <=> vr >= vl + 1).  At k == min_k the entry on the left is either
                // the -1 between two rows (vl + 1 = 0 <= vr) or a diagonal of row d - 1 that the band update dropped while its right
                // neighbour stayed: 2 vl - (k - 1) < best - tol <= 2 vr - (k + 1), i.e. vl + 1 < vr; mirrored at k == max_k (vr is -1, or
                // a dropped diagonal with vr < vl + 1).  So the two edge tests fold into the same max — in the 16-bit forms of add and
                // max, which issue at twice the rate of the 32-bit max (the result is >= 0 and comes back zero-extended).
                int x;
                {
                    int v1;
                    asm("v_add_u16 %0, 1, %1" : "=v"(v1) : "v"(vl));
                    asm("v_max_i16 %0, %1, %2" : "=v"(x) : "v"(v1), "v"(vr));
                    if (CNS) {
                        // the step the reference's traceback will read off V (dw.cpp:176-179: from k + 1 when k == min_k, or k != max_k and
                        // V[k - 1] < V[k + 1]): with the entries the ring holds at a band's two edges (see above) that is "vl >= vr"
                        // everywhere, i.e. vl + 1 > vr: one 16-bit compare of what is in registers anyway (idle lanes: masked by the reader,
                        // which knows the row's slots)
                        unsigned long long fb;
                        asm("v_cmp_gt_i16_e64 %0, %1, %2" : "=s"(fb) : "v"(v1), "v"(vr));
                        // (selects of values, not `if (j == 0) fl0 = fb; else ...`: the compiler sinks the three stores of that form into ONE store
                        // through a selected address, which pins fl0 .. fl2 — and with them every read in log_row — to scratch memory)
                        fl0 = j == 0 ? fb : fl0;
                        fl1 = j == 1 ? fb : fl1;
                        fl2 = j == 2 ? fb : fl2;
                    }
                }
                // 0 <= y <= t_len and x <= q_len on every live diagonal (a diagonal at an end stops the block); idle lanes sit
                // at (q_len, 0), where lim == 0
                x = act ? x : q_len;
                // (idle lanes: lim <= 0 whatever y is, so they never move forward and never ask for another step; their window
                // loads may fall outside the staged block or outside the LDS allocation, where reads return 0: nothing of an
                // idle lane is kept — the stored value, the end mask and m0 below are all guarded by `act`)
                int y = x - k;
                int lim, nn;
#ifdef MECAT_DW_STATS
                snake2 -= 1;
#endif
                do {
#ifdef MECAT_DW_STATS
                    snake2 += 1;
#endif
                    lim = min(q_len - x, t_len - y);
                    // 0..15 equal bases, or >= 16 (0x7fffffff) when the whole window matches.  A lane with exactly 16 bases left
                    // that all match asks for one more step, which then moves nothing.
                    const int m = match16_le(S.Qp, x, S.Tp, y);
                    asm("v_min3_i32 %0, %1, %2, 16" : "=v"(nn) : "v"(m), "v"(lim));
                    x += nn; y += nn;
                } while (BALLOT(nn == 16));
                if (NJ <= 2 || tt <= nslot) ring_st(rbase, lin_l + 64u * (unsigned)j, act ? x : sepv);
                // nothing left of the query or of the target on this diagonal
                e |= BALLOT(lim == nn) & amask;
                mp = m0;
                m0 = act ? x + y : -0x40000000;     // also read by the band update (NJ <= 2); idle lanes never qualify
                mmax = NJ == 1 ? m0 : max(mmax, m0);
            } while (++j < NJ);
            ended = e;
            last_m0 = m0; last_mp = mp;
            rlin_l = lin_l;                      // (the caller moves lin_l behind the row and the entry behind it: lin_l += row_bytes)
            __builtin_amdgcn_wave_barrier();
            // running maximum of x + y (:160-167) = the maximum of this row: the best diagonal k* of the row before qualifies for
            // the band, so k* - 1 and k* + 1 are in this row and start at least one further along
            best_m = half_max(mmax);
        };
#ifdef MECAT_DW_STATS
        auto row_stats = [&](unsigned long long rmask, int ns_a, int ns_b, int NJ) {
            nrows += 1;
            nidle += (rmask == ~0ull) ? 0u : 1u;
            nwide += NJ > 1 ? 1u : 0u;
            n_pass3 += NJ > 2 ? 1u : 0u;
            for (int hq = 0; hq < 2; ++hq) {
                if (!((rmask >> (hq << 5)) & 1ull)) continue;      // the half is not rowing
                const int ns = hq ? ns_b : ns_a;
                nh[ns <= 8 ? 0 : ns <= 16 ? 1 : ns <= 24 ? 2 : ns <= 32 ? 3 : ns <= 48 ? 4 : ns <= 64 ? 5 : ns <= 96 ? 6 : 7] += 1;
                const int p16 = (ns + 15) >> 4;
                q16[p16 <= 4 ? p16 - 1 : p16 <= 6 ? 4 : 5] += 1;
            }
        };
#define DWS_ROW(rmask, a, b, nj) row_stats(rmask, a, b, nj)
#else
#define DWS_ROW(rmask, a, b, nj)
#endif
        // CNS: the record of the row that just ran and of the band update behind it (cns_fwd.h).  Every lane of a half stores the same 16
        // bytes at the same place: no lane predicate to set up, and the coalescer makes one write of it.
        auto log_row = [&](const int NJ, auto both_tag) __attribute__((always_inline)) {
            if constexpr (CNS) {
                constexpr bool BOTH = decltype(both_tag)::value;
                if (BOTH || inblock) {
                    // (a block's rows fit the unit's share: checked when the block is set up.  A record holds a row of up to 96 diagonals —
                    // three passes: 0.01 % of the rows have a third — and its cut, which is below its width; a wider row cannot be
                    // recorded: the unit is handed over.)
                    const uint32_t lo = hh ? (uint32_t)(fl0 >> 32) : (uint32_t)fl0, hi = NJ >= 2 ? (hh ? (uint32_t)(fl1 >> 32) : (uint32_t)fl1) : 0u;
                    const uint32_t top = NJ >= 3 ? (hh ? (uint32_t)(fl2 >> 32) : (uint32_t)fl2) : 0u;
                    if (NJ > 3) cns_bad = true;
                    ca.rowlog[logpos] = CnsRowRec{lo, hi, (uint32_t)last_ns | ((uint32_t)cut << 8), top};
                    logpos += 1;
                }
            }
        };
        int ns_a = __builtin_amdgcn_readlane(nslot, 0), ns_b = __builtin_amdgcn_readlane(nslot, 32);
        if (inmask == ~0ull) {
            // Both halves in a block (98.6 % of the dual rows; it stays that way for as long as the loop runs): the band update writes
            // its per-half values without the `inblock` guard, and the loop control is scalar — the rows the two blocks may still run
            // (d < dlim in both) as one count-down, the band-size limits (:118 "max_k - min_k <= band_size") of the two halves next to
            // the slot counts, which are on the scalar unit anyway.  One-pass rows run in a loop of their own.
            int rows_left = min(__builtin_amdgcn_readlane(dlim - d, 0), __builtin_amdgcn_readlane(dlim - d, 32));
            const int lim_a = __builtin_amdgcn_readlane(band_tol, 0) + 1, lim_b = __builtin_amdgcn_readlane(band_tol, 32) + 1;
            const int one_a = min(lim_a, 32), one_b = min(lim_b, 32);
            const int two_a = min(lim_a, 63), two_b = min(lim_b, 63);
            typedef std::integral_constant<int, 0> fast0;
            typedef std::integral_constant<int, 1> fast1;
            typedef std::integral_constant<int, 2> fast2;
            while (true) {
                // one-pass rows (rows left, and both slot counts within their limits: one sign test)
                while (((rows_left - 1) | (one_a - ns_a) | (one_b - ns_b)) >= 0) {
                    DWS_ROW(~0ull, ns_a, ns_b, 1);
                    rows_left -= 1;
                    const bool full = ((ns_a | ns_b) & 32) != 0;       // a half has all of its 32 lanes on diagonals: nobody stores the entry
                    row_passes(1, ns_a, ns_b, fast1{});                //   behind its row (the other half's idle lane rewrites its own)
                    lin_l += row_bytes;
                    if (full) ring_st(rbase, lin_l - 2u * sl - 2u, sepv);
                    band_update(1, last_m0, last_mp, std::true_type{}, ns_a, ns_b);
                    log_row(1, std::true_type{});
                    // (a row that reached an end leaves the loop through its bound: a second exit makes the compiler build a machine of
                    // flag registers and re-tested branches around every row)
                    const int go = ended ? 0 : 1;
                    rows_left = go ? rows_left : -1;
                    d += go;
                    __builtin_amdgcn_wave_barrier();
                }
                if (ended) break;
                // two-pass rows: a half has 33 .. 63 slots
                while (((rows_left - 1) | (two_a - ns_a) | (two_b - ns_b) | (max(ns_a, ns_b) - 33)) >= 0) {      // (one sign test: rows left, slot counts within 33 .. limit)
                    DWS_ROW(~0ull, ns_a, ns_b, 2);
                    NJ = 2;
                    rows_left -= 1;
                    row_passes(2, ns_a, ns_b, fast2{});
                    band_update(2, last_m0, last_mp, std::true_type{}, ns_a, ns_b);
                    log_row(2, std::true_type{});
                    lin_l += row_bytes;
                    const int go = ended ? 0 : 1;
                    rows_left = go ? rows_left : -1;
                    d += go;
                    __builtin_amdgcn_wave_barrier();
                }
                if (ended) break;
                if (rows_left <= 0 || ns_a > lim_a || ns_b > lim_b) break;
                const int ns_max = max(ns_a, ns_b);
                if (ns_max < 64) continue;                   // back to the two loops above
                // a half with 64 slots or more (0.01 % of the rows): the general form
                rows_left -= 1;
                NJ = (ns_max + 31) >> 5;
                DWS_ROW(~0ull, ns_a, ns_b, NJ);
                row_passes(NJ, ns_a, ns_b, fast0{});
                lin_l += row_bytes;
                if ((ns_max & 31) == 0) ring_st(rbase, lin_l - 2u * sl - 2u, sepv);      // a half fills its last pass: no idle lane for the entry behind its row
                band_update(NJ, last_m0, last_mp, std::true_type{}, ns_a, ns_b);
                log_row(NJ, std::true_type{});
                if (ended) break;
                d += 1;
                __builtin_amdgcn_wave_barrier();
            }
        } else {
            // one half is between blocks (or out of units) while the other one rows on: the general form of everything
            while (true) {
                // the two conditions as masks (a ballot of their conjunction makes the compiler turn the mask into a 0/1 vector and
                // compare it again)
                const unsigned long long rmask = BALLOT(d < dlim) & BALLOT(nslot <= band_tol + 1);
                if (rmask != inmask) break;
                const int ns_max = max(ns_a, ns_b);
                NJ = (ns_max + 31) >> 5;
                DWS_ROW(rmask, ns_a, ns_b, NJ);
                row_passes(NJ, ns_a, ns_b, std::integral_constant<int, 0>{});
                lin_l += row_bytes;
                if ((ns_max & 31) == 0) ring_st(rbase, lin_l - 2u * sl - 2u, sepv);
                band_update(NJ, last_m0, last_mp, std::false_type{}, ns_a, ns_b);
                log_row(NJ, std::false_type{});
                if (ended) break;
                d += 1;
                __builtin_amdgcn_wave_barrier();
            }
        }
        DWS_T(t_c);
        DWS_ADD(tk_rows, t_b, t_c);
#ifdef MECAT_DW_STATS
        n_ended += ended ? 1u : 0u;
#endif
        if (ended) {
            // Once per block, outside the row loop (inside it, the state written here costs register copies on every row): the
            // lowest diagonal that reached an end (:168-169), from the row just stored.  The band update for the row after it has been
            // done already (the other half goes on with it): the band of the end row is what that update started from — its slots
            // from the row's length in the ring, its first diagonal from the cut.
            const int en = (int)((lin_l - rlin_l) >> 1) - 1, emk_l = mk_l - 2 * cut + 1;
            int hkey = 0x7fffffff;
            for (int j = 0; BALLOT(sl + 32 * j < en); ++j) {
                const int k = emk_l + 64 * j, kk = k + k_offset;
                if (sl + 32 * j < en) {
                    const int x = ring_ld(rbase, rlin_l + 64u * (unsigned)j);
                    if (x >= q_len || x - k >= t_len) hkey = min(hkey, (kk << 10) | x);
                }
            }
            hkey = half_min(hkey);
            if (inblock && hkey != 0x7fffffff) {
                aligned = 1; end_k = (hkey >> 10) - k_offset; end_x = hkey & 1023; end_d = d;
                end_mk = emk_l - 2 * sl; end_ns = en;
                dlim = 0;
            }
            d += 1;
            __builtin_amdgcn_wave_barrier();
        }
        row_ok = d < dlim && nslot <= band_tol + 1;
        DWS_T(t_d);
        DWS_ADD(tk_ended, t_c, t_d);

        const bool fin = inblock && !row_ok;

        // ---- 4. tail traceback == trim_mismatch_end(.., 4, ..) (gapalign.cpp:47-68), for the halves whose rows just ended
        bool has_aln = fin && aligned;
        // rows ran without reaching an end: needs the best point (dw_extend).  CNS: Align has failed, the direction ends (dw.cpp:302-304)
        bool handover = !CNS && fin && d > 0 && !aligned;
        const int end_y = end_x - end_k;
        const int aln_size = (end_x + end_y + end_d) / 2;
        int cd = end_d, ck = end_k, cx2 = end_x;
        int qcnt = 0, tcnt = 0, acnt = 0, found = 0;
        bool tracing = CNS ? (has_aln && !last_block) : has_aln;      // (CNS: a last block keeps all of its columns, dw.cpp:353-358)
        // Rows are walked back over the entries between them: behind row r sits ~(slots of r | left cut of row r - 1 << 8).  clin / cmin /
        // cns: ring position, first diagonal and slots of row cd (the row that reached the end is the row that ran last).
        const unsigned int lin_u = lin_l - 2u * sl;
        unsigned int clin = rlin_l - 2u * sl;
        int cmin = end_mk, cns = end_ns;
        while (BALLOT(tracing)) {
            if (tracing) {
                int x1 = 0, pre_k = 0, takes_q = 0;
                if (cd > 0) {
                    const int sc = ~ring_ld(rbase, clin + 2u * (unsigned)cns), sp = ~ring_ld(rbase, clin - 2u);
                    const int pcut = (sc >> 8) & 127, pns = sp & 255;
                    const unsigned int plin = clin - 2u - 2u * (unsigned)pns;
                    // (128: what the idle lanes of the last row's passes may have written behind it, see row_passes; a cut of 127 or
                    // more is not recorded)
                    if (pcut == 127 || lin_u - plin > 2 * RCAP - 128) { handover = true; tracing = false; }
                    else {
                        const int pmin = cmin - 2 * pcut + 1, pmax = pmin + 2 * (pns - 1), cmax = cmin + 2 * (cns - 1);
                        const int kl = ck - 1, kr = ck + 1;
                        int vl = 0, vr = 0;
                        if (kl >= pmin && kl <= pmax) vl = ring_ld(rbase, plin + (unsigned)(kl - pmin));      // entry (kl - pmin) / 2, two bytes each
                        if (kr >= pmin && kr <= pmax) vr = ring_ld(rbase, plin + (unsigned)(kr - pmin));
                        if (ck == cmin || (ck != cmax && vl < vr)) { x1 = vr; pre_k = kr; takes_q = 0; }
                        else { x1 = vl + 1; pre_k = kl; takes_q = 1; }
                        clin = plin; cmin = pmin; cns = pns;
                    }
                }
                if (tracing) {
                    const int sn = cx2 - x1;
                    if (sn >= 4) { acnt += 4; qcnt += 4; tcnt += 4; found = 1; tracing = false; }
                    else {
                        acnt += sn; qcnt += sn; tcnt += sn;
                        if (cd == 0) tracing = false;
                        else {
                            acnt += 1;
                            if (takes_q) { qcnt += 1; cx2 = x1 - 1; } else { tcnt += 1; cx2 = x1; }
                            ck = pre_k;
                            --cd;
                        }
                    }
                }
            }
        }
        const int trim_ok = has_aln && found && (aln_size - acnt >= 2);
        DWS_T(t_e);
        DWS_ADD(tk_trace, t_d, t_e);

        // ---- 5. block accounting (dw_in_one_direction, diff_gapalign.cpp:259-290); a block this kernel cannot finish (0.07 %: the
        // tail needs a row that left the ring, or no end was reached) hands the unit over to dw_extend, which redoes that block
        // and the rest of the unit with every row kept
        if (CNS && fin) {
            // dw_in_one_direction, dw.cpp:333-373.  A block that reached an end contributes the columns in front of its last run of four
            // matches (all of them when it is the direction's last block) and the next block starts there; a block without such a run, or
            // one that would contribute no query base, ends the direction without contributing.
            bool done = true;
            if (handover || cns_bad) {
                if (sl == 0) {
                    ca.hand_units[1u + atomicAdd(ca.hand_units, 1u)] = unit;
                    CnsDir D = {0, 0, 0, 0, 0, -1};
                    ca.dres[unit] = D;
                    nhand += 1;
                }
            } else {
                nblocks += (sl == 0) ? 1u : 0u;
                cells += (sl == 0) ? ((lin_l - 6u) >> 1) - (unsigned)d : 0u;
                bool keep = has_aln && (last_block || found);
                const int kept_q = end_x - qcnt, kept_t = end_y - tcnt, kept_cols = aln_size - acnt;      // (last block: nothing was walked)
                keep = keep && kept_q != 0;
                if (keep && Rc + kept_cols > ca.dir_cols_cap) { keep = false; if (sl == 0) atomicExch(ca.err_flag, 1); }
                if (keep && blkpos >= blkend) {
                    // the unit's share of block records ((ext >> 8) + 4, cns_caps) is used up — blocks that each advance by fewer than 256
                    // bases: like the other limits of the record format, the unit goes to cns_extend, which redoes it from its start
                    // (cns_trace skips the blocks of a handed-over unit)
                    if (sl == 0) {
                        ca.hand_units[1u + atomicAdd(ca.hand_units, 1u)] = unit;
                        CnsDir D = {0, 0, 0, 0, 0, -1};
                        ca.dres[unit] = D;
                        nhand += 1;
                    }
                } else {
                    if (keep) {
                        if (sl == 0) {
                            CnsBlockRec B = {unit, qidx, tidx, qblk, end_d, end_k, end_mk, end_x, Rc, kept_cols, blk_log0, 0};
                            ca.blocks[blkpos] = B;
                        }
                        blkpos += 1;
                        Rc += kept_cols; Rq += kept_q; Rt += kept_t;
                        if (!last_block) { qidx += kept_q; tidx += kept_t; done = false; }
                    }
                    if (done && sl == 0) { CnsDir D = {Rc, Rq, Rt, 0, 0, 0}; ca.dres[unit] = D; }
                }
            }
            if (done) need_unit = true;
            inblock = false;
            setup = true;
            dlim = 0;
        }
        if (!CNS && fin) {
            bool stop;
            if (handover) {
                if (sl == 0) {
                    DwHandover h;
                    h.unit = unit; h.qidx = qidx; h.tidx = tidx;
                    h.R.qbases = Rq; h.R.tbases = Rt; h.R.matches = Rm; h.R.columns = Rc; h.R.blocks = Rb; h.R.pad = 0;
                    hand[atomicAdd(hand_count, 1u)] = h;
                    nhand += 1;
                }
                need_unit = true;
            } else {
                nblocks += (sl == 0) ? 1u : 0u;
                cells += (sl == 0) ? ((lin_l - 6u) >> 1) - (unsigned)d : 0u;      // diagonals visited in this block (d rows, one -1 behind each)
                Rb += 1;
                stop = !has_aln || !trim_ok;
                if (!stop) {
                    const int full_map = (qblk - end_x <= 20 || tblk - end_y <= 20);
                    const bool last = last_block || !full_map;
                    if (last) { qcnt -= 4; tcnt -= 4; acnt -= 4; }
                    Rc += (end_x + end_y + end_d) / 2 - acnt;
                    Rm += (end_x + end_y - end_d) / 2 - (qcnt + tcnt - acnt);
                    Rq += end_x - qcnt;
                    Rt += end_y - tcnt;
                    if (last) stop = true;
                    else { qidx += end_x - qcnt; tidx += end_y - tcnt; }
                }
                if (stop) {
                    if (sl == 0) { DirResult R = {Rq, Rt, Rm, Rc, Rb, 0}; dres[unit] = R; }
                    need_unit = true;
                }
            }
            inblock = false;
            setup = true;
            dlim = 0;           // a half without a block must never look like it is rowing (its rows may have ended on the band limit)
        }
        DWS_T(t_f);
        DWS_ADD(tk_acct, t_e, t_f);
    }
    unsigned long long c64 = cells, b64 = nblocks, h64 = nhand;
    for (int off = 32; off > 0; off >>= 1) { b64 += __shfl_xor(b64, off); c64 += __shfl_xor(c64, off); h64 += __shfl_xor(h64, off); }
    if (lane == 0) {
        atomicAdd(&counters[3], b64);
        atomicAdd(&counters[4], c64);
        atomicAdd(&counters[8], h64);          // units handed over
#ifdef MECAT_DW_STATS
        atomicAdd(&counters[9], nrows);          // dual rows
        atomicAdd(&counters[10], nidle);         // dual rows with one idle half
        atomicAdd(&counters[11], nwide);         // dual rows with more than 32 diagonals in a half
        atomicAdd(&counters[16], wall_clock64() - tk_start);      // wave life, ticks
        atomicAdd(&counters[17], tk_setup);
        atomicAdd(&counters[18], tk_rows);
        atomicAdd(&counters[19], tk_ended);
        atomicAdd(&counters[20], tk_trace);
        atomicAdd(&counters[21], tk_acct);
        atomicAdd(&counters[22], n_outer);
        atomicAdd(&counters[23], n_setup);
        atomicAdd(&counters[24], n_ended);
        atomicAdd(&counters[25], n_pass3);
        atomicAdd(&counters[26], 1ull);                           // waves
        atomicAdd(&counters[27], snake2);
        for (int i = 0; i < 8; ++i) atomicAdd(&counters[32 + i], nh[i]);
        for (int i = 0; i < 6; ++i) atomicAdd(&counters[40 + i], q16[i]);
#endif
    }
}

// stitch the two directions (diff_gapalign.cpp:309-348)
__global__ void dw_stitch(const mhip_aln_job* __restrict__ jobs, const DirResult* __restrict__ dres, int n, int min_aln,
                          mhip_aln_result* __restrict__ out, unsigned long long* __restrict__ counters) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const DirResult L = dres[2 * i], R = dres[2 * i + 1];
    mhip_aln_result r;
    r.query_start = jobs[i].qstart - L.qbases;
    r.target_start = jobs[i].sstart - L.tbases;
    r.query_end = jobs[i].qstart + R.qbases;
    r.target_end = jobs[i].sstart + R.tbases;
    r.matches = L.matches + R.matches;
    r.columns = L.columns + R.columns;
    r.blocks = L.blocks + R.blocks;
    r.ok = r.columns >= min_aln;
    out[i] = r;
    if (r.ok) {
        atomicAdd(&counters[6], (unsigned long long)(r.query_end - r.query_start));
        atomicAdd(&counters[7], 1ull);
    }
}

// candidate table -> job list (pw_impl.cpp:674-686): exclusive scan of the per-read counts (one block, running carry
// over 1024-read tiles), then one thread per (read, slot).
__global__ __launch_bounds__(1024) void dw_job_scan(const int32_t* __restrict__ counts, int n_reads, int part_index, int part_count,
                                                    unsigned int* __restrict__ first, int* __restrict__ num_jobs) {
    __shared__ unsigned int wtot[16];
    __shared__ unsigned int carry_all;
    if (threadIdx.x == 0) carry_all = 0;
    __syncthreads();
    for (int t0 = 0; t0 < n_reads; t0 += 1024) {
        const int i = t0 + threadIdx.x;
        const unsigned int c = i < n_reads ? (unsigned int)counts[i] : 0u;
        unsigned int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            unsigned int v = __shfl_up(incl, o);
            if (lane_id() >= o) incl += v;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned int base = carry_all;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
        if (i < n_reads) first[i] = base + incl - c;         // global index of this read's first candidate
        __syncthreads();
        if (threadIdx.x == 1023) carry_all = base + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const unsigned int tot = carry_all;
        unsigned int mine = tot;
        if (part_count > 1) mine = tot / (unsigned)part_count + ((unsigned)part_index < tot % (unsigned)part_count ? 1u : 0u);
        *num_jobs = (int)mine;
    }
}

__global__ __launch_bounds__(256) void dw_make_jobs(const mhip_candidate* __restrict__ cands, const int32_t* __restrict__ counts,
                                                    const unsigned int* __restrict__ first, int n_reads, int maxc, int rid_begin,
                                                    int rid_stride, int ref_start_id, int part_index, int part_count,
                                                    mhip_aln_job* __restrict__ jobs) {
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int i = (int)(t / (unsigned)maxc), j = (int)(t % (unsigned)maxc);
    if (i >= n_reads || j >= counts[i]) return;
    const unsigned int g = first[i] + (unsigned)j;
    if (part_count > 1 && (int)(g % (unsigned)part_count) != part_index) return;
    const unsigned int slot = part_count > 1 ? g / (unsigned)part_count : g;
    const mhip_candidate cd = cands[(size_t)i * maxc + j];
    mhip_aln_job jb;
    jb.qid_local = rid_begin + i * rid_stride;
    jb.sid_local = cd.readno - ref_start_id;
    jb.chain = cd.chain;
    int qstart = cd.loc2, sstart = cd.loc1;
    if (qstart && sstart) { qstart += MHIP_KMER_SIZE / 2; sstart += MHIP_KMER_SIZE / 2; }
    jb.qstart = qstart;
    jb.sstart = sstart;
    jobs[slot] = jb;
}

// the forward pass of the mecat2cns re-aligner (cns_fwd.h): dw_extend2 under mecat2cns' block rules, 24 waves per CU
int cns_forward_launch(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n, const CnsFwdArgs& args) {
    unsigned int* d_cur;
    if (c->scratch("cnf_cursor", 64, (void**)&d_cur)) return -1;
    HIPCHK(hipMemsetAsync(d_cur, 0, 16, c->stream));
    const int waves = getenv("MECAT_CNS_WAVES") ? std::max(4, std::min(32, atoi(getenv("MECAT_CNS_WAVES")))) : 4 * CNS_FWD_WAVES_PER_SIMD;
    const int grid = std::min(c->num_cus * waves / AL_WAVES, (n + AL_WAVES - 1) / AL_WAVES);
    LAUNCH(c, "cns_forward", dw_extend2<true>, grid, AL_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
           (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, (DirResult*)nullptr,
           (DwHandover*)nullptr, (unsigned int*)nullptr, d_cur, (unsigned long long*)c->d_counters, args);
    return 0;
}

extern "C" {

int mhip_jobs_from_candidates_dev(mhip_ctx* c, const void* d_cands, const void* d_counts, int n_reads, int maxc, int rid_begin,
                                  int rid_stride, int ref_start_read_id, int part_index, int part_count, void* d_jobs, int* num_jobs) {
    HIPCHK(hipSetDevice(c->device));
    *num_jobs = 0;
    if (n_reads <= 0) return 0;
    if (part_count < 1) part_count = 1;
    int* d_n;
    if (c->scratch("al_njobs", 64, (void**)&d_n)) return -1;
    unsigned int* d_first;
    if (c->scratch("al_jobfirst", sizeof(unsigned int) * (size_t)n_reads, (void**)&d_first)) return -1;
    LAUNCH(c, "dw_job_scan", dw_job_scan, 1, 1024, 0, (const int32_t*)d_counts, n_reads, part_index, part_count, d_first, d_n);
    const size_t nthreads = (size_t)n_reads * (size_t)maxc;
    LAUNCH(c, "dw_make_jobs", dw_make_jobs, (unsigned)((nthreads + 255) / 256), 256, 0, (const mhip_candidate*)d_cands,
           (const int32_t*)d_counts, (const unsigned int*)d_first, n_reads, maxc, rid_begin, rid_stride, ref_start_read_id, part_index,
           part_count, (mhip_aln_job*)d_jobs);
    HIPCHK(hipMemcpyAsync(num_jobs, d_n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

int mhip_align_candidates_dev(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n,
                              int min_align_size, void* d_out) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    int waves_per_cu = 16;                              // 4 waves per SIMD (VGPR budget); LDS 9.2 KB per wave
    if (const char* e = getenv("MECAT_DW_WAVES")) waves_per_cu = std::max(4, std::min(32, atoi(e)));   // tuning/debug knob
    const int max_waves = c->num_cus * waves_per_cu;
    int grid = max_waves / AL_WAVES;
    grid = std::min(grid, (2 * n + AL_WAVES - 1) / AL_WAVES);
    DirResult* d_dres;
    uint16_t* d_g;
    unsigned int* d_cur;
    DwHandover* d_hand;
    if (c->scratch("al_dres", sizeof(DirResult) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("al_rows", sizeof(uint16_t) * GROW_STRIDE * (size_t)max_waves, (void**)&d_g)) return -1;
    if (c->scratch("al_cursor", 64, (void**)&d_cur)) return -1;
    HIPCHK(hipMemsetAsync(d_cur, 0, 16, c->stream));                 // [0] unit cursor, [1] units handed over, [2] cursor of the second launch
    const char* kv = getenv("MECAT_DW_KERNEL");      // debug knob: 1 = one unit per wave, default 2 = one unit per half-wave
    if (kv && atoi(kv) == 1) {
        LAUNCH(c, "dw_extend", dw_extend, grid, AL_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_g, d_cur,
               (unsigned long long*)c->d_counters, (const DwHandover*)nullptr, (const unsigned int*)nullptr);
    } else {
        if (c->scratch("al_hand", sizeof(DwHandover) * 2 * (size_t)n, (void**)&d_hand)) return -1;
        int waves2 = DW2_WAVES_PER_SIMD * 4;                  // LDS 20 KB per four waves, 64 VGPRs
        if (const char* e = getenv("MECAT_DW_WAVES")) waves2 = std::max(4, std::min(32, atoi(e)));
        const int grid2 = std::min(c->num_cus * waves2 / AL_WAVES, (n + AL_WAVES - 1) / AL_WAVES);
        if (getenv("MECAT_TRACE")) {
            int nb = 0;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, dw_extend2<false>, AL_BLOCK, 0);
            fprintf(stderr, "[dw trace] occupancy query: %d blocks of %d threads per CU; grid %d; HalfLds %zu bytes\n", nb, AL_BLOCK, grid2, sizeof(HalfLds));
        }
        CnsFwdArgs none;
        memset(&none, 0, sizeof(none));
        LAUNCH(c, "dw_extend2", dw_extend2<false>, grid2, AL_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_hand, d_cur + 1,
               d_cur, (unsigned long long*)c->d_counters, none);
        // the units it handed over (a fraction of a percent), finished by the one-unit kernel with every row kept; the launch reads
        // their number on the device, so nothing waits for the host
        LAUNCH(c, "dw_extend", dw_extend, grid, AL_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_g, d_cur + 2,
               (unsigned long long*)c->d_counters, (const DwHandover*)d_hand, (const unsigned int*)(d_cur + 1));
    }
    LAUNCH(c, "dw_stitch", dw_stitch, (n + 255) / 256, 256, 0, (const mhip_aln_job*)d_jobs, (const DirResult*)d_dres, n,
           min_align_size, (mhip_aln_result*)d_out, (unsigned long long*)c->d_counters);
    HIPCHK(hipGetLastError());
    return 0;
}

int mhip_align_candidates(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs, int n,
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
        if (j.qstart > qs || j.sstart > ts) {      // (a negative start point is a job without extension, see dw_extend)
            mhip_set_error("alignment job %d: start point outside the reads", i);
            return -1;
        }
    }
    mhip_aln_job* d_jobs;
    mhip_aln_result* d_out;
    if (c->scratch("al_jobs", sizeof(mhip_aln_job) * (size_t)n, (void**)&d_jobs)) return -1;
    if (c->scratch("al_out", sizeof(mhip_aln_result) * (size_t)n, (void**)&d_out)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, sizeof(mhip_aln_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (mhip_align_candidates_dev(c, ref, reads, d_jobs, n, min_align_size, d_out)) return -1;
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_aln_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
