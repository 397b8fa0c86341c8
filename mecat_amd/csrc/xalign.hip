// xalign.hip — BLAST-style affine X-drop gapped extension in 500-bp blocks, the nanopore-mode aligner
// (used by mecat2pw -x 1 -j 1).
//
// Replaces XdropAligner::go (common/xdrop_gapalign.cpp:359-439), align_ex (:263-357), xdrop_align (:10-213) and
// script_to_aligned_string (:215-261) with reward 1, penalty -1, gap_open 0, gap_extend 1, X = 30, block 500
// (common/xdrop_gapalign.h:98-114).
//
// One WAVE per (candidate, direction) unit, lanes = cells of a row of the dynamic program (xd_extend_w below).  As in dw, only
// what mecat2pw consumes is produced: end coordinates and identity counts; the traceback walks the script bytes once and does
// trim_mismatch_end and the match/column counting on the fly.  Two instantiations of the same row formulation exist: the
// 128-cell LDS ring (default) and the WIDE one (scores indexed by b, 768-byte rows, byte-by-byte traceback) that takes over the
// blocks whose window outgrows the ring; MECAT_XD_WIDE=1 runs every block through the WIDE one (the at-scale cross-check of
// tests/test_gpu_xalign.py; the sequential restatement both are checked against is test infrastructure, not part of this library).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

#define X_SEG 500
#define X_MAXN 736                       // last block: one side < 600, the other <= 718 (gapalign.cpp:24-30)
#define X_MIN_SCORE (-100000000)

enum { XS_SUB = 3, XS_GAP_IN_A = 0, XS_GAP_IN_B = 6, XS_OP_MASK = 0x07, XS_EXT_A = 0x10, XS_EXT_B = 0x40 };   // xdrop_gapalign.h:13-34

struct XDir {
    int32_t qbases, tbases, matches, columns;     // kept alignment of one direction
    int32_t last_q, last_t, last_m;               // type of its LAST column (the left half drops it, xdrop_gapalign.cpp:401-402)
    int32_t blocks;
};


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
    int cells, rows;            // work counters: DP cells visited (window widths summed over the rows), rows
#ifdef MECAT_XD_STATS
    unsigned long long tk_stage, tk_rows, tk_trace, n_rows2, n_win, n_steps, n_qfill;      // section clocks (s_memtime ticks) and event counts of the block
#endif
};

// ------------------------------------------------------------------------------------------------------------------
// xd_extend_w — one WAVE per (candidate, direction).  A row of the dynamic program is per-cell independent work plus two
// prefix-max scans over the window (the tests check the equivalence with the sequential row state by state):
//   diag_b = H'[b-1] + s(A_a, B_{b-1}) (MIN for the row's first cell),  M_b = max(diag_b, F'[b]),
//   E_b = max_{j<b}(M_j + j) - b,  H_b = max(M_b, E_b),  best_b = max(best, max_{j<b} H_j),  dropped <=> best_b - H_b > X;
// a value carried across a dropped cell is below every later kept cell's score, so it only shows in the op bits of dropped
// cells (there E = H of the nearest kept cell to the left - 1, no decay).  Lanes = 64 consecutive cells, a second chunk with
// scalar carries when the window is wider.  Scores live in an LDS ring of XW_RING cells; the op bytes go to a per-wave global
// scratch with a fixed row stride (coalesced 64-byte stores), plus bit 7 = "the diagonal step into this cell is a match" so
// that the traceback needs nothing but these bytes.  The traceback is a scalar walk: a row's bytes sit in two registers
// (one cell per lane), fetched one row ahead from a 4 KB LDS window over the scratch, and are read with v_readlane.
// A block whose window grows wider than XW_RING - 2 cells is redone at once by the same wave with the WIDE instantiation of the
// round-1 code: scores indexed by b, rows of XW_WSTRIDE bytes, traceback reading byte by byte through a window — with its whole
// state (scores, bases, row starts, window) in the wave's share of a global buffer instead of LDS: slower per row, but only a
// fraction of a percent of the units (low-complexity sequence) have such a block, and the unit then goes on in the ring.
// (Round 1-3 handed such units to a second launch — overflow list: unit, position and partial result — which was as long as its
// slowest unit: 17.6 ms of 136 on 134 k jobs.  That launch is still there behind MECAT_XD_HANDOVER=1, and MECAT_XD_WIDE=1 runs
// every block through the WIDE code in LDS: the tests compare all three.)
#define XW_WAVES 4
#define XW_BLOCK (XW_WAVES * 64)
#define XW_RING 128
#define XW_STRIDE 128                    // script bytes per row in the scratch
#define XW_WIN 4096                      // = 32 rows of XW_STRIDE bytes
#define XW_STATE_BYTES ((size_t)(X_MAXN + 2) * XW_STRIDE)
#define XW_WSTRIDE 768                   // WIDE: any window fits a row (N + 1 <= 737 cells)
#define XW_WIDE_BYTES ((size_t)(X_MAXN + 2) * XW_WSTRIDE)
#define XW_WIDE_HF (X_MAXN + 8)
#define XW_INPLACE_STATE 16384           // bytes in front of a ring wave's wide rows: its XwLds<XW_WIDE_HF> image
#define XW_INPLACE_BYTES (XW_INPLACE_STATE + XW_WIDE_BYTES)
#define XW_NEG (-(1 << 30))
#define XS_MATCH 0x80
template <int HFN> struct XwLds {
    int2 HF[HFN];                                        // (H, F) of cell b at b mod XW_RING; WIDE blocks: at b
    uint8_t Qb[X_MAXN + 8], Tb[X_MAXN + 72];             // Tb[b + 1] = target base b (one pad in front, 64 behind for idle lanes)
    int16_t rstart[HFN == XW_RING ? 2 : X_MAXN + 2];     // WIDE blocks only; 128-byte rows carry their first column in bytes 126-127
    uint32_t win[XW_WIN / 4];
};

static_assert(sizeof(XwLds<XW_WIDE_HF>) <= XW_INPLACE_STATE, "the in-place wide state holds one XwLds<XW_WIDE_HF>");

template <int CTRL, int RMASK> __device__ __forceinline__ int xw_dpp_max(int v) {
    // old = the identity of max: lets the DPP combiner fuse mov_dpp + max into one v_max_i32_dpp and schedule around the hazard
    return max(v, __builtin_amdgcn_update_dpp((int)0x80000000, v, CTRL, RMASK, 0xf, false));
}
__device__ __forceinline__ int xw_scan_max(int v) {          // inclusive prefix max over the 64 lanes, all lanes active
    v = xw_dpp_max<0x111, 0xf>(v);                            // row_shr:1,2,4,8
    v = xw_dpp_max<0x112, 0xf>(v);
    v = xw_dpp_max<0x114, 0xf>(v);
    v = xw_dpp_max<0x118, 0xf>(v);
    v = xw_dpp_max<0x142, 0xa>(v);                            // row_bcast:15 into rows 1 and 3
    v = xw_dpp_max<0x143, 0xc>(v);                            // row_bcast:31 into rows 2 and 3
    return v;
}
__device__ __forceinline__ int xw_shr1(int v, int fill) {    // lane l gets lane l-1, lane 0 gets fill
    return __builtin_amdgcn_update_dpp(fill, v, 0x138, 0xf, 0xf, false);   // wave_shr:1
}

template <bool WIDE, int HFN>
__device__ void xdrop_block_w(XwLds<HFN>& S, const XView& q, int qidx, int M, const XView& t, int tidx, int N, uint8_t* __restrict__ st,
                              XBlockOut& o) {
    constexpr int STRIDE = WIDE ? XW_WSTRIDE : XW_STRIDE;
    static_assert(!WIDE || HFN >= XW_WIDE_HF, "WIDE blocks index the score array by b");
    auto slot = [&](int b) -> int { return WIDE ? min(b, HFN - 1) : b & (XW_RING - 1); };      // (idle lanes read a valid slot)
    auto ldHF = [&](int b, bool in) -> int2 {
        int2 v = S.HF[slot(b)];
        asm volatile("" : "+v"(v.x), "+v"(v.y));          // every lane loads (the slot is always valid): no exec round trip
        return in ? v : make_int2(X_MIN_SCORE, X_MIN_SCORE);
    };
    auto stHF = [&](int b, int h, int f) { S.HF[slot(b)] = make_int2(h, f); };
    o.ae = o.be = 0; o.n = o.nmatch = 0; o.qcnt = o.tcnt = o.acnt = o.mtail = 0; o.trim_ok = 0; o.overflow = 0;
    o.l0q = o.l0t = o.l0m = o.l1q = o.l1t = o.l1m = 0;
    o.cells = o.rows = 0;
    if (M <= 0 || N <= 0) return;
    const int lane = lane_id();
    const int X = 30;
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < M; i += 64) S.Qb[i] = (uint8_t)xv_at(q, qidx + i);
    for (int i = lane; i < N; i += 64) S.Tb[i + 1] = (uint8_t)xv_at(t, tidx + i);
    // row 0 (xdrop_gapalign.cpp:45-57): cells 1.. hold -1, -2, ... while >= -X
    const int n_init = min(N, X);
    if (lane == 0) { if (WIDE) S.rstart[0] = 0; else *(uint16_t*)(st + XW_STRIDE - 2) = 0; }
    if (lane <= n_init) { stHF(lane, -lane, -lane - 1); if (lane) st[lane] = XS_GAP_IN_A; }
    int b_size = n_init + 1, best = 0, first_b = 0, ae = 0, be = 0;      // b_size == N + 1 when N <= X, as in the reference
    __builtin_amdgcn_wave_barrier();
    for (int a = 1; a <= M; ++a) {
        const int AC = S.Qb[a - 1];
        const int f0 = first_b, n0 = b_size;
        o.cells += n0 - f0;
        o.rows = a;
        uint8_t* srow = st + (size_t)a * STRIDE;
        int runP = XW_NEG, bb = best, rowarg = -1, firstkept = -1, lastkept = -1, lastkeptH = 0;
        int prevHp = 0;
        auto pass = [&](const int c0) __attribute__((always_inline)) {
            const int b = c0 + lane;
            const bool in = b < n0;
            const int2 hf = ldHF(b, in);
            const int Hp = hf.x, Fp = hf.y;
            const int left = xw_shr1(Hp, prevHp);
            const bool mt = AC == (int)S.Tb[b];                         // target base b - 1
            const int diag = b == f0 ? X_MIN_SCORE : left + (mt ? 1 : -1);
            const int Mv = max(diag, Fp);
            const int incl = xw_scan_max(in ? Mv + b : XW_NEG);
            const int pex = max(xw_shr1(incl, XW_NEG), runP);         // max over all earlier cells of the row of M_j + j
            const int Ec = (b == f0) ? X_MIN_SCORE : pex - b;
            const int Hc = max(Mv, Ec);
            // best so far before cell b = max(bb, max_{j<b} H_j).  No scan needed: a row's scores exceed the best of the rows before
            // by at most the match reward (every H derives from the previous row, at most +1), so all cells above bb hold the same
            // value and only the position of the first one matters.
            const unsigned long long ex = __builtin_amdgcn_ballot_w64(in && Hc > bb);
            int j1 = 64, nb = bb;
            if (ex) { j1 = __ffsll((long long)ex) - 1; nb = __builtin_amdgcn_readlane(Hc, j1); }
            const int bbefore = lane > j1 ? nb : bb;
            const bool kept = in && !(bbefore - Hc > X);
            const unsigned long long km = __builtin_amdgcn_ballot_w64(kept);
            // op bits: SUB unless the column gap, then the row gap, is strictly better (:99-107)
            int sc = diag, script = XS_SUB;
            if (sc < Fp) { script = XS_GAP_IN_B; sc = Fp; }
            const unsigned long long lower = km & ((1ull << lane) - 1ull);
            const int jsrc = lower ? 63 - __clzll((long long)lower) : 0;
            const int Hj = __shfl(Hc, jsrc);
            // a dropped cell compares with the row gap as the sequential row carries it: H of the nearest kept cell to the left - 1
            const int et = lower ? Hj - 1 : (lastkept >= 0 ? lastkeptH - 1 : X_MIN_SCORE);
            if (sc < (kept ? Ec : et)) script = XS_GAP_IN_A;
            if (kept && Fp >= Hc) script += XS_EXT_A;
            if (kept && Ec >= Hc) script += XS_EXT_B;
            // kept: (H, F') with F' = max(F - 1, H - 1) = H - 1 (H >= F); dropped between kept cells: H = MIN, F stays; a leading
            // dropped cell only moves first_b.  One predicated store.
            if (kept || (in && (lower || firstkept >= 0))) stHF(b, kept ? Hc : X_MIN_SCORE, kept ? Hc - 1 : Fp);
            srow[b - f0] = (uint8_t)(script | (mt ? XS_MATCH : 0));          // lanes past the window write bytes nobody reads
            // carries
            runP = max(runP, __builtin_amdgcn_readlane(incl, 63));
            if (ex && rowarg < 0) rowarg = c0 + j1;
            bb = nb;
            if (km) {
                if (firstkept < 0) firstkept = c0 + __ffsll((long long)km) - 1;
                const int lk = 63 - __clzll((long long)km);
                lastkept = c0 + lk;
                lastkeptH = __builtin_amdgcn_readlane(Hc, lk);
            }
            prevHp = __builtin_amdgcn_readlane(Hp, 63);
        };
        // two in three rows fit one pass: with the carries still at their initial constants the compiler drops their bookkeeping
        if (n0 - f0 <= 64) pass(f0);
        else if (!WIDE) { pass(f0); pass(f0 + 64); }      // the ring bounds a row to 126 cells
        else for (int c0 = f0; c0 < n0; c0 += 64) pass(c0);
        if (bb > best) { best = bb; ae = a; be = rowarg; }
        if (firstkept < 0) { first_b = n0; break; }
        first_b = firstkept;
        // the row's first column (after the passes: their idle lanes may have written over these two bytes)
        if (lane == 0) { if (WIDE) S.rstart[a] = (int16_t)f0; else *(uint16_t*)(srow + XW_STRIDE - 2) = (uint16_t)f0; }
        // The window ends after the last kept cell; if that is the row's last cell, the row gap keeps it open while it stays
        // within X of the best (:139-147; H >= E at a kept cell), and a closing (MIN, MIN) cell follows unless the block ends.
        // (b_size is N + 1 in a block with N <= X, where row 0 ran off the end: nothing is appended then.)
        {
            const bool ext = lastkept >= n0 - 1;
            const int e_end = lastkeptH - 1;
            const int bsz0 = ext ? n0 : lastkept + 1;
            const int cnt = max(min(ext ? e_end - (best - X) + 1 : 0, N - bsz0), 0);
            const int sent = bsz0 + cnt < N ? 1 : 0;
            const int bnew = bsz0 + lane;
            const bool tail = lane < cnt;
            if (lane < cnt + sent) stHF(bnew, tail ? e_end - lane : X_MIN_SCORE, tail ? e_end - lane - 1 : X_MIN_SCORE);
            if (tail && bnew - f0 < STRIDE - 2) srow[bnew - f0] = XS_GAP_IN_A;
            b_size = bsz0 + cnt + sent;
        }
        if (!WIDE && (b_size - first_b > XW_RING - 2 || b_size - f0 > XW_STRIDE - 2)) { o.overflow = 2; return; }
        __builtin_amdgcn_wave_barrier();
    }
    o.ae = ae; o.be = be;
    // ---- traceback (:165-210) fused with script_to_aligned_string + trim_mismatch_end, as in the lane kernel
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    // window reload: the bytes were written by this wave, so they are read past the L1 (agent scope) — all XW_WIN / 256 loads of a
    // lane in flight at once (volatile loads would serialise one memory round trip each)
    auto load_window = [&](int word0) {
        constexpr int NW = XW_WIN / 4 / 64;
        uint32_t v[NW];
#pragma unroll
        for (int j = 0; j < NW; ++j) v[j] = __hip_atomic_load((const uint32_t*)st + word0 + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
        for (int j = 0; j < NW; ++j) S.win[lane + 64 * j] = v[j];
    };
    int wbase = 1 << 30;                                 // first scratch byte in the LDS window; nothing loaded yet
    // row registers: lane l holds the bytes of cells rstart + l and rstart + 64 + l
    auto fetch_row = [&](int a, int& r, int& rs) {
        if (WIDE) { rs = __builtin_amdgcn_readfirstlane((int)S.rstart[a]); return; }      // bytes are read one by one below
        const int off = a * XW_STRIDE;
        if (off < wbase) {
            __builtin_amdgcn_wave_barrier();
            wbase = max(off + XW_STRIDE - XW_WIN, 0);
            load_window(wbase / 4);
            __builtin_amdgcn_wave_barrier();
        }
        const uint8_t* wb = (const uint8_t*)S.win + (off - wbase);
        r = (int)wb[lane] | ((int)wb[64 + lane] << 8);
        rs = (__builtin_amdgcn_readlane(r, 62) >> 8) | (__builtin_amdgcn_readlane(r, 63) & 0xff00);
    };
    auto wide_byte = [&](int a, int b, int rs) -> int {
        const int idx = a * XW_WSTRIDE + (b - rs);
        if (idx < wbase || idx >= wbase + XW_WIN) {
            __builtin_amdgcn_wave_barrier();
            wbase = max(((idx + 4) & ~3) - XW_WIN, 0);
            load_window(wbase / 4);
            __builtin_amdgcn_wave_barrier();
        }
        return __builtin_amdgcn_readfirstlane((int)((const uint8_t*)S.win)[idx - wbase]);
    };
    int a_index = ae, b_index = be;
    // One step = one byte: the op is the previous one while its extension bit is set in this byte, else the byte's own
    // (bit 4 continues GAP_IN_A, bit 6 GAP_IN_B; bit 5 is never set and stands in for "after a substitution").
    int st_op = XS_SUB;
    int n = 0, nmatch = 0, m = 0, found = 0, want_l1 = 0, first = 0, l1 = 0;
    int sn_n = 0, sn_a = 0, sn_b = 0, sn_m = 0;          // the walk's state right after the first run of 4 matches
    if constexpr (WIDE) {
        int cr = 0, crs, nr = 0, nrs = 0;
        fetch_row(a_index, cr, crs);
        if (a_index > 0) fetch_row(a_index - 1, nr, nrs);
        while (a_index > 0 || b_index > 0) {
            const int byte = wide_byte(a_index, b_index, crs);
            const int sh = 4 + (st_op & 1) + ((st_op >> 2) << 1);
            const int op = ((byte >> sh) & 1) ? st_op : (byte & XS_OP_MASK);
            st_op = op;
            const int cq = op != XS_GAP_IN_A, ct = op != XS_GAP_IN_B;
            const int cm = cq & ct & (byte >> 7);
            a_index -= cq;
            b_index -= ct;
            const int pk = cq | (ct << 1) | (cm << 2);
            first = n == 0 ? pk : first;
            l1 = want_l1 ? pk : l1;
            want_l1 = 0;
            ++n;
            nmatch += cm;
            m = cm ? m + 1 : 0;
            if (m == 4 && !found) { found = 1; want_l1 = 1; sn_n = n; sn_a = a_index; sn_b = b_index; sn_m = nmatch; }
            if (cq) {
                cr = nr; crs = nrs;
                if (a_index > 0) fetch_row(a_index - 1, nr, nrs);
            }
        }
    } else {
        // Diagonal runs at once: lane i looks at cell (a - i, b - i) in the LDS window; once a step is a substitution the walk
        // stays on the diagonal for as long as the cells' own op bits say SUB, so the length of the run is one ballot + one
        // bit scan, and the counters (columns, matches, the first run of four matches) follow from the run's match bits.
        // A step that is not a substitution (1 in 8 on ONT-style reads) is taken alone.
        constexpr int WROWS = XW_WIN / XW_STRIDE;
        int wlo = 1 << 29;                               // first row in the window
        while (a_index > 0 || b_index > 0) {
            if (a_index < wlo || a_index >= wlo + WROWS || (wlo > 0 && a_index - wlo < 12)) {
                __builtin_amdgcn_wave_barrier();
                wlo = max(a_index - (WROWS - 1), 0);
                load_window(wlo * (XW_STRIDE / 4));
                __builtin_amdgcn_wave_barrier();
            }
            const int ri = a_index - lane, bi = b_index - lane;
            const bool vl = lane < 32 && ri >= max(wlo, 1) && bi >= 1;
            const uint8_t* wrow = (const uint8_t*)S.win + (vl ? ri - wlo : a_index - wlo) * XW_STRIDE;
            const int rs = *(const uint16_t*)(wrow + XW_STRIDE - 2);
            const int ci = (vl ? bi : b_index) - rs;
            const int byte = wrow[min(max(ci, 0), XW_STRIDE - 3)];
            const unsigned long long sm = __builtin_amdgcn_ballot_w64(vl && ci >= 0 && ci < XW_STRIDE - 2 && (byte & XS_OP_MASK) == XS_SUB);
            const uint32_t mm = (uint32_t)__builtin_amdgcn_ballot_w64((byte & XS_MATCH) != 0);
            const int byte0 = __builtin_amdgcn_readfirstlane(byte);
            const int sh = 4 + (st_op & 1) + ((st_op >> 2) << 1);
            const int op = ((byte0 >> sh) & 1) ? st_op : (byte0 & XS_OP_MASK);
            st_op = op;
            if (op != XS_SUB || !(sm & 1)) {             // a gap step (or a cell the window does not vouch for): alone
                const int cq = op != XS_GAP_IN_A, ct = op != XS_GAP_IN_B;
                const int cm = cq & ct & (byte0 >> 7);
                a_index -= cq;
                b_index -= ct;
                const int pk = cq | (ct << 1) | (cm << 2);
                first = n == 0 ? pk : first;
                l1 = want_l1 ? pk : l1;
                want_l1 = 0;
                ++n;
                nmatch += cm;
                m = cm ? m + 1 : 0;
                if (m == 4 && !found) { found = 1; want_l1 = 1; sn_n = n; sn_a = a_index; sn_b = b_index; sn_m = nmatch; }
                continue;
            }
            const uint32_t s32 = (uint32_t)sm;
            const int r = s32 == 0xffffffffu ? 32 : __builtin_ctz(~s32);      // >= 1
            const uint32_t mask = r >= 32 ? 0xffffffffu : ((1u << r) - 1u);
            const uint32_t M = mm & mask;                                      // bit i: step i is a match
            const int pk0 = 3 | ((M & 1u) << 2);
            first = n == 0 ? pk0 : first;
            l1 = want_l1 ? pk0 : l1;
            want_l1 = 0;
            const uint32_t inv = ~M & mask;                                    // the mismatches of the run
            if (!found) {
                const int z = inv ? __builtin_ctz(inv) : r;                    // leading matches
                int hit = -1;                                                  // the step at which the streak first reaches 4
                if (m + z >= 4) hit = 3 - m;
                else {
                    const uint32_t Q = M & (M >> 1) & (M >> 2) & (M >> 3);
                    if (Q) hit = __builtin_ctz(Q) + 3;
                }
                if (hit >= 0) {
                    found = 1;
                    sn_n = n + hit + 1; sn_a = a_index - (hit + 1); sn_b = b_index - (hit + 1);
                    sn_m = nmatch + __builtin_popcount(M & ((2u << hit) - 1u));
                    if (hit + 1 < r) l1 = 3 | (int)(((M >> (hit + 1)) & 1u) << 2);
                    else want_l1 = 1;
                }
            }
            m = inv ? r - 1 - (31 - __builtin_clz(inv)) : m + r;               // matches after the last mismatch
            n += r;
            nmatch += __builtin_popcount(M);
            a_index -= r;
            b_index -= r;
        }
    }
    o.n = n; o.nmatch = nmatch;
    o.l0q = first & 1; o.l0t = (first >> 1) & 1; o.l0m = first >> 2;
    o.l1q = l1 & 1; o.l1t = (l1 >> 1) & 1; o.l1m = l1 >> 2;
    o.acnt = found ? sn_n : n; o.qcnt = found ? ae - sn_a : ae; o.tcnt = found ? be - sn_b : be; o.mtail = found ? sn_m : nmatch;
    o.trim_ok = found && (n - o.acnt >= 2);
}


// ------------------------------------------------------------------------------------------------------------------
// xdrop_block_ring — the ring instantiation of round 4: same row formulation, organised around what the counters said bounds it
// (profiles/r04a_config5_cell_*: both issue ports at ~40 %, 41 % of the wave time waiting on LDS round trips, three stores per
// row and 907 GB of scratch traffic per 0.75 s launch):
//   * the gap tail behind the window (:139-147) is not a second phase: the lanes behind the window run the same cell code with
//     (H', F') = (MIN, MIN) and no diagonal, their row-gap value is exactly the tail's e_end - l, and "kept" is the tail's own
//     condition (still within X of the best, b < N) — so a row is its passes and a few scalar instructions;
//   * every lane stores its (H, F) and its script byte, no predicated stores: slots of cells outside the next window are never read
//     (the ring holds 128 cells, a row at most 126); the closing (MIN, MIN) cell is one uniform store after the passes;
//   * a row's first column lives in LDS (rs[]), not in bytes 126-127 of its script row: one store per pass is all that leaves the CU,
//     and the script rows are packed: the first 64 cells of row a at byte 64 a of the scratch, cells 64.. of the rows that have them
//     (one in ten) in a second array behind it — the scratch traffic was what capped the kernel (waves 16 -> 28: flat), now two rows
//     share a cache line and a window of 24 rows is 1.5 KB (+ 1.5 KB only when one of its rows has a second chunk);
//   * the score of a dropped cell's row gap (H of the nearest kept cell to the left - 1) is a scalar for every lane behind the last
//     or in front of the first kept cell; only a row with a hole between kept cells takes the per-lane look-up;
//   * the query bases of 64 rows sit in a register (one v_readlane per row instead of an LDS round trip).
// Script cells are 4 bits (round 5; one byte per cell before: 64 bytes per row written and read back were 400 GB per launch on a
// config-5 grid cell against 1.8 GB of algorithmic bytes):
//   bits 0-1  the cell's own op: 0 = substitution over a mismatch, 1 = substitution over a match, 2 = GAP_IN_A, 3 = GAP_IN_B
//   bit 2     EXT_A (the column gap into this cell extends), bit 3 EXT_B (the row gap extends)          (xdrop_gapalign.h:13-34)
// A lane keeps the nibbles of EIGHT consecutive rows in a register (row a in bits 4 (a & 7) ..) and stores the register once per
// group of eight rows: group g of a wave's scratch = 64 dwords, lane l's dword = the cells rs[a] + l of rows 8 g .. 8 g + 7 — one
// coalesced 256-byte store per eight rows instead of a 64-byte store per row.  Cells 64.. of the rows that have them (one row in ten)
// go the same way into a second array of groups behind the first; a bit per group says whether it was written.
#define XR_GROUPS ((X_MAXN + 2 + 7) / 8 + 2)                 // groups of eight rows; two spare groups, because the traceback's first window
                                                             // loads groups (wlo >> 3) + 2 and + 3 with wlo up to (X_MAXN & ~15) - 16
#define XR_STATE_BYTES ((size_t)XR_GROUPS * 256 * 2)        // first chunks, second chunks
#define XN_GA 2
#define XN_GB 3
#define XN_EXT_A 4
#define XN_EXT_B 8
struct XrLds {
    int2 HF[XW_RING];
    uint8_t Tb[X_MAXN + 72 + 64];        // Tb[b] = target base b - 1 (idle and tail lanes read up to 128 cells behind the window)
    uint16_t rs[X_MAXN + 4];             // first column of every row's script cells
    uint32_t win[2][4][64];              // traceback window: [chunk][group & 3][lane], 32 rows
    uint32_t two[(XR_GROUPS + 31) / 32 + 1];   // groups that wrote a second chunk (one spare word: issue_groups reads words (g0 >> 5) and + 1)
};

__device__ void xdrop_block_ring(XrLds& S, const XView& q, int qidx, int M, const XView& t, int tidx, int N, uint8_t* __restrict__ st, XBlockOut& o) {
    o.ae = o.be = 0; o.n = o.nmatch = 0; o.qcnt = o.tcnt = o.acnt = o.mtail = 0; o.trim_ok = 0; o.overflow = 0;
    o.l0q = o.l0t = o.l0m = o.l1q = o.l1t = o.l1m = 0;
    o.cells = o.rows = 0;
#ifdef MECAT_XD_STATS
    o.tk_stage = o.tk_rows = o.tk_trace = o.n_rows2 = o.n_win = o.n_steps = o.n_qfill = 0;
#endif
    if (M <= 0 || N <= 0) return;
    const int lane = lane_id();
    const int X = 30;
#ifdef MECAT_XD_STATS
    unsigned long long tk0 = __builtin_amdgcn_s_memtime();
#define XD_TICK(field) do { const unsigned long long _t = __builtin_amdgcn_s_memtime(); o.field += _t - tk0; tk0 = _t; } while (0)
#define XD_COUNT(field) (++o.field)
#else
#define XD_TICK(field) do { } while (0)
#define XD_COUNT(field) do { } while (0)
#endif
    __builtin_amdgcn_wave_barrier();
    for (int i = lane; i < N; i += 64) S.Tb[i + 1] = (uint8_t)xv_at(t, tidx + i);
    const int n_init = min(N, X);
    uint32_t* __restrict__ st32 = (uint32_t*)st;                    // first chunks: group g at dwords 64 g ..
    uint32_t* __restrict__ st32b = st32 + (size_t)XR_GROUPS * 64;   // second chunks
    if (lane <= n_init) S.HF[lane] = make_int2(-lane, -lane - 1);
    static_assert((((X_MAXN & ~15) - 16) >> 3) + 3 < XR_GROUPS, "the traceback's first window stays inside the wave's group arrays");
    if (lane < (int)(sizeof(S.two) / sizeof(S.two[0]))) S.two[lane] = 0;
    if (lane == 0) S.rs[0] = 0;
    // row 0 (xdrop_gapalign.cpp:45-57): cells 1 .. n_init are GAP_IN_A
    uint32_t acc = (lane >= 1 && lane <= n_init) ? (uint32_t)XN_GA : 0u, acc2 = 0u;
    bool g2 = false;                                     // a row of the current group wrote second-chunk cells
    auto flush = [&](const int g) __attribute__((always_inline)) {
        st32[g * 64 + lane] = acc;
        acc = 0u;
        if (g2) {
            st32b[g * 64 + lane] = acc2;
            acc2 = 0u;
            S.two[g >> 5] |= 1u << (g & 31);              // (every lane, one address)
            g2 = false;
        }
    };
    int b_size = n_init + 1, best = 0, first_b = 0, ae = 0, be = 0;
    int qreg = 0, cellacc = 0;
    const int wmax = N <= X ? -1 : 64;                   // N <= X: row 0 ran off the end of the target (b_size = N + 1), every row through the general code
    __builtin_amdgcn_wave_barrier();
    XD_TICK(tk_stage);
    // One flat loop with one exit: the compiler lays out a loop nest with early exits as a state machine of flag registers and
    // re-tested branches — 30 scalar instructions per row of pure control flow in the round-4 kernel, and the scalar side is what
    // bounds the row (moving five vector instructions of the common row to nine scalar ones cost 10 %).  A row that ends the block
    // (no kept cell, or a window the ring cannot hold) ends the loop through its bound.
    int status = 0;                                      // 1: no kept cell in the row (the block's last), 2: the window outgrew the ring
    int Mlim = M;
    // the query bases of 64 rows in a register, row a in lane a & 63: one v_readlane per row (idle lanes load a valid base: no divergent branch)
    qreg = xv_at(q, qidx + min(max(lane - 1, 0), M - 1));
    asm volatile("v_mov_b32 %0, %0" : "+v"(qreg));      // the load is waited for here, not at the row loop's first v_readlane (vmcnt also counts the groups' stores)
    XD_COUNT(n_qfill);
    int a = 1;
    for (; a <= Mlim; ++a) {
        if ((a & 7) == 0) {
            flush((a >> 3) - 1);
            if ((a & 63) == 0) {
                qreg = xv_at(q, qidx + min(a + lane - 1, M - 1));
                asm volatile("v_mov_b32 %0, %0" : "+v"(qreg));
                XD_COUNT(n_qfill);
            }
        }
        const int sh = (a & 7) << 2;
        const int AC = __builtin_amdgcn_readlane(qreg, a & 63);
        const int f0 = first_b, n0 = b_size;
        S.rs[a] = (uint16_t)f0;                                          // the row's first column (every lane, one address)
        // The common row — one pass, kept cells without a hole between them, gap tail inside the pass — with nothing but what it needs:
        // no carries, no selects for the cases it excludes (checked on the kept mask before anything is stored; a row that fails
        // the check is redone by the general code below).  Masks are compared as masks (s_bfm), "none" is s_ff1's own -1, every lane behind
        // or in front of the kept cells stores (MIN, MIN) — the closing cell is one of them, the others are never read — and the DP cells
        // are counted per lane.
        bool done = false;
        if (__builtin_expect(n0 - f0 <= wmax, 1)) {                     // (wmax = 64; -1 in a special block)
            const int b = f0 + lane;
            const bool in0 = b < n0;
            const bool live = b < N;                                     // (n0 <= N outside the special blocks)
            int2 hf = S.HF[b & (XW_RING - 1)];
            const int tb = S.Tb[b];
            const int Hp = in0 ? hf.x : X_MIN_SCORE, Fp = in0 ? hf.y : X_MIN_SCORE;
            const bool mt = AC == tb;
            const int left = xw_shr1(Hp, 0);
            const int diag = (in0 && lane != 0) ? left + (mt ? 1 : -1) : X_MIN_SCORE;
            const int Mv = max(diag, Fp);
            const int incl = xw_scan_max(Mv + b);
            const int Ec = xw_shr1(incl, XW_NEG) - b;
            const int Hc = max(Mv, Ec);
            const unsigned long long livem = __builtin_amdgcn_ballot_w64(live);
            const unsigned long long exm = __builtin_amdgcn_ballot_w64(Hc > best) & livem;
            int j1;                                                       // first cell above the best so far; -1 (as unsigned: above every lane) if none
            asm("s_ff1_i32_b64 %0, %1" : "=s"(j1) : "s"(exm));
            const int thr = best - X + ((unsigned)lane > (unsigned)j1 ? 1 : 0);
            const bool ge = Hc >= thr;
            const unsigned long long km = __builtin_amdgcn_ballot_w64(ge) & livem;
            int fkl;
            asm("s_ff1_i32_b64 %0, %1" : "=s"(fkl) : "s"(km));
            const int nk = __builtin_popcountll(km);
            unsigned long long want;                                      // nk ones from bit fkl on (0 for nk = 64: such a row fails the test)
            asm("s_bfm_b64 %0, %1, %2" : "=s"(want) : "s"(nk), "s"(fkl));
            const int lk1 = fkl + nk;                                     // one behind the last kept lane; no kept cell: -1, as unsigned above 63
            // kept cells are one run that ends in front of lane 63 (a kept lane 63 may have a gap tail behind it: general code)
            int runend;                                                   // lk1 if the kept cells are one run, else 64 (compare + select: the compiler's version is seven instructions)
            asm("s_cmp_eq_u64 %1, %2\n\ts_cselect_b32 %0, %3, 64" : "=s"(runend) : "s"(km), "s"(want), "s"(lk1) : "scc");
            if (__builtin_expect((unsigned)runend < 64u, 1)) {
                const bool kept = live && ge;
                cellacc += in0 ? 1 : 0;
                const int Hlk = __builtin_amdgcn_readlane(Hc, lk1 - 1);
                const int et = lane >= lk1 ? Hlk - 1 : X_MIN_SCORE;
                int nib = (mt && in0) ? 1 : 0;
                nib = diag < Fp ? XN_GB : nib;
                nib = Mv < (kept ? Ec : et) ? XN_GA : nib;
                const bool xa = kept && Fp >= Hc, xb = kept && in0 && Ec >= Hc;
                nib |= (xa ? XN_EXT_A : 0) | (xb ? XN_EXT_B : 0);
                acc |= (uint32_t)nib << sh;
                S.HF[b & (XW_RING - 1)] = make_int2(kept ? Hc : X_MIN_SCORE, kept ? Hc - 1 : X_MIN_SCORE);
                // a new best (all cells above the old one hold old + 1) moves (ae, be) to the first of them; the window of the next row:
                // from the first kept cell to one behind the last, where lane lk1 wrote the closing (MIN, MIN) cell unless the block ends
                // there.  Written out: the compiler tests each condition twice over (compare, select a mask, compare the mask).
                const int nbe = f0 + j1;
                asm("s_cmp_lg_u64 %[ex], 0\n\ts_cselect_b32 %[ae], %[a], %[ae]\n\ts_cselect_b32 %[be], %[nbe], %[be]\n\ts_addc_u32 %[best], %[best], 0"
                    : [ae] "+s"(ae), [be] "+s"(be), [best] "+s"(best) : [ex] "s"(exm), [a] "s"(a), [nbe] "s"(nbe) : "scc");
                first_b = f0 + fkl;
                asm("s_add_i32 %[bs], %[f0], %[lk1]\n\ts_cmp_lt_i32 %[bs], %[N]\n\ts_addc_u32 %[bs], %[bs], 0" : [bs] "=&s"(b_size) : [f0] "s"(f0), [lk1] "s"(lk1), [N] "s"(N) : "scc");
                done = true;
            }
        }
        if (!done) {
            const int nlim = max(n0, N);                 // cells that may be kept: the window, and behind it the gap tail while b < N
            o.cells += n0 - f0;
            int bb = best, rowarg = -1, fk = -1, lk = -1, lkH = 0;
            int runP = XW_NEG, prevHp = 0, et_carry = X_MIN_SCORE;
            auto pass = [&](auto first_tag, const int c0) __attribute__((always_inline)) {
                constexpr bool FIRST = decltype(first_tag)::value;
                const int b = c0 + lane;
                const bool in0 = b < n0;
                const bool live = b < nlim;
                int2 hf = S.HF[b & (XW_RING - 1)];
                const int tb = S.Tb[b];
                const int Hp = in0 ? hf.x : X_MIN_SCORE, Fp = in0 ? hf.y : X_MIN_SCORE;
                const bool mt = AC == tb;
                const int left = xw_shr1(Hp, FIRST ? 0 : prevHp);
                const bool dok = FIRST ? (in0 && lane != 0) : in0;            // no diagonal into the row's first cell, none into the gap tail
                const int diag = dok ? left + (mt ? 1 : -1) : X_MIN_SCORE;
                const int Mv = max(diag, Fp);
                const int incl = xw_scan_max(Mv + b);                         // (lanes behind the window hold MIN + b: below every real cell)
                int pex = xw_shr1(incl, XW_NEG);
                if (!FIRST) pex = max(pex, runP);
                const int Ec = pex - b;                                       // (the row's first cell: NEG - b, below MIN like the reference's MIN)
                const int Hc = max(Mv, Ec);
                const unsigned long long livem = __builtin_amdgcn_ballot_w64(live);
                const unsigned long long in0m = __builtin_amdgcn_ballot_w64(in0);
                // a row's scores exceed the best of the rows before by at most the match reward: all cells above bb hold bb + 1
                const unsigned long long exm = __builtin_amdgcn_ballot_w64(Hc > bb) & livem;
                const int j1 = exm ? __ffsll((long long)exm) - 1 : 64;
                const int thr = bb - X + (lane > j1 ? 1 : 0);
                const bool ge = Hc >= thr;
                const unsigned long long km = __builtin_amdgcn_ballot_w64(ge) & livem;
                const bool kept = live && ge;
                const int fkl = km ? __ffsll((long long)km) - 1 : 64;
                const int lkl = km ? 63 - __clzll((long long)km) : 64;       // 64: no kept cell in this pass
                const int Hlk = __builtin_amdgcn_readlane(Hc, lkl & 63);
                // the row gap a dropped cell compares with: H of the nearest kept cell to the left - 1 (MIN if none)
                int et;
                const unsigned long long holes = km ? (in0m & ~km & ((1ull << (lkl & 63)) - 1ull) & ~((2ull << (fkl & 63)) - 1ull)) : 0ull;
                if (__builtin_expect(holes == 0ull, 1)) et = lane > lkl ? Hlk - 1 : et_carry;
                else {
                    const unsigned long long lower = km & ((1ull << lane) - 1ull);
                    const int jsrc = lower ? 63 - __clzll((long long)lower) : 0;
                    const int Hj = __shfl(Hc, jsrc);
                    et = lower ? Hj - 1 : et_carry;
                }
                int nib = (mt && in0) ? 1 : 0;
                nib = diag < Fp ? XN_GB : nib;
                nib = Mv < (kept ? Ec : et) ? XN_GA : nib;
                const bool xa = kept && Fp >= Hc, xb = kept && in0 && Ec >= Hc;
                nib |= (xa ? XN_EXT_A : 0) | (xb ? XN_EXT_B : 0);
                S.HF[b & (XW_RING - 1)] = make_int2(kept ? Hc : X_MIN_SCORE, kept ? Hc - 1 : Fp);
                if (FIRST) acc |= (uint32_t)nib << sh;
                else { acc2 |= (uint32_t)nib << sh; g2 = true; }
                // carries
                if (exm) { if (rowarg < 0) rowarg = c0 + j1; bb = bb + 1; }
                if (km) {
                    if (fk < 0) fk = c0 + fkl;
                    lk = c0 + lkl;
                    lkH = Hlk;
                    et_carry = Hlk - 1;
                }
                if (FIRST) { runP = __builtin_amdgcn_readlane(incl, 63); prevHp = __builtin_amdgcn_readlane(Hp, 63); }
            };
            pass(std::integral_constant<bool, true>(), f0);
            // a second pass: the window is wider than 64 cells, or the gap tail runs on behind lane 63
            if (n0 - f0 > 64 || (lk == f0 + 63 && f0 + 64 < N)) pass(std::integral_constant<bool, false>(), f0 + 64);
            XD_COUNT(n_rows2);
            if (bb > best) { best = bb; ae = a; be = rowarg; }
            if (fk < 0) { first_b = n0; status = 1; Mlim = 0; }
            else {
                first_b = fk;
                b_size = lk + 1;
                if (b_size < N) { S.HF[b_size & (XW_RING - 1)] = make_int2(X_MIN_SCORE, X_MIN_SCORE); ++b_size; }      // the closing cell
                if (b_size - first_b > XW_RING - 2 || b_size - f0 > XW_STRIDE - 2) { status = 2; Mlim = 0; }
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    const int arow = a - 1;                              // the last row that was run
    o.rows = arow;
    if (status == 2) { o.overflow = 2; return; }
    flush(arow >> 3);                                    // the rows of the last, partial group
    for (int off = 32; off; off >>= 1) cellacc += __shfl_xor(cellacc, off);
    o.cells += cellacc;
    o.ae = ae; o.be = be;
    XD_TICK(tk_rows);
    // ---- traceback (:165-210) fused with script_to_aligned_string + trim_mismatch_end: whole diagonal runs per step (see xdrop_block_w),
    // and the gap step behind a run in the same turn (the cell the run stopped at is already in a lane's hands).
    __threadfence();
    __builtin_amdgcn_wave_barrier();
    // The groups are read back through a 32-row LDS window (four group slots, group g in slot g & 3) that moves down the rows 16 at a
    // time, so the two groups it needs next are known in advance: their loads are in flight (in registers) while the walk is still
    // inside the current window.  The dwords were written by this wave, so they are read past the L1 (agent scope).
    uint32_t nxt[2], nxt2[2];
    bool nxt_two = false;
    auto issue_groups = [&](const int g0) __attribute__((always_inline)) {         // groups g0, g0 + 1
        nxt[0] = __hip_atomic_load(st32 + g0 * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        nxt[1] = __hip_atomic_load(st32 + g0 * 64 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const uint32_t tw = (uint32_t)((((unsigned long long)S.two[(g0 >> 5) + 1] << 32) | S.two[g0 >> 5]) >> (g0 & 31)) & 3u;
        nxt_two = __builtin_amdgcn_readfirstlane((int)tw) != 0;
        if (nxt_two) {
            nxt2[0] = __hip_atomic_load(st32b + g0 * 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            nxt2[1] = __hip_atomic_load(st32b + g0 * 64 + 64 + lane, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    auto commit_groups = [&](const int g0) __attribute__((always_inline)) {
        S.win[0][g0 & 3][lane] = nxt[0];
        S.win[0][(g0 + 1) & 3][lane] = nxt[1];
        if (nxt_two) {
            S.win[1][g0 & 3][lane] = nxt2[0];
            S.win[1][(g0 + 1) & 3][lane] = nxt2[1];
        }
    };
    int a_index = ae, b_index = be;
    int st_op = 0;                                       // the op of the step before: 0 = substitution, XN_GA, XN_GB
    int n = 0, nmatch = 0, m = 0, found = 0, want_l1 = 0, first = 0, l1 = 0;
    int sn_n = 0, sn_a = 0, sn_b = 0, sn_m = 0;
    int wlo = max((ae & ~15) - 16, 0);                   // first row in the window (a multiple of 16): rows [wlo, wlo + 32)
    issue_groups((wlo >> 3) + 2);
    commit_groups((wlo >> 3) + 2);
    issue_groups(wlo >> 3);
    commit_groups(wlo >> 3);
    if (wlo > 0) issue_groups((wlo >> 3) - 2);
    __builtin_amdgcn_wave_barrier();
    XD_COUNT(n_win);
    // One turn of the walk: the cells (a - i, b - i) of the diagonal in the lanes, the op of the step (the previous one while its extension
    // bit is set in this cell, else the cell's own), then either a step that is not a substitution, or the whole run of substitutions the
    // cells' own ops vouch for plus the gap step behind it.  Two loops share it: the first also looks for the tail's run of four matches
    // (trim_mismatch_end) and notes the column types the stitching needs; once those are known (a turn or two into the walk) the second
    // only counts columns and matches — the walk is a scalar program, and the bookkeeping of the first loop is most of its instructions.
    auto turn = [&](auto lean_tag) __attribute__((always_inline)) {
        constexpr bool LEAN = decltype(lean_tag)::value;
        if (wlo > 0 && a_index - wlo < 8) {              // (a step ends at row wlo - 1 at the lowest: inside the next window)
            __builtin_amdgcn_wave_barrier();
            commit_groups((wlo >> 3) - 2);
            wlo -= 16;
            if (wlo > 0) issue_groups((wlo >> 3) - 2);
            __builtin_amdgcn_wave_barrier();
            XD_COUNT(n_win);
        }
        XD_COUNT(n_steps);
        const int ri = a_index - lane, bi = b_index - lane;
        const bool vl = lane < 32 && ri >= max(wlo, 1) && bi >= 1;
        const int row = vl ? ri : a_index;
        const int rs = S.rs[row];
        const int ci = (vl ? bi : b_index) - rs;
        const int cc = min(max(ci, 0), XW_STRIDE - 1);
        // (a cell on the walk lies inside its row's window, so its chunk is in the window arrays; what an idle lane reads is not used)
        const uint32_t word = S.win[cc >> 6][(row >> 3) & 3][cc & 63];
        const int nib = (int)((word >> ((row & 7) << 2)) & 15u);
        const uint32_t vm = (uint32_t)__builtin_amdgcn_ballot_w64(vl && ci >= 0 && ci < XW_STRIDE);      // cells the window vouches for
        const uint32_t sm = (uint32_t)__builtin_amdgcn_ballot_w64((nib & 2) == 0) & vm;                  // ... whose own op is a substitution
        const uint32_t mm = (uint32_t)__builtin_amdgcn_ballot_w64((nib & 1) != 0);
        const int nib0 = __builtin_amdgcn_readfirstlane(nib);
        const int ext = st_op == XN_GA ? (nib0 & XN_EXT_A) : st_op == XN_GB ? (nib0 & XN_EXT_B) : 0;
        const int own = (nib0 & 2) ? (nib0 & 3) : 0;
        int op = ext ? st_op : own;
        int r = 1, cq = 1, ct = 1;                       // steps of this turn; what the (single) step takes when it is not a run
        uint32_t Mm = (uint32_t)nib0 & 1u;                 // match bits of the turn's substitution steps
        if (op != 0) { cq = op != XN_GA; ct = op != XN_GB; Mm = 0u; }
        else if (sm & 1u) {                              // (else: a substitution at a cell of row or column 0, alone)
            r = sm == 0xffffffffu ? 32 : __builtin_ctz(~sm);
            Mm = mm & (r >= 32 ? 0xffffffffu : ((1u << r) - 1u));
        }
        if (!LEAN) {
            const int pk0 = cq | (ct << 1) | ((int)(Mm & 1u) << 2);
            first = n == 0 ? pk0 : first;
            l1 = want_l1 ? pk0 : l1;
            want_l1 = 0;
            if (op != 0) m = 0;
            else {
                const uint32_t mask = r >= 32 ? 0xffffffffu : ((1u << r) - 1u);
                const uint32_t inv = ~Mm & mask;
                if (!found) {
                    const int z = inv ? __builtin_ctz(inv) : r;
                    int hit = -1;
                    if (m + z >= 4) hit = 3 - m;
                    else {
                        const uint32_t Q = Mm & (Mm >> 1) & (Mm >> 2) & (Mm >> 3);
                        if (Q) hit = __builtin_ctz(Q) + 3;
                    }
                    if (hit >= 0) {
                        found = 1;
                        sn_n = n + hit + 1; sn_a = a_index - (hit + 1); sn_b = b_index - (hit + 1);
                        sn_m = nmatch + __builtin_popcount(Mm & ((2u << hit) - 1u));
                        if (hit + 1 < r) l1 = 3 | (int)(((Mm >> (hit + 1)) & 1u) << 2);
                        else want_l1 = 1;
                    }
                }
                m = inv ? r - 1 - (31 - __builtin_clz(inv)) : m + r;
            }
        }
        n += r;
        nmatch += __builtin_popcount(Mm);
        a_index -= cq ? r : 0;
        b_index -= ct ? r : 0;
        // the cell a run stopped at, when the window vouches for it, holds a gap op of its own (after a substitution the cell's own
        // op counts): that step in the same turn
        if (op == 0 && r < 32 && ((vm >> r) & 1u)) {
            op = __builtin_amdgcn_readlane(nib, r) & 3;
            const int gq = op != XN_GA, gt = op != XN_GB;
            a_index -= gq;
            b_index -= gt;
            if (!LEAN) {
                l1 = want_l1 ? (gq | (gt << 1)) : l1;
                want_l1 = 0;
                m = 0;
            }
            ++n;
        }
        st_op = op;
    };
    while ((a_index > 0 || b_index > 0) && (!found || want_l1)) turn(std::false_type{});
    while (a_index > 0 || b_index > 0) turn(std::true_type{});
    // groups that were asked for and never needed are waited for here: left pending, their loads would put a vmcnt(0) wait — which also
    // waits for the stores of the groups before — at the head of the next block's row loop
    asm volatile("" :: "v"(nxt[0]), "v"(nxt[1]), "v"(nxt2[0]), "v"(nxt2[1]));
    o.n = n; o.nmatch = nmatch;
    o.l0q = first & 1; o.l0t = (first >> 1) & 1; o.l0m = first >> 2;
    o.l1q = l1 & 1; o.l1t = (l1 >> 1) & 1; o.l1m = l1 >> 2;
    o.acnt = found ? sn_n : n; o.qcnt = found ? ae - sn_a : ae; o.tcnt = found ? be - sn_b : be; o.mtail = found ? sn_m : nmatch;
    o.trim_ok = found && (n - o.acnt >= 2);
    XD_TICK(tk_trace);
}

template <bool WIDE>
__global__ __launch_bounds__(XW_BLOCK, 6) void xd_extend_w(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                        const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                        const mhip_aln_job* __restrict__ jobs, int n, XDir* __restrict__ dres,
                                                        uint8_t* __restrict__ scratch, unsigned int* __restrict__ cursor,
                                                        unsigned int* __restrict__ ovf_list, unsigned long long* __restrict__ counters,
                                                        const unsigned int* __restrict__ ulist, unsigned int nunits, int force_wide,
                                                        uint8_t* __restrict__ wide_state, const unsigned int* __restrict__ order) {
    constexpr int HFN = WIDE ? XW_WIDE_HF : XW_RING;
    __shared__ typename std::conditional<WIDE, XwLds<XW_WIDE_HF>, XrLds>::type lds[WIDE ? 1 : XW_WAVES];
    auto& S = lds[threadIdx.x >> 6];
    const int lane = lane_id();
    uint8_t* st = scratch + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * (WIDE ? XW_WIDE_BYTES : XR_STATE_BYTES);
    // the ring kernel's own wide state (a block whose window outgrows the ring is redone in place, below): in global memory
    uint8_t* wst = WIDE ? nullptr : wide_state + (size_t)(blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * XW_INPLACE_BYTES;
    unsigned long long nblocks = 0, ncells = 0, nrows = 0, nredone = 0;
#ifdef MECAT_XD_STATS
    unsigned long long xs[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long tk_life = __builtin_amdgcn_s_memtime();
#endif
    while (true) {
        unsigned int unit = 0;
        if (lane == 0) unit = atomicAdd(cursor, 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= nunits) break;
        if (order) unit = (order[unit >> 1] << 1) | (unit & 1u);          // (jobs longest first)
        int qidx = 0, tidx = 0;
        XDir R = {0, 0, 0, 0, 0, 0, 0, 0};
        if (ulist) {                                     // a handed-over unit resumes at the block that overflowed
            const unsigned int* e = ulist + 3 * (size_t)unit;
            unit = e[0]; qidx = (int)e[1]; tidx = (int)e[2];
            R = dres[unit];
        }
        const mhip_aln_job jb = jobs[unit >> 1];
        const int right = unit & 1;
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        XView q, t;
        q.pac = qpac; q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
        t.pac = rpac; t.off = roffs[jb.sid_local].offset; t.comp = 0;
        const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
        if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
        t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
        // (a start point in front of a read — the reference's wrapped seed numbers, pw_impl.cpp:388 — is a job without extension)
        const bool dead = jb.qstart < 0 || jb.sstart < 0;
        const int query_size = dead ? 0 : (right ? qsize - jb.qstart : jb.qstart);
        const int target_size = dead ? 0 : (right ? tsize - jb.sstart : jb.sstart);
        bool handed_over = false;
        while (true) {      // align_ex (xdrop_gapalign.cpp:263-357)
            const int qleft = query_size - qidx, tleft = target_size - tidx;
            int qblk, tblk, last_block;
            if (qleft < X_SEG + 100 || tleft < X_SEG + 100) {
                qblk = min(qleft, (int)(tleft + tleft * 0.2));
                tblk = min(tleft, (int)(qleft + qleft * 0.2));
                last_block = 1;
            } else { qblk = X_SEG; tblk = X_SEG; last_block = 0; }
            XBlockOut o;
            o.overflow = 1;
            if constexpr (!WIDE) {
                xdrop_block_ring(S, q, qidx, qblk, t, tidx, tblk, st, o);
#ifdef MECAT_XD_STATS
                xs[0] += o.tk_stage; xs[1] += o.tk_rows; xs[2] += o.tk_trace; xs[3] += o.n_rows2; xs[4] += o.n_win; xs[5] += o.n_steps; xs[6] += o.n_qfill;
#endif
            }
            else if (!force_wide) xdrop_block_w<false, HFN>(S, q, qidx, qblk, t, tidx, tblk, st, o);      // (the round-1 ring code: the independent implementation)
            ++nblocks;
            R.blocks += 1;
            ncells += (unsigned)o.cells;
            nrows += (unsigned)o.rows;
            if (o.overflow) {
                if constexpr (!WIDE) {
                    // A window wider than the ring (a fraction of a percent of the units have such a block, low-complexity sequence): the
                    // block again, at once, with the WIDE code — its scores, bases and row starts in this wave's global state instead of
                    // LDS (slower per row, and rare), rows of XW_WSTRIDE bytes behind it.  The unit then goes on in the ring.  (Handing
                    // the unit to a second launch made that launch as long as its slowest unit: 13 % of the kernel time on 134 k jobs.)
                    if (wst) {
                        xdrop_block_w<true, XW_WIDE_HF>(*reinterpret_cast<XwLds<XW_WIDE_HF>*>(wst), q, qidx, qblk, t, tidx, tblk, wst + XW_INPLACE_STATE, o);
                        ncells += (unsigned)o.cells;
                        nrows += (unsigned)o.rows;
                        ++nredone;
                    } else { handed_over = true; R.blocks -= 1; break; }      // (no wide state: the block is counted again when it is redone)
                }
                if constexpr (WIDE) {      // only the blocks that need it
                    xdrop_block_w<true, HFN>(S, q, qidx, qblk, t, tidx, tblk, st, o);
                    ncells += (unsigned)o.cells;
                    nrows += (unsigned)o.rows;
                }
            }
            const int full_map = (qblk - o.ae <= 20 || tblk - o.be <= 20);
            if (!full_map || last_block) {
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
        if (lane == 0) {
            dres[unit] = R;                              // final, or the state in front of the block that overflowed
            if (handed_over) {
                unsigned int* e = ovf_list + 1 + 3 * (size_t)atomicAdd(ovf_list, 1u);
                e[0] = unit; e[1] = (unsigned int)qidx; e[2] = (unsigned int)tidx;
            }
        }
    }
    if (lane == 0) {
        atomicAdd(&counters[3], nblocks);
        atomicAdd(&counters[4], ncells);      // DP cells (the slot dw's d-path cells use)
        atomicAdd(&counters[32], nrows);                     // slots 32 / 33 are X-drop's own (9 and 11 belong to seed_cand and dw)
        if (nredone) atomicAdd(&counters[33], nredone);      // blocks redone in place with the wide window
#ifdef MECAT_XD_STATS
        if (!WIDE) {
            atomicAdd(&counters[16], __builtin_amdgcn_s_memtime() - tk_life);      // wave life, ticks
            for (int i = 0; i < 7; ++i) atomicAdd(&counters[17 + i], xs[i]);         // stage, rows, trace ticks; two-pass rows, window loads, traceback steps, query refills
            atomicAdd(&counters[24], 1ull);
        }
#endif
    }
}

// ---- longest first.  The waves pull units from a cursor; a unit is as long as the shorter of the two sequences on its side of the
// start point allows (its reach), and the kernel ends with its last unit: handing the JOBS out in descending order of their longer side's
// reach takes the long units off the kernel's tail (-3 % on 134 k ONT-style jobs).  The sort is stable and by job — the two units of a job
// and the jobs of one read stay neighbours, which is worth more than the order itself (a per-unit scatter lost 4 % to cache misses): one
// stable counting sort over 64 bins of 1 kb in two passes, every thread counting and then placing a contiguous part of the jobs.
#define XO_BINS 64
__device__ __forceinline__ int xo_bin(const mhip_aln_job& jb, const mhip_offset_t* __restrict__ roffs, const mhip_offset_t* __restrict__ qoffs) {
    int reach = 0;
    if (jb.qstart >= 0 && jb.sstart >= 0) {
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        reach = max(min(qsize - jb.qstart, tsize - jb.sstart), min(jb.qstart, jb.sstart));
    }
    return XO_BINS - 1 - min(max(reach, 0) >> 10, XO_BINS - 1);          // bin 0 = the longest
}
// pass 1 (PLACE = false): jobs per bin of every workgroup's stretch -> cnt[bin][workgroup]; xo_scan: exclusive prefix over (bin, workgroup);
// pass 2 (PLACE = true): every thread places its own contiguous part of the stretch behind the threads before it, bin by bin.
#define XO_THREADS 256
template <bool PLACE>
__global__ __launch_bounds__(XO_THREADS) void xo_order_jobs(const mhip_aln_job* __restrict__ jobs, int n, const mhip_offset_t* __restrict__ roffs,
                                                          const mhip_offset_t* __restrict__ qoffs, unsigned int* __restrict__ cnt, unsigned int* __restrict__ order) {
    __shared__ uint16_t tc[XO_BINS][XO_THREADS];        // jobs of thread t in bin b (a thread's part is < 65 536 jobs: the host checks)
    __shared__ unsigned int wsum[XO_THREADS / 64];
    __shared__ unsigned int base;
    const int t = (int)threadIdx.x, G = (int)gridDim.x;
    const int per_block = (n + G - 1) / G, b0 = min(n, (int)blockIdx.x * per_block), b1 = min(n, b0 + per_block);
    const int per = (b1 - b0 + XO_THREADS - 1) / XO_THREADS, lo = min(b1, b0 + t * per), hi = min(b1, lo + per);
    for (int b = 0; b < XO_BINS; ++b) tc[b][t] = 0;
    for (int i = lo; i < hi; ++i) ++tc[xo_bin(jobs[i], roffs, qoffs)][t];
    __syncthreads();
    unsigned int mypos[XO_BINS];
#pragma unroll
    for (int b = 0; b < XO_BINS; ++b) {
        if (t == 0) base = PLACE ? cnt[(size_t)b * G + blockIdx.x] : 0u;
        const unsigned int c = tc[b][t];
        unsigned int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int v = __shfl_up(incl, o);
            if ((t & 63) >= o) incl += v;
        }
        if ((t & 63) == 63) wsum[t >> 6] = incl;
        __syncthreads();
        unsigned int before = base;
        for (int k = 0; k < (t >> 6); ++k) before += wsum[k];
        mypos[b] = before + incl - c;
        if (!PLACE && t == XO_THREADS - 1) cnt[(size_t)b * G + blockIdx.x] = before + incl;
        __syncthreads();
    }
    if (!PLACE) return;
    for (int i = lo; i < hi; ++i) {
        const int b = xo_bin(jobs[i], roffs, qoffs);
        unsigned int p = 0;
#pragma unroll
        for (int k = 0; k < XO_BINS; ++k) if (k == b) p = mypos[k]++;        // (registers: no dynamic indexing)
        order[p] = (unsigned int)i;
    }
}
__global__ __launch_bounds__(1024) void xo_scan(unsigned int* __restrict__ cnt, int m) {      // counts -> first positions, in place; one workgroup
    __shared__ unsigned int wsum[16];
    __shared__ unsigned int carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int i0 = 0; i0 < m; i0 += 1024) {
        const int i = i0 + (int)threadIdx.x;
        const unsigned int c = i < m ? cnt[i] : 0u;
        unsigned int incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned int v = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += v;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned int before = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) before += wsum[k];
        if (i < m) cnt[i] = before + incl - c;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
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
    XDir* d_dres;
    uint8_t* d_s;
    unsigned int* d_cur;
    if (c->scratch("xa_dres", sizeof(XDir) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    unsigned int* d_ovf;
    if (c->scratch("xa_cursor", 64, (void**)&d_cur)) return -1;
    if (c->scratch("xa_ovf", sizeof(unsigned int) * (6 * (size_t)n + 8), (void**)&d_ovf)) return -1;      // count + {unit, qidx, tidx} each
    HIPCHK(hipMemsetAsync(d_cur, 0, 16, c->stream));
    // MECAT_XD_WIDE=1: every block of every unit through the WIDE instantiation (cross-check of the ring instantiation, tests)
    const bool all_wide = getenv("MECAT_XD_WIDE") && atoi(getenv("MECAT_XD_WIDE")) == 1;
    unsigned int nwide = 0;
    if (!all_wide) {
        const int waves = c->num_cus * (getenv("MECAT_XW_WAVES") ? atoi(getenv("MECAT_XW_WAVES")) : 24);
        const int grid = std::min(waves / XW_WAVES, (2 * n + XW_WAVES - 1) / XW_WAVES);
        const size_t launched = (size_t)grid * XW_WAVES;      // both per-wave buffers are sized for the waves this call launches
        if (c->scratch("xw_state", XR_STATE_BYTES * launched, (void**)&d_s)) return -1;
        // MECAT_XD_HANDOVER=1: blocks that outgrow the ring go to the second launch (the round-1 arrangement; tests compare the two)
        uint8_t* d_wst = nullptr;
        if (!(getenv("MECAT_XD_HANDOVER") && atoi(getenv("MECAT_XD_HANDOVER")) == 1) && c->scratch("xw_inplace", XW_INPLACE_BYTES * launched, (void**)&d_wst)) return -1;
        HIPCHK(hipMemsetAsync(d_ovf, 0, sizeof(unsigned int), c->stream));
        // MECAT_XD_ORDER=0: jobs in candidate order
        unsigned int* d_order = nullptr;
        if (!(getenv("MECAT_XD_ORDER") && atoi(getenv("MECAT_XD_ORDER")) == 0) && (size_t)n > (size_t)waves) {
            const int G = (int)std::min<size_t>((size_t)c->num_cus, ((size_t)n + 4095) / 4096 + 1);      // (a thread's part stays far below 65 536 jobs)
            unsigned int* d_cnt;
            if (c->scratch("xo_cnt", sizeof(unsigned int) * XO_BINS * (size_t)G, (void**)&d_cnt)) return -1;
            if (c->scratch("xo_order", sizeof(unsigned int) * (size_t)n, (void**)&d_order)) return -1;
            if ((size_t)n / ((size_t)G * XO_THREADS) >= 65535) d_order = nullptr;
            else {
                LAUNCH(c, "xo_count", xo_order_jobs<false>, G, XO_THREADS, 0, (const mhip_aln_job*)d_jobs, n, (const mhip_offset_t*)ref->d_offs, (const mhip_offset_t*)reads->d_offs, d_cnt, d_order);
                LAUNCH(c, "xo_scan", xo_scan, 1, 1024, 0, d_cnt, XO_BINS * G);
                LAUNCH(c, "xo_place", xo_order_jobs<true>, G, XO_THREADS, 0, (const mhip_aln_job*)d_jobs, n, (const mhip_offset_t*)ref->d_offs, (const mhip_offset_t*)reads->d_offs, d_cnt, d_order);
            }
        }
        LAUNCH(c, "xd_extend_w", xd_extend_w<false>, grid, XW_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_s, d_cur,
               d_ovf, (unsigned long long*)c->d_counters, (const unsigned int*)nullptr, 2u * (unsigned)n, 0, d_wst, (const unsigned int*)d_order);
        HIPCHK(hipMemcpyAsync(&nwide, d_ovf, sizeof(unsigned int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        if (getenv("MECAT_TRACE")) {
            if (d_wst) {
                unsigned long long redone = 0;
                HIPCHK(hipMemcpy(&redone, (unsigned long long*)c->d_counters + 33, sizeof(redone), hipMemcpyDeviceToHost));
                fprintf(stderr, "[mecat_hip] X-drop: %d units, %llu blocks so far redone in place with the wide window\n", 2 * n, redone);
            } else fprintf(stderr, "[mecat_hip] X-drop: %u of %d units need the wide-window path\n", nwide, 2 * n);
        }
    }
    if (nwide > 0 || all_wide) {
        const unsigned int nu = all_wide ? 2u * (unsigned)n : nwide;
        const int wgrid = (int)std::min(nu, 2048u);
        uint8_t* d_w;
        if (c->scratch("xw_wide", XW_WIDE_BYTES * (size_t)wgrid, (void**)&d_w)) return -1;
        LAUNCH(c, "xd_extend_wide", xd_extend_w<true>, wgrid, 64, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const mhip_aln_job*)d_jobs, n, d_dres, d_w, d_cur + 2,
               d_ovf + 6 * (size_t)n + 4, (unsigned long long*)c->d_counters, all_wide ? (const unsigned int*)nullptr : (const unsigned int*)(d_ovf + 1),
               nu, all_wide ? 1 : 0, (uint8_t*)nullptr, (const unsigned int*)nullptr);
    }
    LAUNCH(c, "xd_stitch", xd_stitch, (n + 255) / 256, 256, 0, (const mhip_aln_job*)d_jobs, (const XDir*)d_dres, n, min_align_size,
           (mhip_aln_result*)d_out, (unsigned long long*)c->d_counters);
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
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
        if (j.qstart > qs || j.sstart > ts) {      // (a negative start point is a job without extension)
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
