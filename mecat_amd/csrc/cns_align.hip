// cns_align.hip — the mecat2cns re-aligner on the device.
//
// Replaces ns_banded_sw::Align / dw_in_one_direction / dw / GetAlignment of the reference's src/mecat2cns/dw.cpp
// (:146-553), which mecat2cns calls for up to 200 candidates per template read (mecat_correction.cpp:419-443).  Same
// O(ND) core as align.hip, different rules around it:
//   * max_d = int(2 * error_rate * (q_len + t_len)) (:163), square blocks: 500 x 500, or the rest when <= 600 (:322-326);
//   * no best-point fallback: a block that does not reach an end of its square ends the direction (:302-304, :333);
//   * the whole path is needed: the consensus stage consumes the aligned strings.  O(ND) paths have no mismatch columns,
//     so a column is one of {both bases, target base only, query base only}; the kernel emits 2 bits per column and the
//     host rebuilds "ACGT-" strings from the reads (cns_expand in hip.py / INTEGRATION.md);
//   * a block's tail is cut in front of its last run of four matches and the next block starts there (:336-351);
//   * GetAlignment trims the merged string to its first and last run of four matches (:495-531).
//
// One wave per (candidate, direction); lanes are the diagonals of a d-row.  Every row's furthest-x values go to a per-wave
// global scratch because the traceback walks all of them; the walk pulls them back into LDS a window of rows at a time
// (coalesced copies), which keeps the LDS footprint at 6.3 KB per wave = 24 waves per CU.
#include <algorithm>

#include "cns_fwd.h"
#include "dw_helpers.h"

#define CN_BLOCK 256
#define CN_WAVES (CN_BLOCK / WAVE)
#define CN_SEG 500
#define CN_MAXLEN 600              // a last block is at most CN_SEG + 100 bases per side
#define CN_MAX_D 480               // int(2 * 0.20 * 1200)
#define CN_VLEN (2 * CN_MAX_D + 8)
#define CN_ROW_W 192               // diagonals per row: band of 2 * int(0.3 * 600) = 360 -> 181 + 2
#define CN_SEQ_WORDS 44            // 600 bases + 32 of window slack = 40 words, one leading pad word, slack
#define CN_RING 512                // cells of the traceback's row window in LDS (~20 rows at 15 %); >= CN_MAX_D (reused per row below)
#define CN_GROW ((size_t)CN_MAX_D * CN_ROW_W)

struct CnsLds {
    uint32_t Qp[CN_SEQ_WORDS];
    uint32_t Tp[CN_SEQ_WORDS];
    union {
        int16_t V[CN_VLEN];        // forward rows
        uint16_t tlen[CN_MAX_D];   // afterwards, the path: snake length of the row, bit 15 = the row's indel is a query-only column
    };
    int16_t rmin[CN_MAX_D], rmax[CN_MAX_D];
    uint16_t woff[64];             // traceback: offset of row r in the window, at r mod 64 (a window holds <= 64 consecutive rows)
    uint16_t ring[CN_RING];        // the row window; after the walk: columns before row r's snake, at r
    uint16_t roff[CN_MAX_D];       // where row d starts in the wave's global scratch, in pairs of cells (rows are packed back to back, even length)
};

static_assert(CN_RING >= CN_MAX_D, "the ring doubles as the per-row column prefix");

__device__ __forceinline__ int cns_cell(const CnsLds& S, int r, int k) {
    return (int)S.ring[(int)S.woff[r & 63] + ((k - (int)S.rmin[r]) >> 1)];
}

// rows in the window ending at row r (going down): as many of r, r - 1, ... (at most 64) as fit CN_RING cells.  The rows lie
// back to back in the global scratch, so the window is one contiguous range: it is copied with full-wave 32-bit loads, all in
// flight at once (the rows were written by this wave: agent scope, past the L1).  Sets woff[]; returns the lowest row loaded.
__device__ __forceinline__ int cns_load_window(CnsLds& S, const uint16_t* grow, int r, int lane) {
    const int ns_r = S.rmax[r] >= S.rmin[r] ? (((int)S.rmax[r] - (int)S.rmin[r]) >> 1) + 1 : 0;
    const int end = 2 * (int)S.roff[r] + ns_r;
    const int rr = r - lane;
    const int start = rr >= 0 ? 2 * (int)S.roff[rr] : 0;
    const unsigned long long fits = __ballot(rr >= 0 && end - start <= CN_RING);
    const int count = __popcll(fits);                      // a prefix of the lanes (start decreases with the lane); >= 1 (a row has <= 182 cells)
    const int base = 2 * (int)S.roff[r - count + 1];
    if (lane < count) S.woff[rr & 63] = (uint16_t)(start - base);
    const int words = (end - base + 1) >> 1;               // <= CN_RING / 2
    const uint32_t* g32 = (const uint32_t*)grow + (base >> 1);
    constexpr int NW = CN_RING / 2 / 64;
    uint32_t v[NW];
#pragma unroll
    for (int j = 0; j < NW; ++j) v[j] = lane + 64 * j < words ? __hip_atomic_load(g32 + lane + 64 * j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int j = 0; j < NW; ++j)
        if (lane + 64 * j < words) ((uint32_t*)S.ring)[lane + 64 * j] = v[j];
    return r - count + 1;
}

// ASM = false: mecat2cns (jobs = mhip_aln_job: one start point, both directions from it, the strand on the x sequence).
// ASM = true: the extension of mecat2canu's mecat2asmpw / mecat2trimpw (mecat2canu/src/mecat2asmpw/mecat2asmpw.c:723-841), which is the
// code dw.cpp was derived from — same blocks, same tail rule, same failure rules — with jobs = mhip_asm_job: each direction has its
// own start point and sizes (the left extension starts on the LAST base of the seed 13-mer, the right one on its first: both contain
// the seed), the x sequence is the subject read of the indexed block (always forward), the strand applies to the y sequence (the
// mapped read), and the band is band_frac * block (0.10: `ErrorRate * s_k`, :749) where mecat2cns has 0.3.
template <bool ASM>
__global__ __launch_bounds__(CN_BLOCK, 6) void cns_extend(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                       const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                       const void* __restrict__ jobs_v, int n, double error_rate, double band_frac, int dir_cols_cap,
                                                       uint32_t* __restrict__ ops, CnsDir* __restrict__ dres, uint16_t* __restrict__ gscratch,
                                                       unsigned int* __restrict__ cursor, int* __restrict__ err_flag,
                                                       const uint32_t* __restrict__ rnpac, const uint32_t* __restrict__ qnpac,
                                                       const unsigned int* __restrict__ ulist /*NULL: every unit; else [0] = how many, then the units*/) {
    __shared__ CnsLds lds[CN_WAVES];
    CnsLds& S = lds[threadIdx.x >> 6];
    const int lane = lane_id();
    // mecat2asmpw / mecat2trimpw only: the planes of the bases that are not A, C, G, T (mhip_volume_set_nplane; NULL without such bases), staged
    // beside the blocks' 2-bit words in the part of V[] these tools never reach (their max_d is 120 of CN_MAX_D = 480: rows and path use the
    // first 496 bytes of the union)
    uint32_t* const Qn = (uint32_t*)&S.V[512];
    uint32_t* const Tn = Qn + CN_SEQ_WORDS;
    static_assert(1024 + 2 * CN_SEQ_WORDS * 4 <= CN_VLEN * 2, "the N planes fit behind the rows of an ASM block");
    uint16_t* grow = gscratch + (size_t)(blockIdx.x * CN_WAVES + (threadIdx.x >> 6)) * CN_GROW;
    const size_t dir_words = ((size_t)dir_cols_cap + 15) / 16;
    while (true) {
        unsigned int unit = 0;
        if (lane == 0) unit = atomicAdd(cursor, 1u);
        unit = __shfl(unit, 0);
        if (unit >= (ulist ? ulist[0] : 2u * (unsigned)n)) break;
        if (ulist) unit = ulist[1u + unit];
        const int right = unit & 1;
        SeqView q, t;
        q.pac = qpac; t.pac = rpac;
        int query_size, target_size;
        if (!ASM) {
            const mhip_aln_job jb = ((const mhip_aln_job*)jobs_v)[unit >> 1];
            const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
            q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
            t.off = roffs[jb.sid_local].offset; t.comp = 0;
            const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
            if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
            t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
            query_size = right ? qsize - jb.qstart : jb.qstart;
            target_size = right ? tsize - jb.sstart : jb.sstart;
        } else {
            const mhip_asm_job jb = ((const mhip_asm_job*)jobs_v)[unit >> 1];
            const int tsize = roffs[jb.yid].size, step = right ? 1 : -1;
            q.off = qoffs[jb.xid].offset; q.comp = 0;
            t.off = roffs[jb.yid].offset; t.comp = jb.chain;
            q.A = right ? jb.rx : jb.lx; q.B = step;
            const int ys0 = right ? jb.ry : jb.ly;            // position on the mapped strand
            if (jb.chain) { t.A = tsize - 1 - ys0; t.B = -step; } else { t.A = ys0; t.B = step; }
            query_size = right ? jb.rnx : jb.lnx;
            target_size = right ? jb.rny : jb.lny;
        }
        uint32_t* uops = ops + (size_t)unit * dir_words;

        int extend1 = 0, extend2 = 0, cols = 0, n_ins = 0, n_del = 0;
        int extend_size = min(query_size, target_size);
        bool more = true;
        while (more) {      // dw_in_one_direction, dw.cpp:319-375
            int seg;
            if (extend_size > CN_SEG + 100) seg = CN_SEG;
            else { seg = extend_size; more = false; }
            if (seg <= 0) break;                         // Align "succeeds" on an empty block, then i == extend_size ends it (:356)
            const int band_tol = (int)(band_frac * seg);
            const int max_d = (int)(2.0 * error_rate * (seg + seg));
            const int koff = max_d, band_size = band_tol * 2;
            __builtin_amdgcn_wave_barrier();
            bool has_n = false;                          // (wave-uniform) the staged range of either sequence holds a base that is not A, C, G, T
            for (int w = lane; w < CN_SEQ_WORDS; w += 64) {
                const bool in = w > 0 && (w - 1) * 16 < seg + 32;
                S.Qp[w] = in ? view_word(q, extend1 + (w - 1) * 16) : 0u;
                S.Tp[w] = in ? view_word(t, extend2 + (w - 1) * 16) : 0u;
                if (ASM && (qnpac || rnpac)) {
                    SeqView qn = q, tn = t;
                    qn.pac = qnpac; qn.comp = 0; tn.pac = rnpac; tn.comp = 0;
                    const uint32_t a = (in && qnpac) ? view_word(qn, extend1 + (w - 1) * 16) : 0u;
                    const uint32_t b = (in && rnpac) ? view_word(tn, extend2 + (w - 1) * 16) : 0u;
                    Qn[w] = a; Tn[w] = b;
                    if (q.comp) S.Qp[w] = unflip_symbols(S.Qp[w], a);
                    if (t.comp) S.Tp[w] = unflip_symbols(S.Tp[w], b);
                    has_n |= (a | b) != 0u;      // (a lane may stage more than one word when the staging size grows)
                }
            }
            if (ASM) has_n = __ballot(has_n) != 0ull;
            for (int i = lane; i < 2 * max_d + 4 && i < CN_VLEN; i += 64) S.V[i] = 0;     // :329-330
            __builtin_amdgcn_wave_barrier();

            // ---- Align, forward rows (:172-210)
            int best_m = -1, min_k = 0, max_k = 0;
            int end_d = -1, end_k = 0, end_x = 0;
            int cum = 0;                                 // cells of the rows written so far (every row padded to an even length)
            for (int d = 0; d < max_d; ++d) {
                if (max_k - min_k > band_size) break;
                const int nslot = max_k >= min_k ? ((max_k - min_k) >> 1) + 1 : 0;
                if (lane == 0) { S.rmin[d] = (int16_t)min_k; S.rmax[d] = (int16_t)max_k; S.roff[d] = (uint16_t)(cum >> 1); }
                int mmax = -1, hkey = 0x7fffffff;
                constexpr int MAXJ = (CN_ROW_W + 63) / 64;
                int us[MAXJ];
                const int NJ = (nslot + 63) >> 6;
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    us[j] = -0x40000000;
                    if (j >= NJ) continue;
                    const int tt = lane + 64 * j;
                    const bool act = tt < nslot;
                    const int k = min_k + 2 * tt, kk = k + koff;
                    const int16_t* vp = &S.V[act ? kk - 1 : 0];
                    const int vl = vp[0], vr = vp[2];
                    int x = (k == min_k || (k != max_k && vl < vr)) ? vr : vl + 1;       // :176-179
                    x = act ? x : seg;
                    int y = act ? x - k : 0;
                    bool again;
                    do {
                        const int lim = min(seg - x, seg - y);
                        // (such a base equals only itself, `align` compares characters: mecat2asmpw.c:316-335 feed it the reads as they are)
                        const int m = min((ASM && has_n) ? match16n(S.Qp, Qn, x, S.Tp, Tn, y) : match16(S.Qp, x, S.Tp, y), lim);
                        const int nn = min(m, 16);
                        x += nn; y += nn;
                        again = m > 16;
                    } while (__ballot(again));
                    us[j] = act ? x + y : -0x40000000;
                    if (act) {
                        grow[cum + tt] = (uint16_t)x;
                        mmax = max(mmax, x + y);
                        if (x >= seg || y >= seg) hkey = min(hkey, (kk << 10) | x);           // lowest diagonal first (:198-199)
                    }
                }
                __builtin_amdgcn_wave_barrier();
                // V is updated after every diagonal of the row has read its neighbours (they have the other parity anyway)
#pragma unroll
                for (int j = 0; j < MAXJ; ++j) {
                    const int tt = lane + 64 * j;
                    if (tt < nslot) { const int k = min_k + 2 * tt; S.V[k + koff] = (int16_t)((us[j] + k) >> 1); }
                }
                cum += (nslot + 1) & ~1;
                best_m = max(best_m, wave_max(mmax));
                if (__ballot(hkey != 0x7fffffff)) {      // some diagonal reached an end: the lowest one ends the block
                    hkey = wave_min(hkey);
                    end_d = d; end_k = (hkey >> 10) - koff; end_x = hkey & 1023;
                    break;
                }
                // band (:202-209): first / last diagonal within band_tol of the best; one ballot per 64 diagonals, bit scans on
                // the scalar unit
                int nmin = max_k, nmax = min_k;
                {
                    int first = -1, last = -1;
#pragma unroll
                    for (int j = 0; j < MAXJ; ++j) {
                        if (j >= NJ) continue;
                        const unsigned long long qb = __ballot(lane + 64 * j < nslot && us[j] >= best_m - band_tol);
                        if (qb) {
                            if (first < 0) first = 64 * j + __builtin_ctzll(qb);
                            last = 64 * j + 63 - __builtin_clzll(qb);
                        }
                    }
                    if (first >= 0) { nmin = min_k + 2 * first; nmax = min_k + 2 * last; }
                }
                max_k = nmax + 1;
                min_k = nmin - 1;
                __builtin_amdgcn_wave_barrier();
            }
            if (end_d < 0) break;                        // no end reached: Align returns 0

            // ---- the path, end to start (:222-240): one scalar walk, every lane the same.  Row cd needs its two neighbours in
            // row cd - 1; the rows come back from the global scratch a window at a time.
            __threadfence();
            __builtin_amdgcn_wave_barrier();
            int rstar = -1, x2s = 0, ks = 0;             // the last row (highest d) whose snake has >= 4 matches (:336-343)
            {
                int ck = end_k, x2 = end_x, w_lo = end_d;        // rows [w_lo, ...] are in the window
                for (int cd = end_d; cd >= 0; --cd) {
                    int x1 = 0, pre = ck, px2 = 0;
                    if (cd > 0) {
                        if (cd - 1 < w_lo) {
                            __builtin_amdgcn_wave_barrier();
                            w_lo = cns_load_window(S, grow, cd - 1, lane);
                            __builtin_amdgcn_wave_barrier();
                        }
                        const int cmin = S.rmin[cd], cmax = S.rmax[cd], pmin = S.rmin[cd - 1], pmax = S.rmax[cd - 1];
                        const int kl = ck - 1, kr = ck + 1;
                        const int vl = (kl >= pmin && kl <= pmax) ? cns_cell(S, cd - 1, kl) : 0;
                        const int vr = (kr >= pmin && kr <= pmax) ? cns_cell(S, cd - 1, kr) : 0;
                        if (ck == cmin || (ck != cmax && vl < vr)) { pre = kr; x1 = vr; px2 = vr; }
                        else { pre = kl; x1 = vl + 1; px2 = vl; }
                    }
                    const int len = x2 - x1;
                    if (rstar < 0 && len >= 4) { rstar = cd; x2s = x2; ks = ck; }
                    if (lane == 0) S.tlen[cd] = (uint16_t)(len | (cd > 0 && ck > pre ? 0x8000 : 0));   // from k - 1: query base only
                    ck = pre;
                    x2 = px2;
                }
            }
            __builtin_amdgcn_wave_barrier();

            // ---- columns before each row's snake
            int carry = 0;
            for (int r0 = 0; r0 <= end_d; r0 += 64) {
                const int r = r0 + lane;
                const bool in = r <= end_d;
                const int len = in ? (int)(S.tlen[r] & 0x7FFF) : 0;
                int incl = len + (in && r > 0 ? 1 : 0);          // the row's indel column + its snake
                for (int o = 1; o < 64; o <<= 1) {
                    const int v = __shfl_up(incl, o);
                    if (lane >= o) incl += v;
                }
                if (in) S.ring[r] = (uint16_t)(carry + incl - len);     // columns before the snake of row r (its indel included)
                carry += __shfl(incl, 63);
            }
            __builtin_amdgcn_wave_barrier();
            const int ncols = carry, end_y = end_x - end_k;
            int kept_cols, kept_q, kept_t;
            if (more) {
                if (rstar < 0) break;
                const int lens = (int)(S.tlen[rstar] & 0x7FFF);
                kept_cols = (int)S.ring[rstar] + lens - 4;
                kept_q = x2s - 4;
                kept_t = x2s - ks - 4;
                if (kept_q == 0) break;                  // "i == ALN_SIZE" (:350)
            } else {
                kept_cols = ncols; kept_q = end_x; kept_t = end_y;
                if (end_x == 0) break;                   // "i == extend_size" (:356)
            }
            if (cols + kept_cols > dir_cols_cap) { if (lane == 0) atomicExch(err_flag, 1); break; }
            // ---- emit: 0 = both bases (the buffer is zero-filled), 1 = target base only, 2 = query base only
            for (int r0 = 1; r0 <= end_d; r0 += 64) {
                const int r = r0 + lane;
                const bool in = r <= end_d;
                const int c = in ? (int)S.ring[r] - 1 : 0x7fffffff;          // column of the row's indel
                const bool put = in && c < kept_cols;
                const int op = put ? ((S.tlen[r] & 0x8000) ? 2 : 1) : 0;
                if (put) {
                    const int gc = cols + c;
                    atomicOr(&uops[gc >> 4], (uint32_t)op << ((gc & 15) << 1));
                }
                n_ins += __popcll(__ballot(op == 1));
                n_del += __popcll(__ballot(op == 2));
            }
            cols += kept_cols;
            extend1 += kept_q;
            extend2 += kept_t;
            extend_size = min(query_size - extend1, target_size - extend2);
        }
        if (lane == 0) { CnsDir D = {cols, extend1, extend2, n_ins, n_del, 0}; dres[unit] = D; }
    }
}


// ---- the two-kernel re-aligner (cns_fwd.h): shares of the row log, and the paths ---------------------------------------------------

// rows a unit may log: an O(ND) block of size s runs at most max_d = 4 e s rows (dw.cpp:163) and typically (e_q + e_t) s of them; the
// share is 2.7 e of the unit's reach plus one block that fails at max_d (a unit that outgrows it is handed over to cns_extend).
// blocks: one per 256 bases of reach + 4 (a block that contributes advances by about its size; a unit with more blocks than records is
// handed over to cns_extend as well).
__global__ __launch_bounds__(256) void cns_caps(const mhip_offset_t* __restrict__ roffs, const mhip_offset_t* __restrict__ qoffs,
                                                const mhip_aln_job* __restrict__ jobs, int n, double error_rate, uint32_t* __restrict__ caps,
                                                uint32_t* __restrict__ bcaps, int bshift, uint32_t bextra) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < 2 * n) {
        const mhip_aln_job jb = jobs[u >> 1];
        const int right = u & 1;
        const int qsize = qoffs[jb.qid_local].size, tsize = roffs[jb.sid_local].size;
        const int qs = right ? qsize - jb.qstart : jb.qstart, ts = right ? tsize - jb.sstart : jb.sstart;
        const int ext = (jb.qstart < 0 || jb.sstart < 0) ? 0 : max(min(qs, ts), 0);
        caps[u] = (uint32_t)(2.7 * error_rate * ext) + (uint32_t)(4.0 * error_rate * (CN_SEG + 100)) + 64u;
        bcaps[u] = (uint32_t)(ext >> bshift) + bextra;      // (8, 4 unless the test knob says otherwise)
    }
}
// caps -> first record of every unit, out[m] = total; one workgroup
__global__ __launch_bounds__(1024) void cns_scan(const uint32_t* __restrict__ caps, int m, uint32_t* __restrict__ out, unsigned long long* __restrict__ total) {
    caps += (size_t)blockIdx.x * (size_t)m;            // workgroup b: array b of m counts -> array b of m + 1 positions, total[b]
    out += (size_t)blockIdx.x * ((size_t)m + 1);
    total += blockIdx.x;
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < m; t0 += 1024) {
        const int i = t0 + (int)threadIdx.x;
        const unsigned long long w = i < m ? (unsigned long long)caps[i] : 0ull;
        unsigned long long incl = w;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long v = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += v;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) before += wsum[k];
        if (i < m) out[i] = (uint32_t)(before + incl - w);
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[m] = (uint32_t)carry; *total = carry; }
}

// cns_trace — one LANE per block record.  Backwards over the block's row records from its end cell: bit (k - min_k) / 2 of a row says
// which neighbour of the row below the cell came from (the reference's traceback reads the same off its stored V values, dw.cpp:222-240),
// min_k of the row below follows from that row's cut; the steps go into a per-lane bit array in LDS.  Forwards again from (0, 0): row d
// is one column with a base of one sequence only — the query's when the step came from k - 1 — and then the snake, re-measured on the
// packed reads (the forward pass compared exactly these bases); columns at or beyond kept_cols (the cut in front of the last four
// matches) are dropped.  Columns: 0 = both bases (the buffer is zero-filled), 1 = target base only, 2 = query base only.
#define CT_BLOCK 256
#define CT_WORDS ((CN_MAX_D + 31) / 32)
__global__ __launch_bounds__(CT_BLOCK) void cns_trace(const uint32_t* __restrict__ rpac, const mhip_offset_t* __restrict__ roffs,
                                                      const uint32_t* __restrict__ qpac, const mhip_offset_t* __restrict__ qoffs,
                                                      const mhip_aln_job* __restrict__ jobs, const CnsBlockRec* __restrict__ blocks, unsigned int nb,
                                                      const CnsRowRec* __restrict__ rowlog, CnsDir* __restrict__ dres, uint32_t* __restrict__ ops,
                                                      size_t dir_words, int* __restrict__ err_flag) {
    __shared__ uint32_t path[CT_BLOCK / 64][CT_WORDS][64];
    const int lane = lane_id(), wv = (int)threadIdx.x >> 6;
    for (unsigned int b0 = blockIdx.x * CT_BLOCK; b0 < nb; b0 += gridDim.x * CT_BLOCK) {
        const unsigned int b = b0 + threadIdx.x;
        if (b >= nb) continue;
        const CnsBlockRec B = blocks[b];
        if (B.unit == 0xffffffffu) continue;                     // (a record of its unit's share that no block took)
        if (dres[B.unit].pad < 0 || B.end_d >= CN_MAX_D) {      // (a unit that was handed over after this block: cns_extend redoes it)
            if (B.end_d >= CN_MAX_D) atomicExch(err_flag, 3);
            continue;
        }
        // ---- backwards: the steps
        // The records of rows end_d .. 1 are read in aligned groups of four (64 bytes, four 16-byte loads of one line): a lane that fetched
        // one record per step pulled the same line in from memory up to four times (the lanes of a wave walk 64 different blocks; r05:
        // 62.9 GB fetched for 19 GB of records).  Row d's first diagonal follows from row d + 1's by row d's own cut, so a record is all
        // a step needs.
        {
            int ck = B.end_k, mk = B.end_mk, wi = B.end_d >> 5;
            uint32_t word = 0;
            bool bad = false;
            unsigned int rec = B.log0 + (unsigned)B.end_d;                 // record of the row in hand
            const unsigned int rec_last = B.log0 + 1u;                     // record of row 1
            const uint4* log4 = (const uint4*)rowlog;
            while (rec >= rec_last && B.end_d >= 1) {
                const unsigned int g = rec & ~3u;
                uint4 R4[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) R4[k] = log4[g + (unsigned)k];      // (records in front of the block or behind its end row: loaded, not used)
#pragma unroll
                for (int k = 3; k >= 0; --k) {
                    const unsigned int r = g + (unsigned)k;
                    if (r > rec || r < rec_last) continue;
                    const uint4 R = R4[k];                                   // {lo, hi, meta, top}
                    const int cd = (int)(r - B.log0);
                    if (cd != B.end_d) mk = mk - 2 * (int)((R.z >> 8) & 0x7fu) + 1;
                    const int t = (ck - mk) >> 1;
                    bad |= t < 0 || t >= (int)(R.z & 0xffu);                 // (never: the path stays inside its rows)
                    const uint32_t bit = ((t < 32 ? R.x : t < 64 ? R.y : R.w) >> (t & 31)) & 1u;
                    if ((cd >> 5) != wi) { path[wv][wi][lane] = word; word = 0; wi = cd >> 5; }
                    word |= bit << (cd & 31);
                    ck += bit ? -1 : 1;
                }
                if (g <= rec_last) break;
                rec = g - 1u;
            }
            path[wv][wi][lane] = word;
            if (bad) atomicExch(err_flag, 3);
        }
        // ---- forwards: the columns
        const mhip_aln_job jb = jobs[B.unit >> 1];
        const int right = (int)(B.unit & 1u);
        SeqView q, t;
        q.pac = qpac; t.pac = rpac;
        {
            const int qsize = qoffs[jb.qid_local].size;
            q.off = qoffs[jb.qid_local].offset; q.comp = jb.chain;
            t.off = roffs[jb.sid_local].offset; t.comp = 0;
            const int qs0 = right ? jb.qstart : jb.qstart - 1, step = right ? 1 : -1;
            if (jb.chain) { q.A = qsize - 1 - qs0; q.B = -step; } else { q.A = qs0; q.B = step; }
            t.A = right ? jb.sstart : jb.sstart - 1; t.B = step;
        }
        uint32_t* uops = ops + (size_t)B.unit * dir_words;
        int x = 0, y = 0, c = 0, n_ins = 0, n_del = 0;
        // 32 bases of either sequence stay in two registers (bases qb .. qb + 31 of the block, first base in the low bits): a row moves
        // along by three or four bases, so a window is refilled once in four rows instead of being fetched and turned for every snake step
        int qb = 0, tb = 0;
        uint32_t q0 = view_word_le(q, B.qidx), q1 = view_word_le(q, B.qidx + 16), t0 = view_word_le(t, B.tidx), t1 = view_word_le(t, B.tidx + 16);
        auto snake = [&]() {
            const int lim = min(B.seg - x, B.seg - y);
            int run = 0;
            while (run < lim) {
                const int xx = x + run, yy = y + run;
                while (xx >= qb + 16) { qb += 16; q0 = q1; q1 = view_word_le(q, B.qidx + qb + 16); }
                while (yy >= tb + 16) { tb += 16; t0 = t1; t1 = view_word_le(t, B.tidx + tb + 16); }
                const uint32_t dlt = __builtin_amdgcn_alignbit(q1, q0, (uint32_t)(xx - qb) << 1) ^ __builtin_amdgcn_alignbit(t1, t0, (uint32_t)(yy - tb) << 1);
                const int m = min(dlt ? (__builtin_ctz(dlt) >> 1) : 16, lim - run);
                run += m;
                if (m < 16) break;
            }
            x += run; y += run; c += run;
        };
        // the columns of a word are gathered in a register and written once: with a plain store when the word is this block's alone, with
        // an atomic OR when it holds the block's first or last kept column (its neighbours in the unit write the rest of it)
        const int gc_lo = B.col0, gc_hi = B.col0 + B.kept_cols - 1;
        int cw = gc_lo >> 4;
        uint32_t wacc = 0;
        auto flush_word = [&]() {
            if (wacc) {
                if (cw == (gc_lo >> 4) || cw == (gc_hi >> 4)) atomicOr(&uops[cw], wacc); else uops[cw] = wacc;
            }
            wacc = 0;
        };
        snake();
        for (int d = 1; d <= B.end_d && c < B.kept_cols; ++d) {
            const uint32_t bit = (path[wv][d >> 5][lane] >> (d & 31)) & 1u;
            x += (int)bit; y += 1 - (int)bit;
            n_del += (int)bit; n_ins += 1 - (int)bit;
            const int gc = B.col0 + c;
            if ((gc >> 4) != cw) { flush_word(); cw = gc >> 4; }
            wacc |= (bit ? 2u : 1u) << ((gc & 15) << 1);
            c += 1;
            snake();
        }
        flush_word();
        if (B.kept_cols == (B.end_x + (B.end_x - B.end_k) + B.end_d) / 2 && x != B.end_x) atomicExch(err_flag, 4);      // (a whole block ends in its end cell)
        if (n_ins) atomicAdd(&dres[B.unit].ins, n_ins);
        if (n_del) atomicAdd(&dres[B.unit].del, n_del);
    }
}

__device__ __forceinline__ int cns_op(const uint32_t* __restrict__ w, int c) { return (int)((w[c >> 4] >> ((c & 15) << 1)) & 3u); }

// dw's merge + GetAlignment's trimming (dw.cpp:397-480, 495-531).  Merged column c: c < L.cols -> left op L.cols - 1 - c, else
// right op c - L.cols.  One thread per candidate; the scans stop at the first run of four matches from either end.
__global__ void cns_stitch(const mhip_aln_job* __restrict__ jobs, const CnsDir* __restrict__ dres, const uint32_t* __restrict__ ops,
                           int dir_cols_cap, int n, int min_aln, mhip_cns_result* __restrict__ out, unsigned long long* __restrict__ counters) {
    const int i0 = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i0 < n;          // (lanes behind the last candidate redo it and keep nothing: the reductions below want whole waves)
    const int i = live ? i0 : n - 1;
    const size_t dir_words = ((size_t)dir_cols_cap + 15) / 16;
    const CnsDir L = dres[2 * i], R = dres[2 * i + 1];
    const uint32_t* lw = ops + (size_t)(2 * i) * dir_words;
    const uint32_t* rw = lw + dir_words;
    mhip_cns_result r;
    memset(&r, 0, sizeof(r));
    r.left_cols = L.cols; r.right_cols = R.cols;
    r.query_start = jobs[i].qstart - L.qbases; r.query_end = jobs[i].qstart + R.qbases;
    r.target_start = jobs[i].sstart - L.tbases; r.target_end = jobs[i].sstart + R.tbases;
    const int size = L.cols + R.cols;
    r.ins = L.ins + R.ins; r.del = L.del + R.del; r.mat = size - r.ins - r.del;
    bool ok = size >= min_aln;
    int first = 0, last = 0, qrb = 0, trb = 0, qre = 0, tre = 0;
    if (ok) {
        int eit = 0, k;
        for (k = 0; k < size && eit < 4; ++k) {
            const int op = k < L.cols ? cns_op(lw, L.cols - 1 - k) : cns_op(rw, k - L.cols);
            if (op != 1) ++qrb;
            if (op != 2) ++trb;
            if (op == 0) ++eit; else eit = 0;
        }
        if (eit < 4) ok = false;
        first = k - 4; qrb -= 4; trb -= 4;
        if (ok) {
            for (k = size - 1, eit = 0; k >= 0 && eit < 4; --k) {
                const int op = k < L.cols ? cns_op(lw, L.cols - 1 - k) : cns_op(rw, k - L.cols);
                if (op != 1) ++qre;
                if (op != 2) ++tre;
                if (op == 0) ++eit; else eit = 0;
            }
            if (eit < 4) ok = false;
            last = k + 4 + 1; qre -= 4; tre -= 4;
        }
    }
    r.ok = ok ? 1 : 0;
    if (ok) {
        r.qoff = r.query_start + qrb; r.qend = r.query_end - qre;
        r.soff = r.target_start + trb; r.send = r.target_end - tre;
        r.first_col = first; r.last_col = last;
    }
    if (live) out[i] = r;
    // work counters (mecat_hip.h): aligned query bases and alignments like dw_stitch; slot 34: the algorithmic bytes of the job (SURVEY.md
    // §8d's extension figure for an aligner that returns its columns): the two spans at 2 bits a base, the columns at 2 bits, the record
    unsigned long long ab = (ok && live) ? (unsigned long long)(r.query_end - r.query_start) : 0ull, cnt = (ok && live) ? 1ull : 0ull;
    unsigned long long by = !live ? 0ull : (unsigned long long)((r.query_end - r.query_start) + (r.target_end - r.target_start) + size) / 4ull + sizeof(mhip_cns_result);
    for (int o = 32; o; o >>= 1) { ab += __shfl_xor(ab, o); cnt += __shfl_xor(cnt, o); by += __shfl_xor(by, o); }
    if ((threadIdx.x & 63) == 0) { atomicAdd(&counters[6], ab); atomicAdd(&counters[7], cnt); atomicAdd(&counters[34], by); }
}

extern "C" {

int mhip_cns_align_candidates_dev(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const void* d_jobs, int n, double error_rate,
                                  int min_align_size, int dir_cols_cap, void* d_results, void* d_ops) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    if (dir_cols_cap < 16 || (dir_cols_cap & 15)) { mhip_set_error("dir_cols_cap must be a positive multiple of 16"); return -1; }
    if (!(error_rate > 0.0) || error_rate > 0.20) { mhip_set_error("error_rate %.3f outside (0, 0.20]", error_rate); return -1; }
    const int waves_per_cu = 24;                                    // cns_extend: 6.3 KB of LDS per wave, <= 80 VGPRs
    const int max_waves = c->num_cus * waves_per_cu;
    CnsDir* d_dres;
    uint16_t* d_g;
    unsigned int* d_cur;
    int* d_err;
    if (c->scratch("cn_dres", sizeof(CnsDir) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("cn_rows", sizeof(uint16_t) * CN_GROW * (size_t)max_waves, (void**)&d_g)) return -1;
    if (c->scratch("cn_cursor", 64, (void**)&d_cur)) return -1;
    d_err = (int*)(d_cur + 8);
    HIPCHK(hipMemsetAsync(d_cur, 0, 64, c->stream));
    const size_t dir_words = (size_t)dir_cols_cap / 16;
    HIPCHK(hipMemsetAsync(d_ops, 0, sizeof(uint32_t) * dir_words * 2 * (size_t)n, c->stream));
    // MECAT_CNS_KERNEL=1: every unit through cns_extend (rounds 1-4: one unit per wave, every row kept; the tests compare the two)
    if (getenv("MECAT_CNS_KERNEL") && atoi(getenv("MECAT_CNS_KERNEL")) == 1) {
        const int grid = std::min(max_waves / CN_WAVES, (2 * n + CN_WAVES - 1) / CN_WAVES);
        LAUNCH(c, "cns_extend", cns_extend<false>, grid, CN_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
               (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const void*)d_jobs, n, error_rate, 0.3, dir_cols_cap,
               (uint32_t*)d_ops, d_dres, d_g, d_cur, d_err, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const unsigned int*)nullptr);
    } else {
        // forward rows (dw_extend2 under mecat2cns' rules, a 16-byte record per row) -> paths (one lane per block) -> the units the
        // forward pass could not finish, redone by cns_extend.  The row log is sized by the units' shares: jobs are taken in slices whose
        // log stays under MECAT_CNS_LOG_GB (default 24).
        const unsigned long long budget = (unsigned long long)(getenv("MECAT_CNS_LOG_GB") ? std::max(1, atoi(getenv("MECAT_CNS_LOG_GB"))) : 24) << 30;
        int slice = std::min(n, 1 << 18);
        unsigned long long* d_tot;
        if (c->scratch("cnf_totals", 64, (void**)&d_tot)) return -1;
        for (int j0 = 0; j0 < n;) {
            const int nj = std::min(slice, n - j0), nu = 2 * nj;
            const mhip_aln_job* jobs = (const mhip_aln_job*)d_jobs + j0;
            uint32_t *d_caps, *d_base, *d_hand;
            if (c->scratch("cnf_caps", sizeof(uint32_t) * 2 * (size_t)nu, (void**)&d_caps)) return -1;
            if (c->scratch("cnf_base", sizeof(uint32_t) * 2 * ((size_t)nu + 1), (void**)&d_base)) return -1;
            uint32_t* d_bbase = d_base + nu + 1;
            if (c->scratch("cnf_hand", sizeof(uint32_t) * ((size_t)nu + 1), (void**)&d_hand)) return -1;
            HIPCHK(hipMemsetAsync(d_tot, 0, 64, c->stream));
            HIPCHK(hipMemsetAsync(d_hand, 0, sizeof(uint32_t), c->stream));
            int bshift = 8, bextra = 4;
            if (const char* e = getenv("MECAT_CNS_BLOCK_SHARE")) {      // test knob "shift,extra": small shares, so that units run out of block records
                if (sscanf(e, "%d,%d", &bshift, &bextra) != 2 || bshift < 0 || bshift > 30 || bextra < 0) { bshift = 8; bextra = 4; }
            }
            LAUNCH(c, "cns_caps", cns_caps, (nu + 255) / 256, 256, 0, (const mhip_offset_t*)ref->d_offs, (const mhip_offset_t*)reads->d_offs, jobs, nj,
                   error_rate, d_caps, d_caps + nu, bshift, (uint32_t)bextra);
            LAUNCH(c, "cns_scan", cns_scan, 2, 1024, 0, (const uint32_t*)d_caps, nu, d_base, d_tot);      // (two workgroups: rows, block records)
            unsigned long long tot[2] = {0, 0};
            HIPCHK(hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            if ((tot[0] * sizeof(CnsRowRec) > budget || tot[0] >= (1ull << 32)) && nj > 1024) { slice = std::max(1024, nj / 2); continue; }
            if (tot[0] >= (1ull << 32)) { mhip_set_error("mecat2cns aligner: the row log of %d jobs does not fit 32-bit record numbers", nj); return -1; }
            CnsFwdArgs a;
            a.error_rate = error_rate;
            a.logbase = d_base;
            if (tot[1] >= (1ull << 32)) { mhip_set_error("mecat2cns aligner: too many block records for %d jobs", nj); return -1; }
            const unsigned int block_cap = (unsigned int)tot[1];
            a.blockbase = d_bbase;
            if (c->scratch("cnf_rowlog", sizeof(CnsRowRec) * ((size_t)tot[0] + 8), (void**)&a.rowlog)) return -1;      // (+ 8: cns_trace reads whole groups of four records)
            if (c->scratch("cnf_blocks", sizeof(CnsBlockRec) * (size_t)std::max(block_cap, 1u), (void**)&a.blocks)) return -1;
            HIPCHK(hipMemsetAsync(a.blocks, 0xff, sizeof(CnsBlockRec) * (size_t)block_cap, c->stream));
            a.hand_units = d_hand;
            a.dres = d_dres + 2 * (size_t)j0;
            a.dir_cols_cap = dir_cols_cap;
            a.err_flag = d_err;
            if (cns_forward_launch(c, ref, reads, jobs, nj, a)) return -1;
            uint32_t* ops_slice = (uint32_t*)d_ops + 2 * (size_t)j0 * dir_words;
            LAUNCH(c, "cns_trace", cns_trace, c->num_cus * 8, CT_BLOCK, 0, (const uint32_t*)ref->d_pac, (const mhip_offset_t*)ref->d_offs,
                   (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, jobs, (const CnsBlockRec*)a.blocks, block_cap,
                   (const CnsRowRec*)a.rowlog, a.dres, ops_slice, dir_words, d_err);
            // the units the forward pass handed over (0.2 %): cns_extend, which reads their number on the device (nothing waits for the host).
            // (On a second stream beside cns_trace it gained nothing: 118 ms either way on 442 k jobs.)
            HIPCHK(hipMemsetAsync(d_cur, 0, sizeof(unsigned int), c->stream));
            LAUNCH(c, "cns_extend", cns_extend<false>, std::min(max_waves / CN_WAVES, (nu + CN_WAVES - 1) / CN_WAVES), CN_BLOCK, 0, (const uint32_t*)ref->d_pac,
                   (const mhip_offset_t*)ref->d_offs, (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, (const void*)jobs, nj, error_rate, 0.3,
                   dir_cols_cap, ops_slice, a.dres, d_g, d_cur, d_err, (const uint32_t*)nullptr, (const uint32_t*)nullptr, (const unsigned int*)d_hand);
            if (getenv("MECAT_TRACE")) {
                unsigned int st = 0;
                HIPCHK(hipStreamSynchronize(c->stream));
                HIPCHK(hipMemcpy(&st, d_hand, sizeof(unsigned int), hipMemcpyDeviceToHost));
                fprintf(stderr, "[mecat_hip] cns re-aligner: jobs %d..%d, row log %.2f GB, room for %u block records, %u of %d units handed over\n", j0, j0 + nj,
                        (double)tot[0] * sizeof(CnsRowRec) / 1e9, block_cap, st, nu);
            }
            j0 += nj;
        }
    }
    LAUNCH(c, "cns_stitch", cns_stitch, (n + 255) / 256, 256, 0, (const mhip_aln_job*)d_jobs, (const CnsDir*)d_dres, (const uint32_t*)d_ops,
           dir_cols_cap, n, min_align_size, (mhip_cns_result*)d_results, (unsigned long long*)c->d_counters);
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, d_err, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (err == 1) { mhip_set_error("mecat2cns aligner: a direction needed more than dir_cols_cap = %d columns", dir_cols_cap); return -1; }
    if (err) { mhip_set_error("mecat2cns aligner: internal error %d (3: a path left its rows; 4: a path missed its end)", err); return -1; }
    return 0;
}

int mhip_cns_align_candidates(mhip_ctx* c, const mhip_volume* ref, const mhip_volume* reads, const mhip_aln_job* jobs, int n, double error_rate,
                              int min_align_size, int dir_cols_cap, mhip_cns_result* results, uint32_t* ops) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    void *d_jobs, *d_res, *d_ops;
    const size_t ops_bytes = sizeof(uint32_t) * ((size_t)dir_cols_cap / 16) * 2 * (size_t)n;
    if (c->scratch("cn_jobs", sizeof(mhip_aln_job) * (size_t)n, &d_jobs)) return -1;
    if (c->scratch("cn_res", sizeof(mhip_cns_result) * (size_t)n, &d_res)) return -1;
    if (c->scratch("cn_ops", ops_bytes, &d_ops)) return -1;
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, sizeof(mhip_aln_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (mhip_cns_align_candidates_dev(c, ref, reads, d_jobs, n, error_rate, min_align_size, dir_cols_cap, d_res, d_ops)) return -1;
    HIPCHK(hipMemcpyAsync(results, d_res, sizeof(mhip_cns_result) * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    if (ops) HIPCHK(hipMemcpyAsync(ops, d_ops, ops_bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"

// The extension of mecat2asmpw / mecat2trimpw for a batch of candidates (mecat2canu/src/mecat2asmpw/mecat2asmpw.c:723-841): per job and
// direction the columns of the alignment the tool strings together in left_store / right_store (2 bits each, in extension order: 0 both
// bases, 1 y base only — a gap in the subject's row —, 2 x base only) and the bases of either sequence they cover.  x = reads of
// `block`, y = reads of `reads`.  dirs[2 * i + d] = {cols, xbases, ybases, ins, del, 0} for direction d (0 left, 1 right) of job i,
// ops[(2 * i + d) * dir_cols_cap / 16 ...] its columns.  What the tool does with them afterwards (string_check, the seed overlap of the
// two directions, coordinates, jscore, the output line) is host work: mecat_amd/host/asmpw.cpp.
namespace {

// exclusive prefix over the directions' word counts (ceil(cols / 16)); one block, the total behind the last entry
__global__ __launch_bounds__(1024) void ae_word_offsets(const CnsDir* __restrict__ dres, int nd, unsigned long long* __restrict__ offs) {
    __shared__ unsigned long long wsum[16];
    __shared__ unsigned long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < nd; t0 += 1024) {
        const int i = t0 + (int)threadIdx.x;
        const unsigned long long w = i < nd ? (unsigned long long)((dres[i].cols + 15) >> 4) : 0ull;
        unsigned long long incl = w;
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned long long v = __shfl_up(incl, o);
            if ((int)(threadIdx.x & 63) >= o) incl += v;
        }
        if ((threadIdx.x & 63) == 63) wsum[threadIdx.x >> 6] = incl;
        __syncthreads();
        unsigned long long before = carry;
        for (int k = 0; k < (int)(threadIdx.x >> 6); ++k) before += wsum[k];
        if (i < nd) offs[i] = before + incl - w;
        __syncthreads();
        if (threadIdx.x == 1023) carry = before + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) offs[nd] = carry;
}

// one wave per direction: its words from the fixed-stride array to their place in the dense one
__global__ __launch_bounds__(256) void ae_pack(const CnsDir* __restrict__ dres, int nd, const uint32_t* __restrict__ ops, size_t dir_words,
                                               const unsigned long long* __restrict__ offs, uint32_t* __restrict__ dense) {
    const int lane = (int)threadIdx.x & 63;
    for (size_t k = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); k < (size_t)nd; k += (size_t)gridDim.x * 4) {
        const int w = (dres[k].cols + 15) >> 4;
        const uint32_t* src = ops + k * dir_words;
        uint32_t* dst = dense + offs[k];
        for (int j = lane; j < w; j += 64) dst[j] = src[j];
    }
}

// validation + launch shared by mhip_asm_extend and mhip_asm_extend_run: results stay in the context's scratch (cn_dres, cn_ops)
int asm_extend_launch(mhip_ctx* c, const mhip_volume* block, const mhip_volume* reads, const mhip_asm_job* jobs, int n, int dir_cols_cap,
                      CnsDir** out_dres, uint32_t** out_ops) {
    if (dir_cols_cap < 16 || (dir_cols_cap & 15)) { mhip_set_error("dir_cols_cap must be a positive multiple of 16"); return -1; }
    for (int i = 0; i < n; ++i) {
        const mhip_asm_job& j = jobs[i];
        if (j.xid < 0 || j.xid >= block->num_reads || j.yid < 0 || j.yid >= reads->num_reads) {
            mhip_set_error("extension job %d: read index out of range", i);
            return -1;
        }
        // start points inside the reads, lengths inside what lies in front of / behind them (the kernel and the host's string
        // rebuild both index the packed reads with these)
        const int xs = block->h_offs[(size_t)j.xid].size, ys = reads->h_offs[(size_t)j.yid].size;
        const bool ok = j.lx >= 0 && j.lx < xs && j.rx >= 0 && j.rx < xs && j.ly >= 0 && j.ly < ys && j.ry >= 0 && j.ry < ys &&
                        j.lnx >= 0 && j.lny >= 0 && j.rnx >= 0 && j.rny >= 0 && j.lnx <= j.lx + 1 && j.lny <= j.ly + 1 &&
                        j.rnx <= xs - j.rx && j.rny <= ys - j.ry;
        if (!ok) {
            mhip_set_error("extension job %d: start points or lengths outside the reads (x %d bases, y %d bases)", i, xs, ys);
            return -1;
        }
    }
    const int max_waves = c->num_cus * 24;
    const int grid = std::min(max_waves / CN_WAVES, (2 * n + CN_WAVES - 1) / CN_WAVES);
    CnsDir* d_dres;
    uint16_t* d_g;
    unsigned int* d_cur;
    void *d_jobs, *d_ops;
    const size_t dir_words = (size_t)dir_cols_cap / 16, ops_bytes = sizeof(uint32_t) * dir_words * 2 * (size_t)n;
    if (c->scratch("cn_dres", sizeof(CnsDir) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("cn_rows", sizeof(uint16_t) * CN_GROW * (size_t)max_waves, (void**)&d_g)) return -1;
    if (c->scratch("cn_cursor", 64, (void**)&d_cur)) return -1;
    if (c->scratch("cn_asmjobs", sizeof(mhip_asm_job) * (size_t)n, &d_jobs)) return -1;
    if (c->scratch("cn_ops", ops_bytes, &d_ops)) return -1;
    int* d_err = (int*)(d_cur + 8);
    HIPCHK(hipMemsetAsync(d_cur, 0, 64, c->stream));
    HIPCHK(hipMemsetAsync(d_ops, 0, ops_bytes, c->stream));
    HIPCHK(hipMemcpyAsync(d_jobs, jobs, sizeof(mhip_asm_job) * (size_t)n, hipMemcpyHostToDevice, c->stream));
    // x = the block (the kernel's "reads" side), y = the mapped reads (its "ref" side); max_d = int(0.10 * (q + t)) = int(2 * 0.05 * ..)
    LAUNCH(c, "asm_extend", cns_extend<true>, grid, CN_BLOCK, 0, (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs,
           (const uint32_t*)block->d_pac, (const mhip_offset_t*)block->d_offs, (const void*)d_jobs, n, 0.05, 0.10, dir_cols_cap,
           (uint32_t*)d_ops, d_dres, d_g, d_cur, d_err, (const uint32_t*)reads->d_npac, (const uint32_t*)block->d_npac, (const unsigned int*)nullptr);
    *out_dres = d_dres;
    *out_ops = (uint32_t*)d_ops;
    return 0;
}

int asm_extend_check(mhip_ctx* c, int dir_cols_cap) {
    unsigned int* d_cur;
    if (c->scratch("cn_cursor", 64, (void**)&d_cur)) return -1;
    int err = 0;
    HIPCHK(hipMemcpyAsync(&err, (int*)(d_cur + 8), sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    HIPCHK(hipGetLastError());
    if (err) { mhip_set_error("mecat2asmpw extension: a direction needed more than dir_cols_cap = %d columns", dir_cols_cap); return -1; }
    return 0;
}

}  // namespace

extern "C" {

int mhip_asm_extend(mhip_ctx* c, const mhip_volume* block, const mhip_volume* reads, const mhip_asm_job* jobs, int n, int dir_cols_cap,
                    int32_t* dirs, uint32_t* ops) {
    HIPCHK(hipSetDevice(c->device));
    if (n <= 0) return 0;
    CnsDir* d_dres;
    uint32_t* d_ops;
    if (asm_extend_launch(c, block, reads, jobs, n, dir_cols_cap, &d_dres, &d_ops)) return -1;
    const size_t ops_bytes = sizeof(uint32_t) * ((size_t)dir_cols_cap / 16) * 2 * (size_t)n;
    HIPCHK(hipMemcpyAsync(dirs, d_dres, sizeof(CnsDir) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(ops, d_ops, ops_bytes, hipMemcpyDeviceToHost, c->stream));
    return asm_extend_check(c, dir_cols_cap);
}

// The same extension with the columns handed over densely, in two calls: _run extends the batch, leaves the results on the device and
// says how many 32-bit words the columns of all directions take (ceil(cols / 16) each); _fetch copies them out — direction k = 2 i + d
// occupies words [word_offs[k], word_offs[k + 1]) of ops_dense.  A direction uses a fraction of dir_cols_cap (the bound is both reads'
// bases on that side), so the fixed-stride form moves 5-10x the bytes over the PCIe link; page-locked host buffers (mhip_host_alloc)
// make the copies asynchronous DMA.  _fetch belongs to the last _run on the context.
int mhip_asm_extend_run(mhip_ctx* c, const mhip_volume* block, const mhip_volume* reads, const mhip_asm_job* jobs, int n, int dir_cols_cap,
                        int64_t* total_words) {
    HIPCHK(hipSetDevice(c->device));
    *total_words = 0;
    c->ae_n = 0;
    if (n <= 0) return 0;
    CnsDir* d_dres;
    uint32_t* d_ops;
    if (asm_extend_launch(c, block, reads, jobs, n, dir_cols_cap, &d_dres, &d_ops)) return -1;
    const size_t dir_words = (size_t)dir_cols_cap / 16;
    unsigned long long* d_offs;
    uint32_t* d_dense;
    if (c->scratch("cn_woffs", sizeof(unsigned long long) * (2 * (size_t)n + 1), (void**)&d_offs)) return -1;
    LAUNCH(c, "ae_word_offsets", ae_word_offsets, 1, 1024, 0, (const CnsDir*)d_dres, 2 * n, d_offs);
    unsigned long long total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_offs + 2 * (size_t)n, sizeof(total), hipMemcpyDeviceToHost, c->stream));
    if (asm_extend_check(c, dir_cols_cap)) return -1;
    if (c->scratch("cn_dense", sizeof(uint32_t) * (size_t)std::max<unsigned long long>(total, 1), (void**)&d_dense)) return -1;
    LAUNCH(c, "ae_pack", ae_pack, (unsigned)std::min<size_t>((2 * (size_t)n + 3) / 4, (size_t)c->num_cus * 32), 256, 0, (const CnsDir*)d_dres, 2 * n,
           (const uint32_t*)d_ops, dir_words, (const unsigned long long*)d_offs, d_dense);
    c->ae_n = n;
    c->ae_total = (int64_t)total;
    *total_words = (int64_t)total;
    return 0;
}

int mhip_asm_extend_fetch(mhip_ctx* c, int n, int32_t* dirs, uint64_t* word_offs, uint32_t* ops_dense) {
    HIPCHK(hipSetDevice(c->device));
    if (n != c->ae_n) { mhip_set_error("mhip_asm_extend_fetch: %d jobs asked for, the last mhip_asm_extend_run on this context had %d", n, c->ae_n); return -1; }
    if (n <= 0) return 0;
    CnsDir* d_dres;
    unsigned long long* d_offs;
    uint32_t* d_dense;
    if (c->scratch("cn_dres", sizeof(CnsDir) * 2 * (size_t)n, (void**)&d_dres)) return -1;
    if (c->scratch("cn_woffs", sizeof(unsigned long long) * (2 * (size_t)n + 1), (void**)&d_offs)) return -1;
    if (c->scratch("cn_dense", sizeof(uint32_t) * (size_t)std::max<int64_t>(c->ae_total, 1), (void**)&d_dense)) return -1;
    HIPCHK(hipMemcpyAsync(dirs, d_dres, sizeof(CnsDir) * 2 * (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(word_offs, d_offs, sizeof(uint64_t) * (2 * (size_t)n + 1), hipMemcpyDeviceToHost, c->stream));
    if (c->ae_total) HIPCHK(hipMemcpyAsync(ops_dense, d_dense, sizeof(uint32_t) * (size_t)c->ae_total, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
