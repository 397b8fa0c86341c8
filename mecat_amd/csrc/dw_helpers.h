// dw_helpers.h — device helpers shared by the O(ND) aligners (align.hip: mecat2pw's dw; cns_align.hip: mecat2cns' dw):
// views of a read as the aligner walks it (forward / backward, complemented), 16-base windows of the packed volume and of
// LDS-staged blocks, packed snake comparison, wave64 DPP reductions.
#pragma once
#include "common.h"

struct SeqView {
    const uint32_t* pac;
    int64_t off;    // volume offset of the read
    int A, B;       // original index of logical extension position i is A + B * i  (B = +1 / -1)
    int comp;       // 1: complement (reverse-complemented query)
};

__device__ __forceinline__ uint32_t rev_groups(uint32_t x) {     // reverse the 16 2-bit groups of a word
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    return __builtin_bswap32(x);
}
// 16 volume bases starting at idx >= 0, first base in the top 2 bits
__device__ __forceinline__ uint32_t pac_win16(const uint32_t* __restrict__ pac, int64_t idx) {
    const int64_t w = idx >> 4;
    const uint64_t W = ((uint64_t)pac_word(pac, w) << 32) | pac_word(pac, w + 1);
    return (uint32_t)((W << ((idx & 15) << 1)) >> 32);
}
// logical bases i0 .. i0+15 of a view, first in the top 2 bits (bases past the block are don't-care)
__device__ __forceinline__ uint32_t view_word(const SeqView& s, int i0) {
    const int64_t o0 = s.off + s.A + (int64_t)s.B * i0;
    uint32_t w;
    if (s.B > 0) w = pac_win16(s.pac, o0);
    else {
        const int64_t lo = o0 - 15;                     // volume bases lo .. o0, to be reversed
        w = lo >= 0 ? pac_win16(s.pac, lo) : (pac_win16(s.pac, 0) >> ((-lo) << 1));
        w = rev_groups(w);
    }
    return s.comp ? ~w : w;
}
// The same 16 logical bases with base i0 in the LOWEST two bits (base i0 + j at bits 2j .. 2j+1): the layout of dw_extend2's
// staged blocks, where a 16-base window at base x is ({P[w+1], P[w]} >> 2(x & 15)) — v_alignbit takes the shift modulo 32, so
// the shift operand is just x + x (no multiply, no pad word) and the run of equal bases is counted from the low end.
__device__ __forceinline__ uint32_t view_word_le(const SeqView& s, int i0) {
    const int64_t o0 = s.off + s.A + (int64_t)s.B * i0;
    uint32_t w;
    if (s.B > 0) w = rev_groups(pac_win16(s.pac, o0));
    else {
        const int64_t lo = o0 - 15;                     // volume bases lo .. o0: the window already has logical base j at bits 2j
        w = lo >= 0 ? pac_win16(s.pac, lo) : (pac_win16(s.pac, 0) >> ((-lo) << 1));
    }
    return s.comp ? ~w : w;
}
// ({hi, lo} << s) >> 32 for s in 0..30, branch-free (HIP's __funnelshift_l lowers to a divergent branch on s == 0)
__device__ __forceinline__ uint32_t funnel_l(uint32_t lo, uint32_t hi, int s) {
    const uint32_t r = __builtin_amdgcn_alignbit(hi, lo, (32 - s) & 31);
    return s ? r : hi;
}
// 16 bases of a staged block starting at base x
__device__ __forceinline__ uint32_t lds_win16(const uint32_t* P, int x) {
    const int w = x >> 4;
    return funnel_l(P[w + 1], P[w], (x & 15) << 1);      // ({P[w], P[w+1]} << s) >> 32
}

// number of leading equal bases (0..32) of the 32-base windows at Q[x..] and T[y..].  Qp/Tp carry one leading pad
// word (base i lives in word (i >> 4) + 1), so the window is ({P[w], P[w+1], P[w+2]} << s) with s in 2..32 and the
// v_alignbit shift 32 - s in 0..30: no special case for a word-aligned start.
__device__ __forceinline__ int match32(const uint32_t* Q, int x, const uint32_t* T, int y) {
    const int xx = x + 15, yy = y + 15;
    const int wq = xx >> 4, wt = yy >> 4;
    const int hq = 30 - ((xx & 15) << 1), ht = 30 - ((yy & 15) << 1);
    const uint32_t q0 = Q[wq], q1 = Q[wq + 1], q2 = Q[wq + 2], t0 = T[wt], t1 = T[wt + 1], t2 = T[wt + 2];
    const uint32_t dh = __builtin_amdgcn_alignbit(q0, q1, hq) ^ __builtin_amdgcn_alignbit(t0, t1, ht);
    const uint32_t dl = __builtin_amdgcn_alignbit(q1, q2, hq) ^ __builtin_amdgcn_alignbit(t1, t2, ht);
    const int nh = __clz(dh) >> 1, nl = 16 + (__clz(dl) >> 1);      // __clz(0) == 32
    return dh ? nh : nl;
}

// 16-base version (0..16): two words per side.  Snakes between two 15 %-error reads are ~3 bases long, so one 16-base
// window settles 99.5 % of the diagonals at two thirds of match32's instruction count.
// The result for 16 equal bases is 0x7fffffff (v_ffbh_u32 of 0 is -1): callers clamp with min(.., lim, 16).
// v_alignbit uses the low 5 bits of its shift operand: 30 - 2 * ((x + 15) & 15) == -2 * x (mod 32), one v_mul_i32_i24
// (a 32-bit v_mul_lo would be quarter rate).
__device__ __forceinline__ int match16(const uint32_t* Q, int x, const uint32_t* T, int y) {
    const int wq = (x + 15) >> 4, wt = (y + 15) >> 4;
    uint32_t sq, st;            // asm: the optimizer rewrites the multiply as a (quarter rate) v_mul_lo_u32 by 30
    asm("v_mul_i32_i24 %0, -2, %1" : "=v"(sq) : "v"(x));
    asm("v_mul_i32_i24 %0, -2, %1" : "=v"(st) : "v"(y));
    const uint32_t dh = __builtin_amdgcn_alignbit(Q[wq], Q[wq + 1], sq) ^ __builtin_amdgcn_alignbit(T[wt], T[wt + 1], st);
    uint32_t lead;
    asm("v_ffbh_u32 %0, %1" : "=v"(lead) : "v"(dh));
    return (int)(lead >> 1);
}

// match16 for sequences with bases outside A, C, G, T: such a base is a symbol (code in Q / T, a non-zero value in the plane Qn / Tn of
// the same layout; the planes are 0 at A, C, G, T) and equals only the same symbol.  Staged strand views hold the stored code at such a
// base (see unflip_symbols), so "equal" is "code and plane equal".
__device__ __forceinline__ int match16n(const uint32_t* Q, const uint32_t* Qn, int x, const uint32_t* T, const uint32_t* Tn, int y) {
    const int wq = (x + 15) >> 4, wt = (y + 15) >> 4;
    uint32_t sq, st;
    asm("v_mul_i32_i24 %0, -2, %1" : "=v"(sq) : "v"(x));
    asm("v_mul_i32_i24 %0, -2, %1" : "=v"(st) : "v"(y));
    const uint32_t dh = __builtin_amdgcn_alignbit(Q[wq], Q[wq + 1], sq) ^ __builtin_amdgcn_alignbit(T[wt], T[wt + 1], st);
    const uint32_t nq = __builtin_amdgcn_alignbit(Qn[wq], Qn[wq + 1], sq), nt = __builtin_amdgcn_alignbit(Tn[wt], Tn[wt + 1], st);
    const uint32_t d = dh | (nq ^ nt);
    uint32_t lead;
    asm("v_ffbh_u32 %0, %1" : "=v"(lead) : "v"(d));
    return (int)(lead >> 1);
}

// A complemented strand view (SeqView::comp) also complements the codes of the bases that are not A, C, G, T, which are their own
// complement (mecat2asmpw.c:583-590): flip them back — word = 16 codes of the view, plane = the same 16 bases of the plane.
__device__ __forceinline__ uint32_t unflip_symbols(uint32_t word, uint32_t plane) {
    uint32_t m = (plane | (plane >> 1)) & 0x55555555u;
    m |= m << 1;
    return word ^ m;
}

// little-endian staged blocks (view_word_le; no pad word: base i lives in word i >> 4)
__device__ __forceinline__ int match16_le(const uint32_t* Q, int x, const uint32_t* T, int y) {
    const int wq = x >> 4, wt = y >> 4;
    // the shift operands 2x and 2y as v_add_u32 x, x (written out: the optimizer turns x + x into a shift left, and on this chip a
    // shift left occupies the SIMD for 4 cycles where the plain 32-bit add takes 2 — DESIGN.md §3.3, valu_peak)
    uint32_t sx, sy;
    asm("v_add_u32 %0, %1, %1" : "=v"(sx) : "v"(x));
    asm("v_add_u32 %0, %1, %1" : "=v"(sy) : "v"(y));
    const uint32_t d = __builtin_amdgcn_alignbit(Q[wq + 1], Q[wq], sx) ^ __builtin_amdgcn_alignbit(T[wt + 1], T[wt], sy);
    uint32_t tz;                                        // v_ffbl_b32 of 0 is -1: 16 equal bases read as 0x7fffffff
    asm("v_ffbl_b32 %0, %1" : "=v"(tz) : "v"(d));
    return (int)(tz >> 1);
}
__device__ __forceinline__ int match32_le(const uint32_t* Q, int x, const uint32_t* T, int y) {
    const int wq = x >> 4, wt = y >> 4;
    const uint32_t sq = (uint32_t)(x + x), st = (uint32_t)(y + y);
    const uint32_t q0 = Q[wq], q1 = Q[wq + 1], q2 = Q[wq + 2], t0 = T[wt], t1 = T[wt + 1], t2 = T[wt + 2];
    const uint32_t dl = __builtin_amdgcn_alignbit(q1, q0, sq) ^ __builtin_amdgcn_alignbit(t1, t0, st);
    const uint32_t dh = __builtin_amdgcn_alignbit(q2, q1, sq) ^ __builtin_amdgcn_alignbit(t2, t1, st);
    const int nl = dl ? (__builtin_ctz(dl) >> 1) : 16, nh = dh ? 16 + (__builtin_ctz(dh) >> 1) : 32;
    return dl ? nl : nh;
}

// ---- wave64 reductions on the DPP network (no LDS traffic): quad swaps, half-row / row mirrors, row broadcasts.
// Fused v_<op>_dpp steps in inline asm (hipcc emits mov + nop + mov_dpp + op per step); a DPP source written by the
// previous VALU instruction needs two wait states, hence the s_nop 1 between steps (cdna_hip_programming.md §5.7).
#define DPP_REDUCE_ASM(OPC)                                                                           \
    asm volatile("s_nop 1\n\t" OPC " %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OPC " %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"   \
                 "s_nop 1\n\t" OPC " %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"        \
                 "s_nop 1\n\t" OPC " %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"             \
                 "s_nop 1\n\t" OPC " %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"           \
                 "s_nop 1\n\t" OPC " %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"           \
                 "s_nop 1"                                                                             \
                 : "+v"(v));                                                                           \
    return __builtin_amdgcn_readlane(v, 63);
__device__ __forceinline__ int wave_max(int v) { DPP_REDUCE_ASM("v_max_i32_dpp") }
__device__ __forceinline__ int wave_min(int v) { DPP_REDUCE_ASM("v_min_i32_dpp") }

