// aln_strings.h — host side of the accept stage (cns_accept.hip): the two gapped strings of an accepted alignment from its packed columns.
//
// Columns are 2-bit ops (0: both sequences advance; 1: gap in the query string; 2: gap in the template string), sixteen per word; the
// left extension's columns come in extension order, i.e. reversed with respect to the merged alignment (reverse(left) then right).
// Column by column (two dependent cursors, two selects, two stores) this was 6 ns per column and core — 4.4 of the 6.3 s of a
// 100 000-template batch on the 16 cores the GPU box gives its host side.  Here four columns at a time: the ops byte indexes a table of
// byte-shuffle masks (which of the next four bases goes where, '-' elsewhere) and of cursor steps; the left part is generated backwards
// from the seam with the cursors walking down, so both parts read their ops in word order.
#pragma once

#include <stdint.h>
#include <string.h>
#include <tmmintrin.h>

namespace alnstr {

struct Lut {
    // forward (right part, columns in order) and backward (left part: output byte j = column 3 - j of the group, bases counted down)
    uint32_t fq_mask[256], fq_fill[256], ft_mask[256], ft_fill[256];
    uint32_t bq_mask[256], bq_fill[256], bt_mask[256], bt_fill[256];
    uint8_t nq[256], nt[256];
    Lut() {
        for (int b = 0; b < 256; ++b) {
            int o[4];
            for (int j = 0; j < 4; ++j) o[j] = (b >> (2 * j)) & 3;
            for (int which = 0; which < 2; ++which) {          // 0: query string (gap at op 1), 1: template string (gap at op 2)
                const int gap = which ? 2 : 1;
                uint8_t fm[4], ff[4], bm[4], bf[4];
                int c = 0;
                for (int j = 0; j < 4; ++j) {
                    const bool has = o[j] != gap;
                    fm[j] = has ? (uint8_t)c : 0x80; ff[j] = has ? 0 : '-';
                    // backward: column j of the group is output byte 3 - j and takes the base (c + 1) below the cursor = byte 3 - c of the
                    // four bytes loaded in front of the cursor
                    bm[3 - j] = has ? (uint8_t)(3 - c) : 0x80; bf[3 - j] = has ? 0 : '-';
                    c += has;
                }
                uint32_t* dst[4] = {which ? ft_mask : fq_mask, which ? ft_fill : fq_fill, which ? bt_mask : bq_mask, which ? bt_fill : bq_fill};
                memcpy(&dst[0][b], fm, 4); memcpy(&dst[1][b], ff, 4); memcpy(&dst[2][b], bm, 4); memcpy(&dst[3][b], bf, 4);
                (which ? nt : nq)[b] = (uint8_t)c;
            }
        }
    }
};
inline const Lut& lut() { static const Lut L; return L; }

__attribute__((target("ssse3"))) inline uint32_t shuffle4(uint32_t four, uint32_t mask) {
    // bytes of `four` picked by the low four bytes of `mask` (0x80: zero); the mask's upper twelve bytes are 0x80
    const __m128i m = _mm_or_si128(_mm_cvtsi32_si128((int)mask), _mm_set_epi32((int)0x80808080, (int)0x80808080, (int)0x80808080, 0));
    return (uint32_t)_mm_cvtsi128_si32(_mm_shuffle_epi8(_mm_cvtsi32_si128((int)four), m));
}

inline int op_at(const uint32_t* w, int k) { return (int)((w[k >> 4] >> ((k & 15) << 1)) & 3u); }

// qs / ts: the bases the columns cover as characters (nq / nt of them, query in the mapped strand's orientation), each with at least 8
// readable bytes in front of and behind the range.  qout / tout: left_cols + right_cols columns, with 8 writable bytes either side.
__attribute__((target("ssse3"))) inline void build(const uint32_t* left, int L, const uint32_t* right, int R, const char* qs, const char* ts, char* qout, char* tout) {
    const Lut& T = lut();
    // bases under the left part = its columns that are not gaps
    int QL = 0, TL = 0;
    {
        int k = 0;
        for (; k + 16 <= L; k += 16) {          // a word at a time: ops 1 and 2 as bit patterns of the sixteen 2-bit groups
            const uint32_t w = left[k >> 4], lo = w & 0x55555555u, hi = (w >> 1) & 0x55555555u;
            QL += 16 - __builtin_popcount(lo & ~hi);
            TL += 16 - __builtin_popcount(hi & ~lo);
        }
        for (; k < L; ++k) { const int op = op_at(left, k); QL += op != 1; TL += op != 2; }
    }
    // right part, forward from the seam
    {
        int qi = QL, ti = TL, k = 0;
        char* qo = qout + L;
        char* to = tout + L;
        for (; k + 4 <= R; k += 4) {
            const uint32_t b = (right[k >> 4] >> ((k & 15) << 1)) & 0xffu;
            uint32_t q4, t4;
            memcpy(&q4, qs + qi, 4); memcpy(&t4, ts + ti, 4);
            const uint32_t qo4 = shuffle4(q4, T.fq_mask[b]) | T.fq_fill[b], to4 = shuffle4(t4, T.ft_mask[b]) | T.ft_fill[b];
            memcpy(qo + k, &qo4, 4); memcpy(to + k, &to4, 4);
            qi += T.nq[b]; ti += T.nt[b];
        }
        for (; k < R; ++k) {
            const int op = op_at(right, k);
            qo[k] = op != 1 ? qs[qi] : '-'; to[k] = op != 2 ? ts[ti] : '-';
            qi += op != 1; ti += op != 2;
        }
    }
    // left part, backwards from the seam: column k of the extension is merged column L - 1 - k
    {
        int qi = QL, ti = TL, k = 0;
        for (; k + 4 <= L; k += 4) {
            const uint32_t b = (left[k >> 4] >> ((k & 15) << 1)) & 0xffu;
            uint32_t q4, t4;
            memcpy(&q4, qs + qi - 4, 4); memcpy(&t4, ts + ti - 4, 4);
            const uint32_t qo4 = shuffle4(q4, T.bq_mask[b]) | T.bq_fill[b], to4 = shuffle4(t4, T.bt_mask[b]) | T.bt_fill[b];
            memcpy(qout + (L - 4 - k), &qo4, 4); memcpy(tout + (L - 4 - k), &to4, 4);
            qi -= T.nq[b]; ti -= T.nt[b];
        }
        for (; k < L; ++k) {
            const int op = op_at(left, k);
            qi -= op != 1; ti -= op != 2;
            qout[L - 1 - k] = op != 1 ? qs[qi] : '-'; tout[L - 1 - k] = op != 2 ? ts[ti] : '-';
        }
    }
}

// the reference form of the same thing (tests): column by column
inline void build_plain(const uint32_t* left, int L, const uint32_t* right, int R, const char* qs, const char* ts, char* qout, char* tout) {
    int qi = 0, ti = 0;
    for (int m = 0; m < L + R; ++m) {
        const int op = m < L ? op_at(left, L - 1 - m) : op_at(right, m - L);
        qout[m] = op != 1 ? qs[qi] : '-'; tout[m] = op != 2 ? ts[ti] : '-';
        qi += op != 1; ti += op != 2;
    }
}

}  // namespace alnstr
