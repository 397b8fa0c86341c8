// cns_strings.hip — the two gapped, gap-normalised strings of every ACCEPTED mecat2cns alignment, built on the device
// (reference: the m5qaln / m5saln strings GetAlignment leaves, dw.cpp:495-531, and normalize_gaps(push = true),
// reads_correction_aux.cpp:3-81, as consensus_one_read_can_* calls it, mecat_correction.cpp:438 / :501).
//
// Until round 5 the host threads rebuilt these strings from the packed columns (aln_strings.h + push_gaps in cns_accept.hip): 24 GB of
// characters per config-2 batch, 2.0 of the batch's 3.7 s on the 16 cores the GPU box gives its host side.  Here:
//
//   cns_strings_build   one WAVE per accepted alignment.  Merged column m (reverse(left) then right, kept columns [first_col, last_col))
//                       is a 2-bit op (0: both sequences advance, 1: gap in the query string, 2: gap in the template string); the base
//                       under a column is the number of non-gap columns in front of it: per lane the count over its 16 columns, a wave
//                       scan, 16 bases of the packed read in one window, 16 characters out as one 16-byte store (lanes own 16-byte
//                       aligned groups of the destination; the two partial groups at a string's ends are written byte by byte, so
//                       neighbouring strings are never touched).
//   cns_push_gaps       one LANE per accepted alignment: normalize_gaps' second loop is a sequential program (a swap at column i moves a
//                       gap to a later column j, which a later iteration reads), so it is replayed as written, in place, column by column.
//                       Each lane keeps the 32-byte block of either string that holds column i in LDS (68 bytes per lane: the lanes of a
//                       wave hit 64 different banks); look-aheads and swaps that leave the block go to memory directly — a lane's own
//                       stores are visible to its later loads — and a block is written back only when it was changed.  A 32-bit mask of the block's
//                       gap bytes lets a lane step from gap to gap instead of from column to column.
//
// O(ND) columns are matches or indels, so normalize_gaps' first loop (mismatch -> two indel columns) changes nothing: aln_size is the
// number of kept columns (as on the host before).
#include <algorithm>
#include <vector>

#include "common.h"
#include "dw_helpers.h"
#include "cns_strings.h"

namespace {

__device__ __forceinline__ uint32_t op_at(const uint32_t* __restrict__ w, int k) { return (w[k >> 4] >> ((k & 15) << 1)) & 3u; }

// ops of merged columns m0 .. m0 + 15 (column m0 + j at bits 2j .. 2j + 1); columns outside [0, L + R) come out as `fill`
__device__ __forceinline__ uint32_t ops16(const uint32_t* __restrict__ left, int L, const uint32_t* __restrict__ right, int R, int m0, uint32_t fill) {
    if (m0 >= L && m0 + 16 <= L + R) {              // inside the right part: its columns in order
        const int k = m0 - L, w = k >> 4, s = (k & 15) << 1;
        const uint32_t lo = right[w];
        if (s == 0) return lo;
        return (uint32_t)((((unsigned long long)right[w + 1] << 32) | lo) >> s);
    }
    if (m0 >= 0 && m0 + 16 <= L) {                  // inside the left part: merged column m is extension column L - 1 - m
        const int k = L - 16 - m0, w = k >> 4, s = (k & 15) << 1;      // extension columns k .. k + 15, to be reversed
        const uint32_t lo = left[w];
        const uint32_t x = s == 0 ? lo : (uint32_t)((((unsigned long long)left[w + 1] << 32) | lo) >> s);
        // rev_groups reverses the groups of a word read MSB-first == LSB-first alike: group i <-> group 15 - i
        return rev_groups(x);
    }
    uint32_t r = 0;                                 // the seam, or the ends of the alignment
    for (int j = 0; j < 16; ++j) {
        const int m = m0 + j;
        uint32_t op = fill;
        if (m >= 0 && m < L) op = op_at(left, L - 1 - m);
        else if (m >= L && m < L + R) op = op_at(right, m - L);
        r |= op << (j << 1);
    }
    return r;
}

__device__ __forceinline__ uint32_t wave_excl_scan_u32(uint32_t v, uint32_t* total) {
    uint32_t incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const uint32_t u = __shfl_up(incl, o);
        if (lane_id() >= o) incl += u;
    }
    *total = __shfl(incl, 63);
    return incl - v;
}

// one string of one alignment: `gap` = the op that is a gap in THIS string (1: query, 2: template), `view` = its bases with logical
// position 0 = the first base under merged column 0
__device__ __forceinline__ void build_one(const uint32_t* __restrict__ left, int L, const uint32_t* __restrict__ right, int R, int first_col, int n,
                                          const uint32_t gap, const SeqView& view, char* __restrict__ dst) {
    const int lane = lane_id();
    // bases under the trimmed-away columns [0, first_col)
    uint32_t carry = 0;
    for (int m0 = 0; m0 < first_col; m0 += 64) {
        const int m = m0 + lane;
        const bool has = m < first_col && (m < L ? op_at(left, L - 1 - m) : op_at(right, m - L)) != gap;
        carry += (uint32_t)__popcll(__ballot(has));
    }
    const int A = (int)((uintptr_t)dst & 15u);
    char* const base = dst - A;                              // 16-byte aligned
    const int ngroups = (A + n + 15) >> 4;
    for (int g0 = 0; g0 < ngroups; g0 += 64) {
        const int g = g0 + lane;
        const int o0 = 16 * g - A;                           // output position of the group's first byte (negative in the first group)
        uint32_t ops = ops16(left, L, right, R, first_col + o0, gap);
        // columns in front of the string / behind it consume no base
        uint32_t valid = 0xffffu;
        if (o0 < 0) valid &= 0xffffu << (-o0);
        if (o0 + 16 > n) valid &= (n - o0 > 0) ? (0xffffu >> (16 - (n - o0))) : 0u;
        if (g >= ngroups) valid = 0;
        // has-base mask: op != gap, per column
        uint32_t hb = 0;
#pragma unroll
        for (int j = 0; j < 16; ++j) hb |= ((((ops >> (j << 1)) & 3u) != gap) ? 1u : 0u) << j;
        hb &= valid;
        uint32_t total;
        const uint32_t excl = wave_excl_scan_u32((uint32_t)__popc(hb), &total) + carry;
        carry += total;
        if (valid == 0) continue;
        uint32_t bw = view_word_le(view, (int)excl);         // 16 bases from this lane's first one, base j at bits 2j
        uint32_t out[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            uint32_t ch = '-';
            if ((hb >> j) & 1u) { ch = (0x54474341u >> ((bw & 3u) << 3)) & 0xffu; bw >>= 2; }      // "ACGT"
            out[j >> 2] |= ch << ((j & 3) << 3);
        }
        char* p = base + 16 * (size_t)g;
        if (valid == 0xffffu) {
            *(uint4*)p = make_uint4(out[0], out[1], out[2], out[3]);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j)
                if ((valid >> j) & 1u) p[j] = (char)((out[j >> 2] >> ((j & 3) << 3)) & 0xffu);
        }
    }
    if (lane == 0) dst[n] = 0;
}

__global__ __launch_bounds__(256) void cns_strings_build(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ offs,
                                                         const mhip_aln_job* __restrict__ jobs, const mhip_cns_result* __restrict__ res,
                                                         const uint32_t* __restrict__ ops, int row_words, const CnsStrItem* __restrict__ items, int n_items,
                                                         char* __restrict__ out) {
    for (size_t a = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6); a < (size_t)n_items; a += (size_t)gridDim.x * 4) {
        const CnsStrItem it = items[a];
        const mhip_cns_result r = res[it.job];
        const mhip_aln_job jb = jobs[it.job];
        const uint32_t* left = ops + (size_t)it.job * (size_t)row_words;
        const uint32_t* right = left + row_words / 2;
        const mhip_offset_t qo = offs[jb.qid_local], so = offs[jb.sid_local];
        SeqView q, t;
        q.pac = pac; q.off = qo.offset;
        if (jb.chain) { q.A = qo.size - 1 - r.query_start; q.B = -1; q.comp = 1; }      // strand position p = read position size - 1 - p, complemented
        else { q.A = r.query_start; q.B = 1; q.comp = 0; }
        t.pac = pac; t.off = so.offset; t.A = r.target_start; t.B = 1; t.comp = 0;
        char* qa = out + it.off;
        build_one(left, r.left_cols, right, r.right_cols, r.first_col, it.aln_size, 1u, q, qa);
        build_one(left, r.left_cols, right, r.right_cols, r.first_col, it.aln_size, 2u, t, qa + it.aln_size + 1);
    }
}

// ---- normalize_gaps, second loop, one lane per alignment ---------------------------------------------------------------------------
#define PG_WAVES 4
#define PG_STRIDE 68        // bytes of LDS per lane: two 32-byte blocks + one pad word (lane * 17 words: 64 different banks)

struct Win {
    uint8_t* lds;           // this lane's 32 bytes
    char* g;                // the string
    int n;                  // its length (the NUL at g[n] is never written)
    uintptr_t blk;          // address >> 5 of the block in the window
    uint32_t gaps;          // bit b: byte b of the block is '-'
    bool dirty;
};
// bit k of the result: byte k of w is '-' (exact zero-byte test on w ^ "----", then the four flag bits gathered by one multiply)
__device__ __forceinline__ uint32_t gap_nibble(uint32_t w) {
    const uint32_t x = w ^ 0x2D2D2D2Du;
    const uint32_t z = ~(((x & 0x7F7F7F7Fu) + 0x7F7F7F7Fu) | x | 0x7F7F7F7Fu);      // 0x80 in every byte that was '-'
    return (((z >> 7) * 0x00204081u) >> 21) & 0xFu;
}
__device__ __forceinline__ void win_load(Win& w, uintptr_t blk) {
    const uint4* p = (const uint4*)(blk << 5);
    const uint4 a = p[0], b = p[1];
    uint32_t* d = (uint32_t*)w.lds;
    d[0] = a.x; d[1] = a.y; d[2] = a.z; d[3] = a.w; d[4] = b.x; d[5] = b.y; d[6] = b.z; d[7] = b.w;
    w.gaps = gap_nibble(a.x) | (gap_nibble(a.y) << 4) | (gap_nibble(a.z) << 8) | (gap_nibble(a.w) << 12) | (gap_nibble(b.x) << 16) | (gap_nibble(b.y) << 20) |
             (gap_nibble(b.z) << 24) | (gap_nibble(b.w) << 28);
    w.blk = blk;
    w.dirty = false;
}
__device__ __forceinline__ void win_flush(Win& w) {
    if (!w.dirty) return;
    char* b = (char*)(w.blk << 5);
    if (b >= w.g && b + 32 <= w.g + w.n) {
        const uint32_t* s = (const uint32_t*)w.lds;
        ((uint4*)b)[0] = make_uint4(s[0], s[1], s[2], s[3]);
        ((uint4*)b)[1] = make_uint4(s[4], s[5], s[6], s[7]);
    } else {
        for (int k = 0; k < 32; ++k)
            if (b + k >= w.g && b + k < w.g + w.n) b[k] = (char)w.lds[k];
    }
    w.dirty = false;
}
__device__ __forceinline__ char win_get(const Win& w, int i) {
    const uintptr_t a = (uintptr_t)(w.g + i);
    if ((a >> 5) == w.blk) return (char)w.lds[a & 31u];
    return *(volatile const char*)a;
}
__device__ __forceinline__ void win_set(Win& w, int i, char v) {
    const uintptr_t a = (uintptr_t)(w.g + i);
    if ((a >> 5) == w.blk) {
        w.lds[a & 31u] = (uint8_t)v;
        const uint32_t bit = 1u << (a & 31u);
        w.gaps = v == '-' ? (w.gaps | bit) : (w.gaps & ~bit);
        w.dirty = true;
    } else *(volatile char*)a = v;
}

__global__ __launch_bounds__(64 * PG_WAVES) void cns_push_gaps(const CnsStrItem* __restrict__ items, int n_items, char* __restrict__ out) {
    __shared__ __attribute__((aligned(16))) uint8_t lds[PG_WAVES][64 * PG_STRIDE];
    const int lane = lane_id(), wv = (int)threadIdx.x >> 6;
    const size_t a = ((size_t)blockIdx.x * PG_WAVES + wv) * 64 + lane;
    int n = 0;
    Win Q, T;
    Q.lds = &lds[wv][lane * PG_STRIDE];
    T.lds = Q.lds + 32;
    Q.g = T.g = out; Q.n = T.n = 0; Q.blk = T.blk = 0; Q.gaps = T.gaps = 0; Q.dirty = T.dirty = false;
    if (a < (size_t)n_items) {
        const CnsStrItem it = items[a];
        n = it.aln_size;
        Q.g = out + it.off; T.g = Q.g + n + 1;
        Q.n = T.n = n;
    }
    if (n >= 2) { win_load(Q, (uintptr_t)Q.g >> 5); win_load(T, (uintptr_t)T.g >> 5); }
    for (int i = 0; i + 1 < n;) {
        const uintptr_t qa = (uintptr_t)(Q.g + i), ta = (uintptr_t)(T.g + i);
        if ((qa >> 5) != Q.blk) { win_flush(Q); win_load(Q, qa >> 5); }
        if ((ta >> 5) != T.blk) { win_flush(T); win_load(T, ta >> 5); }
        // columns without a gap in either string do nothing: skip to the next gap the two blocks show (or to the end of a block)
        const int pq = (int)(qa & 31u), pt = (int)(ta & 31u);
        const uint32_t gaps = (Q.gaps >> pq) | (T.gaps >> pt);
        const int room = min(min(32 - pq, 32 - pt), n - 1 - i);
        const int run = min(gaps ? __builtin_ctz(gaps) : 32, room);
        if (run > 0) { i += run; continue; }
        // push target gaps (reads_correction_aux.cpp:41-52)
        if ((char)T.lds[pt] == '-') {
            int j = i;
            char c;
            do { c = win_get(T, ++j); } while (c == '-');          // (ends at the NUL behind the string at the latest: j <= n)
            if (c == (char)Q.lds[pq]) { win_set(T, i, c); win_set(T, j, '-'); }
        }
        // push query gaps (:54-65)
        if ((char)Q.lds[pq] == '-') {
            int j = i;
            char c;
            do { c = win_get(Q, ++j); } while (c == '-');
            if (c == (char)T.lds[pt]) { win_set(Q, i, c); win_set(Q, j, '-'); }
        }
        ++i;
    }
    win_flush(Q);
    win_flush(T);
}

}  // namespace

// strings of n_items accepted alignments into d_out (device; 32 readable bytes in front of it and 64 behind the last string, see
// cns_strings.h); everything on the context's stream, nothing waits
int cns_strings_launch(mhip_ctx* c, const mhip_volume* vol, const mhip_aln_job* d_jobs, const mhip_cns_result* d_res, const uint32_t* d_ops, int row_words,
                       const CnsStrItem* d_items, int n_items, char* d_out) {
    if (n_items <= 0) return 0;
    if (((uintptr_t)d_out & 31u) != 0) { mhip_set_error("cns strings: the output buffer is not 32-byte aligned"); return -1; }
    LAUNCH(c, "cns_strings_build", cns_strings_build, (unsigned)std::min<size_t>(((size_t)n_items + 3) / 4, (size_t)c->num_cus * 64), 256, 0,
           (const uint32_t*)vol->d_pac, (const mhip_offset_t*)vol->d_offs, d_jobs, d_res, d_ops, row_words, d_items, n_items, d_out);
    LAUNCH(c, "cns_push_gaps", cns_push_gaps, (unsigned)(((size_t)n_items + 64 * PG_WAVES - 1) / (64 * PG_WAVES)), 64 * PG_WAVES, 0, d_items, n_items, d_out);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" {

// TEST HOOK (tests/test_gpu_cns_strings.py): normalize_gaps' gap pushing on `n_pairs` pairs of host strings, in place.  Pair p: q at
// buf + off[p], its partner at q + len[p] + 1, both NUL-terminated (the layout of mhip_cns_accept_templates' string buffer).
int mhip_debug_push_gaps(mhip_ctx* c, char* buf, int64_t bytes, const int64_t* off, const int32_t* len, int n_pairs) {
    HIPCHK(hipSetDevice(c->device));
    if (n_pairs <= 0) return 0;
    std::vector<CnsStrItem> items((size_t)n_pairs);
    for (int p = 0; p < n_pairs; ++p) {
        if (off[p] < 0 || len[p] < 0 || off[p] + 2 * ((int64_t)len[p] + 1) > bytes) { mhip_set_error("push_gaps: pair %d lies outside the buffer", p); return -1; }
        items[(size_t)p].job = p; items[(size_t)p].aln_size = len[p]; items[(size_t)p].off = (unsigned long long)off[p];
    }
    char* d_buf;
    CnsStrItem* d_items;
    if (c->scratch("pg_buf", (size_t)bytes + 128, (void**)&d_buf)) return -1;
    if (c->scratch("pg_items", sizeof(CnsStrItem) * (size_t)n_pairs, (void**)&d_items)) return -1;
    HIPCHK(hipMemsetAsync(d_buf, 0, (size_t)bytes + 128, c->stream));
    HIPCHK(hipMemcpyAsync(d_buf + 32, buf, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemcpyAsync(d_items, items.data(), sizeof(CnsStrItem) * (size_t)n_pairs, hipMemcpyHostToDevice, c->stream));
    LAUNCH(c, "cns_push_gaps", cns_push_gaps, (unsigned)(((size_t)n_pairs + 64 * PG_WAVES - 1) / (64 * PG_WAVES)), 64 * PG_WAVES, 0, (const CnsStrItem*)d_items, n_pairs,
           d_buf + 32);
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(buf, d_buf + 32, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
