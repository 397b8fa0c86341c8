// index_part.hip — the default build of the k = 13 look-up table (index.hip: mhip_index_build): a STABLE two-level partition.
//
// Replaces create_ref_index / fill_ref_index_offsets_func (common/lookup_table.cpp:63-160, 25-61).  The result is the one table of
// index.hip (starts[4^13 + 1], offsets[] ascending inside a bucket, buckets above the cap empty) plus the two side arrays the
// seeding stage reads (slots[], bucket records).
//
// Why stable.  The reference fills a bucket in ascending volume position, and that order is load-bearing (seeding replays it).
// Round 2 partitioned (k-mer, position) pairs with LDS atomics in three 64-way passes of 8-byte entries and sorted every bucket at
// the end: 76 GB of HBM traffic for 7.9 GB of algorithmic bytes, and 15 of its 36 ms in the per-bucket bitonic networks.  Here
// every pass keeps the order of its input, so a bucket comes out ascending with no sort at all, and the entries are 4 bytes:
//
//   level 1   the volume is walked in tiles of 16 384 positions; a k-mer goes to one of 512 bins by its top 9 bits.  A first walk
//             (ix_hist1) counts (tile, bin) occupancies, three small scans turn them into the exact output position of every
//             (tile, bin) run, the second walk (ix_scatter1) writes  r17:17 | position-inside-the-tile:14  (r17 = the other 17 k-mer
//             bits).  The tile number is not stored: entries of a bin are in tile order and the table says where each tile's run starts.
//   level 2   a unit = the runs of 256 consecutive tiles inside one bin (about 16 k entries).  ix_hist2 counts (unit, sub-bin)
//             occupancies for the next 9 k-mer bits, ix_scan2 places them, ix_scatter2 writes the full position (4 bytes) and the
//             last 8 k-mer bits (1 byte) into two arrays, sub-bin major, unit order inside a sub-bin = ascending position.
//   fill      one workgroup per sub-bin (256 k-mer ids, about 6 k entries): per-id counts, buckets above the cap dropped, kept
//             positions placed by id in input order, one coalesced sweep out (positions, 2 kb-segment slots, starts[], records).
//
// Rank inside a tile / chunk / sub-bin: lanes hold CONSECUTIVE entries, so "stable" means "lanes with the same bin keep their lane
// order".  wave_rank() gets that from the wave's private LDS counters: read, no-return add, read again — a lane whose counter moved
// by one is alone in its bin this step (the common case: 64 entries over 512 bins); the few lanes that share a bin are ordered by a
// ballot per shared bin.  Across waves the order is fixed by giving each wave a contiguous slice and adding the waves' totals.
//
// Key-range shards (multi-GPU, comm.hip: mhip_index_build_sharded): a rank that owns bins [bin_lo, bin_hi) runs both volume walks
// in full (they are the cheap part) but writes, partitions and fills only its own bins.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <chrono>
#include <vector>

#include "common.h"
#include "index_part.h"

namespace {

#ifdef MECAT_IX_STATS
// development build only (make ixstats, tools/dev/idx_time.py): s_memrealtime ticks (100 MHz) per section of the three big kernels, summed
// over the workgroups by thread 0
__device__ unsigned long long g_ixstats[32];

#define IXS_T(var) const unsigned long long var = wall_clock64()
#define IXS_ADD(slot, a, b) do { if (threadIdx.x == 0) atomicAdd(&g_ixstats[slot], (b) - (a)); } while (0)
#else
#define IXS_T(var)
#define IXS_ADD(slot, a, b)
#endif

#ifdef MECAT_IX_KNOCK
__device__ int g_ixknock;      // timing experiments (make ixknock, env MECAT_IX_KNOCK): 1 = no copy-out stores, 2 = copy-out to dense addresses (results are wrong)
#endif
constexpr int NB1 = 1 << IXP_L1_BITS;        // 512 level-1 bins
constexpr int NB2 = 1 << IXP_L2_BITS;        // 512 sub-bins per level-1 bin
constexpr int NID = 1 << IXP_ID_BITS;        // 256 k-mer ids per sub-bin
constexpr int T1_POS = 16384;                // positions per level-1 tile (the 14 position bits of a level-1 entry)
constexpr int SUB1 = 4096;                   // positions staged per round of a tile
constexpr int T1_THREADS = 512;
constexpr int G1 = 256;                      // level-1 tiles per level-2 unit
constexpr int T2_ENT = 4096;                 // entries staged per round of a unit
constexpr int T2_THREADS = 512;
constexpr int FILL_THREADS = 256;
constexpr int FILL_CAP = 8192;               // kept positions of a sub-bin that fit its LDS buffer (~4.5 k on average at config 2)
static_assert(IXP_L1_BITS + IXP_L2_BITS + IXP_ID_BITS == 26, "13-mers");
static_assert(T1_THREADS == NB1 && T2_THREADS == NB2, "one thread per bin in the per-bin passes");

// exclusive prefix of v over the NT threads of the workgroup (wtot: NT / 64 words of LDS); total in *total when given
// inclusive prefix over the 64 lanes on the DPP network (no LDS traffic): shifts inside the 16-lane rows, then the row totals
__device__ __forceinline__ uint32_t wave_incl_scan(uint32_t x) {
    int v = (int)x;
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);       // row_shr:1
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);       // row_shr:2
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);       // row_shr:4
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);       // row_shr:8
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);      // row_bcast:15 -> rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);      // row_bcast:31 -> rows 2 and 3
    return (uint32_t)v;
}
template <int NT>
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wtot, uint32_t* total = nullptr) {
    const uint32_t incl = wave_incl_scan(v);
    __syncthreads();                       // wtot may still be read from an earlier call
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, all = 0;
#pragma unroll
    for (int w = 0; w < NT / 64; ++w) {
        const uint32_t t = wtot[w];
        if (w < (int)(threadIdx.x >> 6)) base += t;
        all += t;
    }
    if (total) *total = all;
    return base + incl - v;
}

// Rank of this lane's entry among the entries of bin b its wave has seen so far, the lanes of one step in lane order.
// Called by exactly the lanes that hold an entry; cnt = the wave's private counters.
//   STRICT = false (default build): the value the LDS atomic returns.  Lanes of one instruction that add to the same counter are
//            served in ascending lane order on this hardware — observed, not documented — so nothing rests on it: ix_fill checks
//            that every bucket it writes is strictly ascending (a misordered pair that matters always ends up as a descent inside
//            one bucket; pairs that end up in different buckets do not matter) and raises a flag, and the build is redone STRICT.
//   STRICT = true: read, no-return add, read again (LDS operations of a wave execute in issue order: `before` is the count in front
//            of this step for every lane, `after` the count behind it) — a lane whose counter moved by one is alone in its bin this
//            step; lanes that share a bin are ordered by a ballot per shared bin.
typedef __attribute__((address_space(3))) uint32_t lds_u32_t;      // (an LDS pointer type: the volatile reads must stay ds_read, not flat loads)
template <bool STRICT>
__device__ __forceinline__ uint32_t wave_rank(uint32_t* cnt_generic, const uint32_t b) {
    lds_u32_t* cnt = (lds_u32_t*)cnt_generic;
    if (!STRICT) return __hip_atomic_fetch_add(cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    volatile lds_u32_t* vc = cnt;
    const uint32_t before = vc[b];
    __hip_atomic_fetch_add(cnt + b, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t after = vc[b];
    uint32_t r = before;
    unsigned long long todo = __ballot(after - before > 1u);       // lanes that share their bin with another lane of this step
    const unsigned long long below = (1ull << lane_id()) - 1ull;
    while (todo) {
        const int l = __ffsll(todo) - 1;
        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)b, l);
        const unsigned long long m = __ballot(b == bb);
        if (b == bb) r = before + (uint32_t)__popcll(m & below);
        todo &= ~m;
    }
    return r;
}

// The same with 16-bit counters, two bins per LDS word (a wave adds at most 2^16 - 1 entries to a bin between two resets): half the LDS
// of the per-wave counter tables of the two scatter kernels, which is what decides how many of their workgroups a CU holds.
// cnt16 = the wave's table as words; bin b lives in half (b & 1) of word b >> 1.
template <bool STRICT>
__device__ __forceinline__ uint32_t wave_rank16(uint32_t* cnt_generic, const uint32_t b) {
    lds_u32_t* cnt = (lds_u32_t*)cnt_generic;
    const uint32_t sh = (b & 1u) << 4, one = 1u << sh;
    if (!STRICT) return (__hip_atomic_fetch_add(cnt + (b >> 1), one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) >> sh) & 0xffffu;
    volatile lds_u32_t* vc = cnt;
    const uint32_t before = (vc[b >> 1] >> sh) & 0xffffu;
    __hip_atomic_fetch_add(cnt + (b >> 1), one, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const uint32_t after = (vc[b >> 1] >> sh) & 0xffffu;
    uint32_t r = before;
    unsigned long long todo = __ballot(after - before > 1u);       // lanes that share their bin with another lane of this step
    const unsigned long long below = (1ull << lane_id()) - 1ull;
    while (todo) {
        const int l = __ffsll(todo) - 1;
        const uint32_t bb = (uint32_t)__builtin_amdgcn_readlane((int)b, l);
        const unsigned long long m = __ballot(b == bb);
        if (b == bb) r = before + (uint32_t)__popcll(m & below);
        todo &= ~m;
    }
    return r;
}

// ---- a range of NPOS volume positions starting at q0 (a multiple of 1024): its packed words in LDS, and the bitmap of the positions
// that start no k-mer — the last 12 bases of every read and its pad base (lookup_table.cpp:77-90: the rolling word restarts at every
// read, the first k-mer is emitted at j = 12), i.e. [rend - 12, rend] for every read end `rend` in reach.
template <int NPOS, int NT>
__device__ __forceinline__ void range_prepare(uint32_t* words /*[NPOS / 16 + 2]*/, uint32_t* inval /*[NPOS / 32]*/, const uint32_t* __restrict__ pac,
                                              const mhip_offset_t* __restrict__ offs, int num_reads, int num_bases, const uint32_t* __restrict__ blk,
                                              const int64_t q0) {
    const int tid = threadIdx.x;
    const int64_t w0 = q0 >> 4, wlast = ((int64_t)num_bases + 15) >> 4;       // the volume carries >= 128 zero bytes behind its last base
    __syncthreads();                                                           // (the arrays may still be read from the round before)
    for (int i = tid; i < NPOS / 16 + 1; i += NT) words[i] = (w0 + i <= wlast) ? pac_word(pac, w0 + i) : 0u;
    for (int i = tid; i < NPOS / 32; i += NT) inval[i] = 0u;
    __syncthreads();
    int r = (int)blk[q0 >> 10];                     // the read that holds base q0, or one before it
    while (r + 1 < num_reads && (int64_t)offs[r + 1].offset <= q0) ++r;
    for (int rr = r + tid; rr < num_reads; rr += NT) {
        const int64_t o = offs[rr].offset;
        if (o >= q0 + NPOS + 13) break;             // reads are stored in ascending order
        const int64_t rend = o + offs[rr].size;
        const int lo = (int)(std::max<int64_t>(rend - 12, q0) - q0), hi = (int)(std::min<int64_t>(rend, q0 + NPOS - 1) - q0);
        for (int p = lo; p <= hi; ++p) atomicOr(&inval[p >> 5], 1u << (p & 31));
    }
    __syncthreads();
}
// the 13-mer that starts at local position i (first base in the top bits, as the reference's rolling word holds it)
__device__ __forceinline__ uint32_t kmer_at(const uint32_t* words, int i) {
    const uint32_t hi = words[i >> 4], lo = words[(i >> 4) + 1];
    const int s = (i & 15) << 1;
    const uint32_t w = s ? __builtin_amdgcn_alignbit(hi, lo, 32 - s) : hi;      // ({hi, lo} << s) >> 32
    return w >> 6;
}

// ---- level 1, first walk: occupancy of every (tile, bin)
__global__ __launch_bounds__(T1_THREADS) void ix_hist1(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ offs, int num_reads,
                                                       int num_bases, const uint32_t* __restrict__ blk, uint32_t* __restrict__ hist1,
                                                       uint32_t* __restrict__ nokmer) {
    __shared__ uint32_t words[T1_POS / 16 + 2];
    __shared__ uint32_t inval[T1_POS / 32];
    __shared__ uint32_t h[NB1];
    const int tile = blockIdx.x, tid = threadIdx.x;
    const int64_t q0 = (int64_t)tile * T1_POS;
    h[tid] = 0;
    range_prepare<T1_POS, T1_THREADS>(words, inval, pac, offs, num_reads, num_bases, blk, q0);
    {   // the tile's 512 words of the volume-wide bitmap "no k-mer starts here" (read ends, and everything behind the last base), which
        // the second walk reads instead of going through the read table again (a chain of dependent loads per round: half of its time)
        static_assert(T1_POS / 32 == T1_THREADS, "one bitmap word per thread");
        const int64_t p0 = q0 + 32 * (int64_t)tid, left = (int64_t)num_bases - p0;      // positions p0 .. p0 + 31 of this word
        uint32_t m = inval[tid];
        if (left < 32) m |= left <= 0 ? 0xffffffffu : ~((1u << (int)left) - 1u);
        inval[tid] = m;
        nokmer[(size_t)tile * (T1_POS / 32) + tid] = m;
    }
    __syncthreads();
    for (int j = 0; j < T1_POS / T1_THREADS; ++j) {
        const int i = j * T1_THREADS + tid;
        if (!((inval[i >> 5] >> (i & 31)) & 1u)) atomicAdd(&h[kmer_at(words, i) >> (26 - IXP_L1_BITS)], 1u);
    }
    __syncthreads();
    hist1[(size_t)tile * NB1 + tid] = h[tid];
}

// occupancies -> positions.  (a) inside a group of G1 tiles, per bin, in place; (b) the groups of a bin, then the bins; the group
// totals become the absolute position of each group's first entry of a bin
__global__ __launch_bounds__(NB1) void ix_scan1a(uint32_t* __restrict__ hist1, int ntile, uint32_t* __restrict__ grp) {
    const int g = blockIdx.x, b = threadIdx.x;
    const int t1 = min(ntile, (g + 1) * G1);
    uint32_t run = 0;
    for (int t = g * G1; t < t1; ++t) {
        const uint32_t v = hist1[(size_t)t * NB1 + b];
        hist1[(size_t)t * NB1 + b] = run;
        run += v;
    }
    grp[(size_t)g * NB1 + b] = run;
}
__global__ __launch_bounds__(NB1) void ix_scan1b(uint32_t* __restrict__ grp, int ngroup, uint32_t* __restrict__ binbase /*[NB1 + 1]*/) {
    __shared__ uint32_t wtot[NB1 / 64];
    const int b = threadIdx.x;
    uint32_t run = 0;
    for (int g = 0; g < ngroup; ++g) {
        const uint32_t v = grp[(size_t)g * NB1 + b];
        grp[(size_t)g * NB1 + b] = run;
        run += v;
    }
    uint32_t all;
    const uint32_t base = block_excl_scan<NB1>(run, wtot, &all);
    binbase[b] = base;
    if (b == NB1 - 1) binbase[NB1] = all;
    for (int g = 0; g < ngroup; ++g) grp[(size_t)g * NB1 + b] += base;
}
// the same positions bin-major: base1T[b][t] = first entry of tile t in bin b, base1T[b][ntile] = the end of the bin (level 2 reads a
// bin's tile boundaries as one contiguous run)
__global__ __launch_bounds__(256) void ix_transpose1(const uint32_t* __restrict__ hist1, const uint32_t* __restrict__ grp, const uint32_t* __restrict__ binbase,
                                                     int ntile, uint32_t* __restrict__ base1T) {
    __shared__ uint32_t tl[32][33];
    const int t0 = blockIdx.x * 32, b0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int t = t0 + r, b = b0 + tx;
        tl[r][tx] = t < ntile ? hist1[(size_t)t * NB1 + b] + grp[(size_t)(t / G1) * NB1 + b] : 0u;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int b = b0 + r, t = t0 + tx;
        if (t < ntile) base1T[(size_t)b * (ntile + 1) + t] = tl[tx][r];
    }
    if (blockIdx.x == 0 && threadIdx.x < 32) base1T[(size_t)(b0 + threadIdx.x) * (ntile + 1) + ntile] = binbase[b0 + threadIdx.x + 1];
}

// ---- level 1, second walk: the entries, every (tile, bin) run at its position, input order kept inside a run
#ifndef IXP_S1_WAVES
#define IXP_S1_WAVES 6
#endif
#ifndef IXP_S2_WAVES
#define IXP_S2_WAVES 6
#endif
template <bool STRICT>
__global__ __launch_bounds__(T1_THREADS, IXP_S1_WAVES) void ix_scatter1(const uint32_t* __restrict__ pac, int num_bases, const uint32_t* __restrict__ nokmer,
                                                          const uint32_t* __restrict__ hist1,
                                                          const uint32_t* __restrict__ grp, uint32_t* __restrict__ ent1, int bin_lo, int bin_hi,
                                                          uint32_t ent_off, int xcd_ranges) {
#ifndef IXP_S1_TILE_STAGE
#define IXP_S1_TILE_STAGE 0
#endif
    constexpr int SPOS = IXP_S1_TILE_STAGE ? T1_POS : SUB1;      // positions whose packed words and bitmap (ix_hist1 made it) are staged at a time
    __shared__ uint32_t words[SPOS / 16 + 2];
    __shared__ uint32_t inval[SPOS / 32];
    __shared__ __attribute__((aligned(16))) uint16_t cntw[T1_THREADS / 64][NB1];      // per-wave counters (wave_rank16), then each wave's first stage place of a bin
    __shared__ uint32_t gdel[NB1];
    __shared__ uint32_t ltotal;
    __shared__ uint32_t stage[SUB1];
    __shared__ uint16_t sbin[SUB1];
    __shared__ uint32_t wtot[T1_THREADS / 64];
    constexpr int NW = T1_THREADS / 64, STEPS = SUB1 / T1_THREADS;
    // Which block takes which tile.  Blocks go to the XCDs round robin (block x on XCD x mod 8: observed), so "tile = block" spreads
    // neighbouring tiles — whose runs of a bin are neighbours in memory — over all eight L2s, and every 32-byte run leaves its L2 as
    // partial sectors of its own.  With xcd_ranges an XCD works through a contiguous eighth of the tiles, the tiles its CUs hold at any
    // time are neighbours, and their runs meet in the one L2: 6.85 -> 5.26 ms at config 2 on the boxes where "tile = block" takes 6.85
    // (on others it takes 5.0 - 5.3 as it is; the build times both orders once per context and keeps the faster, see the launch).
    const int per8 = (int)gridDim.x >> 3;                  // (the grid is a multiple of 8 blocks)
    const int tile = xcd_ranges ? (int)(blockIdx.x & 7u) * per8 + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    if ((int64_t)tile * T1_POS >= num_bases) return;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    uint32_t gcur = hist1[(size_t)tile * NB1 + tid] + grp[(size_t)(tile / G1) * NB1 + tid] - ent_off;      // thread b: next entry of (tile, bin b)
    for (int ww = 0; ww < NW; ++ww) cntw[ww][tid] = 0;
    const int64_t wlast = ((int64_t)num_bases + 15) >> 4;       // (the volume carries >= 128 zero bytes behind its last base)
    for (int sub = 0; sub < T1_POS / SUB1; ++sub) {
        const int64_t q0 = (int64_t)tile * T1_POS + (int64_t)sub * SUB1;
        if (q0 >= num_bases) break;
        if (IXP_S1_TILE_STAGE ? sub == 0 : true) {
            IXS_T(s_a);
            const int64_t w0 = q0 >> 4;
            for (int i = tid; i < SPOS / 16 + 1; i += T1_THREADS) words[i] = (w0 + i <= wlast) ? pac_word(pac, w0 + i) : 0u;
            if (tid < SPOS / 32) inval[tid] = nokmer[(size_t)(q0 >> 5) + tid];
            __syncthreads();
            IXS_T(s_b);
            IXS_ADD(0, s_a, s_b);
        }
        IXS_T(s_b);
        uint32_t ent[STEPS], bn[STEPS], rk[STEPS];
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const int il = w * (SUB1 / NW) + j * 64 + lane;      // wave w owns a contiguous eighth of the round, lanes = consecutive positions
            const int i = IXP_S1_TILE_STAGE ? sub * SUB1 + il : il;      // position inside the staged range
            const uint32_t km = kmer_at(words, i);
            const uint32_t b = km >> (26 - IXP_L1_BITS);
            const bool ok = !((inval[i >> 5] >> (i & 31)) & 1u);
            ent[j] = ((km & ((1u << (26 - IXP_L1_BITS)) - 1u)) << 14) | (uint32_t)(sub * SUB1 + il);
            bn[j] = ok ? b : 0xffffu;
            rk[j] = 0;
            if (ok) rk[j] = wave_rank16<STRICT>((uint32_t*)cntw[w], b);
        }
        __syncthreads();
        IXS_T(s_c);
        {   // thread b: the waves' counts of bin b -> each wave's first place of bin b in the stage; where the bin's run goes
            uint32_t c[NW], tot = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { c[ww] = cntw[ww][tid]; tot += c[ww]; }
            uint32_t all;
            uint32_t run = block_excl_scan<T1_THREADS>(tot, wtot, &all);
            gdel[tid] = gcur - run;                   // output index of stage place i of this bin = gdel + i
            gcur += tot;
            if (tid == 0) ltotal = all;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { cntw[ww][tid] = (uint16_t)run; run += c[ww]; }
        }
        __syncthreads();
        IXS_T(s_d);
#pragma unroll
        for (int j = 0; j < STEPS; ++j)
            if (bn[j] != 0xffffu) {
                const uint32_t at = cntw[w][bn[j]] + rk[j];
                stage[at] = ent[j];
                sbin[at] = (uint16_t)bn[j];
            }
        __syncthreads();
        IXS_T(s_e);
        const uint32_t total = ltotal;
        for (uint32_t i = tid; i < total; i += T1_THREADS) {
            const uint32_t b = sbin[i];
#ifdef MECAT_IX_KNOCK
            if (g_ixknock == 1) { if (stage[i] == 0xfffffff1u) ent1[i] = 1; continue; }
            if (g_ixknock == 2) { ent1[(size_t)tile * T1_POS + sub * SUB1 + i] = stage[i] + gdel[b]; continue; }
#endif
            if ((int)b >= bin_lo && (int)b < bin_hi) ent1[(size_t)(uint32_t)(gdel[b] + i)] = stage[i];      // (32-bit wrap-around arithmetic: gdel may be "negative")
        }
        __syncthreads();
        IXS_T(s_f);
        IXS_ADD(1, s_b, s_c); IXS_ADD(2, s_c, s_d); IXS_ADD(3, s_d, s_e); IXS_ADD(4, s_e, s_f);
        for (int ww = 0; ww < NW; ++ww) cntw[ww][tid] = 0;
        __syncthreads();      // (the next round's ranks count from zero)
    }
}

// ---- level 2.  Unit (b1, g) = the entries of bin b1 that come from tiles [g G1, (g + 1) G1).  Blocks are dealt to the XCDs (block x
// runs on XCD x mod 8, observed) so that the units of one bin run on one XCD one after the other: their runs land next to each other
// in the sub-bin streams, and the XCD's L2 merges the short runs into whole lines before they leave for HBM.
__device__ __forceinline__ bool unit_of_block(int x, int nb, int ngroup, int& b1i, int& g) {
    const int xcd = x & 7, slot = x >> 3;
    b1i = (slot / ngroup) * 8 + xcd;
    g = slot % ngroup;
    return b1i < nb;
}
__global__ __launch_bounds__(T2_THREADS) void ix_hist2(const uint32_t* __restrict__ ent1, const uint32_t* __restrict__ base1T, int ntile, int ngroup, int nb,
                                                       int bin_lo, uint32_t ent_off, uint32_t* __restrict__ hist2) {
    __shared__ uint32_t h[NB2];
    int b1i, g;
    if (!unit_of_block(blockIdx.x, nb, ngroup, b1i, g)) return;
    const int tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const uint32_t* bt = base1T + (size_t)(bin_lo + b1i) * (ntile + 1);
    const uint32_t e0 = bt[g * G1] - ent_off, e1 = bt[min(ntile, (g + 1) * G1)] - ent_off;
    // four entries per load: the unit's whole 16-byte words, then the up to three entries either side of them
    const uint32_t a0 = min(e1, (e0 + 3u) & ~3u), a1 = max(a0, e1 & ~3u);
    const uint4* e4 = (const uint4*)ent1;
    for (uint32_t q = (a0 >> 2) + tid; q < (a1 >> 2); q += T2_THREADS) {
        const uint4 v = e4[q];
        atomicAdd(&h[v.x >> (14 + IXP_ID_BITS)], 1u);
        atomicAdd(&h[v.y >> (14 + IXP_ID_BITS)], 1u);
        atomicAdd(&h[v.z >> (14 + IXP_ID_BITS)], 1u);
        atomicAdd(&h[v.w >> (14 + IXP_ID_BITS)], 1u);
    }
    if (tid < 3u && e0 + tid < a0) atomicAdd(&h[ent1[e0 + tid] >> (14 + IXP_ID_BITS)], 1u);
    if (tid >= 32u && tid < 35u && a1 + (tid - 32u) < e1) atomicAdd(&h[ent1[a1 + (tid - 32u)] >> (14 + IXP_ID_BITS)], 1u);
    __syncthreads();
    hist2[((size_t)b1i * ngroup + g) * NB2 + tid] = h[tid];
}
// one workgroup per bin: the units of a sub-bin in order (in place), then the sub-bins of the bin -> sub_ent[] (entry ranges of the sub-bins)
__global__ __launch_bounds__(NB2) void ix_scan2(uint32_t* __restrict__ hist2, int ngroup, int nb, const uint32_t* __restrict__ binbase, int bin_lo,
                                                uint32_t ent_off, uint32_t* __restrict__ sub_ent) {
    __shared__ uint32_t wtot[NB2 / 64];
    const int b1i = blockIdx.x, b2 = threadIdx.x;
    uint32_t run = 0;
    for (int g = 0; g < ngroup; ++g) {
        const size_t at = ((size_t)b1i * ngroup + g) * NB2 + b2;
        const uint32_t v = hist2[at];
        hist2[at] = run;
        run += v;
    }
    const uint32_t ex = block_excl_scan<NB2>(run, wtot);
    sub_ent[(size_t)b1i * NB2 + b2] = binbase[bin_lo + b1i] - ent_off + ex;
    if (b1i == nb - 1 && b2 == NB2 - 1) sub_ent[(size_t)nb * NB2] = binbase[bin_lo + nb] - ent_off;
}
template <bool STRICT>
__global__ __launch_bounds__(T2_THREADS, IXP_S2_WAVES) void ix_scatter2(const uint32_t* __restrict__ ent1, const uint32_t* __restrict__ base1T, int ntile, int ngroup, int nb,
                                                          int bin_lo, uint32_t ent_off, const uint32_t* __restrict__ hist2,
                                                          const uint32_t* __restrict__ sub_ent, uint32_t* __restrict__ pos2, uint8_t* __restrict__ id2) {
    __shared__ uint32_t tstart[G1 + 1];
    __shared__ __attribute__((aligned(16))) uint16_t cntw[T2_THREADS / 64][NB2];      // (wave_rank16)
    __shared__ uint32_t gdel[NB2];
    __shared__ uint32_t spos[T2_ENT];
    __shared__ uint16_t sbin[T2_ENT];
    __shared__ uint8_t sid[T2_ENT];
    __shared__ uint32_t wtot[T2_THREADS / 64];
    constexpr int NW = T2_THREADS / 64, STEPS = T2_ENT / T2_THREADS;
    int b1i, g;
    if (!unit_of_block(blockIdx.x, nb, ngroup, b1i, g)) return;
    const int tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    IXS_T(u_0);
    const uint32_t* bt = base1T + (size_t)(bin_lo + b1i) * (ntile + 1);
    const int t0 = g * G1, nt = min(G1, ntile - t0);
    for (int i = tid; i <= nt; i += T2_THREADS) tstart[i] = bt[t0 + i] - ent_off;
    uint32_t gcur = sub_ent[(size_t)b1i * NB2 + tid] + hist2[((size_t)b1i * ngroup + g) * NB2 + tid];      // thread b2: next place of (unit, sub-bin b2)
    for (int ww = 0; ww < NW; ++ww) cntw[ww][tid] = 0;
    __syncthreads();
    const uint32_t e0 = tstart[0], e1 = tstart[nt];
    IXS_T(u_1);
    IXS_ADD(8, u_0, u_1);
    // the entries of a round are loaded one round ahead: behind the copy-out of the round before, their loads would queue up behind its
    // scattered stores (ent1 carries 64 entries of slack behind its end; what is loaded past e1 is not used)
    uint32_t vn[STEPS];
#pragma unroll
    for (int j = 0; j < STEPS; ++j) {
        const uint32_t e = e0 + (uint32_t)(w * (T2_ENT / NW) + j * 64 + lane);
        vn[j] = e < e1 ? ent1[e] : 0u;
    }
    for (uint32_t c0 = e0; c0 < e1; c0 += T2_ENT) {
        IXS_T(u_a);
        const uint32_t cn = min((uint32_t)T2_ENT, e1 - c0);
        uint32_t pos[STEPS], key[STEPS], rk[STEPS], vc[STEPS];
        int t = 0;
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            vc[j] = vn[j];
            const uint32_t e = c0 + (uint32_t)T2_ENT + (uint32_t)(w * (T2_ENT / NW) + j * 64 + lane);
            vn[j] = e < e1 ? ent1[e] : 0u;
        }
#pragma unroll
        for (int j = 0; j < STEPS; ++j) {
            const uint32_t li = (uint32_t)(w * (T2_ENT / NW) + j * 64 + lane);      // wave w: a contiguous eighth of the round, lanes = consecutive entries
            key[j] = 0xffffffffu;
            pos[j] = 0;
            rk[j] = 0;
            if (li < cn) {
                const uint32_t e = c0 + li;
                const uint32_t v = vc[j];
                // the tile of entry e = the last tile whose run starts at or before e: a binary search for the wave's first step of a
                // round, a few steps forward from the lane's previous tile afterwards (64 entries further on ~ one tile further on)
                if (j == 0) {
                    int lo = 0, hi = nt;          // first index in [0, nt] with tstart[] > e, minus one
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (tstart[mid] <= e) lo = mid + 1; else hi = mid;
                    }
                    t = lo - 1;
                    if (lo == nt && tstart[nt] <= e) t = nt - 1;      // (not reached: e < e1 = tstart[nt])
                } else {
                    while (t + 1 < nt && tstart[t + 1] <= e) ++t;
                }
                pos[j] = (uint32_t)(t0 + t) * (uint32_t)T1_POS + (v & 0x3fffu);
                key[j] = v >> 14;                                       // sub-bin : id
                rk[j] = wave_rank16<STRICT>((uint32_t*)cntw[w], v >> (14 + IXP_ID_BITS));
            }
        }
        __syncthreads();
        IXS_T(u_b);
        {
            uint32_t c[NW], tot = 0;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { c[ww] = cntw[ww][tid]; tot += c[ww]; }
            uint32_t run = block_excl_scan<T2_THREADS>(tot, wtot);
            gdel[tid] = gcur - run;
            gcur += tot;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { cntw[ww][tid] = (uint16_t)run; run += c[ww]; }
        }
        __syncthreads();
        IXS_T(u_c);
#pragma unroll
        for (int j = 0; j < STEPS; ++j)
            if (key[j] != 0xffffffffu) {
                const uint32_t b2 = key[j] >> IXP_ID_BITS;
                const uint32_t at = cntw[w][b2] + rk[j];
                spos[at] = pos[j];
                sbin[at] = (uint16_t)b2;
                sid[at] = (uint8_t)(key[j] & (NID - 1));
            }
        __syncthreads();
        IXS_T(u_d);
        for (uint32_t i = tid; i < cn; i += T2_THREADS) {
#ifdef MECAT_IX_KNOCK
            if (g_ixknock == 1) { if (spos[i] == 0xfffffff1u) pos2[i] = sid[i]; continue; }
            if (g_ixknock == 2) { pos2[c0 + i] = spos[i] + gdel[sbin[i]]; id2[c0 + i] = sid[i]; continue; }
            if (g_ixknock == 3) { pos2[(size_t)(uint32_t)(gdel[sbin[i]] + i)] = spos[i] + sid[i]; continue; }
#endif
            const size_t o = (size_t)(uint32_t)(gdel[sbin[i]] + i);
            pos2[o] = spos[i];
            id2[o] = sid[i];
        }
        __syncthreads();
        IXS_T(u_e);
        IXS_ADD(9, u_a, u_b); IXS_ADD(10, u_b, u_c); IXS_ADD(11, u_c, u_d); IXS_ADD(12, u_d, u_e);
        for (int ww = 0; ww < NW; ++ww) cntw[ww][tid] = 0;
        __syncthreads();
    }
}

// ---- fill.  Kept positions per sub-bin (buckets above the cap dropped), for the scan that places the sub-bins in offsets[]
__global__ __launch_bounds__(FILL_THREADS) void ix_count3(const uint8_t* __restrict__ id2, const uint32_t* __restrict__ sub_ent, uint32_t max_bucket,
                                                          uint32_t* __restrict__ subkept) {
    __shared__ uint32_t h[NID];
    __shared__ uint32_t wsum[FILL_THREADS / 64];
    const uint32_t sb = blockIdx.x, tid = threadIdx.x;
    h[tid] = 0;
    __syncthreads();
    const uint32_t e0 = sub_ent[sb], e1 = sub_ent[sb + 1];
    const uint32_t* id4 = (const uint32_t*)id2;
    for (uint32_t wi = (e0 >> 2) + tid; wi < ((e1 + 3) >> 2); wi += FILL_THREADS) {
        const uint32_t v = id4[wi];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const uint32_t e = wi * 4 + k;
            if (e >= e0 && e < e1) atomicAdd(&h[(v >> (8 * k)) & 0xffu], 1u);
        }
    }
    __syncthreads();
    uint32_t k = h[tid] > max_bucket ? 0u : h[tid];
    for (int o = 32; o > 0; o >>= 1) k += __shfl_xor(k, o);
    if (lane_id() == 0) wsum[tid >> 6] = k;
    __syncthreads();
    if (tid == 0) subkept[sb] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of n u32 (n <= 2^22): per-block sums, scan of the sums by one workgroup, apply; out[n] = total
constexpr int SC_ITEMS = 16, SC_THREADS = 256, SC_TILE = SC_ITEMS * SC_THREADS;
__global__ __launch_bounds__(SC_THREADS) void ix_scan_sum(const uint32_t* __restrict__ in, uint32_t n, uint32_t* __restrict__ part) {
    __shared__ uint32_t wsum[SC_THREADS / 64];
    uint32_t s = 0;
    for (int k = 0; k < SC_ITEMS; ++k) {
        const uint32_t i = blockIdx.x * SC_TILE + k * SC_THREADS + threadIdx.x;
        s += i < n ? in[i] : 0u;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) part[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}
__global__ __launch_bounds__(1024) void ix_scan_parts(uint32_t* __restrict__ part, uint32_t nblk) {      // nblk <= 1024
    __shared__ uint32_t wtot[16];
    const uint32_t v = threadIdx.x < nblk ? part[threadIdx.x] : 0u;
    uint32_t all;
    const uint32_t ex = block_excl_scan<1024>(v, wtot, &all);
    if (threadIdx.x < nblk) part[threadIdx.x] = ex;
    if (threadIdx.x == 0) part[nblk] = all;
}
__global__ __launch_bounds__(SC_THREADS) void ix_scan_apply(const uint32_t* __restrict__ in, uint32_t n, const uint32_t* __restrict__ part, uint32_t nblk,
                                                            uint32_t* __restrict__ out) {
    __shared__ uint32_t wtot[SC_THREADS / 64];
    // thread t holds items t * SC_ITEMS .. of the block's tile (contiguous per thread: the prefix inside a thread is a running sum)
    uint32_t v[SC_ITEMS], s = 0;
    const uint32_t i0 = blockIdx.x * SC_TILE + threadIdx.x * SC_ITEMS;
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) { v[k] = i0 + k < n ? in[i0 + k] : 0u; s += v[k]; }
    uint32_t run = part[blockIdx.x] + block_excl_scan<SC_THREADS>(s, wtot);
#pragma unroll
    for (int k = 0; k < SC_ITEMS; ++k) { if (i0 + k < n) out[i0 + k] = run; run += v[k]; }
    if (blockIdx.x == nblk - 1 && threadIdx.x == SC_THREADS - 1) out[n] = part[nblk];
}

// one workgroup per sub-bin: entries (position, id) in ascending position order -> per-id buckets in the same order.
// Memory-level parallelism is explicit (a workgroup is ~6 k entries: its latency is what a CU's few resident workgroups cannot hide):
// the ids are staged in LDS by one sweep of independent word loads, the counting pass and the placing pass read them from there, and
// the placing pass loads its positions eight steps ahead.
constexpr int FILL_IDS = 12288;              // entries of a sub-bin whose ids fit the LDS stage (larger sub-bins re-read them from global memory)
#ifndef IXP_FILL_T
#define IXP_FILL_T 512
#endif
constexpr int FILL_T = IXP_FILL_T;          // threads of ix_fill (>= the 256 ids of a sub-bin)
static_assert(FILL_T >= NID && FILL_T % 64 == 0, "a thread per k-mer id");
template <bool STRICT>
__global__ __launch_bounds__(FILL_T) void ix_fill(const uint32_t* __restrict__ pos2, const uint8_t* __restrict__ id2, const uint32_t* __restrict__ sub_ent,
                                                        const uint32_t* __restrict__ sub_off, uint32_t max_bucket, uint32_t sub0, uint32_t* __restrict__ starts,
                                                        int32_t* __restrict__ offsets, uint16_t* __restrict__ slots, uint4* __restrict__ recs, int cut_step,
                                                        uint32_t* __restrict__ disorder) {
    constexpr int NW = FILL_T / 64;
    __shared__ uint32_t cntw[NW][NID];       // per-wave counts of an id, then the wave's next place in the id's bucket
    __shared__ uint32_t lstart[NID + 1];
    __shared__ int32_t buf[FILL_CAP];
    __shared__ uint32_t ids[FILL_IDS / 4 + 2];      // the sub-bin's id bytes from the word that holds entry e0 on
    __shared__ uint32_t cc[NID][2];                 // per id: kept entries by position cut (eight byte counters; a kept bucket holds <= 255)
    __shared__ uint32_t isfirst[FILL_CAP / 32];     // places that start a bucket (for the ascending check of the output sweep)
    __shared__ uint32_t wtot[NW];
    const uint32_t sb = blockIdx.x, tid = threadIdx.x, lane = lane_id(), w = tid >> 6;
    IXS_T(f_0);
    const uint32_t e0 = sub_ent[sb], e1 = sub_ent[sb + 1], n = e1 - e0;
    const uint32_t first = sub_off[sb];
    const size_t id0 = ((size_t)sub0 + sb) * NID;           // first k-mer id of the sub-bin
    const uint32_t ew0 = e0 & ~3u;                           // entry of the first staged id byte
    const bool staged = (e1 - ew0) <= (uint32_t)FILL_IDS;
    if (staged) {
        const uint32_t* id4 = (const uint32_t*)id2;
        for (uint32_t wi = tid; wi < ((e1 - ew0 + 3) >> 2); wi += FILL_T) ids[wi] = id4[(ew0 >> 2) + wi];
    }
    // (thread t < NID is also "the thread of k-mer id t" in the per-id steps below; with more threads than ids the others only walk entries)
    if (tid < (uint32_t)NID) {
        for (int ww = 0; ww < NW; ++ww) cntw[ww][tid] = 0;
        cc[tid][0] = 0; cc[tid][1] = 0;
    }
    for (uint32_t i = tid; i < (uint32_t)(FILL_CAP / 32); i += FILL_T) isfirst[i] = 0;
    __syncthreads();
    IXS_T(f_1);
    const uint8_t* idb = (const uint8_t*)ids;
    auto id_of = [&](uint32_t e) -> uint32_t { return staged ? (uint32_t)idb[e - ew0] : (uint32_t)id2[e]; };
    // each wave owns a contiguous quarter of the entries (whole 64-entry steps)
    const uint32_t per = ((n + NW * 64 - 1) / (NW * 64)) * 64;
    const uint32_t wb = e0 + w * per, we = min(e1, wb + per);
    for (uint32_t e = wb + lane; e < we; e += 64) atomicAdd(&cntw[w][id_of(e)], 1u);
    __syncthreads();
    IXS_T(f_2);
    {
        const bool isid = tid < (uint32_t)NID;
        uint32_t c[NW], tot = 0;
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) { c[ww] = isid ? cntw[ww][tid] : 0u; tot += c[ww]; }
        const uint32_t kept = tot > max_bucket ? 0u : tot;
        uint32_t all;
        const uint32_t ex = block_excl_scan<FILL_T>(kept, wtot, &all);
        if (isid) {
            lstart[tid] = ex;
            if (tid == NID - 1) lstart[NID] = all;
            uint32_t run = ex;
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) { cntw[ww][tid] = run; run += c[ww]; }      // (a dropped id's counters are never touched again)
            starts[id0 + tid] = first + ex;
            if (kept && ex < (uint32_t)FILL_CAP) atomicOr(&isfirst[ex >> 5], 1u << (ex & 31));
        }
    }
    __syncthreads();
    IXS_T(f_3);
    const uint32_t total = lstart[NID];
    const bool in_lds = total <= FILL_CAP;
    int32_t* gdst = offsets + first;
    constexpr int AHEAD = 8;
    for (uint32_t eb = wb; eb < we; eb += 64 * AHEAD) {
        uint32_t p[AHEAD];
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const uint32_t e = eb + 64 * k + lane;
            p[k] = e < we ? pos2[e] : 0u;
        }
#pragma unroll
        for (int k = 0; k < AHEAD; ++k) {
            const uint32_t e = eb + 64 * k + lane;
            if (e < we) {
                const uint32_t id = id_of(e);
                if (lstart[id + 1] != lstart[id]) {                  // the id's bucket is kept (an id that has entries and an empty bucket was dropped)
                    const uint32_t at = wave_rank<STRICT>(cntw[w], id);
                    if (in_lds) buf[at] = (int32_t)p[k]; else gdst[at] = (int32_t)p[k];
                    // the bucket record's counts: this position lies below the cuts t + 1 .. 7, t = position / cut_step (the eighth
                    // cut is at or behind the end of the volume)
                    const uint32_t t = min(7u, p[k] / (uint32_t)cut_step);
                    atomicAdd(&cc[id][t >> 2], 1u << ((t & 3u) << 3));
                }
            }
        }
    }
    __threadfence_block();
    __syncthreads();
    IXS_T(f_4);
    IXS_ADD(16, f_0, f_1); IXS_ADD(17, f_1, f_2); IXS_ADD(18, f_2, f_3); IXS_ADD(19, f_3, f_4);
    if (total == 0) {
        if (recs && tid < (uint32_t)NID) recs[id0 + tid] = make_uint4(first, 0u, 0u, 0u);
        return;
    }
    if (recs && tid < (uint32_t)NID) {
        // the bucket record of this thread's k-mer id (index.hip: idx_cut_records): occurrences below each of the seven position cuts =
        // running sums of the per-cut counters the placing pass made
        const uint32_t b0 = lstart[tid], b1 = lstart[tid + 1];
        const uint32_t c0 = cc[tid][0], c1 = cc[tid][1];
        uint32_t below[7], run = 0;
#pragma unroll
        for (int t = 0; t < 7; ++t) {
            run += ((t < 4 ? c0 : c1) >> ((t & 3) << 3)) & 0xffu;
            below[t] = run;
        }
        uint4 r;
        r.x = first + b0;
        r.y = below[0] | (below[1] << 8) | (below[2] << 16) | (below[3] << 24);
        r.z = below[4] | (below[5] << 8) | (below[6] << 16) | ((b1 - b0) << 24);
        r.w = 0;
        recs[id0 + tid] = r;
    }
    // positions out, with each position's slot in the seeding stage's relevance table ((position / ZV) mod 2^15, ZV = 2000), and the
    // check that every bucket is strictly ascending (every pass kept the input order; see wave_rank): a place that does not start
    // a bucket must hold a larger position than the place before it
    bool bad = false;
    if (in_lds) {
        for (uint32_t i = tid; i < total; i += FILL_T) {
            const int32_t pos = buf[i];
            const bool starts_bucket = (isfirst[i >> 5] >> (i & 31)) & 1u;
            bad |= !starts_bucket && pos <= buf[i ? i - 1 : 0];
            gdst[i] = pos;
            slots[first + i] = (uint16_t)(((uint32_t)pos / 2000u) & 0x7FFFu);
        }
    } else {
        // (more kept positions than the LDS buffer holds: they were placed in global memory; checked bucket by bucket)
        const uint32_t b0 = tid < (uint32_t)NID ? lstart[tid] : 0u, b1 = tid < (uint32_t)NID ? lstart[tid + 1] : 0u;
        for (uint32_t i = b0 + 1; i < b1; ++i) bad |= gdst[i] <= gdst[i - 1];
        for (uint32_t i = tid; i < total; i += FILL_T) slots[first + i] = (uint16_t)(((uint32_t)gdst[i] / 2000u) & 0x7FFFu);
    }
    if (bad) atomicOr(disorder, 1u);
    IXS_T(f_5);
    IXS_ADD(20, f_4, f_5);
}

#define S1_GRID(n) ((((n) + 7) / 8) * 8)
double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace

#define TRACE(tag) do { if (trace) { (void)hipStreamSynchronize(c->stream); double t_ = now_ms(); fprintf(stderr, "[idx trace] %-14s %.2f ms\n", tag, t_ - t0); t0 = t_; } } while (0)

// Builds the slice of the table that belongs to level-1 bins [bin_lo, bin_hi) (k-mer ids [bin_lo << 17, bin_hi << 17)):
//   d_starts_slice[((bin_hi - bin_lo) << 17) + 1]  bucket boundaries, RELATIVE to the slice's first kept position
//   offsets / slots / recs of the slice (recs' .x relative like starts), num_kept
// With a balance request (want_parts > 0) it only runs the first walk and returns the level-1 bin totals (host), for the caller to cut
// the key range into shards.
int index_build_partitioned(mhip_ctx* c, const mhip_volume* v, int max_bucket, int bin_lo, int bin_hi, IxpSlice* out) {
    const bool trace = getenv("MECAT_TRACE") != nullptr;
    double t0 = now_ms();
    const int nb = bin_hi - bin_lo;
    const int ntile = (int)(((int64_t)v->num_bases + T1_POS - 1) / T1_POS);
    const int ngroup = (ntile + G1 - 1) / G1;
    out->num_kept = 0;
    out->bin_total.assign((size_t)NB1 + 1, 0u);
    if (ntile == 0 || v->num_reads == 0) {      // nothing to index: an all-empty slice (no position arrays are allocated)
        if (out->d_starts) HIPCHK(hipMemsetAsync(out->d_starts, 0, sizeof(uint32_t) * (((size_t)nb << (26 - IXP_L1_BITS)) + 1), c->stream));
        if (out->d_recs && nb > 0) HIPCHK(hipMemsetAsync(out->d_recs, 0, sizeof(uint4) * ((size_t)nb << (26 - IXP_L1_BITS)), c->stream));
        return 0;
    }
    uint32_t *d_hist1, *d_grp, *d_binbase, *d_base1T;
    if (c->scratch("ixp_hist1", sizeof(uint32_t) * (size_t)ntile * NB1, (void**)&d_hist1)) return -1;
    if (c->scratch("ixp_grp", sizeof(uint32_t) * (size_t)ngroup * NB1, (void**)&d_grp)) return -1;
    if (c->scratch("ixp_binbase", sizeof(uint32_t) * (NB1 + 1), (void**)&d_binbase)) return -1;
    if (c->scratch("ixp_base1T", sizeof(uint32_t) * (size_t)NB1 * ((size_t)ntile + 1), (void**)&d_base1T)) return -1;
    uint32_t* d_nokmer;      // one bit per volume position: no k-mer starts here (ix_hist1 writes it, ix_scatter1 reads it)
    if (c->scratch("ixp_nokmer", sizeof(uint32_t) * (size_t)ntile * (T1_POS / 32), (void**)&d_nokmer)) return -1;
    LAUNCH(c, "ix_hist1", ix_hist1, ntile, T1_THREADS, 0, (const uint32_t*)v->d_pac, (const mhip_offset_t*)v->d_offs, v->num_reads, v->num_bases,
           (const uint32_t*)v->d_blk2read, d_hist1, d_nokmer);
    LAUNCH(c, "ix_scan1a", ix_scan1a, ngroup, NB1, 0, d_hist1, ntile, d_grp);
    LAUNCH(c, "ix_scan1b", ix_scan1b, 1, NB1, 0, d_grp, ngroup, d_binbase);
    HIPCHK(hipMemcpyAsync(out->bin_total.data(), d_binbase, sizeof(uint32_t) * (NB1 + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    TRACE("hist1+scan");
    if (out->count_only) return 0;
    const uint32_t ent_off = out->bin_total[(size_t)bin_lo];
    const uint32_t nent = out->bin_total[(size_t)bin_hi] - ent_off;
    uint32_t *d_ent1, *d_pos2, *d_hist2, *d_sub_ent, *d_subkept, *d_sub_off, *d_part;
    uint8_t* d_id2;
    const int nunit = nb * ngroup;
    const uint32_t nsub = (uint32_t)nb * NB2;
    const uint32_t nblk = (nsub + SC_TILE - 1) / SC_TILE;
    if (c->scratch("ixp_ent1", sizeof(uint32_t) * ((size_t)nent + 64), (void**)&d_ent1)) return -1;
    if (c->scratch("ixp_pos2", sizeof(uint32_t) * ((size_t)nent + 64), (void**)&d_pos2)) return -1;
    if (c->scratch("ixp_id2", (size_t)nent + 64, (void**)&d_id2)) return -1;
    if (c->scratch("ixp_hist2", sizeof(uint32_t) * (size_t)nunit * NB2, (void**)&d_hist2)) return -1;
    if (c->scratch("ixp_sub_ent", sizeof(uint32_t) * ((size_t)nsub + 1), (void**)&d_sub_ent)) return -1;
    if (c->scratch("ixp_subkept", sizeof(uint32_t) * ((size_t)nsub + 1), (void**)&d_subkept)) return -1;
    if (c->scratch("ixp_sub_off", sizeof(uint32_t) * ((size_t)nsub + 1), (void**)&d_sub_off)) return -1;
    if (c->scratch("ixp_part", sizeof(uint32_t) * ((size_t)nblk + 1), (void**)&d_part)) return -1;
    uint32_t* d_flag;
    if (c->scratch("ixp_flag", 64, (void**)&d_flag)) return -1;
#ifdef MECAT_IX_KNOCK
    { const int kn = getenv("MECAT_IX_KNOCK") ? atoi(getenv("MECAT_IX_KNOCK")) : 0; HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ixknock), &kn, sizeof(kn))); }
#endif
    TRACE("scratch");
    LAUNCH(c, "ix_transpose1", ix_transpose1, dim3((unsigned)((ntile + 31) / 32), NB1 / 32), 256, 0, (const uint32_t*)d_hist1, (const uint32_t*)d_grp,
           (const uint32_t*)d_binbase, ntile, d_base1T);
    const unsigned ublocks = (unsigned)(((nb + 7) / 8) * 8 * ngroup);
    const int cut_step = out->cut_step > 0 ? out->cut_step : 0x0fffffff;
    bool allocated = false;
    // MECAT_IDX_STRICT=1: never trust the order in which the LDS serves the lanes of one atomic instruction (wave_rank)
    bool strict = getenv("MECAT_IDX_STRICT") && atoi(getenv("MECAT_IDX_STRICT")) != 0;
    for (;;) {
        HIPCHK(hipMemsetAsync(d_flag, 0, 4, c->stream));
        {
            // ix_scatter1's tile order (see the kernel): MECAT_IDX_S1_XCD=0 / 1 fixes it; otherwise the first large build of a context runs the
            // kernel once in each order (same output) and the context keeps the faster one
            auto launch_s1 = [&](int ranges) {
                if (strict)
                    LAUNCH(c, "ix_scatter1", ix_scatter1<true>, S1_GRID(ntile), T1_THREADS, 0, (const uint32_t*)v->d_pac, v->num_bases, (const uint32_t*)d_nokmer,
                           (const uint32_t*)d_hist1, (const uint32_t*)d_grp, d_ent1, bin_lo, bin_hi, ent_off, ranges);
                else
                    LAUNCH(c, "ix_scatter1", ix_scatter1<false>, S1_GRID(ntile), T1_THREADS, 0, (const uint32_t*)v->d_pac, v->num_bases, (const uint32_t*)d_nokmer,
                           (const uint32_t*)d_hist1, (const uint32_t*)d_grp, d_ent1, bin_lo, bin_hi, ent_off, ranges);
            };
            if (const char* e = getenv("MECAT_IDX_S1_XCD")) c->ix_s1_ranges = atoi(e) != 0;
            if (c->ix_s1_ranges < 0 && ntile >= 8192) {
                hipEvent_t ev[3];
                for (auto& x : ev) HIPCHK(hipEventCreate(&x));
                HIPCHK(hipEventRecord(ev[0], c->stream));
                launch_s1(0);
                HIPCHK(hipEventRecord(ev[1], c->stream));
                launch_s1(1);
                HIPCHK(hipEventRecord(ev[2], c->stream));
                HIPCHK(hipEventSynchronize(ev[2]));
                float ms0 = 0.f, ms1 = 0.f;
                HIPCHK(hipEventElapsedTime(&ms0, ev[0], ev[1]));
                HIPCHK(hipEventElapsedTime(&ms1, ev[1], ev[2]));
                for (auto& x : ev) (void)hipEventDestroy(x);
                c->ix_s1_ranges = ms1 < ms0 ? 1 : 0;
                if (trace) fprintf(stderr, "[idx trace] ix_scatter1: tile = block %.2f ms, XCD ranges %.2f ms: keeping %s\n", ms0, ms1, c->ix_s1_ranges ? "the ranges" : "tile = block");
            } else {
                launch_s1(c->ix_s1_ranges > 0 ? 1 : 0);
            }
        }
        TRACE("scatter1");
        LAUNCH(c, "ix_hist2", ix_hist2, ublocks, T2_THREADS, 0, (const uint32_t*)d_ent1, (const uint32_t*)d_base1T, ntile, ngroup, nb, bin_lo, ent_off, d_hist2);
        LAUNCH(c, "ix_scan2", ix_scan2, nb, NB2, 0, d_hist2, ngroup, nb, (const uint32_t*)d_binbase, bin_lo, ent_off, d_sub_ent);
        if (strict)
            LAUNCH(c, "ix_scatter2", ix_scatter2<true>, ublocks, T2_THREADS, 0, (const uint32_t*)d_ent1, (const uint32_t*)d_base1T, ntile, ngroup, nb, bin_lo, ent_off,
                   (const uint32_t*)d_hist2, (const uint32_t*)d_sub_ent, d_pos2, d_id2);
        else
            LAUNCH(c, "ix_scatter2", ix_scatter2<false>, ublocks, T2_THREADS, 0, (const uint32_t*)d_ent1, (const uint32_t*)d_base1T, ntile, ngroup, nb, bin_lo, ent_off,
                   (const uint32_t*)d_hist2, (const uint32_t*)d_sub_ent, d_pos2, d_id2);
        TRACE("level 2");
        if (!allocated) {
            LAUNCH(c, "ix_count3", ix_count3, nsub, FILL_THREADS, 0, (const uint8_t*)d_id2, (const uint32_t*)d_sub_ent, (uint32_t)max_bucket, d_subkept);
            LAUNCH(c, "ix_scan_sum", ix_scan_sum, nblk, SC_THREADS, 0, (const uint32_t*)d_subkept, nsub, d_part);
            LAUNCH(c, "ix_scan_parts", ix_scan_parts, 1, 1024, 0, d_part, nblk);
            LAUNCH(c, "ix_scan_apply", ix_scan_apply, nblk, SC_THREADS, 0, (const uint32_t*)d_subkept, nsub, (const uint32_t*)d_part, nblk, d_sub_off);
            uint32_t total = 0;
            HIPCHK(hipMemcpyAsync(&total, d_sub_off + nsub, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
            HIPCHK(hipStreamSynchronize(c->stream));
            out->num_kept = total;
            TRACE("count3+scan");
            if (out->alloc(out, (size_t)total)) return -1;
            allocated = true;
            TRACE("alloc");
            HIPCHK(hipMemcpyAsync(out->d_starts + ((size_t)nb << (26 - IXP_L1_BITS)), d_sub_off + nsub, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
        }
        if (strict)
            LAUNCH(c, "ix_fill", ix_fill<true>, nsub, FILL_T, 0, (const uint32_t*)d_pos2, (const uint8_t*)d_id2, (const uint32_t*)d_sub_ent,
                   (const uint32_t*)d_sub_off, (uint32_t)max_bucket, 0u, out->d_starts, out->d_offsets, out->d_slots, out->d_recs, cut_step, d_flag);
        else
            LAUNCH(c, "ix_fill", ix_fill<false>, nsub, FILL_T, 0, (const uint32_t*)d_pos2, (const uint8_t*)d_id2, (const uint32_t*)d_sub_ent,
                   (const uint32_t*)d_sub_off, (uint32_t)max_bucket, 0u, out->d_starts, out->d_offsets, out->d_slots, out->d_recs, cut_step, d_flag);
        uint32_t flag = 0;
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(&flag, d_flag, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        TRACE("fill");
#ifdef MECAT_IX_STATS
        {
            unsigned long long hs[32];
            HIPCHK(hipMemcpyFromSymbol(hs, HIP_SYMBOL(g_ixstats), sizeof(hs)));
            const char* nm[32] = {"s1 prepare", "s1 rank", "s1 scan", "s1 stage", "s1 copy-out", 0, 0, 0, "s2 head", "s2 load+search+rank", "s2 scan", "s2 stage", "s2 copy-out", 0, 0, 0,
                                  "fill stage ids", "fill count", "fill scan", "fill place", "fill out"};
            for (int i = 0; i < 32; ++i) if (nm[i]) fprintf(stderr, "[ix stats] %-22s %10.1f ms summed over workgroups\n", nm[i], hs[i] * 1e-5);
            unsigned long long z[32] = {0};
            HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_ixstats), z, sizeof(z)));
        }
#endif
        if (!flag) break;
        if (strict) { mhip_set_error("index build: a bucket is not in ascending position order (internal error)"); return -1; }
        fprintf(stderr, "[mecat_hip] index build: the LDS did not serve one instruction's lanes in lane order; rebuilding with explicit ranks\n");
        strict = true;
    }
    return 0;
}
