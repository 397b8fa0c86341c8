// index.hip — k = 13 direct-address k-mer index of one 2-bit volume.
//
// Replaces create_ref_index / fill_ref_index_offsets_func (common/lookup_table.cpp:63-160, 25-61):
//   every overlapping 13-mer start of every read is bucketed by its 26-bit id; buckets with more than 128
//   occurrences are emptied (:97); bucket contents are ascending volume positions.
//
// HBM layout: starts[4^13 + 1] u32 (bucket i = offsets[starts[i] .. starts[i+1])), offsets[num_kmers] i32.
// One table replaces the reference's counts[] + pointer table: a probe reads two adjacent words of one line.
//
// Kernels (all HBM-bound integer work, no LDS reuse to exploit except the scan):
//   idx_count   one thread per 16 aligned volume positions: two coalesced 32-bit loads give the 28 bases those
//               16 k-mers need; one no-return global atomic per k-mer into counts[].
//   idx_scan_*  drop (>128), exclusive scan of 4^13 counters (reduce / scan-of-partials / scan) -> starts[].
//   idx_fill    same walk as idx_count; slot = atomic cursor per bucket; writes the position.
//   idx_sort    atomics return slots in arbitrary order, the reference's order is ascending position:
//               one wave per 64 consecutive buckets, each non-trivial bucket sorted by a 64-/128-wide
//               bitonic network in registers (wave shuffles), contiguous loads/stores.
// Algorithmic bytes: 2*(N/4) + 3*4*4^13 + 4*N_kept.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>

#include "common.h"
#include "index_part.h"

#define IDX_BLOCK 256
#define SCAN_ITEMS 16
#define SCAN_TILE (IDX_BLOCK * SCAN_ITEMS)          // 4096 counters per block
#define SCAN_BLOCKS (NKMER / SCAN_TILE)             // 16384

// last read whose offset <= p (reads are stored in ascending offset order, one pad base apart)
__device__ __forceinline__ int find_read(const mhip_offset_t* __restrict__ offs, int n, int p) {
    int lo = 0, hi = n;  // first index with offset > p
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (offs[mid].offset <= p) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

// the same for the 64 ascending positions of a wave: lane 0 searches, the others gallop forward from its answer
// (17 dependent loads per thread otherwise; a wave covers 1024 bases, i.e. one or two reads)
__device__ __forceinline__ int find_read_wave(const mhip_offset_t* __restrict__ offs, int n, int p) {
    int hint = 0;
    if (lane_id() == 0) hint = find_read(offs, n, p);
    hint = __shfl(hint, 0);
    if (n == 0) return -1;
    int lo = hint < 0 ? 0 : hint;                 // offs[lo].offset <= p unless hint == -1
    if (hint < 0 && offs[0].offset > p) return -1;
    int step = 1;
    while (lo + step < n && offs[lo + step].offset <= p) { lo += step; step <<= 1; }
    int hi = lo + step < n ? lo + step : n;       // offs[hi].offset > p or hi == n
    ++lo;                                          // first index with offset > p lies in [lo, hi]
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (offs[mid].offset <= p) lo = mid + 1; else hi = mid;
    }
    return lo - 1;
}

template <bool FILL>
__global__ __launch_bounds__(IDX_BLOCK) void idx_walk(const uint32_t* __restrict__ pac,
                                                      const mhip_offset_t* __restrict__ offs, int num_reads,
                                                      int num_bases, uint32_t* __restrict__ counts,
                                                      const uint32_t* __restrict__ starts, int32_t* __restrict__ offsets) {
    int64_t t = (int64_t)blockIdx.x * IDX_BLOCK + threadIdx.x;
    int64_t p0 = t << 4;
    if (p0 >= num_bases || num_reads == 0) return;
    uint64_t W = ((uint64_t)pac_word(pac, t) << 32) | pac_word(pac, t + 1);
    int r = find_read(offs, num_reads, (int)p0);
    int rend = r >= 0 ? offs[r].offset + offs[r].size : -1;       // one past the last base of read r
    int next_off = (r + 1 < num_reads) ? offs[r + 1].offset : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        int p = (int)p0 + j;
        if (p >= next_off) {
            ++r;
            rend = offs[r].offset + offs[r].size;
            next_off = (r + 1 < num_reads) ? offs[r + 1].offset : 0x7fffffff;
            // a read of size >= 1 plus its pad spans >= 2 positions, the loop advances one position per step,
            // so a single increment per position is enough
        }
        if (p + MHIP_KMER_SIZE <= rend) {
            uint32_t k = (uint32_t)(W >> (64 - 26 - 2 * j)) & KMER_MASK;
            if (!FILL) {
                atomicAdd(&counts[k], 1u);
            } else {
                uint32_t s0 = starts[k], s1 = starts[k + 1];
                if (s1 > s0) {
                    uint32_t slot = atomicAdd(&counts[k], 1u);
                    offsets[s0 + slot] = p;
                }
            }
        }
    }
}

__device__ __forceinline__ uint32_t kept(uint32_t c) { return c > MAX_BUCKET ? 0u : c; }

// per-block sum of kept counts
__global__ __launch_bounds__(IDX_BLOCK) void idx_scan_reduce(const uint32_t* __restrict__ counts, uint32_t* __restrict__ partial) {
    __shared__ uint32_t wsum[IDX_BLOCK / WAVE];
    const uint4* c4 = (const uint4*)(counts + (size_t)blockIdx.x * SCAN_TILE) + threadIdx.x * (SCAN_ITEMS / 4);
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
        uint4 v = c4[i];
        s += kept(v.x) + kept(v.y) + kept(v.z) + kept(v.w);
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane_id() == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of the SCAN_BLOCKS partials by one block; total -> partial[SCAN_BLOCKS]
__global__ __launch_bounds__(1024) void idx_scan_partials(uint32_t* __restrict__ partial) {
    __shared__ uint32_t wtot[16];
    const int per = SCAN_BLOCKS / 1024;   // 16
    uint32_t v[per];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < per; ++i) { v[i] = partial[threadIdx.x * per + i]; s += v[i]; }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
    uint32_t run = base + incl - s;
#pragma unroll
    for (int i = 0; i < per; ++i) { partial[threadIdx.x * per + i] = run; run += v[i]; }
    if (threadIdx.x == 1023) partial[SCAN_BLOCKS] = run;
}

// starts[i] = exclusive scan of kept counts ; counts[] is zeroed for reuse as the fill cursor
__global__ __launch_bounds__(IDX_BLOCK) void idx_scan_apply(uint32_t* __restrict__ counts, const uint32_t* __restrict__ partial,
                                                            uint32_t* __restrict__ starts) {
    __shared__ uint32_t wtot[IDX_BLOCK / WAVE];
    uint4* c4 = (uint4*)(counts + (size_t)blockIdx.x * SCAN_TILE) + threadIdx.x * (SCAN_ITEMS / 4);
    uint32_t v[SCAN_ITEMS];
    uint32_t s = 0;
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
        uint4 q = c4[i];
        v[4 * i] = kept(q.x); v[4 * i + 1] = kept(q.y); v[4 * i + 2] = kept(q.z); v[4 * i + 3] = kept(q.w);
        s += v[4 * i] + v[4 * i + 1] + v[4 * i + 2] + v[4 * i + 3];
        c4[i] = make_uint4(0, 0, 0, 0);
    }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = partial[blockIdx.x];
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
    uint32_t run = base + incl - s;
    uint4* s4 = (uint4*)(starts + (size_t)blockIdx.x * SCAN_TILE) + threadIdx.x * (SCAN_ITEMS / 4);
#pragma unroll
    for (int i = 0; i < SCAN_ITEMS / 4; ++i) {
        uint4 o;
        o.x = run; run += v[4 * i];
        o.y = run; run += v[4 * i + 1];
        o.z = run; run += v[4 * i + 2];
        o.w = run; run += v[4 * i + 3];
        s4[i] = o;
    }
    if (blockIdx.x == SCAN_BLOCKS - 1 && threadIdx.x == IDX_BLOCK - 1) starts[NKMER] = run;
}

__device__ __forceinline__ int cmpswap(int v, int partner, bool keep_min) {
    return keep_min ? min(v, partner) : max(v, partner);
}

// ascending bitonic sort of 64 values, one per lane
__device__ __forceinline__ int bitonic64(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            int partner = __shfl_xor(v, j);
            bool up = (lane & k) == 0;           // k == 64: always ascending
            bool lower = (lane & j) == 0;
            v = cmpswap(v, partner, lower == up);
        }
    }
    return v;
}

// ascending bitonic sort inside every aligned group of W lanes (W = 16 or 32), one value per lane
template <int W>
__device__ __forceinline__ int bitonic_group(int v) {
    const int lane = lane_id();
#pragma unroll
    for (int k = 2; k <= W; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            int partner = __shfl_xor(v, j);
            bool up = k == W || (lane & k) == 0;
            bool lower = (lane & j) == 0;
            v = cmpswap(v, partner, lower == up);
        }
    }
    return v;
}

// Sort, in place, the buckets selected by `todo` (one bucket per lane: [s0, e1) of that lane) with G = 64 / W buckets per
// pass, each in its own group of W lanes (bucket sizes <= W).  `data` may point to LDS or to global memory.
template <int W>
__device__ __forceinline__ void sort_small_buckets(unsigned long long todo, uint32_t s0, uint32_t e1, int32_t* data) {
    constexpr int G = 64 / W;
    const int lane = lane_id(), grp = lane / W, lig = lane % W;
    while (todo) {
        int src = -1;
        unsigned long long t = todo;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int bb = t ? __ffsll(t) - 1 : -1;
            if (g == grp) src = bb;
            t &= t - 1;          // t == 0 stays 0
        }
        todo = t;
        const uint32_t st = __shfl(s0, src < 0 ? 0 : src), en = __shfl(e1, src < 0 ? 0 : src);
        const int n = src < 0 ? 0 : (int)(en - st);
        int x = lig < n ? data[st + lig] : 0x7fffffff;
        x = bitonic_group<W>(x);
        if (lig < n) data[st + lig] = x;
    }
}

__device__ __forceinline__ void bitonic128(int& a, int& b);
__device__ __noinline__ void bitonic256(int32_t* p, const int n);

// every bucket of the wave's 64 lanes ([s0, e1) per lane, <= 128 entries) ascending: the reference's fill order
__device__ __forceinline__ void sort_wave_buckets(uint32_t s0, uint32_t e1, int32_t* data) {
    const int lane = lane_id();
    const uint32_t nb = e1 - s0;
    // buckets average ~22 positions: four 16-lane or two 32-lane sorts per pass; whole-wave sorts for the long ones
    sort_small_buckets<16>(__ballot(nb >= 2u && nb <= 16u), s0, e1, data);
    sort_small_buckets<32>(__ballot(nb > 16u && nb <= 32u), s0, e1, data);
    uint64_t todo = __ballot(nb > 32u);
    while (todo) {
        const int bb = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        const uint32_t st = __shfl(s0, bb), en = __shfl(e1, bb);
        const int n = (int)(en - st);
        if (n <= 64) {
            int x = lane < n ? data[st + lane] : 0x7fffffff;
            x = bitonic64(x);
            if (lane < n) data[st + lane] = x;
        } else if (n <= 128) {
            int a = data[st + lane];
            int c2 = lane + 64 < n ? data[st + 64 + lane] : 0x7fffffff;
            bitonic128(a, c2);
            data[st + lane] = a;
            if (lane + 64 < n) data[st + 64 + lane] = c2;
        } else {
            bitonic256(data + st, n);       // only an index built with a bucket cap above 128 has these (mhip_index_build_ex)
        }
    }
}

// ascending sort of n <= 256 non-negative values by one wave: bitonic network, element q * 64 + lane in register v[q]
__device__ __noinline__ void bitonic256(int32_t* p, const int n) {
    const int lane = lane_id();
    int v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i = q * 64 + lane; v[q] = i < n ? p[i] : 0x7fffffff; }
#pragma unroll
    for (int k = 2; k <= 256; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dq = j >> 6;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if ((q & dq) == 0) {
                        const int a = v[q], b = v[q | dq];
                        const bool asc = ((q * 64) & k) == 0;
                        v[q] = asc ? min(a, b) : max(a, b);
                        v[q | dq] = asc ? max(a, b) : min(a, b);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int o = __shfl_xor(v[q], j);
                    const bool asc = (((q * 64 + lane) & k) == 0), lower = (lane & j) == 0;
                    v[q] = (asc == lower) ? min(v[q], o) : max(v[q], o);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int i = q * 64 + lane; if (i < n) p[i] = v[q]; }
}

// ascending bitonic sort of 128 values: element i in lane i (a) and element 64+i in lane i (b)
__device__ __forceinline__ void bitonic128(int& a, int& b) {
    const int lane = lane_id();
    // sort a ascending and b descending (bitonic sequences of 64), then merge across
#pragma unroll
    for (int k = 2; k <= 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            int pa = __shfl_xor(a, j), pb = __shfl_xor(b, j);
            bool lower = (lane & j) == 0;
            bool up_a = (lane & k) == 0;                 // element index i = lane
            bool up_b = ((lane + 64) & k) == 0;          // element index i = lane + 64 ; for k == 64 -> descending
            if (k == 64) { up_a = true; up_b = false; }
            a = cmpswap(a, pa, lower == up_a);
            b = cmpswap(b, pb, lower == up_b);
        }
    }
    // final merge k = 128: j = 64 pairs (a_i, b_i), then j = 32..1 inside each half, all ascending
    int lo = min(a, b), hi = max(a, b);
    a = lo; b = hi;
#pragma unroll
    for (int j = 32; j > 0; j >>= 1) {
        int pa = __shfl_xor(a, j), pb = __shfl_xor(b, j);
        bool lower = (lane & j) == 0;
        a = cmpswap(a, pa, lower);
        b = cmpswap(b, pb, lower);
    }
}

// one wave per 64 consecutive buckets; sorts every bucket with >= 2 entries
__global__ __launch_bounds__(IDX_BLOCK) void idx_sort(const uint32_t* __restrict__ starts, int32_t* __restrict__ offsets) {
    const int lane = lane_id();
    const uint32_t wave = (uint32_t)(blockIdx.x * (IDX_BLOCK / WAVE) + (threadIdx.x >> 6));
    const uint32_t b0 = wave * 64u;
    uint32_t s0 = starts[b0 + lane], s1 = starts[b0 + lane + 1];
    uint64_t todo = __ballot(s1 - s0 >= 2u);
    while (todo) {
        int b = __ffsll((unsigned long long)todo) - 1;
        todo &= todo - 1;
        uint32_t st = __shfl(s0, b), en = __shfl(s1, b);
        int n = (int)(en - st);
        if (n <= 64) {
            int v = lane < n ? offsets[st + lane] : 0x7fffffff;
            v = bitonic64(v);
            if (lane < n) offsets[st + lane] = v;
        } else {
            int a = offsets[st + lane];
            int c = lane + 64 < n ? offsets[st + 64 + lane] : 0x7fffffff;
            bitonic128(a, c);
            offsets[st + lane] = a;
            if (lane + 64 < n) offsets[st + 64 + lane] = c;
        }
    }
}

// ==================================================================================================================
// Binned build (default).  The direct-address table is 268 MB, so the count/fill walks above are 1.6e9 read-modify-write
// round trips to memory-side atomics plus 4-byte scattered stores (~100 GB of HBM traffic each for 6 GB of algorithmic
// bytes).  Here the (k-mer, position) pairs are partitioned by the top 18 k-mer bits in three LDS-staged 64-way scatter
// passes (runs of ~512 bytes per bin and tile): 64 coarse bins -> 4096 fine bins (2^14 k-mer ids; one workgroup per fine bin
// counts occurrences per id in LDS, drops > 128 and scans -> starts[]) -> 262144 sub-bins of 256 ids, whose ~4.5 k kept
// positions fit LDS: one workgroup per sub-bin gathers them per bucket, sorts every bucket and writes its contiguous slice
// of offsets[] in one sweep.  (Two levels with the fill done per fine bin straight into global memory kept 512 x 1.5 MB
// of half-written lines in flight: 58 GB of HBM writes for 4.7 GB of positions.)  All global traffic is sequential or
// run-coalesced.
#define FINE_BITS 12
#define NFINE (1 << FINE_BITS)            // 4096 fine bins
#define NCOARSE 64
#define IDS_PER_FINE (NKMER >> FINE_BITS) // 16384 k-mer ids per fine bin
#define TILE_POS 4096                     // positions (level 1) / entries (level 2) per scatter tile
#define HIST_TILES 64                     // tiles per block in the histogram pass
#define BIN_THREADS 1024

// the 16 k-mer start positions of aligned window t (positions 16t .. 16t+15): f(j, kmer, pos) for each valid one (j = slot 0..15)
template <typename F>
__device__ __forceinline__ void walk16(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ offs, int num_reads,
                                       int num_bases, int64_t t, const uint32_t* __restrict__ blk, F f) {
    const int64_t p0 = t << 4;
    const bool live = p0 < num_bases && num_reads != 0;
    const uint64_t W = live ? ((uint64_t)pac_word(pac, t) << 32) | pac_word(pac, t + 1) : 0ull;
    if (!live) return;
    // the read of the window's first base: the volume's block table (entry b = the read that holds base 1024 b, or the one before
    // it) and a step or two forward, instead of a 17-step binary search over the offsets per wave
    int r = (int)blk[p0 >> 10];
    while (r + 1 < num_reads && offs[r + 1].offset <= (int)p0) ++r;
    int rend = r >= 0 ? offs[r].offset + offs[r].size : -1;
    int next_off = (r + 1 < num_reads) ? offs[r + 1].offset : 0x7fffffff;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int p = (int)p0 + j;
        if (p >= next_off) {
            ++r;
            rend = offs[r].offset + offs[r].size;
            next_off = (r + 1 < num_reads) ? offs[r + 1].offset : 0x7fffffff;
        }
        if (p + MHIP_KMER_SIZE <= rend) f(j, (uint32_t)(W >> (64 - 26 - 2 * j)) & KMER_MASK, p);
    }
}

// occurrences per fine bin
__global__ __launch_bounds__(IDX_BLOCK) void idx_hist(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ offs,
                                                      int num_reads, int num_bases, uint32_t* __restrict__ fine_hist, const uint32_t* __restrict__ blk) {
    __shared__ uint32_t h[NFINE];
    for (int i = threadIdx.x; i < NFINE; i += IDX_BLOCK) h[i] = 0;
    __syncthreads();
    for (int tile = 0; tile < HIST_TILES; ++tile) {
        const int64_t t = ((int64_t)blockIdx.x * HIST_TILES + tile) * IDX_BLOCK + threadIdx.x;
        walk16(pac, offs, num_reads, num_bases, t, blk, [&](int, uint32_t k, int) { atomicAdd(&h[k >> (26 - FINE_BITS)], 1u); });
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NFINE; i += IDX_BLOCK)
        if (h[i]) atomicAdd(&fine_hist[i], h[i]);
}

// exclusive scan of 4096 values (one block of 1024 threads, 4 items each); out[4096] = total
__global__ __launch_bounds__(1024) void idx_scan4096(const uint32_t* __restrict__ in, uint32_t* __restrict__ out) {
    __shared__ uint32_t wtot[16];
    uint32_t v[4], s = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = in[threadIdx.x * 4 + i]; s += v[i]; }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) base += wtot[w];
    uint32_t run = base + incl - s;
#pragma unroll
    for (int i = 0; i < 4; ++i) { out[threadIdx.x * 4 + i] = run; run += v[i]; }
    if (threadIdx.x == 1023) out[NFINE] = run;
}

// cursors of a scatter level: cur[b] = base of bin b's region
__global__ void idx_init_cursors(const uint32_t* __restrict__ fine_base, uint32_t* __restrict__ cur1, uint32_t* __restrict__ cur2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < NFINE) cur2[i] = fine_base[i];
    if (i < NCOARSE) cur1[i] = fine_base[i * (NFINE / NCOARSE)];
}

// LDS-staged 64-way scatter of up to TILE_POS entries held 16 per thread (slot j valid iff bit j of vmask, bins in binv[]):
// rank inside the tile by LDS atomics, one global reservation per bin and tile, then run-coalesced copy out.
__device__ __forceinline__ void scatter_tile64(const uint64_t* ent, const uint32_t* binv, uint32_t vmask, uint32_t* __restrict__ cursors,
                                               uint64_t* __restrict__ out, uint32_t* hist /*[64]*/, uint32_t* lbase /*[65]*/,
                                               uint32_t* gbase /*[64]*/, uint64_t* stage /*[TILE_POS]*/, uint8_t* sbin /*[TILE_POS]*/) {
    if (threadIdx.x < NCOARSE) hist[threadIdx.x] = 0;
    __syncthreads();
    uint32_t rank[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) rank[j] = ((vmask >> j) & 1u) ? atomicAdd(&hist[binv[j]], 1u) : 0u;
    __syncthreads();
    if (threadIdx.x < 64) {     // exclusive scan of the 64 counts by the first wave + global reservation
        const uint32_t c = hist[threadIdx.x];
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            uint32_t n = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += n;
        }
        lbase[threadIdx.x] = incl - c;
        if (threadIdx.x == 63) lbase[64] = incl;
        gbase[threadIdx.x] = c ? atomicAdd(&cursors[threadIdx.x], c) : 0u;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j)
        if ((vmask >> j) & 1u) {
            const uint32_t at = lbase[binv[j]] + rank[j];
            stage[at] = ent[j];
            sbin[at] = (uint8_t)binv[j];
        }
    __syncthreads();
    const uint32_t total = lbase[64];
    for (uint32_t i = threadIdx.x; i < total; i += IDX_BLOCK) {
        const uint32_t b = sbin[i];
        out[(size_t)gbase[b] + (i - lbase[b])] = stage[i];
    }
    __syncthreads();
}

// level 1: volume walk -> entries (kmer << 32 | pos) partitioned by the top 6 k-mer bits
__global__ __launch_bounds__(IDX_BLOCK) void idx_scatter1(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ offs,
                                                          int num_reads, int num_bases, uint32_t* __restrict__ cur1,
                                                          uint64_t* __restrict__ ent1, const uint32_t* __restrict__ blk) {
    __shared__ uint32_t hist[NCOARSE], lbase[NCOARSE + 1], gbase[NCOARSE];
    __shared__ uint64_t stage[TILE_POS];
    __shared__ uint8_t sbin[TILE_POS];
    uint64_t ent[16];
    uint32_t binv[16];
    uint32_t vmask = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { ent[j] = 0; binv[j] = 0; }
    const int64_t t = (int64_t)blockIdx.x * IDX_BLOCK + threadIdx.x;
    walk16(pac, offs, num_reads, num_bases, t, blk, [&](int j, uint32_t k, int p) {
        ent[j] = ((uint64_t)k << 32) | (uint32_t)p;
        binv[j] = k >> 20;
        vmask |= 1u << j;
    });
    scatter_tile64(ent, binv, vmask, cur1, ent1, hist, lbase, gbase, stage, sbin);
}

// level 2: inside coarse bin c (its own 64 fine cursors), by the next 6 k-mer bits.  blockIdx.y = coarse bin.
__global__ __launch_bounds__(IDX_BLOCK) void idx_scatter2(const uint64_t* __restrict__ ent1, const uint32_t* __restrict__ fine_base,
                                                          uint32_t* __restrict__ cur2, uint64_t* __restrict__ ent2, int coarse0) {
    __shared__ uint32_t hist[NCOARSE], lbase[NCOARSE + 1], gbase[NCOARSE];
    __shared__ uint64_t stage[TILE_POS];
    __shared__ uint8_t sbin[TILE_POS];
    const int c = coarse0 + blockIdx.y;      // (ent2 is addressed by volume-wide entry positions: the caller passes the group's base pointer shifted)
    const uint32_t cb = fine_base[c * 64], ce = fine_base[(c + 1) * 64];
    const uint64_t first = (uint64_t)cb + (uint64_t)blockIdx.x * TILE_POS;
    if (first >= ce) return;
    uint64_t ent[16];
    uint32_t binv[16];
    uint32_t vmask = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint64_t i = first + (uint64_t)j * IDX_BLOCK + threadIdx.x;      // coalesced reads
        const bool ok = i < ce;
        const uint64_t e = ok ? ent1[i] : 0ull;
        ent[j] = e;
        binv[j] = (uint32_t)(e >> (32 + 14)) & 63u;
        vmask |= ok ? (1u << j) : 0u;
    }
    scatter_tile64(ent, binv, vmask, cur2 + c * 64, ent2, hist, lbase, gbase, stage, sbin);
}

#define NSUB 64                           // sub-bins per fine bin (level 3)
#define IDS_PER_SUB (IDS_PER_FINE / NSUB)  // 256 k-mer ids per sub-bin
#define SUB_THREADS 256
#define SUB_CAP 8192                      // positions a sub-bin may hold in LDS (32 KB); ~4.5 k on average at config 2

// one workgroup per fine bin: kept occurrence counts of its 2^14 k-mer ids -> starts[] slice (temporarily counts), bin
// total, and the entry ranges of its 64 sub-bins (all occurrences, dropped buckets included) for the level-3 scatter
__global__ __launch_bounds__(BIN_THREADS) void idx_bin_count(const uint64_t* __restrict__ ent2, const uint32_t* __restrict__ fine_base,
                                                             uint32_t* __restrict__ starts, uint32_t* __restrict__ bintot,
                                                             uint32_t* __restrict__ sub_base, uint32_t* __restrict__ cur3, int fine0,
                                                             uint32_t max_bucket) {
    __shared__ uint32_t cnt[IDS_PER_FINE];     // 64 KB
    __shared__ uint32_t wtot[BIN_THREADS / WAVE];
    __shared__ uint32_t subtot[NSUB];
    const int b = fine0 + blockIdx.x;
    for (int i = threadIdx.x; i < IDS_PER_FINE; i += BIN_THREADS) cnt[i] = 0;
    __syncthreads();
    const uint32_t eb = fine_base[b], ee = fine_base[b + 1];
    for (uint32_t i = eb + threadIdx.x; i < ee; i += BIN_THREADS) atomicAdd(&cnt[(uint32_t)(ent2[i] >> 32) & (IDS_PER_FINE - 1)], 1u);
    __syncthreads();
    uint32_t s = 0;
    for (int i = threadIdx.x; i < IDS_PER_FINE; i += BIN_THREADS) {
        const uint32_t k = cnt[i] > max_bucket ? 0u : cnt[i];
        starts[(size_t)b * IDS_PER_FINE + i] = k;
        s += k;
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane_id() == 0) wtot[threadIdx.x >> 6] = s;
    {
        // sub-bin j = ids [256 j, 256 j + 256): 16 threads per sub-bin, thread u sums ids 256 j + u + 16 v
        const int j = threadIdx.x >> 4, u = threadIdx.x & 15;
        uint32_t t = 0;
#pragma unroll
        for (int v = 0; v < 16; ++v) t += cnt[j * IDS_PER_SUB + u + 16 * v];
        for (int o = 8; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (u == 0) subtot[j] = t;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int w = 0; w < BIN_THREADS / WAVE; ++w) tot += wtot[w];
        bintot[b] = tot;
    }
    if (threadIdx.x < NSUB) {
        const uint32_t c = subtot[threadIdx.x];
        uint32_t incl = c;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t n = __shfl_up(incl, o);
            if ((int)threadIdx.x >= o) incl += n;
        }
        const uint32_t at = eb + incl - c;
        sub_base[(size_t)b * NSUB + threadIdx.x] = at;
        cur3[(size_t)b * NSUB + threadIdx.x] = at;
        if (b == NFINE - 1 && threadIdx.x == NSUB - 1) sub_base[(size_t)NFINE * NSUB] = ee;
    }
}

// level 3: inside fine bin b (its own 64 sub-bin cursors), by k-mer bits 13..8.  blockIdx.y = fine bin.
__global__ __launch_bounds__(IDX_BLOCK) void idx_scatter3(const uint64_t* __restrict__ ent2, const uint32_t* __restrict__ fine_base,
                                                          uint32_t* __restrict__ cur3, uint64_t* __restrict__ ent3, int fine0) {
    __shared__ uint32_t hist[NCOARSE], lbase[NCOARSE + 1], gbase[NCOARSE];
    __shared__ uint64_t stage[TILE_POS];
    __shared__ uint8_t sbin[TILE_POS];
    const int b = fine0 + blockIdx.y;
    const uint32_t cb = fine_base[b], ce = fine_base[b + 1];
    const uint64_t first = (uint64_t)cb + (uint64_t)blockIdx.x * TILE_POS;
    if (first >= ce) return;
    uint64_t ent[16];
    uint32_t binv[16];
    uint32_t vmask = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const uint64_t i = first + (uint64_t)j * IDX_BLOCK + threadIdx.x;      // coalesced reads
        const bool ok = i < ce;
        const uint64_t e = ok ? ent2[i] : 0ull;
        ent[j] = e;
        binv[j] = (uint32_t)(e >> (32 + 8)) & 63u;
        vmask |= ok ? (1u << j) : 0u;
    }
    scatter_tile64(ent, binv, vmask, cur3 + (size_t)b * NSUB, ent3, hist, lbase, gbase, stage, sbin);
}

// one workgroup per fine bin: kept counts -> absolute starts[] slice
__global__ __launch_bounds__(BIN_THREADS) void idx_bin_starts(const uint32_t* __restrict__ binout, uint32_t* __restrict__ starts) {
    __shared__ uint32_t wtot[BIN_THREADS / WAVE];
    const int b = blockIdx.x;
    const int per = IDS_PER_FINE / BIN_THREADS;     // 16 consecutive ids per thread
    uint4* sp = (uint4*)(starts + (size_t)b * IDS_PER_FINE + (size_t)threadIdx.x * per);
    uint32_t v[per], s = 0;
#pragma unroll
    for (int i = 0; i < per / 4; ++i) {
        const uint4 q = sp[i];
        v[4 * i] = q.x; v[4 * i + 1] = q.y; v[4 * i + 2] = q.z; v[4 * i + 3] = q.w;
        s += q.x + q.y + q.z + q.w;
    }
    uint32_t incl = s;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t run = binout[b] + incl - s;
    for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) run += wtot[w];
#pragma unroll
    for (int i = 0; i < per / 4; ++i) {
        uint4 o;
        o.x = run; run += v[4 * i];
        o.y = run; run += v[4 * i + 1];
        o.z = run; run += v[4 * i + 2];
        o.w = run; run += v[4 * i + 3];
        sp[i] = o;
    }
}

// one workgroup per sub-bin (256 k-mer ids, one contiguous slice of offsets[]): the kept positions are gathered per bucket
// in LDS (cursors handed out by LDS atomics), every bucket is sorted ascending (atomics hand out slots in arbitrary order;
// the reference's fill order is ascending position and is load-bearing), and the slice is written out in one coalesced
// sweep.  A sub-bin with more kept positions than SUB_CAP (possible up to 256 x 128) does the same directly in global memory.
__global__ __launch_bounds__(SUB_THREADS) void idx_sub_fill(const uint64_t* __restrict__ ent3, const uint32_t* __restrict__ sub_base,
                                                            const uint32_t* __restrict__ starts, int32_t* __restrict__ offsets,
                                                            uint16_t* __restrict__ slots, uint4* __restrict__ recs, int cut_step) {
    __shared__ uint32_t lstart[IDS_PER_SUB + 1];
    __shared__ uint32_t cursor[IDS_PER_SUB];
    __shared__ int32_t buf[SUB_CAP];
    const uint32_t sb = blockIdx.x;
    const size_t id0 = (size_t)sb * IDS_PER_SUB;
    const uint32_t first = starts[id0];
    const uint32_t mine = starts[id0 + threadIdx.x] - first;
    lstart[threadIdx.x] = mine;
    cursor[threadIdx.x] = mine;
    if (threadIdx.x == 0) lstart[IDS_PER_SUB] = starts[id0 + IDS_PER_SUB] - first;
    __syncthreads();
    const uint32_t total = lstart[IDS_PER_SUB];
    if (total == 0) {
        if (recs) recs[id0 + threadIdx.x] = make_uint4(first, 0u, 0u, 0u);
        return;
    }
    const bool in_lds = total <= SUB_CAP;
    int32_t* gdst = offsets + first;
    const uint32_t eb = sub_base[sb], ee = sub_base[sb + 1];
    for (uint32_t i = eb + threadIdx.x; i < ee; i += SUB_THREADS) {
        const uint64_t e = ent3[i];
        const uint32_t id = (uint32_t)(e >> 32) & (IDS_PER_SUB - 1);
        if (lstart[id + 1] != lstart[id]) {
            const uint32_t slot = atomicAdd(&cursor[id], 1u);
            if (in_lds) buf[slot] = (int32_t)(uint32_t)e;
            else gdst[slot] = (int32_t)(uint32_t)e;
        }
    }
    __threadfence_block();
    __syncthreads();
    {
        const int id = threadIdx.x;      // wave w sorts the buckets of ids 64 w .. 64 w + 63
        if (in_lds) sort_wave_buckets(lstart[id], lstart[id + 1], buf);
        else sort_wave_buckets(lstart[id], lstart[id + 1], gdst);
    }
    // positions out, and with them each position's slot in the seeding stage's relevance table: (position / ZV) mod 2^15, ZV = 2000
    __syncthreads();
    if (recs) {
        // the bucket record of this thread's k-mer id (see idx_cut_records): the sorted bucket is at hand, one pass against the
        // ascending cuts
        const uint32_t b0 = lstart[threadIdx.x], b1 = lstart[threadIdx.x + 1];
        uint32_t below[7];
        int t = 0;
        for (uint32_t i = b0; i < b1 && t < 7; ++i) {
            const int pos = in_lds ? buf[i] : gdst[i];
            while (t < 7 && pos >= (t + 1) * cut_step) below[t++] = i - b0;
        }
        while (t < 7) below[t++] = b1 - b0;
        uint4 r;
        r.x = first + b0;
        r.y = below[0] | (below[1] << 8) | (below[2] << 16) | (below[3] << 24);
        r.z = below[4] | (below[5] << 8) | (below[6] << 16) | ((b1 - b0) << 24);
        r.w = 0;
        recs[id0 + threadIdx.x] = r;
    }
    if (in_lds) {
        for (uint32_t i = threadIdx.x; i < total; i += SUB_THREADS) {
            const int32_t pos = buf[i];
            gdst[i] = pos;
            slots[first + i] = (uint16_t)(((uint32_t)pos / 2000u) & 0x7FFFu);
        }
    } else {
        __threadfence_block();
        for (uint32_t i = threadIdx.x; i < total; i += SUB_THREADS) slots[first + i] = (uint16_t)(((uint32_t)gdst[i] / 2000u) & 0x7FFFu);
    }
}

#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define TRACE(tag) do { if (trace) { (void)hipStreamSynchronize(c->stream); double t_ = now_ms(); fprintf(stderr, "[idx trace] %-14s %.2f ms\n", tag, t_ - t0); t0 = t_; } } while (0)

static int index_build_binned(mhip_ctx* c, const mhip_volume* v, mhip_index* idx) {
    const bool trace = getenv("MECAT_TRACE") != nullptr;
    double t0 = now_ms();
    uint32_t *d_hist, *d_fbase, *d_cur1, *d_cur2, *d_bintot, *d_binout;
    if (c->scratch("ix_hist", sizeof(uint32_t) * (NFINE + 1), (void**)&d_hist)) return -1;
    if (c->scratch("ix_fbase", sizeof(uint32_t) * (NFINE + 1), (void**)&d_fbase)) return -1;
    if (c->scratch("ix_cur1", sizeof(uint32_t) * NCOARSE, (void**)&d_cur1)) return -1;
    if (c->scratch("ix_cur2", sizeof(uint32_t) * NFINE, (void**)&d_cur2)) return -1;
    if (c->scratch("ix_bintot", sizeof(uint32_t) * (NFINE + 1), (void**)&d_bintot)) return -1;
    if (c->scratch("ix_binout", sizeof(uint32_t) * (NFINE + 1), (void**)&d_binout)) return -1;
    HIPCHK(hipMemsetAsync(d_hist, 0, sizeof(uint32_t) * (NFINE + 1), c->stream));
    const int64_t nthreads = ((int64_t)v->num_bases + 15) / 16;
    const unsigned tiles = (unsigned)((nthreads + IDX_BLOCK - 1) / IDX_BLOCK);
    if (tiles == 0) {
        HIPCHK(hipMemsetAsync(idx->d_starts, 0, sizeof(uint32_t) * ((size_t)NKMER + 1), c->stream));
        idx->num_kmers = 0;
        return 0;
    }
    LAUNCH(c, "idx_hist", idx_hist, (tiles + HIST_TILES - 1) / HIST_TILES, IDX_BLOCK, 0, (const uint32_t*)v->d_pac, (const mhip_offset_t*)v->d_offs,
           v->num_reads, v->num_bases, d_hist, (const uint32_t*)v->d_blk2read);
    LAUNCH(c, "idx_scan4096", idx_scan4096, 1, 1024, 0, (const uint32_t*)d_hist, d_fbase);
    TRACE("hist+scan");
    // Entries: level 1 scatters the whole volume into ent1 (64 coarse bins).  Levels 2 and 3 run per group of IDX_GROUP coarse
    // bins: level 2 moves the group's entries into the small ping buffer ent2, level 3 moves them back into the same range of
    // ent1 (fine bins keep their ranges through both levels) — so the ping buffer is a group's size, not the volume's.
    std::vector<uint32_t> fb(NFINE + 1);
    HIPCHK(hipMemcpyAsync(fb.data(), d_fbase, sizeof(uint32_t) * (NFINE + 1), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    const uint32_t nent = fb[NFINE];
    constexpr int IDX_GROUP = 16, NGROUP = NCOARSE / IDX_GROUP;
    uint32_t gmax = 0;
    for (int g = 0; g < NGROUP; ++g) gmax = std::max(gmax, fb[(size_t)(g + 1) * IDX_GROUP * 64] - fb[(size_t)g * IDX_GROUP * 64]);
    uint64_t *d_e1, *d_e2;
    if (c->scratch("ix_ent1", sizeof(uint64_t) * ((size_t)nent + 64), (void**)&d_e1)) return -1;
    if (c->scratch("ix_ent2", sizeof(uint64_t) * ((size_t)gmax + 64), (void**)&d_e2)) return -1;
    TRACE("scratch");
    uint32_t *d_subbase, *d_cur3;
    if (c->scratch("ix_subbase", sizeof(uint32_t) * ((size_t)NFINE * NSUB + 1), (void**)&d_subbase)) return -1;
    if (c->scratch("ix_cur3", sizeof(uint32_t) * (size_t)NFINE * NSUB, (void**)&d_cur3)) return -1;
    LAUNCH(c, "idx_init_cursors", idx_init_cursors, NFINE / 256, 256, 0, (const uint32_t*)d_fbase, d_cur1, d_cur2);
    LAUNCH(c, "idx_scatter1", idx_scatter1, tiles, IDX_BLOCK, 0, (const uint32_t*)v->d_pac, (const mhip_offset_t*)v->d_offs, v->num_reads,
           v->num_bases, d_cur1, d_e1, (const uint32_t*)v->d_blk2read);
    for (int g = 0; g < NGROUP; ++g) {
        const int c0 = g * IDX_GROUP, f0 = c0 * 64, nf = IDX_GROUP * 64;
        const uint32_t gbase = fb[(size_t)f0];
        // grid.x covers the largest bin of the group; blocks past a bin's end exit immediately
        uint32_t mx = 0, mx3 = 0;
        for (int cc = c0; cc < c0 + IDX_GROUP; ++cc) mx = std::max(mx, fb[(size_t)(cc + 1) * 64] - fb[(size_t)cc * 64]);
        for (int f = f0; f < f0 + nf; ++f) mx3 = std::max(mx3, fb[(size_t)f + 1] - fb[(size_t)f]);
        const unsigned gx = (mx + TILE_POS - 1) / TILE_POS, gx3 = (mx3 + TILE_POS - 1) / TILE_POS;
        uint64_t* e2v = d_e2 - gbase;                    // the kernels index ent2 by volume-wide entry positions
        if (gx) LAUNCH(c, "idx_scatter2", idx_scatter2, dim3(gx, IDX_GROUP), IDX_BLOCK, 0, (const uint64_t*)d_e1, (const uint32_t*)d_fbase, d_cur2, e2v, c0);
        LAUNCH(c, "idx_bin_count", idx_bin_count, nf, BIN_THREADS, 0, (const uint64_t*)e2v, (const uint32_t*)d_fbase, idx->d_starts, d_bintot,
               d_subbase, d_cur3, f0, (uint32_t)idx->max_bucket);
        if (gx3) LAUNCH(c, "idx_scatter3", idx_scatter3, dim3(gx3, nf), IDX_BLOCK, 0, (const uint64_t*)e2v, (const uint32_t*)d_fbase, d_cur3, d_e1, f0);
    }
    TRACE("scatter");
    LAUNCH(c, "idx_scan4096", idx_scan4096, 1, 1024, 0, (const uint32_t*)d_bintot, d_binout);
    LAUNCH(c, "idx_bin_starts", idx_bin_starts, NFINE, BIN_THREADS, 0, (const uint32_t*)d_binout, idx->d_starts);
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_binout + NFINE, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    idx->num_kmers = total;
    TRACE("count+scatter3");
    if (dev_alloc_recycled(c->device, sizeof(int32_t) * ((size_t)total + 64), (void**)&idx->d_offsets, &idx->cap_offsets)) return -1;
    TRACE("malloc offsets");
    HIPCHK(hipMemcpyAsync(idx->d_starts + NKMER, d_binout + NFINE, sizeof(uint32_t), hipMemcpyDeviceToDevice, c->stream));
    if (dev_alloc_recycled(c->device, sizeof(uint16_t) * ((size_t)total + 64), (void**)&idx->d_slots, &idx->cap_slots)) return -1;
    // bucket records with the prefix counts at seven position cuts (idx_cut_records): written by the kernel that has every sorted
    // bucket in LDS anyway
    if (idx->max_bucket <= 255 && dev_alloc_recycled(c->device, sizeof(uint4) * (size_t)NKMER, (void**)&idx->d_recs, &idx->cap_recs) == 0) {
        const int segs = (idx->num_bases + 2000 - 1) / 2000;
        idx->cut_step = ((segs + 7) / 8) * 2000;
    } else {
        idx->d_recs = nullptr;
    }
    LAUNCH(c, "idx_sub_fill", idx_sub_fill, NFINE * NSUB, SUB_THREADS, 0, (const uint64_t*)d_e1, (const uint32_t*)d_subbase,
           (const uint32_t*)idx->d_starts, idx->d_offsets, idx->d_slots, idx->d_recs, idx->cut_step);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    TRACE("bin_fill");
    return 0;
}

// slot of a k-mer position in the seeding stage's relevance table: (position / ZV) mod 2^15, ZV = 2000 (seed.hip)
__global__ __launch_bounds__(256) void idx_slots(const int32_t* __restrict__ offsets, int64_t n, uint16_t* __restrict__ slots) {
    const int64_t i0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
    if (i0 + 3 < n) {
        const int4 p = *(const int4*)(offsets + i0);
        ushort4 o;
        o.x = (uint16_t)(((uint32_t)p.x / 2000u) & 0x7FFFu); o.y = (uint16_t)(((uint32_t)p.y / 2000u) & 0x7FFFu);
        o.z = (uint16_t)(((uint32_t)p.z / 2000u) & 0x7FFFu); o.w = (uint16_t)(((uint32_t)p.w / 2000u) & 0x7FFFu);
        *(ushort4*)(slots + i0) = o;
    } else {
        for (int64_t i = i0; i < n; ++i) slots[i] = (uint16_t)(((uint32_t)offsets[i] / 2000u) & 0x7FFFu);
    }
}

// The relevance filter of the seeding stage only needs each position's 2 kb-segment slot (15 bits): a second array at half the
// bytes halves the sectors its bucket walk touches.
// ---- bucket records with prefix counts at seven position cuts (seed.hip: the probe of a diagonal grid cell).
// A query read of the reference volume itself only needs the bucket entries in front of its own copy (+ a few segments):
// everything behind belongs to reads with higher ids, whose candidates get_candidates drops (pw_impl.cpp:370).  Bucket entries are
// ascending positions, so "in front of position P" is a prefix of every bucket, and its length for seven fixed cuts P fits the
// 16-byte record the probe reads anyway instead of two words of starts[]: x = start, the bytes of y:z = occurrences below cut
// 1..7 and, last, all occurrences (<= 255), w = 0.
__global__ __launch_bounds__(256) void idx_cut_records(const uint32_t* __restrict__ starts, const int32_t* __restrict__ offsets, int cut_step,
                                                      uint4* __restrict__ recs) {
    const uint32_t id = blockIdx.x * 256u + threadIdx.x;
    const uint32_t s0 = starts[id], s1 = starts[id + 1];
    uint32_t below[7] = {0, 0, 0, 0, 0, 0, 0};
    for (uint32_t i = s0; i < s1; ++i) {
        const int pos = offsets[i];
#pragma unroll
        for (int t = 0; t < 7; ++t) below[t] += pos < (t + 1) * cut_step ? 1u : 0u;
    }
    uint4 r;
    r.x = s0;
    r.y = below[0] | (below[1] << 8) | (below[2] << 16) | (below[3] << 24);
    r.z = below[4] | (below[5] << 8) | (below[6] << 16) | ((s1 - s0) << 24);
    r.w = 0;
    recs[id] = r;
}

const uint4* index_ensure_cuts(mhip_ctx* c, const mhip_index* cidx) {
    mhip_index* idx = const_cast<mhip_index*>(cidx);          // a cache inside the handle
    if (idx->max_bucket > 255) return nullptr;
    std::lock_guard<std::mutex> lk(idx->recs_mu);
    if (idx->d_recs) return idx->d_recs;
    if (hipSetDevice(c->device) != hipSuccess) return nullptr;
    uint4* p = nullptr;
    if (dev_alloc_recycled(c->device, sizeof(uint4) * (size_t)NKMER, (void**)&p, &idx->cap_recs)) return nullptr;   // no room: no cuts
    // cuts at multiples of a whole number of segments, the eighth at or behind the end of the volume
    const int segs = (idx->num_bases + 2000 - 1) / 2000;
    idx->cut_step = ((segs + 7) / 8) * 2000;
    LAUNCH(c, "idx_cut_records", idx_cut_records, NKMER / 256, 256, 0, (const uint32_t*)idx->d_starts, (const int32_t*)idx->d_offsets, idx->cut_step, p);
    if (hipStreamSynchronize(c->stream) != hipSuccess) { dev_free_recycled(c->device, p, idx->cap_recs); return nullptr; }
    idx->d_recs = p;
    return p;
}

static int index_add_slots(mhip_ctx* c, mhip_index* idx) {
    if (idx->num_kmers <= 0 || idx->d_slots) return 0;      // the binned build writes the slots with the positions
    if (dev_alloc_recycled(c->device, sizeof(uint16_t) * ((size_t)idx->num_kmers + 64), (void**)&idx->d_slots, &idx->cap_slots)) return -1;
    LAUNCH(c, "idx_slots", idx_slots, (unsigned)((idx->num_kmers + 1023) / 1024), 256, 0, (const int32_t*)idx->d_offsets, idx->num_kmers,
           idx->d_slots);
    HIPCHK(hipGetLastError());
    return 0;
}

// the whole table through the stable two-level partition (index_part.hip): level-1 bins [0, 512)
static int index_build_whole(mhip_ctx* c, const mhip_volume* v, mhip_index* idx) {
    IxpSlice sl;
    sl.d_starts = idx->d_starts;
    sl.user = idx;
    const bool with_recs = idx->max_bucket <= 255;
    if (with_recs) {
        const int segs = (idx->num_bases + 2000 - 1) / 2000;
        idx->cut_step = ((segs + 7) / 8) * 2000;
        sl.cut_step = idx->cut_step;
    }
    struct Ctx { mhip_ctx* c; mhip_index* idx; bool with_recs; } cx{c, idx, with_recs};
    sl.user = &cx;
    sl.alloc = [](IxpSlice* s, size_t kept) -> int {
        Ctx* x = (Ctx*)s->user;
        mhip_index* I = x->idx;
        if (dev_alloc_recycled(x->c->device, sizeof(int32_t) * (kept + 64), (void**)&I->d_offsets, &I->cap_offsets)) return -1;
        if (dev_alloc_recycled(x->c->device, sizeof(uint16_t) * (kept + 64), (void**)&I->d_slots, &I->cap_slots)) return -1;
        if (x->with_recs && dev_alloc_recycled(x->c->device, sizeof(uint4) * (size_t)NKMER, (void**)&I->d_recs, &I->cap_recs) != 0) I->d_recs = nullptr;
        s->d_offsets = I->d_offsets;
        s->d_slots = I->d_slots;
        s->d_recs = I->d_recs;
        return 0;
    };
    if (index_build_partitioned(c, v, idx->max_bucket, 0, IXP_NB1, &sl)) return -1;
    idx->num_kmers = sl.num_kept;
    if (sl.num_kept == 0 && !idx->d_offsets) {      // empty volume: starts[] is all zero, nothing else exists
        HIPCHK(hipMemsetAsync(idx->d_starts, 0, sizeof(uint32_t) * ((size_t)NKMER + 1), c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

extern "C" {

int mhip_index_build(mhip_ctx* c, const mhip_volume* v, mhip_index** out) { return mhip_index_build_ex(c, v, MAX_BUCKET, out); }

int mhip_index_build_ex(mhip_ctx* c, const mhip_volume* v, int max_bucket, mhip_index** out) {
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    if (max_bucket < 1 || max_bucket > 256) { mhip_set_error("bucket cap %d outside 1..256", max_bucket); return -1; }
    {
        const char* e = getenv("MECAT_IDX_BUILD");
        if (e && atoi(e) == 1 && max_bucket != MAX_BUCKET) { mhip_set_error("MECAT_IDX_BUILD=1 (direct atomic walks) only builds with the default bucket cap"); return -1; }
    }
    const double tb0 = now_ms();
    mhip_index* idx = new mhip_index();
    idx->device = c->device;
    idx->num_bases = v->num_bases;
    idx->max_bucket = max_bucket;
    uint32_t* d_counts = nullptr;
    uint32_t* d_partial = nullptr;
    if (c->scratch("idx_counts", sizeof(uint32_t) * (size_t)NKMER, (void**)&d_counts)) { delete idx; return -1; }
    if (c->scratch("idx_partial", sizeof(uint32_t) * (SCAN_BLOCKS + 1), (void**)&d_partial)) { delete idx; return -1; }
    if (dev_alloc_recycled(c->device, sizeof(uint32_t) * ((size_t)NKMER + 1), (void**)&idx->d_starts, &idx->cap_starts)) {
        delete idx;
        return -1;
    }
    {
        const char* e = getenv("MECAT_IDX_BUILD");       // debug knob: 1 = direct atomic walks, default = binned build
        if (e && atoi(e) == 2) {           // debug knob: round 2's build (three 64-way passes of 8-byte entries + per-bucket sorts)
            if (index_build_binned(c, v, idx)) { mhip_index_free(idx); return -1; }
            if (index_add_slots(c, idx)) { mhip_index_free(idx); return -1; }
            *out = idx;
            return 0;
        }
        if (!(e && atoi(e) == 1)) {
            if (getenv("MECAT_TRACE")) fprintf(stderr, "[idx trace] pre-alloc      %.2f ms\n", now_ms() - tb0);
            if (index_build_whole(c, v, idx)) { mhip_index_free(idx); return -1; }
            *out = idx;
            return 0;
        }
    }
    HIPCHK(hipMemsetAsync(d_counts, 0, sizeof(uint32_t) * (size_t)NKMER, c->stream));
    int64_t nthreads = ((int64_t)v->num_bases + 15) / 16;
    unsigned grid = (unsigned)((nthreads + IDX_BLOCK - 1) / IDX_BLOCK);
    if (grid > 0)
        LAUNCH(c, "idx_count", (idx_walk<false>), grid, IDX_BLOCK, 0, v->d_pac, v->d_offs, v->num_reads, v->num_bases,
               d_counts, (const uint32_t*)nullptr, (int32_t*)nullptr);
    LAUNCH(c, "idx_scan_reduce", idx_scan_reduce, SCAN_BLOCKS, IDX_BLOCK, 0, d_counts, d_partial);
    LAUNCH(c, "idx_scan_partials", idx_scan_partials, 1, 1024, 0, d_partial);
    LAUNCH(c, "idx_scan_apply", idx_scan_apply, SCAN_BLOCKS, IDX_BLOCK, 0, d_counts, d_partial, idx->d_starts);
    uint32_t total = 0;
    HIPCHK(hipMemcpyAsync(&total, d_partial + SCAN_BLOCKS, sizeof(uint32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    idx->num_kmers = total;
    if (dev_alloc_recycled(c->device, sizeof(int32_t) * ((size_t)total + 64), (void**)&idx->d_offsets, &idx->cap_offsets)) {
        mhip_index_free(idx);
        return -1;
    }
    if (grid > 0 && total > 0) {
        LAUNCH(c, "idx_fill", (idx_walk<true>), grid, IDX_BLOCK, 0, v->d_pac, v->d_offs, v->num_reads, v->num_bases,
               d_counts, (const uint32_t*)idx->d_starts, idx->d_offsets);
        LAUNCH(c, "idx_sort", idx_sort, NKMER / 64 / (IDX_BLOCK / WAVE), IDX_BLOCK, 0, (const uint32_t*)idx->d_starts, idx->d_offsets);
    }
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(c->stream));
    if (index_add_slots(c, idx)) { mhip_index_free(idx); return -1; }
    *out = idx;
    return 0;
}

void mhip_index_free(mhip_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    // the arrays are parked for the next build, which may run on another context / stream: nothing queued anywhere on the
    // device may still read them (once per grid row, so a device-wide wait costs nothing measurable)
    (void)hipDeviceSynchronize();
    dev_free_recycled(idx->device, idx->d_starts, idx->cap_starts);
    dev_free_recycled(idx->device, idx->d_offsets, idx->cap_offsets);
    dev_free_recycled(idx->device, idx->d_slots, idx->cap_slots);
    dev_free_recycled(idx->device, idx->d_recs, idx->cap_recs);
    delete idx;
}

int64_t mhip_index_num_kmers(const mhip_index* idx) { return idx->num_kmers; }

int mhip_index_download(mhip_ctx* c, const mhip_index* idx, int32_t* counts, int32_t* offsets) {
    HIPCHK(hipSetDevice(c->device));
    if (counts) {
        std::vector<uint32_t> st((size_t)NKMER + 1);
        HIPCHK(hipMemcpyAsync(st.data(), idx->d_starts, sizeof(uint32_t) * st.size(), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
        for (size_t i = 0; i < (size_t)NKMER; ++i) counts[i] = (int32_t)(st[i + 1] - st[i]);
    }
    if (offsets && idx->num_kmers) {
        HIPCHK(hipMemcpyAsync(offsets, idx->d_offsets, sizeof(int32_t) * (size_t)idx->num_kmers, hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));
    }
    return 0;
}

}  // extern "C"
