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
#include "common.h"

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

extern "C" {

int mhip_index_build(mhip_ctx* c, const mhip_volume* v, mhip_index** out) {
    *out = nullptr;
    HIPCHK(hipSetDevice(c->device));
    mhip_index* idx = new mhip_index();
    idx->device = c->device;
    idx->num_bases = v->num_bases;
    uint32_t* d_counts = nullptr;
    uint32_t* d_partial = nullptr;
    if (c->scratch("idx_counts", sizeof(uint32_t) * (size_t)NKMER, (void**)&d_counts)) { delete idx; return -1; }
    if (c->scratch("idx_partial", sizeof(uint32_t) * (SCAN_BLOCKS + 1), (void**)&d_partial)) { delete idx; return -1; }
    if (hipMalloc((void**)&idx->d_starts, sizeof(uint32_t) * ((size_t)NKMER + 1)) != hipSuccess) {
        mhip_set_error("hipMalloc starts failed");
        delete idx;
        return -1;
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
    if (hipMalloc((void**)&idx->d_offsets, sizeof(int32_t) * ((size_t)total + 64)) != hipSuccess) {
        mhip_set_error("hipMalloc of %zu bytes for k-mer positions failed", sizeof(int32_t) * (size_t)total);
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
    *out = idx;
    return 0;
}

void mhip_index_free(mhip_index* idx) {
    if (!idx) return;
    (void)hipSetDevice(idx->device);
    if (idx->d_starts) (void)hipFree(idx->d_starts);
    if (idx->d_offsets) (void)hipFree(idx->d_offsets);
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
