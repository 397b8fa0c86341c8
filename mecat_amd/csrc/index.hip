// index.hip — k = 13 direct-address k-mer index of one 2-bit volume.
//
// Replaces create_ref_index / fill_ref_index_offsets_func (common/lookup_table.cpp:63-160, 25-61):
//   every overlapping 13-mer start of every read is bucketed by its 26-bit id; buckets with more than 128
//   occurrences are emptied (:97); bucket contents are ascending volume positions.
//
// HBM layout: starts[4^13 + 1] u32 (bucket i = offsets[starts[i] .. starts[i+1])), offsets[num_kmers] i32.
// One table replaces the reference's counts[] + pointer table: a probe reads two adjacent words of one line.
//
// The default build is the stable two-level partition of index_part.hip.  The kernels in THIS file are the first, direct
// formulation (MECAT_IDX_BUILD=1: one global atomic per k-mer into the 268 MB table; ~100 GB of HBM traffic per walk, 230 ms at
// config 2), kept as the independent implementation tests/test_gpu_fullsize.py compares the default build with:
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

#include <chrono>
static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

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

int index_add_slots(mhip_ctx* c, mhip_index* idx) {
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
        const char* e = getenv("MECAT_IDX_BUILD");       // debug knob: 1 = direct atomic walks (the independent implementation the full-size test compares with), default = index_part.hip
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

int mhip_index_download_aux(mhip_ctx* c, const mhip_index* idx, uint16_t* slots, uint32_t* recs, int* cut_step) {
    HIPCHK(hipSetDevice(c->device));
    if (cut_step) *cut_step = idx->d_recs ? idx->cut_step : 0;
    if (slots && idx->num_kmers && idx->d_slots) HIPCHK(hipMemcpyAsync(slots, idx->d_slots, sizeof(uint16_t) * (size_t)idx->num_kmers, hipMemcpyDeviceToHost, c->stream));
    if (recs && idx->d_recs) HIPCHK(hipMemcpyAsync(recs, idx->d_recs, sizeof(uint4) * (size_t)NKMER, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return (recs && !idx->d_recs) ? 1 : 0;
}

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
