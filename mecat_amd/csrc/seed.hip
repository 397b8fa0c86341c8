// seed.hip — k-mer probe, DDF block-scoring seed filter and candidate selection.
//
// Replaces, for a batch of query reads and both strands of each:
//   extract_kmers      mecat2pw/pw_impl.cpp:83-97      query k-mers at stride 10
//   seeding            mecat2pw/pw_impl.cpp:241-286    bucket walk, per-2kb-segment seed lists (Back_List), index_score
//   insert_loc         mecat2pw/pw_impl.cpp:121-159    >40-seeds-per-segment overflow policy
//   find_location      mecat2pw/pw_impl.cpp:161-239    DDF vote inside a segment pair
//   get_candidates     mecat2pw/pw_impl.cpp:288-465    gate, subject lookup, self-hit scrub, neighbour sweeps, top-MAXC
//
// The reference keeps a 168-byte Back_List per 2 kb of reference per thread (126 MB at 1.5 Gbase) and updates it in
// hit order.  That state is order dependent (SURVEY.md §7 "hard parts" 1-3), so the GPU formulation is a replay:
//
//   seed_probe  (block / strand)  2-bit read -> 13-mers (reverse strand by index arithmetic) -> bucket (start, count)
//                                 from the starts[] table, hits per strand
//   seed_filter (block / strand)  relevance filter: one walk over the buckets counts hits per (hashed) 2 kb segment in
//                                 LDS; segments that can reach the index_score gate, +- the sweep reach, form the
//                                 relevance bitmap; every other hit (6 of 7 at config 2) is never expanded
//   seed_scan   (one block)       exclusive scan of kept hits per strand -> region of every strand in the batch arrays
//   seed_emit   (block / strand)  expand the kept hits: key = seg:21 | km:16 | off:11 written in (km, position) order,
//                                 i.e. in the order the reference visits the hits (64 buckets per step: count, scan, write)
//   seed_sort   (block / strand)  stable LSD radix sort on the seg bits only (7-8 bits per pass): ranking by
//                                 wave64 ballot multi-split, per-wave cursors in LDS, no atomics in the scatter loop.
//                                 Stability keeps (km, position) order inside a segment == the reference's visit order.
//   seed_build  (block / strand)  "recorded" events = first hit of each (segment, km) (pw_impl.cpp:265,282);
//                                 ballot/prefix compaction into per-segment seed lists; segments with > 40 recorded
//                                 events replay insert_loc one wave per segment (exact-diagonal fast path for the
//                                 read hitting itself); index_score incl. the left neighbour's score at the time of
//                                 the segment's last event; gated segments ordered by first-touch time
//   seed_cand   (wave / read)     sequential replay of get_candidates over the gated segments of F then R strand:
//                                 lanes evaluate the O(n^2) DDF votes, the sweeps and the list shifts in parallel
//
// FP: find_location is all f32, insert_loc divides in f32 and compares in f64, the sweeps are f64;
// built with -ffp-contract=off and hipcc's default correctly-rounded f32 division.
//
// Reads longer than 327 670 bases: the reference's seed numbers are `short` and wrap.  ent_seed() sign-extends the stored 16
// bits like the reference's loads do, the recording rule follows the wrapped `seednum` (seed_build, phase A), and such a strand always takes the
// kernel chain (seed_strand stops at K = 32 767): candidates equal the reference's also there (test_reads_beyond_the_short_seed_numbers).
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "common.h"

#define SEED_BLOCK 256
#define SEED_WAVES (SEED_BLOCK / WAVE)
#define KEY_OFF_BITS 11
#define KEY_KM_BITS 16
#define KEY_SEG_SHIFT (KEY_OFF_BITS + KEY_KM_BITS)   // 27
#define MAXC_LDS 1024                 // top-MAXC lists up to this size live in LDS; longer ones in the output array itself
#define MAXC_LIMIT (1 << 20)
#define FLT_BITS 15                 // relevance filter: 2^15 8-bit hit counters per strand, segment ids hashed by their low bits
#define FLT_M (1 << FLT_BITS)
#define REL_WORDS (FLT_M / 32)      // relevance bitmap over the same hashed ids

struct SeedArrays {
    // per batch
    const uint32_t* km_base;     // [ns]   first k-mer slot of the strand
    uint32_t* km_bstart;         // [sumK] bucket start in index offsets[]
    uint32_t* km_cnt;            // [sumK] bucket size
    uint32_t* strand_hits_all;   // [ns]   bucket hits of the strand (before the relevance filter)
    uint32_t* strand_hits;       // [ns]   hits kept (emitted, sorted, built)
    uint32_t* rel_bits;          // [ns * REL_WORDS] relevance bitmap of the strand (filtered strands only)
    int32_t* filtered;           // [ns]   1 = only relevant hits are kept, 0 = every hit is kept
    int32_t* fused;              // [ns]   1 = the strand went through seed_strand (its tables live in the fused arrays)
    uint32_t rel_mask;           // relevance bitmap bit of segment seg: ((seg & rel_mask) >> rel_shift): 2^15 - 1 and 0 for seed_filter,
    uint32_t rel_shift;          //   2^18 - 1 and 3 for seed_filter_wide
    uint64_t* hit_base;          // [ns + 1]
    uint64_t* keysA;             // [Htot]
    uint64_t* keysB;             // [Htot]
    uint32_t* ent;               // [Htot] recorded events: off << 16 | (uint16)(km + 1)
    uint32_t* ent_fin;           // [Htot] final 40-entry lists of overflowed segments (same indexing as ent)
    uint16_t* escore;            // [Htot] score after each recorded event (overflowed segments only)
    uint32_t* seg_id;            // [Htot] segment table, ascending seg id
    uint32_t* seg_start;         // [Htot] first recorded event of the segment in ent[]
    int32_t* seg_score;          // [Htot] live Back_List.score (bit 30 set: lists live in ent_fin)
    uint32_t* seg_kmlast;        // [Htot] km of the segment's last recorded event
    uint64_t* seg_tfirst;        // [Htot] first-touch time (km << 32 | position)
    uint32_t* gated;             // [Htot] segment-table indices passing the index_score gate, in first-touch order
    uint32_t* nseg;              // [ns]
    uint32_t* nrec;              // [ns]
    uint32_t* ngated;            // [ns]
};

#define OVF_FLAG 0x40000000
#define GATE_LDS 2048                // gated segments whose first-touch times fit the LDS stage of seed_build

__device__ __forceinline__ int kmers_of(int L) { return L < MHIP_KMER_SIZE ? 0 : (L - MHIP_KMER_SIZE) / BC + 1; }

// block-wide exclusive scan of one value per thread (SEED_BLOCK threads); returns the exclusive prefix, total in *total
__device__ __forceinline__ uint32_t block_excl_scan(uint32_t v, uint32_t* wtot /*[SEED_WAVES]*/, uint32_t* total) {
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    __syncthreads();   // protect wtot from the previous use
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < SEED_WAVES; ++w) {
        if (w < (int)(threadIdx.x >> 6)) base += wtot[w];
        tot += wtot[w];
    }
    *total = tot;
    return base + incl - v;
}

// Which reads a call works on: local index i -> read rid0 + (i / chunk) * cstride + i % chunk.  chunk == 1 is a plain stride
// (a contiguous range has cstride == 1); chunk > 1 is the multi-GPU shard of a grid cell by chunks of reads (SURVEY.md §8e:
// the reference hands out chunks of 500 reads, mecat2pw/pw_impl.cpp:612-621; rank r of P owns every P-th chunk).
struct ReadSel { int rid0, chunk, cstride; };
__host__ __device__ __forceinline__ int sel_rid(const ReadSel& s, int i) {
    return s.chunk == 1 ? s.rid0 + i * s.cstride : s.rid0 + (i / s.chunk) * s.cstride + i % s.chunk;
}

// ------------------------------------------------------------------------------------------------ probe
// `recs` (index.hip: idx_cut_records) replaces starts[] when reference and query volume are the same one (a diagonal grid cell):
// the bucket of a query k-mer is cut at the first of seven fixed positions behind the read's own copy + 5 segments.  What lies
// behind belongs to reads with higher ids: get_candidates drops their candidates at `sid > read_id` (pw_impl.cpp:370) before it writes
// anything, the sweeps of a kept candidate stay within 3 segments of its subject read's end (num2 <= send - loc_list, :399-403,
// 424-438), index_score looks at the left neighbour only (:270-280) — so those hits cannot reach the output, and half of the
// bucket walk of a diagonal cell is not done.  counters[1] still counts every bucket hit (the H of SURVEY.md §8d).
__global__ __launch_bounds__(SEED_BLOCK) void seed_probe(const uint32_t* __restrict__ pac, const mhip_offset_t* __restrict__ roffs,
                                                         ReadSel sel, int ib, const uint32_t* __restrict__ starts, SeedArrays A,
                                                         unsigned long long* __restrict__ counters, const uint4* __restrict__ recs, int cut_step) {
    __shared__ uint32_t wtot[SEED_WAVES];
    const int s = blockIdx.x;
    const int rid = sel_rid(sel, ib + (s >> 1));
    const bool rev = s & 1;
    const int off = roffs[rid].offset, L = roffs[rid].size;
    const int K = kmers_of(L);
    const uint32_t kb = A.km_base[s];
    // byte of the record's y:z that holds the bucket length for this read: cut t = ceil((end of the read + 5 segments) / step)
    int cut_byte = 7;
    if (recs) {
        const long long need = (long long)off + L + 1 + 5 * ZV;
        const long long t = (need + cut_step - 1) / cut_step;
        if (t <= 7) cut_byte = (int)t - 1;
    }
    uint32_t run = 0, run_all = 0;
    for (int t0 = 0; t0 < K; t0 += SEED_BLOCK) {
        int km = t0 + threadIdx.x;
        uint32_t cnt = 0, all = 0, bs = 0;
        if (km < K) {
            uint32_t id;
            if (!rev) id = pac_kmer(pac, (int64_t)off + (int64_t)km * BC);
            else id = kmer_revcomp(pac_kmer(pac, (int64_t)off + L - MHIP_KMER_SIZE - (int64_t)km * BC));
            if (recs) {
                const uint4 r = recs[id];
                const unsigned long long yz = ((unsigned long long)r.z << 32) | r.y;
                bs = r.x;
                all = r.z >> 24;
                cnt = (uint32_t)(yz >> (8 * cut_byte)) & 0xFFu;
            } else {
                bs = starts[id];
                all = cnt = starts[id + 1] - bs;
            }
        }
        uint32_t tot;
        if (recs) {
            (void)block_excl_scan(cnt | (all << 16), wtot, &tot);      // both sums in one scan: <= 256 x 255 each
            run += tot & 0xFFFFu;
            run_all += tot >> 16;
        } else {
            (void)block_excl_scan(cnt, wtot, &tot);
            run += tot;
            run_all += tot;
        }
        if (km < K) {
            A.km_bstart[kb + km] = bs;
            A.km_cnt[kb + km] = cnt;
        }
    }
    if (threadIdx.x == 0) {
        A.strand_hits_all[s] = run;
        atomicAdd(&counters[0], (unsigned long long)K);
        atomicAdd(&counters[1], (unsigned long long)run_all);
        if (recs) atomicAdd(&counters[15], (unsigned long long)run);      // debug slot 15: bucket hits walked with the cuts on
    }
}

// exclusive scan of strand_hits[ns] -> hit_base[ns + 1] (single block, sequential over tiles)
__global__ __launch_bounds__(1024) void seed_scan(const uint32_t* __restrict__ hits, int ns, uint64_t* __restrict__ base) {
    __shared__ uint64_t wtot[16];
    __shared__ uint64_t carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int t0 = 0; t0 < ns; t0 += 1024) {
        int i = t0 + threadIdx.x;
        uint64_t v = i < ns ? hits[i] : 0;
        uint64_t incl = v;
        for (int o = 1; o < 64; o <<= 1) {
            uint64_t n = __shfl_up(incl, o);
            if (lane_id() >= o) incl += n;
        }
        if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
        __syncthreads();
        uint64_t b = carry;
        for (int w = 0; w < (int)(threadIdx.x >> 6); ++w) b += wtot[w];
        if (i < ns) base[i] = b + incl - v;
        __syncthreads();
        if (threadIdx.x == 1023) carry = b + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) base[ns] = carry;
}

// ------------------------------------------------------------------------------------------------ hit iteration
// 16 lanes per query k-mer: a bucket (<= 128 ascending positions, ~22 on average) is read in 64-byte pieces, lane `sub`
// takes entries sub, sub + 16, ...  `f(km, r, pos, valid)` is called by all 16 lanes of the group together (group-uniform
// control flow), so it may use the group's 16 bits of a wave ballot.
// Four buckets per group are in flight at once (their (start, size) and first two 64-byte pieces are loaded before any is
// consumed): the walk is a gather of ~88-byte runs, so throughput comes from outstanding loads, not from arithmetic.
// `begin(slot, km)`, `f(slot, km, r, pos, valid)`, `end(slot, km)`; slot 0..3 is a compile-time index for per-bucket state.
template <typename T, typename B, typename F, typename E>
__device__ __forceinline__ void for_each_hit16(const SeedArrays& A, const T* __restrict__ offsets, const uint32_t kb, const int K,
                                               B begin, F f, E end) {
    const int g = threadIdx.x >> 4, sub = threadIdx.x & 15;
    constexpr int G = SEED_BLOCK / 16;
    for (int km0 = g; km0 < K; km0 += 4 * G) {
        uint32_t bs[4], cnt[4], p0[4], p1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int km = km0 + q * G;
            bs[q] = km < K ? A.km_bstart[kb + km] : 0u;
            cnt[q] = km < K ? A.km_cnt[kb + km] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            p0[q] = (uint32_t)sub < cnt[q] ? (uint32_t)offsets[bs[q] + sub] : 0u;
            p1[q] = (uint32_t)sub + 16u < cnt[q] ? (uint32_t)offsets[bs[q] + 16 + sub] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int km = km0 + q * G;
            if (km < K) {
                begin(q, km);
                if (cnt[q] > 0) f(q, km, (uint32_t)sub, p0[q], (uint32_t)sub < cnt[q]);
                if (cnt[q] > 16) f(q, km, (uint32_t)sub + 16u, p1[q], (uint32_t)sub + 16u < cnt[q]);
                for (uint32_t r0 = 32; r0 < cnt[q]; r0 += 16) {
                    const uint32_t r = r0 + sub;
                    const bool valid = r < cnt[q];
                    f(q, km, r, valid ? (uint32_t)offsets[bs[q] + r] : 0u, valid);
                }
                end(q, km);
            }
        }
    }
}
__device__ __forceinline__ uint32_t group_bits(unsigned long long ballot) { return (uint32_t)(ballot >> (threadIdx.x & 48)) & 0xFFFFu; }

// sweeps of get_candidates reach ceil(num / ZV) segments, num <= read length + 12 (pw_impl.cpp:388-425); +1 for seg - 1
__device__ __forceinline__ int sweep_reach(int L) { return (L + MHIP_KMER_SIZE - 1 + ZV - 1) / ZV + 1; }

__device__ __forceinline__ bool rel_test(const uint32_t* rel, uint32_t seg) {
    const uint32_t e = seg & (FLT_M - 1);
    return (rel[e >> 5] >> (e & 31u)) & 1u;
}
// the same with the bitmap's geometry as data (seed_emit: bitmaps of seed_filter and of seed_filter_wide)
__device__ __forceinline__ bool rel_test(const uint32_t* rel, uint32_t seg, uint32_t mask, uint32_t shift) {
    const uint32_t e = (seg & mask) >> shift;
    return (rel[e >> 5] >> (e & 31u)) & 1u;
}

// ------------------------------------------------------------------------------------------------ relevance filter
// Most bucket hits are random 13-mer matches that land alone in their 2 kb segment and can never matter:
// get_candidates only looks at segments whose index_score (own + left neighbour's seed count) reaches 2 * min_kmer_match
// (pw_impl.cpp:309) and, from those, at most ceil((L + 12) / 2000) segments to either side (:405-438).  With h(e) = bucket
// hits whose segment id is e modulo 2^15 (an upper bound of the seed count of every segment hashing to e), slot e is HOT
// when h(e) > 0 and h(e-1) + h(e) or h(e) + h(e+1) reaches the gate; a hit is kept when its slot lies within `sweep_reach`
// slots of a hot one.  Dropped hits touch no state that is ever read, so the result is unchanged: the filter only errs
// towards keeping (collisions inflate h and spread relevance).  One walk over the buckets fills the byte counters in LDS;
// hot slots, the relevance bitmap and the exact number of kept hits (sum of h over relevant slots: the strand's room in
// the key arrays) come from passes over the 32 K-entry table.  A counter that would wrap (>= 256 hits in one slot: repeats)
// turns the filter off for the strand.
__global__ __launch_bounds__(SEED_BLOCK) void seed_filter(const mhip_offset_t* __restrict__ roffs, ReadSel sel, int ib,
                                                          const uint16_t* __restrict__ slots, SeedArrays A, int gate, int enable) {
    static_assert(ZV == 2000 && FLT_M == (1 << 15), "idx_slots (index.hip) computes (position / 2000) mod 2^15");
    __shared__ uint32_t cnt[FLT_M / 4];          // 32 KB
    __shared__ uint32_t rel[REL_WORDS];          // 4 KB
    __shared__ uint32_t wtot[SEED_WAVES];
    __shared__ uint32_t s_wrap;
    const int s = blockIdx.x;
    const int rid = sel_rid(sel, ib + (s >> 1));
    const int L = roffs[rid].size;
    const int K = kmers_of(L);
    const uint32_t kb = A.km_base[s];
    const uint32_t Hall = A.strand_hits_all[s];
    if (A.fused[s]) {                 // done by seed_strand: nothing for the kernels of this path
        if (threadIdx.x == 0) { A.strand_hits[s] = 0; A.filtered[s] = 0; }
        return;
    }
    bool filtered = enable && Hall > 0;
    uint32_t kept = Hall;
    if (filtered) {
        for (int i = threadIdx.x; i < FLT_M / 4; i += SEED_BLOCK) cnt[i] = 0;
        for (int i = threadIdx.x; i < REL_WORDS; i += SEED_BLOCK) rel[i] = 0;
        if (threadIdx.x == 0) s_wrap = 0;
        __syncthreads();
        auto nop = [](int, int) {};
        for_each_hit16(A, slots, kb, K, nop, [&](int, int, uint32_t, uint32_t e, bool valid) {      // e = the hit's table slot
            if (valid) {
                const uint32_t sh = (e & 3u) * 8u;
                const uint32_t old = atomicAdd(&cnt[e >> 2], 1u << sh);
                if (((old >> sh) & 255u) == 255u) s_wrap = 1;
            }
        }, nop);
        __syncthreads();
        if (s_wrap) filtered = false;
    }
    if (filtered) {
        const uint8_t* cb = (const uint8_t*)cnt;
        const uint32_t reach = (uint32_t)min(sweep_reach(L), FLT_M / 2 - 1);
        for (int i = 0; i < FLT_M / SEED_BLOCK; ++i) {
            const uint32_t e = (uint32_t)i * SEED_BLOCK + threadIdx.x;
            const int c = cb[e];
            if (c > 0 && (c + (int)cb[(e + 1) & (FLT_M - 1)] >= gate || c + (int)cb[(e - 1) & (FLT_M - 1)] >= gate)) {
                uint32_t lo = (e - reach) & (FLT_M - 1), left = 2 * reach + 1;      // bits lo .. lo + left - 1, circular
                while (left > 0) {
                    const uint32_t b = lo & 31u, take = min(32u - b, left);
                    const uint32_t m = (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << b;
                    atomicOr(&rel[lo >> 5], m);
                    lo = (lo + take) & (FLT_M - 1);
                    left -= take;
                }
            }
        }
        __syncthreads();
        uint32_t mine = 0;
        for (int i = 0; i < FLT_M / SEED_BLOCK; ++i) {
            const uint32_t e = (uint32_t)i * SEED_BLOCK + threadIdx.x;
            if (rel_test(rel, e)) mine += cb[e];
        }
        uint32_t tot;
        (void)block_excl_scan(mine, wtot, &tot);
        kept = tot;
        for (int i = threadIdx.x; i < REL_WORDS; i += SEED_BLOCK) A.rel_bits[(size_t)s * REL_WORDS + i] = rel[i];
    }
    if (threadIdx.x == 0) { A.strand_hits[s] = kept; A.filtered[s] = filtered ? 1 : 0; }
}

#ifdef WF_PROF
// development build (make wfprof): s_memrealtime ticks per section of seed_filter_wide [0..7] and seed_emit [8..15], summed over strands
__device__ unsigned long long g_wfprof[16];
#define WF_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&g_wfprof[i], t_ - t_prev); t_prev = t_; } } while (0)
#define WF_T0 unsigned long long t_prev = wall_clock64(); (void)t_prev
#else
#define WF_MARK(i) do { } while (0)
#define WF_T0
#endif
// ------------------------------------------------------------------------------------------------ emit
// keys of the kept hits in (km, position) order = the order the reference visits them: key = seg:21 | km:16 | off:11.
// The strand's k-mers are taken 64 at a time (16 lanes per bucket, four buckets per group as in for_each_hit16): count the
// kept hits per bucket, scan the 64 counts, write.  The first 32 entries of a bucket stay in registers between the two
// steps; longer buckets are read again (from cache).
#ifndef EMIT_NP
#define EMIT_NP 4                     // 16-entry pieces of a bucket loaded together (the rest, if any, one by one)
#endif
__global__ __launch_bounds__(SEED_BLOCK) void seed_emit(const mhip_offset_t* __restrict__ roffs, ReadSel sel, int ib,
                                                        const int32_t* __restrict__ offsets, SeedArrays A) {
    __shared__ uint32_t rel[REL_WORDS];
    __shared__ uint32_t ccnt[64];
    __shared__ uint32_t coff[64];
    __shared__ uint32_t s_tot;
    const int s = blockIdx.x;
    const int rid = sel_rid(sel, ib + (s >> 1));
    const int L = roffs[rid].size;
    const int K = kmers_of(L);
    const uint32_t kb = A.km_base[s];
    if (A.strand_hits[s] == 0) return;
    WF_T0;
    uint64_t* __restrict__ out = A.keysA + A.hit_base[s];
    const bool flt = A.filtered[s] != 0;
    if (flt) {
        for (int i = threadIdx.x; i < REL_WORDS; i += SEED_BLOCK) rel[i] = A.rel_bits[(size_t)s * REL_WORDS + i];
    }
    __syncthreads();
    WF_MARK(8);
    const int g = threadIdx.x >> 4;
    const uint32_t sub = threadIdx.x & 15;
    const uint32_t below = (1u << sub) - 1u;
    uint32_t run = 0;
    // EMIT_NP pieces of 16 entries of every bucket are requested together, before any is used: a piece that is loaded when the one in
    // front of it has been consumed costs the group a memory latency of its own, and at nanopore volume sizes (32 entries per bucket on
    // average, many at the cap of 128) those serial loads — in a wave, whenever one of its four groups has a long bucket — were the
    // kernel's time.  The verdict of the count pass on each entry of those pieces stays in a register for the write pass.
    for (int c0 = 0; c0 < K; c0 += 64) {
        uint32_t bs[4], cn[4], pc[4][EMIT_NP];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int km = c0 + q * 16 + g;
            bs[q] = km < K ? A.km_bstart[kb + km] : 0u;
            cn[q] = km < K ? A.km_cnt[kb + km] : 0u;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int p = 0; p < EMIT_NP; ++p) pc[q][p] = sub + 16u * p < cn[q] ? (uint32_t)offsets[bs[q] + 16 * p + sub] : 0u;
        // count
        uint32_t keep = 0;          // bit EMIT_NP q + p: this lane's entry of piece p of bucket q is kept
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            uint32_t kept = 0;
#pragma unroll
            for (int p = 0; p < EMIT_NP; ++p)
                if (16u * p < cn[q]) {
                    const bool k1 = sub + 16u * p < cn[q] && (!flt || rel_test(rel, pc[q][p] / ZV, A.rel_mask, A.rel_shift));
                    kept += __popc(group_bits(__ballot(k1)));
                    keep |= k1 ? 1u << (EMIT_NP * q + p) : 0u;
                }
            for (uint32_t r0 = 16 * EMIT_NP; r0 < cn[q]; r0 += 16) {
                const uint32_t r = r0 + sub;
                const bool valid = r < cn[q];
                const uint32_t pos = valid ? (uint32_t)offsets[bs[q] + r] : 0u;
                kept += __popc(group_bits(__ballot(valid && (!flt || rel_test(rel, pos / ZV, A.rel_mask, A.rel_shift)))));
            }
            if (sub == 0) ccnt[q * 16 + g] = kept;
        }
        __syncthreads();
        WF_MARK(9);
        if (threadIdx.x < 64) {
            const uint32_t v = ccnt[threadIdx.x];
            uint32_t incl = v;
            for (int o = 1; o < 64; o <<= 1) {
                const uint32_t nb = __shfl_up(incl, o);
                if ((int)threadIdx.x >= o) incl += nb;
            }
            coff[threadIdx.x] = incl - v;
            if (threadIdx.x == 63) s_tot = incl;
        }
        __syncthreads();
        WF_MARK(10);
        // write
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int km = c0 + q * 16 + g;
            uint32_t at = run + coff[q * 16 + g];
            auto put = [&](uint32_t pos, bool k1) {
                const uint32_t seg = pos / ZV, so = pos - seg * ZV;
                const uint32_t bits = group_bits(__ballot(k1));
                if (k1) out[at + __popc(bits & below)] = ((uint64_t)seg << KEY_SEG_SHIFT) | ((uint64_t)km << KEY_OFF_BITS) | so;
                at += __popc(bits);
            };
#pragma unroll
            for (int p = 0; p < EMIT_NP; ++p)
                if (16u * p < cn[q]) put(pc[q][p], (keep >> (EMIT_NP * q + p)) & 1u);
            for (uint32_t r0 = 16 * EMIT_NP; r0 < cn[q]; r0 += 16) {
                const uint32_t r = r0 + sub;
                const bool valid = r < cn[q];
                const uint32_t pos = valid ? (uint32_t)offsets[bs[q] + r] : 0u;
                put(pos, valid && (!flt || rel_test(rel, pos / ZV, A.rel_mask, A.rel_shift)));
            }
        }
        WF_MARK(11);
        run += s_tot;
    }
    // the number of keys written: exact, where seed_filter_wide's strand_hits was the room it asked for (an upper bound)
    if (threadIdx.x == 0) A.strand_hits[s] = run;
}

// ------------------------------------------------------------------------------------------------ sort (one LSD pass)
// digit = (key >> shift) & (2^bits - 1), bits <= SORT_MAXBITS (two passes cover the 20-21 segment bits of a volume).  Wave w
// of the block owns the w-th quarter of the strand's keys.
#define SORT_MAXBITS 11
#define SORT_BINS (1 << SORT_MAXBITS)
__global__ __launch_bounds__(SEED_BLOCK) void seed_sort_pass(SeedArrays A, int from_b, int shift, int bits) {
    __shared__ uint32_t hist[SEED_WAVES][SORT_BINS];      // 32 KB
    __shared__ uint32_t wtot[SEED_WAVES];
    const int s = blockIdx.x;
    const uint32_t H = A.strand_hits[s];
    if (H == 0) return;
    const uint64_t hb = A.hit_base[s];
    const uint64_t* __restrict__ src = (from_b ? A.keysB : A.keysA) + hb;
    uint64_t* __restrict__ dst = (from_b ? A.keysA : A.keysB) + hb;
    const int w = threadIdx.x >> 6, lane = lane_id();
    const uint32_t mask = (1u << bits) - 1u;
    const uint32_t chunk = (((H + SEED_WAVES - 1) / SEED_WAVES) + 63u) & ~63u;
    const uint32_t lo = min(H, w * chunk), hi = min(H, lo + chunk);
    const uint32_t nbins = 1u << bits;
    for (uint32_t i = threadIdx.x; i < nbins; i += SEED_BLOCK)
#pragma unroll
        for (int q = 0; q < SEED_WAVES; ++q) hist[q][i] = 0;
    __syncthreads();
    for (uint32_t i = lo + lane; i < hi; i += 64) atomicAdd(&hist[w][(uint32_t)(src[i] >> shift) & mask], 1u);
    __syncthreads();
    {
        // bases in (bin major, wave minor) order, SEED_BLOCK bins per round
        uint32_t carry = 0;
        for (uint32_t b0 = 0; b0 < nbins; b0 += SEED_BLOCK) {
            const uint32_t bin = b0 + threadIdx.x;
            const bool in = bin < nbins;
            uint32_t c[SEED_WAVES], tot = 0;
#pragma unroll
            for (int q = 0; q < SEED_WAVES; ++q) { c[q] = in ? hist[q][bin] : 0u; tot += c[q]; }
            uint32_t all;
            uint32_t ex = carry + block_excl_scan(tot, wtot, &all);
            if (in) {
#pragma unroll
                for (int q = 0; q < SEED_WAVES; ++q) { hist[q][bin] = ex; ex += c[q]; }
            }
            carry += all;
        }
    }
    __syncthreads();
    volatile uint32_t* cur = hist[w];
    const uint64_t lt = (1ull << lane) - 1ull;
    for (uint32_t i0 = lo; i0 < hi; i0 += 64) {
        uint32_t i = i0 + lane;
        bool valid = i < hi;
        uint64_t key = valid ? src[i] : 0;
        uint32_t d = (uint32_t)(key >> shift) & mask;
        uint64_t peers = __ballot(valid);
        for (int b = 0; b < bits; ++b) {
            uint64_t m = __ballot((d >> b) & 1u);
            peers &= ((d >> b) & 1u) ? m : ~m;
        }
        uint32_t rank = __popcll(peers & lt);
        uint32_t base = valid ? cur[d] : 0;
        if (valid) dst[base + rank] = key;
        if (valid && rank == 0) cur[d] = base + (uint32_t)__popcll(peers);
    }
}

// ------------------------------------------------------------------------------------------------ build
__device__ __forceinline__ uint32_t key_seg(uint64_t k) { return (uint32_t)(k >> KEY_SEG_SHIFT); }
__device__ __forceinline__ uint32_t key_km(uint64_t k) { return (uint32_t)(k >> KEY_OFF_BITS) & 0xFFFFu; }
__device__ __forceinline__ uint32_t key_off(uint64_t k) { return (uint32_t)k & 0x7FFu; }
__device__ __forceinline__ int ent_loc(uint32_t e) { return (int)(e >> 16); }
__device__ __forceinline__ int ent_seed(uint32_t e) { return (int)(int16_t)(e & 0xFFFFu); }

// f32 divide, f64 compare: the DDF test of insert_loc (pw_impl.cpp:135)
__device__ __forceinline__ bool ddf_insert(int dloc, int dseed, double cutoff) {
    float r = (float)dloc / ((float)dseed * 10.0f);
    return fabs((double)r - 1.0) < cutoff;
}
// all-f32 DDF test of find_location (pw_impl.cpp:165,196,222)
__device__ __forceinline__ bool ddf_find(int dloc, int dseed, double cutoff) {
    float r = (float)dloc / ((float)dseed * 10.0f) - 1.0f;
    return (double)fabsf(r) < cutoff;
}

// ddf_find for cutoff == 0.25 (the reference's value in every mode) without the division.  dloc, dseed > 0 are small
// integers, so x = dloc / (10 dseed) is never within a rounding error of 0.75 or 1.25 unless it equals them
// (|x - 1.25| >= 1 / (40 dseed) >= 7.6e-7 for dseed < 32768, half an f32 ulp there is 6e-8): q = RN(x) satisfies
// 0.75 < q < 1.25 exactly when x does; q - 1 is exact for q in [0.5, 2] (Sterbenz) and |RN(q - 1)| >= 0.5 outside.
__device__ __forceinline__ bool ddf_find_quarter(int dloc, int dseed) {        // |dseed| < 2^15: 24-bit multiplies
    return (int)(2 * dloc > __mul24(15, dseed)) & (int)(2 * dloc < __mul24(25, dseed));
}

// insert_loc replay for one overflowed segment by one wave.  Events e = 40.. c-1 (0-based) arrive one by one.
// lane i (< 40) holds list entry i.  Writes the final 40 entries to fin[] and the score after each event to esc[].
//
// While the 40-entry list is one exact chain (strictly increasing seed numbers, loc advancing by BC per seed from entry to
// entry) an event that continues the chain after the last entry passes every pair test: the minimum is SM and the event just
// replaces the last entry (pw_impl.cpp:121-159 with every score equal).  Runs of such events are skipped in one step — a read
// against its own copy in the volume is the common overflow and is such a chain from end to end, bar the odd random hit,
// which is looked at, thrown out again, and leaves the list the chain it was.
__device__ void replay_overflow(const uint32_t* __restrict__ ev, int c, uint32_t* __restrict__ fin, uint16_t* __restrict__ esc,
                                double cutoff, int* score_out) {
    const int lane = lane_id();
    int loc = 0, seed = 0;
    if (lane < SM) { const uint32_t e0 = ev[lane]; loc = ent_loc(e0); seed = ent_seed(e0); }
    if (lane < SM) esc[lane] = (uint16_t)(lane + 1);
    int score = SM;
    int e = SM;
    auto is_chain = [&]() {
        const int nl = __shfl_down(loc, 1), ns = __shfl_down(seed, 1);
        return (bool)__all(lane >= SM - 1 || (ns - seed > 0 && nl - loc == (ns - seed) * BC));
    };
    bool chain = is_chain();
    while (e < c) {
        if (chain) {
            const int lloc = __builtin_amdgcn_readlane(loc, SM - 1), lseed = __builtin_amdgcn_readlane(seed, SM - 1);
            int nb = c;                              // first event that does not continue the chain
            for (int i0 = e; i0 < c && nb == c; i0 += 64) {
                const int i = i0 + lane;
                bool bad = false;
                if (i < c) {
                    const uint32_t b = ev[i];
                    int pl = lloc, ps = lseed;
                    if (i > e) { const uint32_t a = ev[i - 1]; pl = ent_loc(a); ps = ent_seed(a); }
                    const int ds = ent_seed(b) - ps, dl = ent_loc(b) - pl;
                    bad = !(ds > 0 && dl == ds * BC);
                }
                const unsigned long long bm = __ballot(bad);
                if (bm) nb = i0 + __builtin_ctzll(bm);
            }
            if (nb > e) {
                for (int x = e + lane; x < nb; x += 64) esc[x] = (uint16_t)(score + (x - e) + 1);
                score += nb - e;
                if (lane == SM - 1) { const uint32_t le = ev[nb - 1]; loc = ent_loc(le); seed = ent_seed(le); }
                e = nb;
                if (e >= c) break;
            }
        }
        const uint32_t ne = ev[e];
        const int nloc = ent_loc(ne), nseed = ent_seed(ne);
        ++score;   // loc = ++spr->score (pw_impl.cpp:267)
        // element 40 of the 41-entry working list is the new seed
        const int myloc = lane < SM ? loc : nloc, myseed = lane < SM ? seed : nseed;
        int sc = 0;
        for (int i = 0; i < SM; ++i) {
            const int li = __builtin_amdgcn_readlane(myloc, i), si = __builtin_amdgcn_readlane(myseed, i);      // entry i lives in lane i
            const bool pass = lane > i && lane <= SM && myseed - si > 0 && myloc - li > 0 && ddf_insert(myloc - li, myseed - si, cutoff);
            const uint64_t b = __ballot(pass);
            sc += pass ? 1 : 0;
            if (lane == i) sc += __popcll(b);
        }
        const int v = lane <= SM ? sc : 0x7fffffff;
        int m = v;
        for (int o = 32; o > 0; o >>= 1) m = min(m, __shfl_xor(m, o));
        const int minval = m;
        const int mini = __ffsll((unsigned long long)__ballot(v == m)) - 1;   // first index holding the minimum
        if (minval == SM) {
            if (lane == SM - 1) { loc = nloc; seed = nseed; }
        } else if (minval < SM && mini < SM) {
            // delete entry mini, shift left, append the new one
            const int sl = __shfl_down(myloc, 1), ss = __shfl_down(myseed, 1);
            if (lane >= mini && lane < SM) { loc = sl; seed = ss; }
            --score;
        }
        if (lane == 0) esc[e] = (uint16_t)score;
        ++e;
        chain = is_chain();
    }
    if (lane < SM) fin[lane] = ((uint32_t)loc << 16) | ((uint32_t)seed & 0xFFFFu);
    *score_out = score;
}

__device__ __forceinline__ int read_id_lookup(const mhip_offset_t* __restrict__ a, int n, const uint32_t* __restrict__ blk, int offset);

// what the segment builders need to know about the reference volume: a gated segment whose every possible subject read has a
// higher id than the query read is not listed (get_candidates would vote on it and then drop it at `sid > read_id`,
// pw_impl.cpp:370, before anything is written: half of the gated segments of a diagonal grid cell)
struct RefReads { const mhip_offset_t* offs; const uint32_t* blk; int nreads, start_id, reads_start_id, enable, same_volume; };
// every location find_location can return for segment `seg` (with its left neighbour) lies at or behind the start of segment
// seg - 1, and the read lookup is monotone in the position (a pad base belongs to the read before it).  own_end = the first position
// behind the query read's own copy and its pad, when the query volume is the reference volume: no lookup needed then.
__device__ __forceinline__ bool subjects_all_higher(const RefReads& R, uint32_t seg, int rid, int own_end) {
    const int lo = (int)((seg > 0 ? seg - 1u : 0u) * (uint32_t)ZV);
    if (R.same_volume) return lo >= own_end;
    return read_id_lookup(R.offs, R.nreads, R.blk, lo) + R.start_id > rid + R.reads_start_id;
}

__global__ __launch_bounds__(SEED_BLOCK) void seed_build(SeedArrays A, int sorted_in_b, int min_kmer_match, double cutoff, ReadSel sel, int ib,
                                                         RefReads R) {
    __shared__ uint32_t wtot[SEED_WAVES];
    __shared__ uint32_t s_cnt[4];          // 0: overflow count, 1: gated count
    __shared__ uint64_t s_tf[GATE_LDS];    // first-touch times of the gated segments (phase E)
    const int s = blockIdx.x;
    const uint32_t H = A.strand_hits[s];
    const int own_rid = sel_rid(sel, ib + (s >> 1));
    const int own_end = R.same_volume ? R.offs[own_rid].offset + R.offs[own_rid].size + 1 : 0;
    const uint64_t hb = A.hit_base[s];
    if (H == 0) {
        if (threadIdx.x == 0) { A.nseg[s] = 0; A.nrec[s] = 0; A.ngated[s] = 0; }
        return;
    }
    const uint64_t* __restrict__ S = (sorted_in_b ? A.keysB : A.keysA) + hb;
    uint32_t* ovf_list = (uint32_t*)((sorted_in_b ? A.keysA : A.keysB) + hb);   // free sort buffer: overflow work list
    uint64_t* gate_tmp = (sorted_in_b ? A.keysA : A.keysB) + hb;                // later: unsorted gated keys
    uint32_t* ent = A.ent + hb;
    uint32_t* fin = A.ent_fin + hb;
    uint16_t* esc = A.escore + hb;
    uint32_t* seg_id = A.seg_id + hb;
    uint32_t* seg_start = A.seg_start + hb;
    int32_t* seg_score = A.seg_score + hb;
    uint32_t* seg_kmlast = A.seg_kmlast + hb;
    uint64_t* seg_tfirst = A.seg_tfirst + hb;
    uint32_t* gated = A.gated + hb;

    // ---- phase A: recorded events and segment heads
    uint32_t rec_run = 0, seg_run = 0;
    for (uint32_t t0 = 0; t0 < H; t0 += SEED_BLOCK) {
        uint32_t i = t0 + threadIdx.x;
        bool in = i < H;
        uint64_t key = in ? S[i] : 0, prev = (in && i > 0) ? S[i - 1] : 0;
        bool head = in && (i == 0 || key_seg(key) != key_seg(prev));
        // a hit is recorded when the segment is new or its last seed number is below this k-mer's (pw_impl.cpp:265): the first hit of each
        // (segment, km) — until the reference's `short seednum` wraps: from seed number 32 768 on (reads beyond 327 670 bases) it is
        // negative, below every k-mer number, and every further hit of the segment is recorded
        bool rec = in && (head || key_km(key) != key_km(prev) || key_km(prev) >= 32767u);
        // one scan for both counts: recorded events in the low half, segment heads in the high half (<= 256 each per round)
        uint32_t both;
        const uint32_t ex = block_excl_scan((rec ? 1u : 0u) | (head ? 0x10000u : 0u), wtot, &both);
        const uint32_t rtot = both & 0xFFFFu, stot = both >> 16;
        uint32_t rpos = rec_run + (ex & 0xFFFFu);
        uint32_t spos = seg_run + (ex >> 16);
        if (rec) ent[rpos] = (key_off(key) << 16) | ((key_km(key) + 1u) & 0xFFFFu);
        if (head) {
            seg_id[spos] = key_seg(key);
            seg_start[spos] = rpos;
            seg_tfirst[spos] = ((uint64_t)key_km(key) << 32) | (uint64_t)(key_seg(key) * (uint32_t)ZV + key_off(key));
            if (spos > 0) seg_kmlast[spos - 1] = key_km(prev);
        }
        rec_run += rtot;
        seg_run += stot;
    }
    const uint32_t nrec = rec_run, nseg = seg_run;
    if (threadIdx.x == 0) {
        seg_kmlast[nseg - 1] = key_km(S[H - 1]);
        A.nseg[s] = nseg;
        A.nrec[s] = nrec;
        s_cnt[0] = 0;
        s_cnt[1] = 0;
    }
    __syncthreads();

    // ---- phase B: scores; collect overflowed segments
    for (uint32_t g = threadIdx.x; g < nseg; g += SEED_BLOCK) {
        uint32_t st = seg_start[g], en = (g + 1 < nseg) ? seg_start[g + 1] : nrec;
        uint32_t c = en - st;
        if (c > SM) { uint32_t k = atomicAdd(&s_cnt[0], 1u); ovf_list[k] = g; seg_score[g] = OVF_FLAG; }
        else seg_score[g] = (int32_t)c;
    }
    __syncthreads();

    // ---- phase C: replay insert_loc, one wave per overflowed segment
    {
        const uint32_t novf = s_cnt[0];
        for (uint32_t k = threadIdx.x >> 6; k < novf; k += SEED_WAVES) {
            uint32_t g = ovf_list[k];
            uint32_t st = seg_start[g], en = (g + 1 < nseg) ? seg_start[g + 1] : nrec;
            int sc;
            replay_overflow(ent + st, (int)(en - st), fin + st, esc + st, cutoff, &sc);
            if (lane_id() == 0) seg_score[g] = sc | OVF_FLAG;
        }
    }
    __syncthreads();

    // ---- phase D: index_score = own score + left neighbour's score at the time of the last event (pw_impl.cpp:270-280)
    for (uint32_t g = threadIdx.x; g < nseg; g += SEED_BLOCK) {
        int own = seg_score[g] & ~OVF_FLAG;
        int s_k = own;
        uint32_t sid = seg_id[g];
        if (g > 0 && sid > 0 && seg_id[g - 1] == sid - 1) {
            // events of the left neighbour with km <= km_last happened before this segment's last event
            uint32_t st = seg_start[g - 1], en = seg_start[g];
            uint32_t t = seg_kmlast[g] + 1u;     // compare on km + 1 as stored
            uint32_t lo = st, hi = en;           // first event with (km + 1) > t
            while (lo < hi) {
                uint32_t mid = (lo + hi) >> 1;
                if ((ent[mid] & 0xFFFFu) <= t) lo = mid + 1; else hi = mid;
            }
            int left = 0;
            if (lo > st) left = (seg_score[g - 1] & OVF_FLAG) ? (int)esc[lo - 1] : (int)(lo - st);
            s_k += left;
        }
        if ((int)(int16_t)s_k >= 2 * min_kmer_match && !(R.enable && subjects_all_higher(R, sid, own_rid, own_end))) {
            uint32_t k = atomicAdd(&s_cnt[1], 1u);
            gate_tmp[k] = (uint64_t)g;   // index only; ordered below by seg_tfirst
        }
    }
    __syncthreads();

    // ---- phase E: order gated segments by first-touch time (rank sort; first-touch times are unique)
    {
        const uint32_t ng = s_cnt[1];
        if (ng <= GATE_LDS) {       // the usual case (~250 gated segments): times staged in LDS, ranks from broadcast reads
            for (uint32_t a = threadIdx.x; a < ng; a += SEED_BLOCK) s_tf[a] = seg_tfirst[(uint32_t)gate_tmp[a]];
            __syncthreads();
            for (uint32_t a = threadIdx.x; a < ng; a += SEED_BLOCK) {
                const uint64_t ta = s_tf[a];
                uint32_t rank = 0;
                for (uint32_t b = 0; b < ng; ++b) rank += s_tf[b] < ta ? 1u : 0u;
                gated[rank] = (uint32_t)gate_tmp[a];
            }
        } else {
            for (uint32_t a = threadIdx.x; a < ng; a += SEED_BLOCK) {
                uint32_t ga = (uint32_t)gate_tmp[a];
                uint64_t ta = seg_tfirst[ga];
                uint32_t rank = 0;
                for (uint32_t b = 0; b < ng; ++b) rank += seg_tfirst[(uint32_t)gate_tmp[b]] < ta ? 1u : 0u;
                gated[rank] = ga;
            }
        }
        if (threadIdx.x == 0) A.ngated[s] = ng;
    }
}

// ------------------------------------------------------------------------------------------------ the strand-resident pipeline
// seed_strand = seed_filter + seed_emit + seed_sort_pass x 2 + seed_build of one strand inside one workgroup, for the strands whose
// kept hits fit the LDS of a CU (all of them at BASELINE config 2: ~4 k kept hits of ~39 k bucket hits per strand, ~2.7 k
// segments).  The kept hits never travel through HBM: what leaves the workgroup is what seed_cand reads (recorded events, the
// segment table in ascending segment order, the gated segments in first-touch order).  Everything else falls back to the
// kernels above, strand by strand (A.fused[s] == 0): nanopore mode (no relevance filter), strands with more than FS_CAP kept
// hits or FS_SEGCAP segments, repeats that pile > FS_BIGN hits into one table slot, and strands that find the shared output
// arrays full.
//
//   walk 1  (slots[])    16-bit hit counters per hashed segment id ("table slot", 2^15 of them), LDS atomics, no return value
//   relevance            as seed_filter; then occ = relevant & non-empty slots, compact index ci(slot) = word prefix + popcount,
//                        region start of every occupied slot = prefix sum of its counter
//   walk 2  (offsets[])  every kept hit takes the next place of its slot's region (LDS atomic): payload = seg_hi:6 | km:15 |
//                        off:11, slot kept beside it.  Regions come out in slot order; inside a region the order is whatever
//                        the atomics made it
//   region sort          ascending payload inside each region = (segment, km, position) order, the order the reference visits
//                        the hits of a segment in (keys are unique).  Regions are tiny (a hit or two) except where a read
//                        meets itself (~200 hits per segment): every element of a short region counts the smaller ones (all reads,
//                        barrier, all writes); a wave sorts a long region in registers (bitonic network)
//   build                phases A-E of seed_build on the LDS arrays.  The elements stay in slot-major order; only the segment
//                        table is permuted to ascending segment id (stable split on the segment bits above the slot bits)
//
// Both walks are software pipelined: bucket headers two steps ahead, bucket data one step ahead of the hits being consumed.
// They are gathers of ~50 / ~90 byte runs (mecat_amd/tools/gather_peak.hip measures what the chip delivers on such runs: 35 G / s of
// <= 64 bytes, 26 G / s of 128 bytes), and they cost per bucket rather than per hit: see the FS_LPB / FS_NP table below.
#define FS_THREADS 1024
#define FS_WAVES (FS_THREADS / WAVE)
#define FS_CAP 8192                  // kept hits
#define FS_SEGCAP 6144               // segments
#define FS_SMALLN 16                 // regions up to this length: every element ranks itself
#define FS_BIGN 512                  // longest region (8 elements per lane of a wave)
#define FS_BIGCAP 768                // regions longer than FS_SMALLN (FS_CAP / (FS_SMALLN + 1) at most: 630)
#define FS_GATECAP 1024              // gated segments
#define FS_OVF16 0x8000u
#ifndef FS_Q1
// walk 1 / walk 2: lanes per bucket, pieces of a bucket loaded ahead, buckets per group and stage, stages in flight.  Measured at
// config 2 (seed_strand per pass): 16 lanes x 3 pieces 46.1 ms, 8 x 4 43.4 ms, 8 x 3 45.0, 8 x 2 48.9, 4 x 8 50.4; three buckets per
// stage or three stages in flight 45.3 - 45.8: a walk costs per bucket (its loads are issued whatever the bucket holds), not per hit
#define FS_LPB1 8
#define FS_NP1 4
#define FS_Q1 2
#define FS_D1 2
#define FS_LPB2 8
#define FS_NP2 4
#define FS_Q2 2
#define FS_D2 2
#endif

struct FsLds {
    union {
        uint32_t cnt32[FLT_M / 2];                       // walk 1 .. relevance: 16-bit counters, two per word (64 KB)
        struct {
            uint32_t pay[FS_CAP];                        // walk 2 ..: payload, then (phase A, in place) recorded events
            uint16_t eslot[FS_CAP];
            uint16_t big[FS_BIGCAP];                     // long regions (compact slot index); later: overflowed segments
            uint32_t hist[FS_WAVES][64];                 // segment permutation: per-wave cursors
            uint64_t tf[FS_GATECAP];                     // first-touch times of the gated segments
        } e;
    } x;
    union {
        uint32_t cur32[FS_CAP / 2];                      // region cursors, 16 bit each (walk 2, region sort)
        uint16_t perm[FS_SEGCAP];                        // build: segment-order index -> slot-order index
    } y;
    uint32_t occ[REL_WORDS];
    uint32_t rel[REL_WORDS];
    uint16_t base_ci[REL_WORDS];
    uint32_t sq_id[FS_SEGCAP];                           // segment table in slot-major order q
    uint16_t sq_st[FS_SEGCAP + 2];                       // first recorded event of the segment; [nseg] = nrec
    uint16_t sq_score[FS_SEGCAP];                        // live score | FS_OVF16
    uint16_t gate[FS_GATECAP];
    uint32_t qpos[FS_WAVES][128];                        // walk 2: kept hits waiting for a full wave (position, km)
    uint16_t qkm[FS_WAVES][128];
    uint32_t wtot[FS_WAVES];
    uint32_t misc[8];                                    // 0 big count, 1 overflow count, 2 gated count, 3 fail flag, 4/5 carry, 6 hb lo, 7 hb hi
};
static_assert(sizeof(FsLds) <= 160 * 1024 - 512, "one workgroup per CU");

struct FusedCtl { unsigned long long alloc; unsigned int n_fallback; unsigned int pad; unsigned long long prof[32]; };
#ifdef FS_PROF
#define FS_MARK(i) do { __syncthreads(); if (threadIdx.x == 0) { const unsigned long long t_ = wall_clock64(); atomicAdd(&ctl->prof[(i) + 16 * (blockIdx.x & 1)], t_ - t_prev); t_prev = t_; } } while (0)
#else
#define FS_MARK(i) do { } while (0)
#endif

// block-wide exclusive scan, FS_THREADS threads
__device__ __forceinline__ uint32_t fs_excl_scan(uint32_t v, uint32_t* wtot, uint32_t* total) {
    uint32_t incl = v;
    for (int o = 1; o < 64; o <<= 1) {
        uint32_t n = __shfl_up(incl, o);
        if (lane_id() >= o) incl += n;
    }
    __syncthreads();
    if (lane_id() == 63) wtot[threadIdx.x >> 6] = incl;
    __syncthreads();
    uint32_t base = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < FS_WAVES; ++w) {
        const uint32_t x = wtot[w];
        if (w < (int)(threadIdx.x >> 6)) base += x;
        tot += x;
    }
    *total = tot;
    return base + incl - v;
}

// pipelined walk over the buckets of a strand; f(km, value) per hit.  LPB lanes per bucket, Q buckets per group and stage, D stages
// of bucket data in flight plus one stage of bucket headers (start, size) ahead of them.  The first NP * LPB entries of a bucket are
// loaded by the pipeline (NP pieces of LPB); a longer bucket reads the rest when it is consumed.
template <int LPB, int NP, int Q, int D, bool UNIFORM, typename T, typename F>
__device__ __forceinline__ void fs_walk(const uint32_t* __restrict__ kbs, const uint32_t* __restrict__ kcn, const T* __restrict__ arr, const int K, F f) {
    constexpr int G = FS_THREADS / LPB, STEP = Q * G;
    const int g = threadIdx.x / LPB;
    const uint32_t sub = threadIdx.x % LPB;
    uint32_t hb[Q], hc[Q];                          // headers of stage D (no data requested yet)
    uint32_t sb[D][Q], sc[D][Q], sd[D][Q][NP];      // stages 0 .. D-1: header and data
#define FS_HDR(BASE, BS, CN)                                                      \
    _Pragma("unroll") for (int q = 0; q < Q; ++q) {                               \
        const int km_ = (BASE) + q * G + g;                                       \
        BS[q] = km_ < K ? kbs[km_] : 0u;                                          \
        CN[q] = km_ < K ? kcn[km_] : 0u;                                          \
    }
#define FS_DAT(BS, CN, DD)                                                        \
    _Pragma("unroll") for (int q = 0; q < Q; ++q)                                 \
        _Pragma("unroll") for (int p = 0; p < NP; ++p)                            \
            DD[q][p] = sub + (uint32_t)(p * LPB) < CN[q] ? (uint32_t)arr[BS[q] + (uint32_t)(p * LPB) + sub] : 0u;
#pragma unroll
    for (int d = 0; d < D; ++d) { FS_HDR(d * STEP, sb[d], sc[d]) }
    FS_HDR(D * STEP, hb, hc)
#pragma unroll
    for (int d = 0; d < D; ++d) { FS_DAT(sb[d], sc[d], sd[d]) }
    for (int base = 0; base < K; base += STEP) {
        uint32_t cb[Q], cc[Q], cd[Q][NP];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            cb[q] = sb[0][q]; cc[q] = sc[0][q];
#pragma unroll
            for (int p = 0; p < NP; ++p) cd[q][p] = sd[0][q][p];
        }
#pragma unroll
        for (int d = 0; d + 1 < D; ++d)
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                sb[d][q] = sb[d + 1][q]; sc[d][q] = sc[d + 1][q];
#pragma unroll
                for (int p = 0; p < NP; ++p) sd[d][q][p] = sd[d + 1][q][p];
            }
#pragma unroll
        for (int q = 0; q < Q; ++q) { sb[D - 1][q] = hb[q]; sc[D - 1][q] = hc[q]; }
        FS_DAT(sb[D - 1], sc[D - 1], sd[D - 1])
        FS_HDR(base + (D + 1) * STEP, hb, hc)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            const int km = base + q * G + g;
            if constexpr (UNIFORM) {            // f(km, value, valid) is called by the whole wave together
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const bool v = sub + (uint32_t)(p * LPB) < cc[q];
                    if (__any(v)) f(km, cd[q][p], v);
                }
                // (what the pipeline did not cover, four pieces per round trip: one by one each piece is a memory latency for the whole wave)
                for (uint32_t r = (uint32_t)(NP * LPB) + sub; __any(r < cc[q]); r += 4 * LPB) {
                    uint32_t tv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) tv[u] = r + (uint32_t)(u * LPB) < cc[q] ? (uint32_t)arr[cb[q] + r + (uint32_t)(u * LPB)] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const bool v = r + (uint32_t)(u * LPB) < cc[q]; if (__any(v)) f(km, tv[u], v); }
                }
            } else {
#pragma unroll
                for (int p = 0; p < NP; ++p)
                    if (sub + (uint32_t)(p * LPB) < cc[q]) f(km, cd[q][p]);
                for (uint32_t r = (uint32_t)(NP * LPB) + sub; r < cc[q]; r += 4 * LPB) {      // (four pieces per round trip, see above)
                    uint32_t tv[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) tv[u] = r + (uint32_t)(u * LPB) < cc[q] ? (uint32_t)arr[cb[q] + r + (uint32_t)(u * LPB)] : 0u;
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (r + (uint32_t)(u * LPB) < cc[q]) f(km, tv[u]);
                }
            }
        }
    }
#undef FS_HDR
#undef FS_DAT
}

// ascending sort of p[0 .. n) (LDS, n <= 64 NQ) by one wave: bitonic network, element q * 64 + lane in register v[q]
template <int NQ>
__device__ __noinline__ void fs_bitonic(uint32_t* p, const uint32_t n, const int lane) {
    uint32_t v[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const uint32_t idx = q * 64 + lane; v[q] = idx < n ? p[idx] : 0xFFFFFFFFu; }
#pragma unroll
    for (int k = 2; k <= NQ * 64; k <<= 1) {
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            if (j >= 64) {
                const int dq = j >> 6;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    if ((q & dq) == 0) {
                        const uint32_t a = v[q], b = v[q | dq];
                        const bool asc = ((q * 64) & k) == 0;         // k > j >= 64: the k bit of the index is a bit of q
                        v[q] = asc ? min(a, b) : max(a, b);
                        v[q | dq] = asc ? max(a, b) : min(a, b);
                    }
                }
            } else {
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const uint32_t o = (uint32_t)__shfl_xor((int)v[q], j);
                    const bool asc = (((q * 64 + lane) & k) == 0), lower = (lane & j) == 0;
                    v[q] = (asc == lower) ? min(v[q], o) : max(v[q], o);
                }
            }
        }
    }
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int q = 0; q < NQ; ++q) { const uint32_t idx = q * 64 + lane; if (idx < n) p[idx] = v[q]; }
}

__device__ __forceinline__ uint32_t fs_cnt(const uint32_t* cnt32, uint32_t e) { return (cnt32[e >> 1] >> ((e & 1u) * 16u)) & 0xFFFFu; }

// ------------------------------------------------------------------------------------------------ relevance filter, low gates
// Nanopore mode gates at index_score >= 4 (min_kmer_match 2), and a 2.14 Gbase volume puts ~64 k bucket hits of a strand on ~1.07 M
// segments: folded onto seed_filter's 2^15 slots that is two hits per slot, every second slot pair passes a gate of 4, and the reach
// of the sweeps (+- 11 segments for a 20 kb read) makes everything relevant — which is why that filter is only used from a gate of 6
// up, and why nanopore strands used to sort and build all of their hits in HBM (0.66 of the 0.73 s of a config-5 grid cell).
// Same construction with 2^18 slots (a quarter hit per slot: 0.2 % of the slot pairs pass by chance) and 4-bit counters in the LDS of
// a whole CU; the relevance bitmap has one bit per 8 slots, so it is the 4 KB per strand seed_emit already reads.  A slot of a true
// overlap collects 20-30 hits, so a 4-bit counter wraps there: the add that wraps it (its return value says so, and which neighbours
// the carry ran into) marks the slot relevant on the spot and counts the wrap.  Counters next to a wrapped one are too high by the
// carry: the bitmap only grows, and the number of kept hits becomes an upper bound (sum over the relevant slots + 16 per wrap) — the
// room of the strand in the key arrays; seed_emit writes the exact number when it has placed the keys.
#ifndef WF_LPB
// the walk of seed_filter_wide: lanes per bucket, pieces loaded ahead, buckets per group and stage, stages in flight (see FS_LPB1)
#define WF_LPB 8
#define WF_NP 4
#define WF_Q 2
#define WF_D 2
#endif
#define WF_BITS 18
#define WF_M (1 << WF_BITS)
#define WF_CELL 3                    // log2 slots per relevance bit
static_assert((WF_M >> WF_CELL) == FLT_M, "the bitmap is REL_WORDS words");
__global__ __launch_bounds__(FS_THREADS) void seed_filter_wide(const mhip_offset_t* __restrict__ roffs, ReadSel sel, int ib,
                                                               const int32_t* __restrict__ offsets, SeedArrays A, int gate, int same_volume,
                                                               unsigned long long* __restrict__ counters) {
    __shared__ uint32_t cnt[WF_M / 8];           // 128 KB: eight 4-bit counters per word
    __shared__ uint32_t rel[REL_WORDS];
    __shared__ uint32_t hotb[3][REL_WORDS];      // cells with a hot slot; of those, the ones whose reach needs one more cell on the left / right
    __shared__ uint32_t wtot[FS_WAVES];
    __shared__ uint32_t s_wraps, s_self;
    const int s = blockIdx.x, tid = threadIdx.x;
    const int rid = sel_rid(sel, ib + (s >> 1));
    const int L = roffs[rid].size;
    const int K = kmers_of(L);
    const uint32_t kb = A.km_base[s];
    const uint32_t Hall = A.strand_hits_all[s];
    if (A.fused[s] || Hall == 0) {
        if (tid == 0) { A.strand_hits[s] = 0; A.filtered[s] = 0; }
        return;
    }
    WF_T0;
    for (int i = tid; i < WF_M / 8; i += FS_THREADS) cnt[i] = 0;
    rel[tid] = 0;                                 // REL_WORDS == FS_THREADS
    static_assert(REL_WORDS == FS_THREADS, "one bitmap word per thread");
    if (tid == 0) { s_wraps = 0; s_self = 0; }
    __syncthreads();
    WF_MARK(0);
    const uint32_t reach = (uint32_t)min(sweep_reach(L), WF_M / 2 - 1);
    auto mark = [&](int lo, int hi) {            // slots lo .. hi (unwrapped, hi - lo < WF_M) -> their bitmap bits
        uint32_t cell = (uint32_t)(lo >> WF_CELL) & (FLT_M - 1), left = min((uint32_t)((hi >> WF_CELL) - (lo >> WF_CELL) + 1), (uint32_t)FLT_M);
        while (left > 0) {
            const uint32_t b = cell & 31u, take = min(32u - b, left);
            const uint32_t m = (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << b;
            atomicOr(&rel[cell >> 5], m);
            cell = (cell + take) & (FLT_M - 1);
            left -= take;
        }
    };
    // The forward strand of a read against the volume it comes from hits its own copy with every k-mer (200 hits per segment: every
    // counter there would wrap).  Those hits are known without counting them: position = the read's offset + 10 km.  They are
    // counted apart, and the read's own segments are relevant whatever the counters say.
    const bool own = same_volume && !(s & 1);
    const uint32_t own_off = (uint32_t)roffs[rid].offset;
    fs_walk<WF_LPB, WF_NP, WF_Q, WF_D, false>(A.km_bstart + kb, A.km_cnt + kb, offsets, K, [&](int km, uint32_t pos) {
#if defined(WF_KNOCK) && WF_KNOCK == 1
        if (pos == 0xfffffff1u) atomicAdd(&s_self, 1u);      // timing experiment: the gather alone (results are wrong)
        return;
#elif defined(WF_KNOCK) && WF_KNOCK == 2
        { const uint32_t e_ = (pos / (uint32_t)ZV) & (WF_M - 1); atomicAdd(&cnt[e_ >> 3], 1u << ((e_ & 7u) * 4u)); return; }      // no return value, no wrap test
#endif
        if (own && pos == own_off + (uint32_t)km * BC) { atomicAdd(&s_self, 1u); return; }
        const uint32_t e = (pos / (uint32_t)ZV) & (WF_M - 1);
        const uint32_t old = atomicAdd(&cnt[e >> 3], 1u << ((e & 7u) * 4u));
        // the counters this add wrapped: slot e if it stood at 15, and the neighbours at 15 the carry went on through
        for (uint32_t n = e & 7u; n < 8u && ((old >> (4u * n)) & 15u) == 15u; ++n) {
            const int w = (int)((e & ~7u) + n);
            mark(w - (int)reach, w + (int)reach);
            atomicAdd(&s_wraps, 1u);
        }
    });
    __syncthreads();
    WF_MARK(1);
    // hot slots -> relevance bitmap (one bit per cell of 8 slots = one counter word)
    if (own && tid == 0) mark((int)(own_off / (uint32_t)ZV) - (int)reach, (int)((own_off + (uint32_t)L) / (uint32_t)ZV) + (int)reach);
    {
        // Eight counters at a time: the even and the odd nibbles of a word as the bytes of two registers, the four sums inside the pairs
        // (n0+n1 ..) and the four across them (n1+n2 .., n7 + the next word's n0) as byte adds, "sum >= gate" as the carry into bit 7 of
        // every byte (sums are at most 30).  A slot is hot when it is occupied and one of its two pairs reaches the gate.  All hot slots
        // of a word lie in one bitmap cell c, and with reach = 8 a + b their reaches cover the cells c - a .. c + a, one more on the left
        // when the lowest hot slot of the word is below b, one more on the right when the highest is at 8 - b or above: three cell
        // bitmaps (hot, needs-left, needs-right), written as wave ballots (lanes = consecutive words: a thread that walks 32 consecutive
        // words of its own puts all 64 lanes of every read on one LDS bank), and one dilation pass, thread t = bitmap word t.
        // (Slot by slot, a branch per nibble and an atomicOr loop per hot slot: 22 of the kernel's 56 us per strand; this: 4.)
        const uint32_t kb7 = (uint32_t)(128 - min(gate, 31)) * 0x01010101u;
        const int ra = (int)(reach >> 3), rb = (int)(reach & 7u);
        const bool by_bitmaps = ra <= 16;            // (longer reaches — reads beyond 250 kb — mark hot word by hot word)
#pragma unroll 4
        for (int j = 0; j < 32; ++j) {
            const uint32_t wi = (uint32_t)(j * FS_THREADS + tid);
            // (all three reads unconditionally: no branch in front of the arithmetic, the reads of the unrolled steps overlap)
            const uint32_t w = cnt[wi], wprev = cnt[(wi - 1u) & (WF_M / 8 - 1)], wnext = cnt[(wi + 1u) & (WF_M / 8 - 1)];
            const uint32_t E = w & 0x0F0F0F0Fu, O = (w >> 4) & 0x0F0F0F0Fu;
            // bit 7 of every byte: the pair's sum reaches the gate / the counter is not zero
            const uint32_t H1 = E + O + kb7;                                                              // pairs (0,1) (2,3) (4,5) (6,7)
            const uint32_t H2 = O + __builtin_amdgcn_alignbit(wnext & 15u, E, 8) + kb7;                   // pairs (1,2) (3,4) (5,6) (7,next)
            const uint32_t hp = ((wprev >> 28) + (w & 15u)) >= (uint32_t)gate ? 0x80u : 0u;              // pair (previous word's 7, 0)
            const uint32_t hotE = (E + 0x7F7F7F7Fu) & (H1 | (H2 << 8) | hp) & 0x80808080u;
            const uint32_t hotO = (O + 0x7F7F7F7Fu) & (H1 | H2) & 0x80808080u;
            const uint32_t m = (hotE >> 7) | (hotO >> 6);          // bit 8 i + j <-> nibble 2 i + j
            const unsigned long long bh = __ballot(m != 0);
            if (by_bitmaps) {
                unsigned long long bl = 0, br = 0;
                if (bh) {               // (a wave with a hot word: one step in ten)
                    const int l = __builtin_ctz(m | 0x80000000u), h = 31 - __builtin_clz(m | 1u);
                    const int lo = ((l >> 3) << 1) | (l & 1), hi = ((h >> 3) << 1) | (h & 1);
                    bl = __ballot(m != 0 && lo < rb); br = __ballot(m != 0 && hi + rb >= 8);
                }
                if ((tid & 63) == 0) {
                    const uint32_t bw = wi >> 5;          // the wave's 64 words = two bitmap words
                    hotb[0][bw] = (uint32_t)bh; hotb[0][bw + 1] = (uint32_t)(bh >> 32);
                    hotb[1][bw] = (uint32_t)bl; hotb[1][bw + 1] = (uint32_t)(bl >> 32);
                    hotb[2][bw] = (uint32_t)br; hotb[2][bw + 1] = (uint32_t)(br >> 32);
                }
            } else if (m) {
                const int l = __builtin_ctz(m), h = 31 - __builtin_clz(m);
                const int e0 = (int)wi * 8;
                mark(e0 + (((l >> 3) << 1) | (l & 1)) - (int)reach, e0 + (((h >> 3) << 1) | (h & 1)) + (int)reach);
            }
        }
        if (by_bitmaps) {
            __syncthreads();
            // word t of bitmap B moved d cells up (d < 0: down), circular
            auto moved = [&](const uint32_t* B, int d) -> uint32_t {
                if (d >= 0) {
                    const int q = d >> 5, r = d & 31;
                    const uint32_t x = B[(tid - q) & (REL_WORDS - 1)];
                    return r ? (x << r) | (B[(tid - q - 1) & (REL_WORDS - 1)] >> (32 - r)) : x;
                }
                const int q = (-d) >> 5, r = (-d) & 31;
                const uint32_t x = B[(tid + q) & (REL_WORDS - 1)];
                return r ? (x >> r) | (B[(tid + q + 1) & (REL_WORDS - 1)] << (32 - r)) : x;
            };
            uint32_t acc = moved(hotb[2], ra + 1) | moved(hotb[1], -(ra + 1));
            for (int d = -ra; d <= ra; ++d) acc |= moved(hotb[0], d);
            if (acc) atomicOr(&rel[tid], acc);          // (the wraps of the walk and the read's own segments are in there already)
        }
    }
    __syncthreads();
    WF_MARK(2);
    // kept hits = hits of the slots whose bit is set
    uint32_t mine = 0;
    {
        const uint32_t r = rel[tid];
        for (uint32_t m = r; m; m &= m - 1u) {
            const uint32_t w = cnt[32 * tid + (uint32_t)__builtin_ctz(m)];
            // sum of the eight nibbles
            const uint32_t a = (w & 0x0F0F0F0Fu) + ((w >> 4) & 0x0F0F0F0Fu);
            mine += (a * 0x01010101u) >> 24;
        }
    }
    uint32_t kept;
    (void)fs_excl_scan(mine, wtot, &kept);
    A.rel_bits[(size_t)s * REL_WORDS + tid] = rel[tid];
    if (tid == 0) {
        const uint32_t room = min(Hall, kept + s_self + 16u * s_wraps);
        A.strand_hits[s] = room;
        A.filtered[s] = 1;
        atomicAdd(&counters[12], (unsigned long long)room);      // debug slot 12: room asked for by this filter
    }
    WF_MARK(3);
}

__global__ __launch_bounds__(FS_THREADS) void seed_strand(const mhip_offset_t* __restrict__ roffs, ReadSel sel, int ib,
                                                          const uint16_t* __restrict__ slots, const int32_t* __restrict__ offsets, SeedArrays A,
                                                          int gate, int hi_bits, int min_kmer_match, double cutoff, unsigned long long cap,
                                                          FusedCtl* __restrict__ ctl, unsigned long long* __restrict__ counters, RefReads R) {
    __shared__ FsLds L;
    const int s = blockIdx.x;
    const int tid = threadIdx.x, lane = lane_id(), wv = threadIdx.x >> 6;
    const int rid = sel_rid(sel, ib + (s >> 1));
    const int rlen = roffs[rid].size;
    const int K = kmers_of(rlen);
    const uint32_t kb = A.km_base[s];
    const uint32_t Hall = A.strand_hits_all[s];
    const uint32_t* __restrict__ kbs = A.km_bstart + kb;
    const uint32_t* __restrict__ kcn = A.km_cnt + kb;
    auto fail = [&]() {                     // called by all threads together
        if (tid == 0) { A.fused[s] = 0; atomicAdd(&ctl->n_fallback, 1u); atomicAdd(&counters[14], 1ull); }      // debug slot 14: strands left to the kernel chain
    };
    if (Hall == 0) {
        if (tid == 0) { A.fused[s] = 1; A.strand_hits[s] = 0; A.hit_base[s] = 0; A.nseg[s] = 0; A.nrec[s] = 0; A.ngated[s] = 0; }
        return;
    }
    if (Hall >= 65536u || K > 32767) { fail(); return; }        // a 16-bit counter could wrap; km has 15 bits in the payload

    unsigned long long t_prev = wall_clock64(); (void)t_prev;
    // ---- walk 1: hits per table slot
    for (int i = tid; i < FLT_M / 2; i += FS_THREADS) L.x.cnt32[i] = 0;
    for (int i = tid; i < REL_WORDS; i += FS_THREADS) L.rel[i] = 0u;
    if (tid < 8) L.misc[tid] = 0;
    __syncthreads();
    FS_MARK(0);
    fs_walk<FS_LPB1, FS_NP1, FS_Q1, FS_D1, false>(kbs, kcn, slots, K, [&](int, uint32_t e) { atomicAdd(&L.x.cnt32[e >> 1], 1u << ((e & 1u) * 16u)); });
    __syncthreads();
    FS_MARK(1);

    // ---- relevance: hot slots, +- reach (seed_filter).  A thread owns one word of the bitmaps = 32 slots = 16 counter words,
    // which it keeps in registers from here to the region starts.
    static_assert(FLT_M / FS_THREADS == 32, "one bitmap word per thread");
    uint32_t cw[18];                   // cw[1 .. 16]: the thread's counters; cw[0], cw[17]: the neighbours' last / first word
    {
        const uint4* c4 = (const uint4*)&L.x.cnt32[16 * tid];
#pragma unroll
        for (int q = 0; q < 4; ++q) { const uint4 v = c4[q]; cw[1 + 4 * q] = v.x; cw[2 + 4 * q] = v.y; cw[3 + 4 * q] = v.z; cw[4 + 4 * q] = v.w; }
        cw[0] = L.x.cnt32[(16 * tid - 1) & (FLT_M / 2 - 1)];
        cw[17] = L.x.cnt32[(16 * tid + 16) & (FLT_M / 2 - 1)];
        uint32_t hot = 0;
        {
#pragma unroll
        for (int j = 1; j <= 16; ++j) {
            const int lo = (int)(cw[j] & 0xFFFFu), hi = (int)(cw[j] >> 16), pl = (int)(cw[j - 1] >> 16), nx = (int)(cw[j + 1] & 0xFFFFu);
            const bool hl = lo > 0 && (lo + hi >= gate || lo + pl >= gate);
            const bool hh = hi > 0 && (hi + nx >= gate || hi + lo >= gate);
            hot |= (hl ? 1u : 0u) << (2 * (j - 1)) | (hh ? 1u : 0u) << (2 * (j - 1) + 1);
        }
        const uint32_t reach = (uint32_t)min(sweep_reach(rlen), FLT_M / 2 - 1);
        for (uint32_t m = hot; m; m &= m - 1u) {
            const uint32_t e = (uint32_t)tid * 32u + (uint32_t)__builtin_ctz(m);
            uint32_t lo = (e - reach) & (FLT_M - 1), left = 2 * reach + 1;
            while (left > 0) {
                const uint32_t b = lo & 31u, take = min(32u - b, left);
                const uint32_t msk = (take == 32u ? 0xFFFFFFFFu : ((1u << take) - 1u)) << b;
                atomicOr(&L.rel[lo >> 5], msk);
                lo = (lo + take) & (FLT_M - 1);
                left -= take;
            }
        }
        }
    }
    __syncthreads();
    FS_MARK(2);
    // ---- occupied relevant slots, region starts
    uint32_t kept;
    {
        const uint32_t r = L.rel[tid];
        uint32_t occ = 0, mine = 0;
        bool toolong = false;
#pragma unroll
        for (int j = 0; j < 32; ++j) {
            const uint32_t h = (cw[1 + (j >> 1)] >> ((j & 1) * 16)) & 0xFFFFu;
            if (((r >> j) & 1u) && h) { occ |= 1u << j; mine += h; toolong |= h > FS_BIGN; }
        }
        // one scan for both prefixes: hits in the low half (Hall < 2^16), occupied slots in the high half (<= 2^15)
        uint32_t both;
        const uint32_t exb = fs_excl_scan(mine | ((uint32_t)__popc(occ) << 16), L.wtot, &both);
        const uint32_t ex = exb & 0xFFFFu, cx = exb >> 16;
        kept = both & 0xFFFFu;
        if (kept > FS_CAP) { fail(); return; }
        L.occ[tid] = occ;
        L.base_ci[tid] = (uint16_t)cx;
        uint16_t* cur16 = (uint16_t*)L.y.cur32;
        uint32_t at = ex, ci = cx;
        if (occ) {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                const uint32_t h = (cw[1 + (j >> 1)] >> ((j & 1) * 16)) & 0xFFFFu;
                if ((occ >> j) & 1u) { cur16[ci++] = (uint16_t)at; at += h; }
            }
        }
        if (toolong) L.misc[3] = 1;
        __syncthreads();
        if (L.misc[3]) { fail(); return; }
    }
    if (kept == 0) {
        if (tid == 0) { A.fused[s] = 1; A.strand_hits[s] = 0; A.hit_base[s] = 0; A.nseg[s] = 0; A.nrec[s] = 0; A.ngated[s] = 0; }
        return;
    }
    // room in the shared output arrays
    if (tid == 0) {
        const unsigned long long hb0 = atomicAdd(&ctl->alloc, (unsigned long long)kept);
        if (hb0 + kept > cap) L.misc[3] = 1;
        L.misc[6] = (uint32_t)hb0;
        L.misc[7] = (uint32_t)(hb0 >> 32);
    }
    __syncthreads();                                      // also: cnt32 is dead from here, the payload arrays take its place
    if (L.misc[3]) { fail(); return; }
    const uint64_t hb = ((uint64_t)L.misc[7] << 32) | L.misc[6];

    FS_MARK(3);
    // ---- walk 2: kept hits into their slot's region
    // A kept hit is rare (one in nine): the wave parks kept hits in its queue and places 64 of them at a time, so that the
    // cursor arithmetic runs with every lane busy.
    {
        uint32_t qn = 0;                                  // wave-uniform: hits in this wave's queue
        auto place = [&](uint32_t qi) {
            const uint32_t pos = L.qpos[wv][qi], km = L.qkm[wv][qi];
            const uint32_t seg = pos / (uint32_t)ZV, off = pos - seg * (uint32_t)ZV;
            const uint32_t e = seg & (FLT_M - 1), w = e >> 5, b = e & 31u;
            const uint32_t oc = L.occ[w];
            const uint32_t ci = (uint32_t)L.base_ci[w] + (uint32_t)__popc(oc & ((1u << b) - 1u));
            const uint32_t sh = (ci & 1u) * 16u;
            const uint32_t p = (atomicAdd(&L.y.cur32[ci >> 1], 1u << sh) >> sh) & 0xFFFFu;
            L.x.e.pay[p] = ((seg >> FLT_BITS) << 26) | (km << 11) | off;
            L.x.e.eslot[p] = (uint16_t)e;
        };
        fs_walk<FS_LPB2, FS_NP2, FS_Q2, FS_D2, true>(kbs, kcn, offsets, K, [&](int km, uint32_t pos, bool valid) {
            const uint32_t seg = pos / (uint32_t)ZV;
            const uint32_t e = seg & (FLT_M - 1);
            const uint32_t oc = valid ? L.occ[e >> 5] : 0u;
            const bool keep = (bool)((oc >> (e & 31u)) & 1u);
            const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
            if (m) {
                if (keep) {
                    const uint32_t at = qn + (uint32_t)__popcll(m & ((1ull << lane) - 1ull));
                    L.qpos[wv][at] = pos;
                    L.qkm[wv][at] = (uint16_t)km;
                }
                qn += (uint32_t)__popcll(m);
                if (qn >= 64u) { qn -= 64u; place(qn + (uint32_t)lane); }
            }
        });
        if ((uint32_t)lane < qn) place((uint32_t)lane);
    }
    __syncthreads();
    FS_MARK(4);
    // ---- region sort.  After walk 2 cursor[ci] = end of region ci = start of region ci + 1.
    {
        const uint16_t* cur16 = (const uint16_t*)L.y.cur32;
        // short regions: every element counts the smaller elements of its region (all reads, barrier, all writes)
        uint32_t val[FS_CAP / FS_THREADS], dst[FS_CAP / FS_THREADS];
#pragma unroll
        for (int rd = 0; rd < FS_CAP / FS_THREADS; ++rd) {
            const uint32_t i = (uint32_t)rd * FS_THREADS + tid;
            dst[rd] = 0xFFFFFFFFu;
            val[rd] = 0;
            if ((uint32_t)rd * FS_THREADS < kept && i < kept) {
                const uint32_t v = L.x.e.pay[i], e = L.x.e.eslot[i], w = e >> 5, oc = L.occ[w];
                const uint32_t ci = (uint32_t)L.base_ci[w] + (uint32_t)__popc(oc & ((1u << (e & 31u)) - 1u));
                const uint32_t st = ci ? cur16[ci - 1] : 0u, en = cur16[ci], n = en - st;
                if (n > FS_SMALLN) {
                    if (i == st) L.x.e.big[atomicAdd(&L.misc[0], 1u)] = (uint16_t)ci;
                } else if (n > 1) {
                    uint32_t rank = 0;
                    for (uint32_t k = st; k < en; ++k) rank += L.x.e.pay[k] < v ? 1u : 0u;
                    val[rd] = v;
                    dst[rd] = st + rank;
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int rd = 0; rd < FS_CAP / FS_THREADS; ++rd)
            if (dst[rd] != 0xFFFFFFFFu) L.x.e.pay[dst[rd]] = val[rd];
        FS_MARK(11);
        // long regions (a read meeting itself: ~200 hits per segment): one wave each, bitonic network in registers
        const uint32_t nbig = L.misc[0];
        for (uint32_t k = wv; k < nbig; k += FS_WAVES) {
            const uint32_t ci = L.x.e.big[k];
            const uint32_t st = ci ? cur16[ci - 1] : 0u, n = (uint32_t)cur16[ci] - st;      // FS_SMALLN < n <= FS_BIGN
            if (n <= 64) fs_bitonic<1>(L.x.e.pay + st, n, lane);
            else if (n <= 256) fs_bitonic<4>(L.x.e.pay + st, n, lane);
            else fs_bitonic<8>(L.x.e.pay + st, n, lane);
        }
        __syncthreads();
    }

    FS_MARK(5);
    // ---- build, phase A: recorded events (first hit of each (segment, km)) and segment heads, events compacted in place.
    // A thread takes a run of consecutive elements into registers; one scan; then everybody writes.
    uint32_t nrec, nseg;
    {
        constexpr int CHMAX = FS_CAP / FS_THREADS;
        const uint32_t ch = (kept + FS_THREADS - 1) / FS_THREADS;
        const uint32_t i0 = (uint32_t)tid * ch;
        uint32_t pv[CHMAX], sv[CHMAX];
        uint32_t ppay = 0, psl = 0xFFFFFFFFu;                       // no element before the first: a head
        if (i0 > 0 && i0 < kept) { ppay = L.x.e.pay[i0 - 1]; psl = L.x.e.eslot[i0 - 1]; }
        uint32_t hm = 0, rm = 0;
#pragma unroll
        for (int c = 0; c < CHMAX; ++c) {
            const uint32_t i = i0 + c;
            const bool in = (uint32_t)c < ch && i < kept;
            pv[c] = in ? L.x.e.pay[i] : 0u;
            sv[c] = in ? L.x.e.eslot[i] : 0u;
            const bool head = in && (sv[c] != psl || (pv[c] >> 26) != (ppay >> 26));
            const bool rec = in && (head || ((pv[c] >> 11) & 0x7FFFu) != ((ppay >> 11) & 0x7FFFu));
            hm |= (head ? 1u : 0u) << c;
            rm |= (rec ? 1u : 0u) << c;
            if (in) { ppay = pv[c]; psl = sv[c]; }
        }
        uint32_t both;
        const uint32_t ex = fs_excl_scan((uint32_t)__popc(rm) | ((uint32_t)__popc(hm) << 16), L.wtot, &both);     // its barriers come after all the reads
        uint32_t rpos = ex & 0xFFFFu, spos = ex >> 16;
        nrec = both & 0xFFFFu;
        nseg = both >> 16;
        if (nseg > FS_SEGCAP) { fail(); return; }
#pragma unroll
        for (int c = 0; c < CHMAX; ++c) {
            if ((rm >> c) & 1u) {
                if ((hm >> c) & 1u) {
                    L.sq_id[spos] = ((pv[c] >> 26) << FLT_BITS) | sv[c];
                    L.sq_st[spos] = (uint16_t)rpos;
                    ++spos;
                }
                L.x.e.pay[rpos++] = ((pv[c] & 0x7FFu) << 16) | ((((pv[c] >> 11) & 0x7FFFu) + 1u) & 0xFFFFu);
            }
        }
        if (tid == 0) L.sq_st[nseg] = (uint16_t)nrec;
    }
    uint16_t* perm = L.y.perm;                           // the cursors are dead
    __syncthreads();

    FS_MARK(6);
    // ---- segment order: stable split of the slot-major table on the segment bits above the slot bits
    if (hi_bits == 0) {
        for (uint32_t q = tid; q < nseg; q += FS_THREADS) perm[q] = (uint16_t)q;
    } else {
        uint32_t* hist = &L.x.e.hist[0][0];
        hist[tid] = 0;                                    // FS_WAVES * 64 == FS_THREADS
        __syncthreads();
        const uint32_t chunk = (((nseg + FS_WAVES - 1) / FS_WAVES) + 63u) & ~63u;
        const uint32_t lo = min(nseg, (uint32_t)wv * chunk), hi = min(nseg, lo + chunk);
        for (uint32_t q = lo + lane; q < hi; q += 64) atomicAdd(&L.x.e.hist[wv][L.sq_id[q] >> FLT_BITS], 1u);
        __syncthreads();
        {
            const uint32_t bin = tid >> 4, w = tid & 15;         // (bin major, wave minor)
            static_assert(FS_WAVES == 16, "16 waves");
            const uint32_t v = L.x.e.hist[w][bin];
            uint32_t all;
            const uint32_t ex = fs_excl_scan(v, L.wtot, &all);
            L.x.e.hist[w][bin] = ex;
        }
        __syncthreads();
        volatile uint32_t* cur = L.x.e.hist[wv];
        const uint64_t lt = (1ull << lane) - 1ull;
        for (uint32_t q0 = lo; q0 < hi; q0 += 64) {
            const uint32_t q = q0 + lane;
            const bool valid = q < hi;
            const uint32_t d = valid ? L.sq_id[q] >> FLT_BITS : 0u;
            uint64_t peers = __ballot(valid);
            for (int b = 0; b < hi_bits; ++b) {
                const uint64_t m = __ballot((d >> b) & 1u);
                peers &= ((d >> b) & 1u) ? m : ~m;
            }
            const uint32_t rank = __popcll(peers & lt);
            const uint32_t base = valid ? cur[d] : 0u;
            if (valid) perm[base + rank] = (uint16_t)q;
            if (valid && rank == 0) cur[d] = base + (uint32_t)__popcll(peers);
        }
    }
    FS_MARK(7);
    // ---- phase B: scores; overflowed segments
    uint16_t* ovf = L.x.e.big;
    for (uint32_t q = tid; q < nseg; q += FS_THREADS) {
        const uint32_t c = (uint32_t)L.sq_st[q + 1] - (uint32_t)L.sq_st[q];
        if (c > SM) {
            const uint32_t k = atomicAdd(&L.misc[1], 1u);
            if (k < FS_BIGCAP) ovf[k] = (uint16_t)q;
            L.sq_score[q] = (uint16_t)FS_OVF16;
        } else {
            L.sq_score[q] = (uint16_t)c;
        }
    }
    __syncthreads();
    if (L.misc[1] > FS_BIGCAP) { fail(); return; }
    // ---- phase C: insert_loc replay, one wave per overflowed segment (final lists and running scores go to HBM)
    {
        const uint32_t novf = L.misc[1];
        for (uint32_t k = wv; k < novf; k += FS_WAVES) {
            const uint32_t q = ovf[k];
            const uint32_t st = L.sq_st[q], en = L.sq_st[q + 1];
            int sc;
            replay_overflow(L.x.e.pay + st, (int)(en - st), A.ent_fin + hb + st, A.escore + hb + st, cutoff, &sc);
            if (lane == 0) L.sq_score[q] = (uint16_t)((uint32_t)sc | FS_OVF16);
        }
    }
    __syncthreads();
    FS_MARK(8);
    // ---- phase D: index_score and the gate, in segment order
    for (uint32_t g = tid; g < nseg; g += FS_THREADS) {
        const uint32_t q = perm[g];
        const uint32_t sid = L.sq_id[q];
        int s_k = (int)(L.sq_score[q] & 0x7FFFu);
        if (g > 0 && sid > 0) {
            const uint32_t ql = perm[g - 1];
            if (L.sq_id[ql] == sid - 1u) {
                const uint32_t st = L.sq_st[ql], en = L.sq_st[ql + 1];
                const uint32_t t = L.x.e.pay[(uint32_t)L.sq_st[q + 1] - 1u] & 0xFFFFu;       // km + 1 of this segment's last event
                uint32_t lo = st, hi = en;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((L.x.e.pay[mid] & 0xFFFFu) <= t) lo = mid + 1; else hi = mid;
                }
                if (lo > st) s_k += (L.sq_score[ql] & FS_OVF16) ? (int)A.escore[hb + lo - 1] : (int)(lo - st);
            }
        }
        bool listed = (int)(int16_t)s_k >= 2 * min_kmer_match;
        if (listed && R.enable) listed = !subjects_all_higher(R, sid, rid, roffs[rid].offset + rlen + 1);
        if (listed) {
            const uint32_t k = atomicAdd(&L.misc[2], 1u);
            if (k < FS_GATECAP) {
                const uint32_t e0 = L.x.e.pay[L.sq_st[q]];
                L.gate[k] = (uint16_t)g;
                L.x.e.tf[k] = ((uint64_t)((e0 & 0xFFFFu) - 1u) << 32) | (uint64_t)(sid * (uint32_t)ZV + (e0 >> 16));
            }
        }
    }
    __syncthreads();
    const uint32_t ng = L.misc[2];
    if (ng > FS_GATECAP) { fail(); return; }
    FS_MARK(9);
    // ---- phase E + output
    for (uint32_t a = tid; a < ng; a += FS_THREADS) {
        const uint64_t ta = L.x.e.tf[a];
        uint32_t rank = 0;
        for (uint32_t b = 0; b < ng; ++b) rank += L.x.e.tf[b] < ta ? 1u : 0u;
        A.gated[hb + rank] = (uint32_t)L.gate[a];
    }
    for (uint32_t i = tid; i < nrec; i += FS_THREADS) A.ent[hb + i] = L.x.e.pay[i];
    for (uint32_t g = tid; g < nseg; g += FS_THREADS) {
        const uint32_t q = perm[g];
        const uint32_t sc = L.sq_score[q];
        A.seg_id[hb + g] = L.sq_id[q];
        A.seg_start[hb + g] = L.sq_st[q];
        A.seg_score[hb + g] = (int32_t)((sc & 0x7FFFu) | ((sc & FS_OVF16) ? OVF_FLAG : 0u));
    }
    FS_MARK(10);
    if (tid == 0) {
        atomicAdd(&counters[13], 1ull);                   // debug slot 13: strands that went through this kernel with hits
        A.fused[s] = 1;
        A.strand_hits[s] = kept;
        A.hit_base[s] = hb;
        A.nseg[s] = nseg;
        A.nrec[s] = nrec;
        A.ngated[s] = ng;
    }
}

// ------------------------------------------------------------------------------------------------ candidates
struct CandLds {
    int t_loc[2 * SM + 10];
    int t_seed[2 * SM + 10];
};

// index of segment `seg` in the strand's segment table, or -1
__device__ __forceinline__ int seg_find(const uint32_t* __restrict__ seg_id, int nseg, uint32_t seg) {
    int lo = 0, hi = nseg;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (seg_id[mid] < seg) lo = mid + 1; else hi = mid;
    }
    return (lo < nseg && seg_id[lo] == seg) ? lo : -1;
}

// get_read_id_from_offset_list (common/split_database.cpp:15-35), literally
__device__ __forceinline__ int read_id_from_offset(const mhip_offset_t* __restrict__ a, int n, int offset) {
    int left = 0, right = n - 1, mid = (left + right) / 2;
    if (a[right].offset < offset) return right;
    while (left <= right) {
        int o = a[mid].offset, z = a[mid].size;
        if (o <= offset && o + z > offset) return mid;
        if (o + z <= offset) left = mid + 1;
        else right = mid - 1;
        mid = (left + right) / 2;
    }
    return mid;
}

// The same answer from the volume's block table for every position inside a read (the only positions a k-mer hit can have):
// entry b is the read holding base 1024 b, so the read of `offset` is that one or one of the next few.  Anything else (a pad
// base, the "unset" location 0 of find_location landing between reads) takes the literal search.
__device__ __forceinline__ int read_id_lookup(const mhip_offset_t* __restrict__ a, int n, const uint32_t* __restrict__ blk, int offset) {
    if (offset >= 0) {
        int r = (int)blk[offset >> 10];
        while (r + 1 < n && a[r].offset + a[r].size <= offset) ++r;
        const int o = a[r].offset, z = a[r].size;
        if (o <= offset && o + z > offset) return r;
    }
    return read_id_from_offset(a, n, offset);
}

// lane `l` (wave-uniform) of `old` replaced by the wave-uniform value `v`
__device__ __forceinline__ int write_lane(int v, int l, int old) {
    // two scalar operands exceed the constant bus: the lane select goes through m0 (the compiler sets m0 anew before each of its own uses)
    asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(old) : "s"(v), "s"(l));
    return old;
}

// one wave per read: F strand then R strand into one top-MAXC list kept in LDS (12 ints per entry)
__global__ __launch_bounds__(WAVE) __attribute__((amdgpu_waves_per_eu(7, 7))) void seed_cand(SeedArrays AF, SeedArrays AB, const mhip_offset_t* __restrict__ ref_offs, const uint32_t* __restrict__ ref_blk, int ref_nreads,
                                                  int ref_start_id, const mhip_offset_t* __restrict__ roffs, ReadSel sel, int ib,
                                                  int reads_start_id, mhip_params P, mhip_candidate* __restrict__ out,
                                                  int32_t* __restrict__ out_counts, unsigned long long* __restrict__ counters) {
    extern __shared__ __attribute__((aligned(16))) int smem[];
    const int lane = lane_id();
    const int r = blockIdx.x;
    // the list: [maxc][12] in LDS; a list too long for that (-n above 1024: the reference takes any positive -n, pw_options.cpp:9) is
    // built in place in the read's slice of the output table — the same code on global memory, read past the L1 (the wave reads its
    // own earlier stores)
    const bool big = P.maxc > MAXC_LDS;
    int* clist = big ? (int*)(out + (size_t)r * P.maxc) : smem;
    CandLds* T = (CandLds*)(smem + (big ? 0 : P.maxc * 12));
    auto cl_ld = [&](int i) -> int { return big ? __hip_atomic_load(clist + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : clist[i]; };
    auto cl_st = [&](int i, int v) { if (big) __hip_atomic_store(clist + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else clist[i] = v; };
    const int rid = sel_rid(sel, ib + r);
    const int read_id = rid + reads_start_id;
    const int read_size = roffs[rid].size;
    const int MAXC = P.maxc;
    const double cutoff = P.ddfs_cutoff;
    int ncand = 0;

    for (int strand = 0; strand < 2; ++strand) {
        const int s = 2 * r + strand;
        const SeedArrays& A = AF.fused[s] ? AF : AB;          // the strand's tables: from seed_strand or from the kernel chain
        const uint64_t hb = A.hit_base[s];
        const int nseg = (int)A.nseg[s];
        const int ng = (int)A.ngated[s];
        if (A.strand_hits[s] == 0) continue;
        uint32_t* ent = A.ent + hb;
        uint32_t* fin = A.ent_fin + hb;
        const uint32_t* seg_id = A.seg_id + hb;
        const uint32_t* seg_start = A.seg_start + hb;
        int32_t* seg_score = A.seg_score + hb;
        const uint32_t* gated = A.gated + hb;
#define SEG_LIST(g) (((seg_score[g] & OVF_FLAG) ? fin : ent) + seg_start[g])
#define SEG_SCORE(g) (seg_score[g] & ~OVF_FLAG)
#define SET_SCORE(g, v) seg_score[g] = (seg_score[g] & OVF_FLAG) | (v)

        // The immutable part of the next gated segment's record (its index, segment ids, list starts) is fetched one iteration
        // ahead: the walk is a chain of dependent loads (gated -> ids/starts -> scores -> lists), and the scores alone have to
        // be read fresh (earlier iterations zero them).
        int g_n = 0, seg_n = 0, segm1_n = -1;
        uint32_t st_n = 0, stm1_n = 0;
        auto prefetch = [&](int gi) {
            g_n = (int)gated[gi];
            seg_n = (int)seg_id[g_n];
            st_n = seg_start[g_n];
            segm1_n = g_n > 0 ? (int)seg_id[g_n - 1] : -1;
            stm1_n = g_n > 0 ? seg_start[g_n - 1] : 0u;
        };
        if (ng > 0) prefetch(0);
        for (int gi = 0; gi < ng; ++gi) {
            const int g = g_n, seg = seg_n, segm1 = segm1_n;
            const uint32_t st_g = st_n, st_gm1 = stm1_n;
            if (gi + 1 < ng) prefetch(gi + 1);
#ifdef CAND_STATS
            if (lane == 0) atomicAdd(&counters[9], 1ull);
#endif
            const int raw_g = seg_score[g];
            int s_k = raw_g & ~OVF_FLAG;
            if (s_k == 0) continue;
#ifdef CAND_STATS
            if (lane == 0) atomicAdd(&counters[10], 1ull);
#endif
            int start_loc = seg * ZV;
            int loc = 0, raw_gm1 = 0;
            const bool has_left = seg > 0 && g > 0 && segm1 == seg - 1;
            if (has_left) { raw_gm1 = seg_score[g - 1]; loc = raw_gm1 & ~OVF_FLAG; }
            if (loc > 0) start_loc = (seg - 1) * ZV;
            const int n1 = loc > 0 ? min(loc, SM) : 0, n2 = min(s_k, SM);
            const int k = n1 + n2;
            const uint32_t* list_g = ((raw_g & OVF_FLAG) ? fin : ent) + st_g;
            const uint32_t* list_gm1 = ((raw_gm1 & OVF_FLAG) ? fin : ent) + st_gm1;
            __syncthreads();
            for (int i = lane; i < k; i += 64) {
                uint32_t e = i < n1 ? list_gm1[i] : list_g[i - n1];
                T->t_loc[i] = ent_loc(e) + ((i >= n1 && loc > 0) ? ZV : 0);
                T->t_seed[i] = ent_seed(e);
            }
            __syncthreads();
            // ---- find_location (pw_impl.cpp:161-239): lane owns entries i0 = lane and i1 = lane + 64
            const int i0 = lane, i1 = lane + 64;
            const int l0 = i0 < k ? T->t_loc[i0] : 0, d0 = i0 < k ? T->t_seed[i0] : 0;
            const int l1 = i1 < k ? T->t_loc[i1] : 0, d1 = i1 < k ? T->t_seed[i1] : 0;
            int temp0 = d0, temp1 = d1, sc0 = 0, sc1 = 0;
            // entries 64.. exist only when both lists are long (k <= 80); most segment pairs fit one entry per lane
            const int k0 = min(k, 64);
            // Branch-free: every condition is evaluated for every lane and combined with &; the votes an entry receives from
            // the entries before it go through a lane write (each j is written once) and are added after the loops.
            int in0 = 0, in1 = 0;
            auto vote_loops = [&](auto ddf) {
                for (int j = 1; j < k0; ++j) {
                    const int lj = __builtin_amdgcn_readlane(l0, j), dj = __builtin_amdgcn_readlane(d0, j);     // entry j lives in lane j
                    const int ds = dj - d0, dl = lj - l0;
                    const bool v0 = (i0 < j) & (temp0 != dj) & (ds > 0) & (dl > 0) & (dl < read_size) & ddf(dl, ds);
                    sc0 += v0 ? 1 : 0;
                    temp0 = v0 ? dj : temp0;
                    in0 = write_lane(__popcll(__builtin_amdgcn_ballot_w64(v0)), j, in0);
                }
                for (int j = 64; j < k; ++j) {
                    const int lj = __builtin_amdgcn_readlane(l1, j - 64), dj = __builtin_amdgcn_readlane(d1, j - 64);
                    const int ds0 = dj - d0, dl0 = lj - l0, ds1 = dj - d1, dl1 = lj - l1;
                    const bool v0 = (temp0 != dj) & (ds0 > 0) & (dl0 > 0) & (dl0 < read_size) & ddf(dl0, ds0);
                    const bool v1 = (i1 < j) & (temp1 != dj) & (ds1 > 0) & (dl1 > 0) & (dl1 < read_size) & ddf(dl1, ds1);
                    sc0 += v0 ? 1 : 0;
                    temp0 = v0 ? dj : temp0;
                    sc1 += v1 ? 1 : 0;
                    temp1 = v1 ? dj : temp1;
                    in1 = write_lane(__popcll(__builtin_amdgcn_ballot_w64(v0)) + __popcll(__builtin_amdgcn_ballot_w64(v1)), j - 64, in1);
                }
            };
            if (cutoff == 0.25) vote_loops([](int dloc, int dseed) { return ddf_find_quarter(dloc, dseed); });
            else vote_loops([&](int dloc, int dseed) { return (dloc > 0) & (dseed > 0) & ddf_find(dloc, dseed, cutoff); });
            sc0 += in0;
            sc1 += in1;
            int mv = max(i0 < k ? sc0 : -1, i1 < k ? sc1 : -1);
            for (int o = 32; o > 0; o >>= 1) mv = max(mv, __shfl_xor(mv, o));
            const int maxval = mv;
#ifdef CAND_STATS
            if (lane == 0 && maxval < 5) atomicAdd(&counters[11], 1ull);
            if (lane == 0) atomicAdd(&counters[12], (unsigned long long)k);
#endif
            if (maxval < 5) continue;
            const uint64_t eq0 = __ballot(i0 < k && sc0 == maxval), eq1 = __ballot(i1 < k && sc1 == maxval);
            const int maxi = eq0 ? __ffsll((unsigned long long)eq0) - 1 : 64 + __ffsll((unsigned long long)eq1) - 1;
            const int rep = __popcll(eq0) + __popcll(eq1) - 1;
            int rep_loc;
            if (rep == maxval) {
                rep_loc = maxi;
            } else {
                const int lm = T->t_loc[maxi], dm = T->t_seed[maxi];
                // selected = DDF-consistent with maxi (before: "< read_len", after: "<= read_len"), plus maxi itself
                bool q0 = false, q1 = false;
                if (i0 < k) {
                    if (i0 < maxi) q0 = dm - d0 > 0 && lm - l0 > 0 && lm - l0 < read_size && ddf_find(lm - l0, dm - d0, cutoff);
                    else if (i0 == maxi) q0 = true;
                    else q0 = d0 - dm > 0 && l0 - lm > 0 && l0 - lm <= read_size && ddf_find(l0 - lm, d0 - dm, cutoff);
                }
                if (i1 < k) {
                    if (i1 < maxi) q1 = dm - d1 > 0 && lm - l1 > 0 && lm - l1 < read_size && ddf_find(lm - l1, dm - d1, cutoff);
                    else if (i1 == maxi) q1 = true;
                    else q1 = d1 - dm > 0 && l1 - lm > 0 && l1 - lm <= read_size && ddf_find(l1 - lm, d1 - dm, cutoff);
                }
                const uint64_t s0m = __ballot(q0), s1m = __ballot(q1);
                const uint64_t z0 = __ballot(q0 && l0 != 0), z1 = __ballot(q1 && l1 != 0);
                // loc[0] == 0 doubles as "unset" (pw_impl.cpp:198,211,224): first selected entry with a non-zero
                // location wins; if every selected location is 0 the last selected entry wins
                if (z0) rep_loc = __ffsll((unsigned long long)z0) - 1;
                else if (z1) rep_loc = 64 + __ffsll((unsigned long long)z1) - 1;
                else if (s1m) rep_loc = 64 + (63 - __clzll((unsigned long long)s1m));
                else rep_loc = 63 - __clzll((unsigned long long)s0m);
            }
            const int vote = rep_loc < 64 ? __shfl(sc0, rep_loc) : __shfl(sc1, rep_loc - 64);
            if (vote < 2 * P.min_kmer_match + 2) continue;
            const int loc_seed = T->t_seed[rep_loc];
            const int loc_list = start_loc + T->t_loc[rep_loc];
            int sid = read_id_lookup(ref_offs, ref_nreads, ref_blk, loc_list);
            const int sstart = ref_offs[sid].offset, ssize = ref_offs[sid].size;
            const int send = sstart + ssize + 1;
            sid += ref_start_id;
            if (sid > read_id) continue;
            if (sid == read_id) {
                // scrub the read's own region (pw_impl.cpp:371-383); only loczhi is compacted, seedno is not
                int u_k = sstart / ZV;
                int gg = seg_find(seg_id, nseg, (uint32_t)u_k);
                if (gg >= 0) {
                    const int lim = sstart % ZV, cnt = min(SEG_SCORE(gg), SM);
                    uint32_t* L = SEG_LIST(gg);
                    uint32_t e = lane < cnt ? L[lane] : 0;
                    bool keep = lane < cnt && ent_loc(e) < lim;
                    uint64_t km = __ballot(keep);
                    int dstp = __popcll(km & ((1ull << lane) - 1ull));
                    __syncthreads();
                    if (keep) L[dstp] = (L[dstp] & 0xFFFFu) | (e & 0xFFFF0000u);
                    __syncthreads();
                    if (lane == 0) SET_SCORE(gg, __popcll(km));
                }
                ++u_k;
                const int kend = send / ZV;
                {
                    // segments strictly between: score = 0
                    int lo = 0, hi = nseg;
                    while (lo < hi) { int mid = (lo + hi) >> 1; if ((int)seg_id[mid] < u_k) lo = mid + 1; else hi = mid; }
                    for (int x = lo + lane; x < nseg && (int)seg_id[x] < kend; x += 64) SET_SCORE(x, 0);
                }
                const int tail_seg = max(u_k, kend);
                gg = seg_find(seg_id, nseg, (uint32_t)tail_seg);
                __syncthreads();
                if (gg >= 0) {
                    const int lim = send % ZV, cnt = min(SEG_SCORE(gg), SM);
                    uint32_t* L = SEG_LIST(gg);
                    uint32_t e = lane < cnt ? L[lane] : 0;
                    bool keep = lane < cnt && ent_loc(e) > lim;
                    uint64_t km = __ballot(keep);
                    int dstp = __popcll(km & ((1ull << lane) - 1ull));
                    __syncthreads();
                    if (keep) L[dstp] = (L[dstp] & 0xFFFFu) | (e & 0xFFFF0000u);
                    __syncthreads();
                    if (lane == 0) SET_SCORE(gg, __popcll(km));
                }
                __syncthreads();
                continue;
            }
            // ---- geometry (pw_impl.cpp:386-403)
            const int loc2 = (loc_seed - 1) * BC;
            const int left1 = loc_list - sstart + MHIP_KMER_SIZE - 1, right1 = send - loc_list;
            const int left2 = loc2 + MHIP_KMER_SIZE - 1, right2 = read_size - loc2;
            const int num1 = left1 > left2 ? left2 : left1, num2 = right1 > right2 ? right2 : right1;
            if (num1 + num2 < P.min_kmer_dist) continue;
            // ---- neighbour sweeps (pw_impl.cpp:405-438), f64
            int seedcount = 0;
            {
                int nlb = (num1 + ZV - 1) / ZV;
                const int lowest = max(0, seg - nlb);          // segments seg-1 .. lowest
                for (int x = g - 1; x >= 0 && (int)seg_id[x] >= lowest; --x) {
                    const int sc = SEG_SCORE(x);
                    if (sc <= 0) continue;
                    const int sl = (int)seg_id[x] * ZV, scnt = min(sc, SM);
                    bool ok = false;
                    if (lane < scnt) {
                        uint32_t e = SEG_LIST(x)[lane];
                        ok = fabs((double)(loc_list - sl - ent_loc(e)) / ((double)((loc_seed - ent_seed(e)) * BC) * 1.0) - 1.0) < cutoff;
                    }
                    const int agree = __popcll(__ballot(ok));
                    seedcount += agree;
                    if (agree * 1.0 / scnt > 0.4) { if (lane == 0) SET_SCORE(x, 0); }
                    __syncthreads();
                }
                int nrb = (num2 + ZV - 1) / ZV;
                const int highest = seg + nrb;                  // segments seg+1 .. highest
                for (int x = g + 1; x < nseg && (int)seg_id[x] <= highest; ++x) {
                    const int sc = SEG_SCORE(x);
                    if (sc <= 0) continue;
                    const int sl = (int)seg_id[x] * ZV, scnt = min(sc, SM);
                    bool ok = false;
                    if (lane < scnt) {
                        uint32_t e = SEG_LIST(x)[lane];
                        ok = fabs((double)(sl + ent_loc(e) - loc_list) / ((double)((ent_seed(e) - loc_seed) * BC) * 1.0) - 1.0) < cutoff;
                    }
                    const int agree = __popcll(__ballot(ok));
                    seedcount += agree;
                    if (agree * 1.0 / scnt > 0.4) { if (lane == 0) SET_SCORE(x, 0); }
                    __syncthreads();
                }
            }
            const int cscore = vote + seedcount;
            // ---- stable insertion into the descending top-MAXC list (pw_impl.cpp:442-455)
            int ge = 0;
            for (int i = lane; i < ncand; i += 64) ge += cl_ld(i * 12 + 6) >= cscore ? 1 : 0;
            for (int o = 32; o > 0; o >>= 1) ge += __shfl_xor(ge, o);
            const int pos = ge;                                  // == high + 1
            const int last_src = (ncand < MAXC) ? ncand - 1 : ncand - 2;
            __syncthreads();
            for (int top = last_src; top >= pos; top -= 64) {
                const int i = top - lane;
                int v[12];
                if (i >= pos) {
#pragma unroll
                    for (int q = 0; q < 12; ++q) v[q] = cl_ld(i * 12 + q);
                }
                __syncthreads();
                if (i >= pos) {
#pragma unroll
                    for (int q = 0; q < 12; ++q) cl_st((i + 1) * 12 + q, v[q]);
                }
                __syncthreads();
            }
            if (pos < MAXC && lane == 0) {
                const int cv[12] = {loc_list - sstart, loc2, left1, left2, right1, right2, cscore, num1, num2, sid, sstart, strand};
#pragma unroll
                for (int q = 0; q < 12; ++q) cl_st(pos * 12 + q, cv[q]);
            }
            if (ncand < MAXC) ++ncand;
            __syncthreads();
        }
#undef SEG_LIST
#undef SEG_SCORE
#undef SET_SCORE
    }
    __syncthreads();
    int* o = (int*)(out + (size_t)r * MAXC);
    if (!big)
        for (int i = lane; i < ncand * 12; i += 64) o[i] = clist[i];
    if (lane == 0) {
        out_counts[r] = ncand;
        atomicAdd(&counters[2], (unsigned long long)ncand);
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int bits_for(uint32_t maxv) {
    int b = 0;
    while ((1ull << b) <= maxv) ++b;
    return b < 1 ? 1 : b;
}

static bool filter_enabled(const mhip_params* P);
static bool fused_enabled(const mhip_params* P);
static bool wide_filter_enabled(const mhip_params* P);
static bool predrop_enabled();
static bool cuts_enabled(const mhip_params* P);

// the reads with local index in [ib, ie) of the selection
static int seed_batch(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, const ReadSel sel,
                      int ib, int ie, const mhip_params* P, mhip_candidate* d_out, int32_t* d_counts) {
    const int nr = ie - ib, ns = 2 * nr;
    std::vector<uint32_t> kmb((size_t)ns);
    uint64_t sumK = 0;
    for (int r = 0; r < nr; ++r) {
        int L = reads->h_offs[(size_t)sel_rid(sel, ib + r)].size;
        int K = L < MHIP_KMER_SIZE ? 0 : (L - MHIP_KMER_SIZE) / BC + 1;
        kmb[(size_t)2 * r] = (uint32_t)sumK; sumK += (uint64_t)K;
        kmb[(size_t)2 * r + 1] = (uint32_t)sumK; sumK += (uint64_t)K;
    }
    if (sumK >= 0xFFFFFFFFull) { mhip_set_error("seed batch too large"); return -1; }
    SeedArrays A;
    memset(&A, 0, sizeof(A));
    uint32_t* d_kmb;
    if (c->scratch("sd_kmbase", sizeof(uint32_t) * (size_t)ns, (void**)&d_kmb)) return -1;
    if (c->scratch("sd_kmbstart", sizeof(uint32_t) * (size_t)(sumK + 1), (void**)&A.km_bstart)) return -1;
    if (c->scratch("sd_kmcnt", sizeof(uint32_t) * (size_t)(sumK + 1), (void**)&A.km_cnt)) return -1;
    if (c->scratch("sd_hitsall", sizeof(uint32_t) * (size_t)ns, (void**)&A.strand_hits_all)) return -1;
    if (c->scratch("sd_fused", sizeof(int32_t) * (size_t)ns, (void**)&A.fused)) return -1;
    A.km_base = d_kmb;
    HIPCHK(hipMemcpyAsync(d_kmb, kmb.data(), sizeof(uint32_t) * (size_t)ns, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipMemsetAsync(A.fused, 0, sizeof(int32_t) * (size_t)ns, c->stream));
    // a diagonal grid cell (the query volume IS the reference volume): buckets cut behind each read's own copy
    const uint4* recs = (ref == reads && cuts_enabled(P)) ? index_ensure_cuts(c, idx) : nullptr;
    LAUNCH(c, "seed_probe", seed_probe, ns, SEED_BLOCK, 0, (const uint32_t*)reads->d_pac, (const mhip_offset_t*)reads->d_offs, sel, ib,
           (const uint32_t*)idx->d_starts, A, (unsigned long long*)c->d_counters, recs, recs ? idx->cut_step : 1);
    const int gate = 2 * P->min_kmer_match;
    const int nbits = bits_for((uint32_t)(ref->num_bases / ZV));
    // (no subject can have a higher id than a query when the reference volume lies entirely before the query volume: nothing to drop)
    const RefReads RR{(const mhip_offset_t*)ref->d_offs, (const uint32_t*)ref->d_blk2read, ref->num_reads, ref->start_read_id, reads->start_read_id,
                      predrop_enabled() && !(ref != reads && ref->start_read_id + ref->num_reads <= reads->start_read_id) ? 1 : 0, ref == reads ? 1 : 0};

    // ---- the strand-resident pipeline (seed_strand): tables of the strands it takes, in arrays handed out by an atomic cursor
    SeedArrays F = A;
    SeedArrays B = A;
    unsigned int nfallback = (unsigned int)ns;
    // (PacBio gates only.  A form of this kernel behind seed_filter_wide, for the low gates of nanopore mode, was built in round 5 and measured
    // slower than the chain it replaced — 59.4 ms against seed_emit 26.6 + seed_sort_pass 13.7 + seed_build 8.6 = 48.9 ms on 100 000 ONT-style
    // reads of 20 kb, its second bucket walk costing more than the two sort passes and the build it saved — and was removed in round 6.)
    auto run_strand_pipeline = [&]() -> int {
        // room: the relevance filter keeps about one bucket hit in nine; a strand that finds the arrays full takes the kernel chain
        const double hits_per_lookup = (double)idx->num_kmers / (double)NKMER + 2.0;
        size_t capF = (size_t)((double)sumK * hits_per_lookup / 5.0) + (size_t)ns * 256 + 64;
        if (const char* e = getenv("MECAT_SEED_FUSED_ROOM")) capF = (size_t)std::max(64.0, atof(e));      // test knob: entries in the shared arrays
        FusedCtl* d_ctl;
        if (c->scratch("sf_ctl", sizeof(FusedCtl), (void**)&d_ctl)) return -1;
        if (c->scratch("sf_hits", sizeof(uint32_t) * (size_t)ns, (void**)&F.strand_hits)) return -1;
        if (c->scratch("sf_hbase", sizeof(uint64_t) * (size_t)(ns + 1), (void**)&F.hit_base)) return -1;
        if (c->scratch("sf_nseg", sizeof(uint32_t) * (size_t)ns, (void**)&F.nseg)) return -1;
        if (c->scratch("sf_nrec", sizeof(uint32_t) * (size_t)ns, (void**)&F.nrec)) return -1;
        if (c->scratch("sf_ngated", sizeof(uint32_t) * (size_t)ns, (void**)&F.ngated)) return -1;
        if (c->scratch("sf_ent", sizeof(uint32_t) * capF, (void**)&F.ent)) return -1;
        if (c->scratch("sf_entfin", sizeof(uint32_t) * capF, (void**)&F.ent_fin)) return -1;
        if (c->scratch("sf_escore", sizeof(uint16_t) * capF, (void**)&F.escore)) return -1;
        if (c->scratch("sf_segid", sizeof(uint32_t) * capF, (void**)&F.seg_id)) return -1;
        if (c->scratch("sf_segstart", sizeof(uint32_t) * capF, (void**)&F.seg_start)) return -1;
        if (c->scratch("sf_segscore", sizeof(int32_t) * capF, (void**)&F.seg_score)) return -1;
        if (c->scratch("sf_gated", sizeof(uint32_t) * capF, (void**)&F.gated)) return -1;
        HIPCHK(hipMemsetAsync(d_ctl, 0, sizeof(FusedCtl), c->stream));
        LAUNCH(c, "seed_strand", seed_strand, ns, FS_THREADS, 0, (const mhip_offset_t*)reads->d_offs, sel, ib, (const uint16_t*)idx->d_slots,
               (const int32_t*)idx->d_offsets, F, gate, std::max(0, nbits - FLT_BITS), (int)P->min_kmer_match, P->ddfs_cutoff,
               (unsigned long long)capF, d_ctl, (unsigned long long*)c->d_counters, RR);
        FusedCtl ctl;
        HIPCHK(hipMemcpyAsync(&ctl, d_ctl, sizeof(ctl), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));   // also orders the kmb host buffer
        nfallback = ctl.n_fallback;
#ifdef FS_PROF
        for (int i = 0; i < 12; ++i) fprintf(stderr, "FS_PROF phase %d: F %.1f R %.1f us per strand\n", i, (double)ctl.prof[i] / 50.0 / ns, (double)ctl.prof[i + 16] / 50.0 / ns);
#endif
        if (getenv("MECAT_SEED_STATS")) {       // development (tools/dev/seed_stats.py): per-strand sizes of the strand pipeline
            std::vector<uint32_t> h[5];
            const uint32_t* src[5] = {A.strand_hits_all, F.strand_hits, F.nseg, F.nrec, F.ngated};
            for (int i = 0; i < 5; ++i) { h[i].resize((size_t)ns); HIPCHK(hipMemcpy(h[i].data(), src[i], sizeof(uint32_t) * (size_t)ns, hipMemcpyDeviceToHost)); }
            std::vector<int32_t> fu((size_t)ns);
            HIPCHK(hipMemcpy(fu.data(), A.fused, sizeof(int32_t) * (size_t)ns, hipMemcpyDeviceToHost));
            for (int s = 0; s < ns; ++s)
                if (fu[(size_t)s]) fprintf(stderr, "SEEDSTAT %d %u %u %u %u %u\n", s & 1, h[0][(size_t)s], h[1][(size_t)s], h[2][(size_t)s], h[3][(size_t)s], h[4][(size_t)s]);
        }
        return 0;
    };
    if (fused_enabled(P) && run_strand_pipeline()) return -1;

    // ---- the kernel chain for every other strand (with the low gates of nanopore mode its own filter comes first)
    if (nfallback > 0) {
        if (c->scratch("sd_hits", sizeof(uint32_t) * (size_t)ns, (void**)&B.strand_hits)) return -1;
        if (c->scratch("sd_filtered", sizeof(int32_t) * (size_t)ns, (void**)&B.filtered)) return -1;
        if (c->scratch("sd_relbits", sizeof(uint32_t) * (size_t)ns * REL_WORDS, (void**)&B.rel_bits)) return -1;
        if (c->scratch("sd_hbase", sizeof(uint64_t) * (size_t)(ns + 1), (void**)&B.hit_base)) return -1;
        if (c->scratch("sd_nseg", sizeof(uint32_t) * (size_t)ns, (void**)&B.nseg)) return -1;
        if (c->scratch("sd_nrec", sizeof(uint32_t) * (size_t)ns, (void**)&B.nrec)) return -1;
        if (c->scratch("sd_ngated", sizeof(uint32_t) * (size_t)ns, (void**)&B.ngated)) return -1;
        if (wide_filter_enabled(P)) {           // low gates (nanopore mode): 2^18 slots, one workgroup per strand and CU
            B.rel_mask = WF_M - 1;
            B.rel_shift = WF_CELL;
            LAUNCH(c, "seed_filter_wide", seed_filter_wide, ns, FS_THREADS, 0, (const mhip_offset_t*)reads->d_offs, sel, ib,
                   (const int32_t*)idx->d_offsets, B, gate, ref == reads ? 1 : 0, (unsigned long long*)c->d_counters);
        } else {
            B.rel_mask = FLT_M - 1;
            B.rel_shift = 0;
            LAUNCH(c, "seed_filter", seed_filter, ns, SEED_BLOCK, 0, (const mhip_offset_t*)reads->d_offs, sel, ib,
                   (const uint16_t*)idx->d_slots, B, gate, filter_enabled(P) ? 1 : 0);
        }
    }
    if (nfallback > 0) {
        LAUNCH(c, "seed_scan", seed_scan, 1, 1024, 0, (const uint32_t*)B.strand_hits, ns, B.hit_base);
        uint64_t Htot = 0;
        HIPCHK(hipMemcpyAsync(&Htot, B.hit_base + ns, sizeof(uint64_t), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));   // also orders the kmb host buffer
        const size_t Hc = (size_t)Htot + 64;
        if (c->scratch("sd_keysA", sizeof(uint64_t) * Hc, (void**)&B.keysA)) return -1;
        if (c->scratch("sd_keysB", sizeof(uint64_t) * Hc, (void**)&B.keysB)) return -1;
        if (c->scratch("sd_ent", sizeof(uint32_t) * Hc, (void**)&B.ent)) return -1;
        if (c->scratch("sd_entfin", sizeof(uint32_t) * Hc, (void**)&B.ent_fin)) return -1;
        if (c->scratch("sd_escore", sizeof(uint16_t) * Hc, (void**)&B.escore)) return -1;
        if (c->scratch("sd_segid", sizeof(uint32_t) * Hc, (void**)&B.seg_id)) return -1;
        if (c->scratch("sd_segstart", sizeof(uint32_t) * Hc, (void**)&B.seg_start)) return -1;
        if (c->scratch("sd_segscore", sizeof(int32_t) * Hc, (void**)&B.seg_score)) return -1;
        if (c->scratch("sd_segkmlast", sizeof(uint32_t) * Hc, (void**)&B.seg_kmlast)) return -1;
        if (c->scratch("sd_segtfirst", sizeof(uint64_t) * Hc, (void**)&B.seg_tfirst)) return -1;
        if (c->scratch("sd_gated", sizeof(uint32_t) * Hc, (void**)&B.gated)) return -1;
        int in_b = 0;
        if (Htot > 0) {
            LAUNCH(c, "seed_emit", seed_emit, ns, SEED_BLOCK, 0, (const mhip_offset_t*)reads->d_offs, sel, ib, (const int32_t*)idx->d_offsets, B);
#ifdef WF_PROF
            {
                unsigned long long h[16], z[16] = {0};
                HIPCHK(hipStreamSynchronize(c->stream));
                HIPCHK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_wfprof), sizeof(h)));
                HIPCHK(hipMemcpyToSymbol(HIP_SYMBOL(g_wfprof), z, sizeof(z)));
                const char* nm[16] = {"wf zero", "wf walk", "wf hot", "wf kept+out", 0, 0, 0, 0, "emit bitmap", "emit load+count", "emit scan", "emit write"};
                for (int i = 0; i < 16; ++i) if (nm[i]) fprintf(stderr, "[wf prof] %-16s %8.1f us per strand (%d strands)\n", nm[i], (double)h[i] / 100.0 / ns, ns);
            }
#endif
            const int npass = (nbits + SORT_MAXBITS - 1) / SORT_MAXBITS;
            const int per = (nbits + npass - 1) / npass;
            int done = 0;
            for (int p = 0; p < npass; ++p) {
                int b = std::min(per, nbits - done);
                LAUNCH(c, "seed_sort_pass", seed_sort_pass, ns, SEED_BLOCK, 0, B, in_b, KEY_SEG_SHIFT + done, b);
                done += b;
                in_b ^= 1;
            }
        }
        LAUNCH(c, "seed_build", seed_build, ns, SEED_BLOCK, 0, B, in_b, (int)P->min_kmer_match, P->ddfs_cutoff, sel, ib, RR);
    }
    const size_t lds = sizeof(int) * 12 * (size_t)(P->maxc > MAXC_LDS ? 0 : P->maxc) + sizeof(CandLds);
    LAUNCH(c, "seed_cand", seed_cand, nr, WAVE, lds, F, B, (const mhip_offset_t*)ref->d_offs, (const uint32_t*)ref->d_blk2read, ref->num_reads,
           ref->start_read_id,
           (const mhip_offset_t*)reads->d_offs, sel, ib, reads->start_read_id, *P, d_out, d_counts, (unsigned long long*)c->d_counters);
    HIPCHK(hipGetLastError());
    return 0;
}

static bool filter_enabled(const mhip_params* P) {
    // the filter needs a gate high enough to separate signal from random hits; below that every hit is kept
    const char* fe = getenv("MECAT_SEED_FILTER");      // debug knob: 0 disables the relevance filter
    return 2 * P->min_kmer_match >= 6 && !(fe && atoi(fe) == 0);
}

static bool cuts_enabled(const mhip_params* P) {
    (void)P;
    const char* e = getenv("MECAT_SEED_CUTS");         // debug knob: 0 walks every bucket to its end
    return !(e && atoi(e) == 0);
}

static bool predrop_enabled() {
    const char* e = getenv("MECAT_SEED_PREDROP");      // debug knob: 0 lists every gated segment
    return !(e && atoi(e) == 0);
}

static bool wide_filter_enabled(const mhip_params* P) {
    const char* fe = getenv("MECAT_SEED_FILTER");      // the same knob: 0 disables every relevance filter
    const int gate = 2 * P->min_kmer_match;
    return gate >= 4 && gate < 6 && !(fe && atoi(fe) == 0);
}

static bool fused_enabled(const mhip_params* P) {
    const char* fe = getenv("MECAT_SEED_FUSED");       // debug knob: 0 sends every strand through the kernel chain
    return filter_enabled(P) && !(fe && atoi(fe) == 0);
}

// reads per launch: bounded by an estimate of the bucket hits they produce.  The batch arrays cost ~54 bytes per KEPT
// hit (about one hit in seven survives the relevance filter); seed_cand runs one wave per read and is latency bound,
// so larger batches are what fills the chip.
static int next_batch_end(const mhip_index* idx, const mhip_volume* reads, const ReadSel sel, int rb, int re, const mhip_params* P) {
    const double hits_per_lookup = (double)idx->num_kmers / (double)NKMER + 2.0;
    double budget = filter_enabled(P) ? 3.2e9 : wide_filter_enabled(P) ? 1.6e9 : 400e6;   // ~25 GB of batch arrays either way (1.6e9: 5 ms more per config-2 pass, tail of the one-wave-per-read kernel)
    // With the relevance filter on and memory to spare the whole of a config-2 cell goes through in one launch of each kernel instead of
    // three (~60 GB of batch arrays; 57.7 against 58.8 ms per pass: two tails of the one-wave-per-read kernel and two host round trips less)
    if (filter_enabled(P)) {
        size_t fr = 0, tot = 0;
        // in proportion to what is free right now (the batch arrays cost ~6 bytes per estimated hit; a third of the free memory at most:
        // several ranks folded onto one device, or another process on it, each see less and ask for less — ADVICE r03)
        if (hipMemGetInfo(&fr, &tot) == hipSuccess) budget = std::max(budget, std::min(1.0e10, (double)fr / 3.0 / 6.0));
    }
    if (const char* e = getenv("MECAT_SEED_BATCH_HITS")) budget = std::max(1e6, atof(e));   // tuning knob
    double acc = 0;
    int r = rb;
    while (r < re) {
        int L = reads->h_offs[(size_t)sel_rid(sel, r)].size;
        double k = L < MHIP_KMER_SIZE ? 0 : (double)((L - MHIP_KMER_SIZE) / BC + 1);
        acc += 2.0 * k * hits_per_lookup;
        ++r;
        if (acc > budget || r - rb >= (1 << 20)) break;
    }
    return r;
}

extern "C" {

static int seed_selection(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, const ReadSel sel, int n,
                          const mhip_params* P, void* d_out, void* d_out_counts) {
    HIPCHK(hipSetDevice(c->device));
    if (n < 0 || sel.chunk < 1 || sel.cstride < 1 || sel.rid0 < 0 || (n > 0 && sel_rid(sel, n - 1) >= reads->num_reads)) {
        mhip_set_error("bad read selection begin %d chunk %d stride %d n %d", sel.rid0, sel.chunk, sel.cstride, n);
        return -1;
    }
    if (P->maxc < 1 || P->maxc > MAXC_LIMIT) { mhip_set_error("maxc %d outside 1..%d", P->maxc, MAXC_LIMIT); return -1; }
    if (ref->num_reads == 0) { mhip_set_error("empty reference volume"); return -1; }
    int ib = 0;
    while (ib < n) {
        int ie = next_batch_end(idx, reads, sel, ib, n, P);
        if (seed_batch(c, idx, ref, reads, sel, ib, ie, P, (mhip_candidate*)d_out + (size_t)ib * P->maxc, (int32_t*)d_out_counts + ib))
            return -1;
        ib = ie;
    }
    return 0;
}

int mhip_seed_reads_strided_dev(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int rid_begin,
                                int rid_stride, int n, const mhip_params* P, void* d_out, void* d_out_counts) {
    if (n > 0 && (int64_t)rid_begin + (int64_t)(n - 1) * rid_stride >= reads->num_reads) {
        mhip_set_error("bad read selection begin %d stride %d n %d", rid_begin, rid_stride, n);
        return -1;
    }
    return seed_selection(c, idx, ref, reads, ReadSel{rid_begin, 1, rid_stride}, n, P, d_out, d_out_counts);
}

// local reads of a chunked shard (mecat_hip.h: mhip_shard_*): local index i -> read rid0 + (i / chunk) * chunk * nranks + i % chunk
int mhip_seed_reads_chunked_dev(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int rid0, int chunk,
                                int nranks, int n, const mhip_params* P, void* d_out, void* d_out_counts) {
    if (chunk < 1 || nranks < 1 || (int64_t)chunk * nranks > 0x7fffffff) { mhip_set_error("bad shard: chunk %d ranks %d", chunk, nranks); return -1; }
    return seed_selection(c, idx, ref, reads, ReadSel{rid0, chunk, chunk * nranks}, n, P, d_out, d_out_counts);
}

int mhip_seed_reads_dev(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int rid_begin,
                        int rid_end, const mhip_params* P, void* d_out, void* d_out_counts) {
    if (rid_begin < 0 || rid_end > reads->num_reads || rid_begin > rid_end) { mhip_set_error("bad read range [%d,%d)", rid_begin, rid_end); return -1; }
    return mhip_seed_reads_strided_dev(c, idx, ref, reads, rid_begin, 1, rid_end - rid_begin, P, d_out, d_out_counts);
}

int mhip_seed_reads(mhip_ctx* c, const mhip_index* idx, const mhip_volume* ref, const mhip_volume* reads, int rid_begin,
                    int rid_end, const mhip_params* P, mhip_candidate* out, int32_t* out_counts) {
    HIPCHK(hipSetDevice(c->device));
    const size_t n = (size_t)std::max(0, rid_end - rid_begin);
    if (n == 0) return 0;
    mhip_candidate* d_out;
    int32_t* d_cnt;
    if (c->scratch("sd_out", sizeof(mhip_candidate) * n * (size_t)P->maxc, (void**)&d_out)) return -1;
    if (c->scratch("sd_outcnt", sizeof(int32_t) * n, (void**)&d_cnt)) return -1;
    if (mhip_seed_reads_dev(c, idx, ref, reads, rid_begin, rid_end, P, d_out, d_cnt)) return -1;
    HIPCHK(hipMemcpyAsync(out_counts, d_cnt, sizeof(int32_t) * n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipMemcpyAsync(out, d_out, sizeof(mhip_candidate) * n * (size_t)P->maxc, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return 0;
}

}  // extern "C"
