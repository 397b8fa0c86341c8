// common.h — shared host/device declarations of libmecat_hip (gfx950 only; wave = 64 lanes).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "mecat_hip.h"

#define WAVE 64
#define NKMER (1u << 26)          // 4^13 direct-address buckets
#define KMER_MASK 0x3FFFFFFu
#define ZV 2000                   // reference segment length, mecat2pw/pw_impl.h:16
#define BC 10                     // query k-mer stride, pw_impl.h:12
#define SM 40                     // seeds kept per segment, pw_impl.h:13
#define MAX_BUCKET 128            // lookup_table.cpp:97

void mhip_set_error(const char* fmt, ...);

#define HIPCHK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t _e = (expr);                                                                        \
        if (_e != hipSuccess) {                                                                        \
            mhip_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e));       \
            return -1;                                                                                 \
        }                                                                                              \
    } while (0)

struct KStat {
    int64_t launches = 0;
    double ms = 0.0;
};

struct PendingEv {
    std::string name;
    hipEvent_t a, b;
};

struct DevBuf {
    void* p = nullptr;
    size_t cap = 0;
    uint64_t gen = 0;             // bumped by every (re)allocation: what state that outlives a call is keyed on (an address can come back)
};

struct mhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    bool profiling = false;
    std::map<std::string, KStat> stats;
    std::vector<PendingEv> pending;
    std::vector<hipEvent_t> ev_pool;
    std::map<std::string, DevBuf> bufs;   // named scratch buffers, grown on demand, freed with the context
    std::mutex bufs_mu;                   // scratch() vs mhip_ctx_reserve_index on another thread
    int64_t* d_counters = nullptr;        // 8 work counters (see mecat_hip.h)
    int num_cus = 256;
    // asm_seed's per-wave directories and record pools (scratch "as_dir", "as_pool", "as_hw"): a completed call leaves them in the state
    // the next one needs, so they are set up once per buffer and layout (asm_seed.hip); as_clean_nrec = waves whose directories are zero
    const void* as_clean_base = nullptr;
    const void* as_clean_dir = nullptr;
    uint64_t as_clean_gen[3] = {0, 0, 0};   // allocation generations of as_pool, as_dir, as_hw the clean state belongs to
    int as_clean_nseg = 0, as_clean_pcap = 0;
    size_t as_clean_nrec = 0;
    int ae_n = 0;                         // jobs and dense words of the last mhip_asm_extend_run (what mhip_asm_extend_fetch copies)
    int64_t ae_total = 0;
    int ix_s1_ranges = -1;                // index build, ix_scatter1's tile order: -1 not measured yet, 0 tile = block, 1 a contiguous eighth of the tiles per XCD

    // returns a device buffer of at least `bytes` (contents undefined)
    int scratch(const char* name, size_t bytes, void** out);
    uint64_t scratch_generation(const char* name);      // 0: never allocated
    int drain_events();
    hipEvent_t get_event();
};

struct mhip_volume {
    int device = 0;
    int num_reads = 0;
    int num_bases = 0;          // incl. one pad base per read
    int start_read_id = 0;
    uint32_t* d_pac = nullptr;  // 2-bit store as big-endian-in-byte bytes, padded with >= 64 zero bytes
    uint32_t* d_npac = nullptr; // optional second plane in the same layout: 3 at every base that is not A, C, G or T (code 0 in d_pac), else 0;
                                // only the overlappers for corrected reads set it (mhip_volume_set_nplane), and only for input that holds an N
    mhip_offset_t* d_offs = nullptr;
    uint32_t* d_blk2read = nullptr;   // [num_bases / 1024 + 2] read holding base 1024 b (or the one before it when that base is a pad)
    std::vector<mhip_offset_t> h_offs;
    size_t pac_bytes = 0;
};

struct mhip_index {
    int device = 0;
    uint32_t* d_starts = nullptr;   // [NKMER + 1] bucket boundaries into d_offsets (dropped buckets are empty)
    int32_t* d_offsets = nullptr;   // [num_kmers] k-mer start positions, ascending inside a bucket
    uint16_t* d_slots = nullptr;    // [num_kmers] (position / 2000) mod 2^15: all the relevance filter's bucket walk needs, at half the bytes
    int64_t num_kmers = 0;
    int num_bases = 0;
    // bucket records for the seeding of a diagonal grid cell (index.hip: index_ensure_cuts): per k-mer id {start, occurrences below
    // each of seven position cuts, occurrences}; built on first use
    uint4* d_recs = nullptr;
    int cut_step = 0;               // cut t (1..7) = position t * cut_step
    std::mutex recs_mu;
    int max_bucket = MAX_BUCKET;    // buckets with more occurrences are empty (128: lookup_table.cpp:97; 256: mecat2asmpw.c:311)
    size_t cap_starts = 0, cap_offsets = 0, cap_slots = 0, cap_recs = 0;   // allocation sizes, for the recycler below
};

// Recycler for the index's two large device arrays: a per-read-volume rebuild (one per `-j` grid row) otherwise pays a
// multi-GB hipMalloc + hipFree (~0.2 s) every time.  Freed blocks are parked per device (at most four) and handed back
// to the next request they fit without more than 2x slack; mhip_ctx_destroy releases what is parked.
int dev_alloc_recycled(int device, size_t bytes, void** p, size_t* cap);
void dev_free_recycled(int device, void* p, size_t cap);
void dev_recycler_release(int device);
// the bucket records of `idx` (built by the first caller; nullptr when the index cannot have them: bucket cap above 255)
const uint4* index_ensure_cuts(mhip_ctx* c, const mhip_index* idx);

// launch helper: optional HIP-event timing per kernel name on the context's stream
struct LaunchTimer {
    mhip_ctx* ctx;
    hipEvent_t a = nullptr, b = nullptr;
    const char* name;
    LaunchTimer(mhip_ctx* c, const char* n) : ctx(c), name(n) {
        if (ctx->profiling) {
            a = ctx->get_event();
            b = ctx->get_event();
            (void)hipEventRecord(a, ctx->stream);
        }
    }
    ~LaunchTimer() {
        if (ctx->profiling) {
            (void)hipEventRecord(b, ctx->stream);
            ctx->pending.push_back({name, a, b});
        }
    }
};

#define LAUNCH(ctx, name, kernel, grid, block, shmem, ...)                                   \
    do {                                                                                     \
        LaunchTimer _lt((ctx), (name));                                                      \
        hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), (shmem), (ctx)->stream, __VA_ARGS__); \
    } while (0)

#ifdef __HIPCC__
// ---------------------------------------------------------------- device helpers

// 16 bases per 32-bit word, first base in the most significant bits (packed_db.h:98-107 read as big-endian words)
__device__ __forceinline__ uint32_t pac_word(const uint32_t* __restrict__ pac, int64_t w) {
    return __builtin_bswap32(pac[w]);
}
// code (0..3) of volume base `idx`
__device__ __forceinline__ uint32_t pac_base(const uint32_t* __restrict__ pac, int64_t idx) {
    uint32_t w = pac_word(pac, idx >> 4);
    return (w >> ((~(uint32_t)idx & 15u) << 1)) & 3u;
}
// the 13-mer starting at volume base `idx` (MSB-first id, lookup_table.cpp:77-90 / pw_impl.cpp:83-97)
__device__ __forceinline__ uint32_t pac_kmer(const uint32_t* __restrict__ pac, int64_t idx) {
    int64_t w = idx >> 4;
    uint64_t W = ((uint64_t)pac_word(pac, w) << 32) | pac_word(pac, w + 1);
    int sh = 64 - 26 - (int)((idx & 15) << 1);
    return (uint32_t)(W >> sh) & KMER_MASK;
}
// reverse complement of a 13-mer id
__device__ __forceinline__ uint32_t kmer_revcomp(uint32_t k) {
    uint32_t x = ~k;                                        // complement every 2-bit code (3 - c)
    x = ((x >> 2) & 0x33333333u) | ((x & 0x33333333u) << 2);
    x = ((x >> 4) & 0x0F0F0F0Fu) | ((x & 0x0F0F0F0Fu) << 4);
    x = __builtin_bswap32(x);                               // now the 16 2-bit groups are reversed
    return (x >> 6) & KMER_MASK;                            // the 13 wanted groups were the low 26 bits -> top 26 bits
}
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
#endif
