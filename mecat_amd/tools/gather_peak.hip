// gather_peak.hip — how fast can an MI355X fetch short runs at random addresses?  The ceiling of the bucket walks of
// seed_filter / seed_emit (mecat_amd/csrc/seed.hip): a 13-mer bucket is ~22 ascending entries somewhere in a multi-GB array,
// read by 16 lanes, 2 bytes (filter: table slots) or 4 bytes (emit: positions) per entry.
//
//   gather_peak [array_MB] [iters]      prints one JSON line: G runs/s and GB/s of useful bytes per variant
//
// Variants: run length in bytes (32 .. 128, aligned to 4 only, so a run straddles 64-byte lines as buckets do), runs in flight per
// 16-lane group (1, 2, 4, 8), waves per SIMD (2, 4, 8).  Addresses come from a per-group LCG (no address loads: this is the
// ceiling WITHOUT the bucket-table indirection), every loaded word is folded into a checksum so nothing is dropped.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int INFL, int WORDS /*4-byte words per lane: run = 16 lanes * WORDS * 4 B*/>
__global__ __launch_bounds__(256) void gather(const uint32_t* __restrict__ a, uint64_t nwords, int iters, uint32_t* __restrict__ out) {
    const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) >> 4, sub = threadIdx.x & 15u;
    uint64_t st = 0x9E3779B97F4A7C15ull * (grp + 1u);
    uint32_t acc = 0;
    const uint64_t span = nwords - 64;
    for (int it = 0; it < iters; ++it) {
        uint32_t v[INFL][WORDS];
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t base = (uint64_t)(((st >> 24) * (unsigned __int128)span) >> 40);      // uniform word index
#pragma unroll
            for (int w = 0; w < WORDS; ++w) v[q][w] = a[base + (uint64_t)w * 16u + sub];
        }
#pragma unroll
        for (int q = 0; q < INFL; ++q)
#pragma unroll
            for (int w = 0; w < WORDS; ++w) acc += v[q][w];
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int INFL, int WORDS>
static double run(const uint32_t* d, uint64_t nwords, int blocks, int iters, uint32_t* d_out) {
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    gather<INFL, WORDS><<<blocks, 256>>>(d, nwords, 2, d_out);
    CHK(hipEventRecord(e0));
    gather<INFL, WORDS><<<blocks, 256>>>(d, nwords, iters, d_out);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double runs = (double)blocks * 16.0 * iters * INFL;
    return runs / (ms * 1e-3) * 1e-9;          // G runs / s
}

int main(int argc, char** argv) {
    const uint64_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 6144;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const uint64_t nwords = mb * 1024 * 1024 / 4;
    uint32_t *d, *d_out;
    CHK(hipMalloc(&d, nwords * 4));
    CHK(hipMemset(d, 1, nwords * 4));
    CHK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"array_MB\": %llu, \"cus\": %d, \"unit\": \"G runs/s (run = 16 lanes x bytes_per_lane at a random 4-byte-aligned address)\", \"rows\": [", (unsigned long long)mb, cus);
    bool first = true;
    auto row = [&](int wps, int infl, int words, double g) {
        printf("%s{\"waves_per_simd\": %d, \"in_flight\": %d, \"run_bytes\": %d, \"g_runs_per_s\": %.2f, \"useful_GBps\": %.0f}", first ? "" : ", ", wps, infl, words * 64,
               g, g * words * 64);
        first = false;
    };
    for (int wps : {2, 4, 8}) {
        const int blocks = cus * wps * 8;      // wps waves per SIMD = wps blocks of 4 waves per CU, 8 rounds of blocks
#define R(I, W) row(wps, I, W, run<I, W>(d, nwords, blocks, iters, d_out))
        R(1, 1); R(2, 1); R(4, 1); R(8, 1);
        R(1, 2); R(2, 2); R(4, 2); R(8, 2);
#undef R
    }
    printf("]}\n");
    return 0;
}
