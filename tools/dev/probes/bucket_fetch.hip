// bucket_fetch.hip — dev probe: how should a group of lanes fetch one 128-byte bucket at a random 4-byte-aligned address?
// Same bytes, different shapes: LANES lanes x VEC dwords per lane x PIECES load instructions.  Prints G buckets/s per shape.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
template <int VEC> struct V;
template <> struct V<1> { typedef uint32_t t; };
template <> struct V<2> { typedef uint2 t; };
template <> struct V<4> { typedef uint4 t; };
__device__ inline uint32_t fold(uint32_t v) { return v; }
__device__ inline uint32_t fold(uint2 v) { return v.x + v.y; }
__device__ inline uint32_t fold(uint4 v) { return v.x + v.y + v.z + v.w; }
template <int LANES, int VEC, int PIECES, int INFL, int ALIGN_WORDS = 1>
__global__ __launch_bounds__(256) void fetch(const uint32_t* __restrict__ a, uint64_t nwords, int iters, uint32_t* __restrict__ out) {
    const uint32_t grp = (blockIdx.x * 256u + threadIdx.x) / LANES, sub = threadIdx.x % LANES;
    uint64_t st = 0x9E3779B97F4A7C15ull * (grp + 1u);
    uint32_t acc = 0;
    const uint64_t span = nwords - 256;
    typedef typename V<VEC>::t T;
    for (int it = 0; it < iters; ++it) {
        T v[INFL][PIECES];
#pragma unroll
        for (int q = 0; q < INFL; ++q) {
            st = st * 6364136223846793005ull + 1442695040888963407ull;
            const uint64_t base = (uint64_t)(((st >> 24) * (unsigned __int128)span) >> 40) & ~(uint64_t)(ALIGN_WORDS - 1);      // (ALIGN_WORDS = 8: buckets that start on 32-byte boundaries)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) {
                const uint32_t* src = a + base + (uint64_t)(p * LANES + sub) * VEC;
                if (VEC == 1) v[q][p] = *(const T*)src;
                else __builtin_memcpy(&v[q][p], src, sizeof(T));      // 4-byte aligned only
            }
        }
#pragma unroll
        for (int q = 0; q < INFL; ++q)
#pragma unroll
            for (int p = 0; p < PIECES; ++p) acc += fold(v[q][p]);
    }
    if (acc == 0x12345678u) out[0] = acc;
}
template <int LANES, int VEC, int PIECES, int INFL, int ALIGN_WORDS = 1>
static void run(const uint32_t* d, uint64_t nwords, int cus, int wps, int iters, uint32_t* d_out) {
    const int blocks = cus * wps * 8;
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    fetch<LANES, VEC, PIECES, INFL, ALIGN_WORDS><<<blocks, 256>>>(d, nwords, 2, d_out);
    CHK(hipEventRecord(e0));
    fetch<LANES, VEC, PIECES, INFL, ALIGN_WORDS><<<blocks, 256>>>(d, nwords, iters, d_out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double buckets = (double)blocks * (256.0 / LANES) * iters * INFL;
    printf("align %2d B  bytes %4d  lanes %2d x %2d B x %d pieces, %d in flight, %d waves/SIMD: %7.2f G buckets/s  %6.0f GB/s\n", ALIGN_WORDS * 4, LANES * VEC * 4 * PIECES, LANES, VEC * 4, PIECES, INFL, wps,
           buckets / (ms * 1e-3) * 1e-9, buckets / (ms * 1e-3) * 1e-9 * LANES * VEC * 4 * PIECES);
}
int main(int argc, char** argv) {
    const uint64_t mb = argc > 1 ? strtoull(argv[1], 0, 10) : 8192;
    const int iters = argc > 2 ? atoi(argv[2]) : 64;
    const uint64_t nwords = mb * 1024 * 1024 / 4;
    uint32_t *d, *d_out;
    CHK(hipMalloc(&d, nwords * 4)); CHK(hipMemset(d, 1, nwords * 4)); CHK(hipMalloc(&d_out, 64));
    hipDeviceProp_t prop; CHK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    for (int wps : {4, 8}) {
        // 128-byte buckets
        run<8, 1, 4, 2>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 2, 2>(d, nwords, cus, wps, iters, d_out);
        run<16, 2, 1, 2>(d, nwords, cus, wps, iters, d_out);
        run<8, 4, 1, 2>(d, nwords, cus, wps, iters, d_out);
        run<8, 4, 1, 4>(d, nwords, cus, wps, iters, d_out);
        run<4, 4, 2, 2>(d, nwords, cus, wps, iters, d_out);
        // 64-byte buckets
        run<8, 1, 2, 2>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 1, 2>(d, nwords, cus, wps, iters, d_out);
        run<8, 2, 1, 2>(d, nwords, cus, wps, iters, d_out);
        run<4, 4, 1, 2>(d, nwords, cus, wps, iters, d_out);
        run<4, 4, 1, 4>(d, nwords, cus, wps, iters, d_out);
        // 32-byte and 256-byte
        run<8, 1, 1, 4>(d, nwords, cus, wps, iters, d_out);
        run<2, 4, 1, 4>(d, nwords, cus, wps, iters, d_out);
        run<16, 4, 1, 2>(d, nwords, cus, wps, iters, d_out);
        // the same bytes from 32- and 64-byte aligned starts
        run<8, 1, 1, 4, 8>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 1, 2, 8>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 1, 2, 16>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 2, 2, 8>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 2, 2, 16>(d, nwords, cus, wps, iters, d_out);
        run<16, 1, 2, 2, 32>(d, nwords, cus, wps, iters, d_out);
    }
    return 0;
}
