// dev probe: what a large host buffer costs as a D2H target — hipHostMalloc vs posix_memalign + THP + parallel first touch + hipHostRegister,
// and the D2H rate into each (and into pageable memory).   hipcc -O2 pin_probe.hip -o pin_probe -lpthread ; ./pin_probe [GB]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <chrono>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv) {
    const size_t gb = argc > 1 ? atol(argv[1]) : 8;
    const size_t bytes = gb << 30;
    void* d; CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 1, bytes)); CK(hipDeviceSynchronize());
    hipStream_t s; CK(hipStreamCreate(&s));
    auto d2h = [&](void* h, const char* what) {
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s);
            double t = now() - t0;
            printf("  D2H %zu GB into %s: %.3f s = %.1f GB/s\n", gb, what, t, gb * 1.073741824 / t);
        }
        return 0;
    };
    { double t0 = now(); void* h; CK(hipHostMalloc(&h, bytes, hipHostMallocDefault)); printf("hipHostMalloc %zu GB: %.3f s\n", gb, now() - t0); d2h(h, "hipHostMalloc");
      t0 = now(); CK(hipHostFree(h)); printf("hipHostFree: %.3f s\n", now() - t0); }
    { void* h = nullptr; double t0 = now(); if (posix_memalign(&h, 2 << 20, bytes)) return 1; madvise(h, bytes, MADV_HUGEPAGE);
      const int T = 32; std::vector<std::thread> th;
      for (int t = 0; t < T; ++t) th.emplace_back([=] { char* p = (char*)h + bytes / T * t; for (size_t o = 0; o < bytes / T; o += 4096) p[o] = 0; });
      for (auto& x : th) x.join();
      printf("posix_memalign + THP + first touch on %d threads: %.3f s\n", T, now() - t0);
      d2h(h, "pageable (touched)");
      t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); printf("hipHostRegister (touched pages): %.3f s\n", now() - t0);
      d2h(h, "registered");
      t0 = now(); CK(hipHostUnregister(h)); printf("hipHostUnregister: %.3f s\n", now() - t0); free(h); }
    { void* h = nullptr; if (posix_memalign(&h, 2 << 20, bytes)) return 1; madvise(h, bytes, MADV_HUGEPAGE);
      double t0 = now(); CK(hipHostRegister(h, bytes, hipHostRegisterDefault)); printf("hipHostRegister (untouched pages): %.3f s\n", now() - t0);
      d2h(h, "registered");
      CK(hipHostUnregister(h)); free(h); }
    // host memcpy from a pinned staging buffer to a touched pageable one, 16 threads
    { void *a, *b; CK(hipHostMalloc(&a, 1ull << 30, 0)); b = malloc(1ull << 30); memset(b, 0, 1ull << 30); memset(a, 1, 1ull << 30);
      for (int T : {4, 16, 32}) { double t0 = now(); std::vector<std::thread> th; const size_t n = 1ull << 30;
        for (int t = 0; t < T; ++t) th.emplace_back([=] { memcpy((char*)b + n / T * t, (char*)a + n / T * t, n / T); }); for (auto& x : th) x.join();
        printf("host memcpy 1 GB pinned -> pageable on %d threads: %.1f GB/s\n", T, 1.0737 / (now() - t0)); } }
    return 0;
}
