// dev probe: does a wave64 VALU instruction get cheaper when whole 16-lane groups are switched off in EXEC?
// hipcc --offload-arch=gfx950 -O3 exec_skip.hip -o exec_skip && ./exec_skip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
template <int OP>
__global__ __launch_bounds__(256) void k(unsigned long long mask, uint32_t* out, int iters) {
    uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = blockIdx.x | 3;
    uint32_t* dst = out + blockIdx.x * 256 + threadIdx.x;
    asm volatile("" : "+v"(dst), "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(b));      // everything in registers before EXEC is cut
    unsigned long long saved;
    asm volatile("s_mov_b64 %0, exec\n\ts_and_b64 exec, exec, %1" : "=&s"(saved) : "s"(mask) : "scc");
    for (int i = 0; i < iters; ++i) {
        if (OP == 0) {
            asm volatile("v_alignbit_b32 %0, %0, %8, %9\n\tv_alignbit_b32 %1, %1, %8, %9\n\tv_alignbit_b32 %2, %2, %8, %9\n\tv_alignbit_b32 %3, %3, %8, %9\n\t"
                         "v_alignbit_b32 %4, %4, %8, %9\n\tv_alignbit_b32 %5, %5, %8, %9\n\tv_alignbit_b32 %6, %6, %8, %9\n\tv_alignbit_b32 %7, %7, %8, %9"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b), "v"(b));
        } else {
            asm volatile("v_add_u32 %0, %0, %8\n\tv_add_u32 %1, %1, %8\n\tv_add_u32 %2, %2, %8\n\tv_add_u32 %3, %3, %8\n\t"
                         "v_add_u32 %4, %4, %8\n\tv_add_u32 %5, %5, %8\n\tv_add_u32 %6, %6, %8\n\tv_add_u32 %7, %7, %8"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(b));
        }
    }
    asm volatile("s_mov_b64 exec, %0" : : "s"(saved));
    *dst = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
}
int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount, blocks = cus * 8, iters = 20000;      // 8 blocks of 4 waves per CU: 8 waves per SIMD
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    const unsigned long long masks[] = {~0ull, 0x00000000ffffffffull, 0x0000ffff0000ffffull, 0x000000000000ffffull, 0x00ff00ff00ff00ffull, 0x1ull};
    const char* names[] = {"all 64", "low 32", "16 of each half", "low 16", "8 of each 16", "one lane"};
    for (int op = 0; op < 2; ++op)
        for (int m = 0; m < 6; ++m) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (op == 0) k<0><<<blocks, 256>>>(masks[m], d, iters); else k<1><<<blocks, 256>>>(masks[m], d, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double insts = (double)blocks * 4 * iters * 8;
            printf("%-12s exec %-16s %8.3f ms  %7.1f G wave-instr/s  (%.2f cycles per instr and SIMD at 2.4 GHz)\n", op ? "v_add_u32" : "v_alignbit", names[m], ms,
                   insts / ms / 1e6, 2.4e9 * (ms / 1e3) * (cus * 4) / insts);
        }
    return 0;
}
