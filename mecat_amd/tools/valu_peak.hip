// valu_peak — calibrates the instruction-issue ceilings that bench.py prices the dw kernel against.
//
// The d-path kernels (align.hip) are integer VALU work on LDS-resident state; their HBM fraction is tiny by construction
// (SURVEY.md §8d), so the interpretable ceiling is the rate at which the chip issues wave64 instructions.  This program
// measures that rate on the device it runs on, per instruction class, for 1, 2, 4 and 8 resident waves per SIMD: eight
// independent register streams per wave, 64 instructions per loop iteration.  Output: one JSON object on stdout, rates in
// G wave-instructions/s over the whole chip (1024 SIMDs x clock / rate = issue cycles per instruction and SIMD).
//
//   hipcc --offload-arch=gfx950 -O3 mecat_amd/tools/valu_peak.hip -o mecat_amd/bin/valu_peak
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define REP8(X) X X X X X X X X

// operands: %0..%7 = vector registers a0..a7 (read-write), %8..%11 = scalar registers s0..s3 (read-write),
//           %12 = vector register b, %13 = 64-bit scalar mask m, %14 = vector register holding an LDS byte address
#define DEF_KERNEL_PRE(NAME, PRE, BODY)                                                                                            \
    __global__ void __launch_bounds__(256) k_##NAME(int iters, uint32_t* out) {                                             \
        __shared__ uint32_t lds[1040];                                                                                    \
        uint32_t a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
        uint32_t b = (blockIdx.x & 7) | 1, s0 = blockIdx.x, s1 = 1, s2 = 2, s3 = 3;                                          \
        uint64_t m = 0x5555aaaa3333ccccull ^ blockIdx.x;                                                                   \
        lds[threadIdx.x] = a0; lds[threadIdx.x + 256] = a1; lds[threadIdx.x + 512] = a2; lds[threadIdx.x + 768] = a3;       \
        __syncthreads();                                                                                                   \
        uint32_t la = (threadIdx.x & 255) * 4;                                                                             \
        asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 s[24:25], %0, %1\n s_mov_b64 s[26:27], %2\n" PRE : : "v"(a0), "v"(b), "s"(m) : "vcc", "s24", "s25", "s26", "s27"); \
        for (int i = 0; i < iters; ++i) {                                                                                  \
            REP8(asm volatile(BODY : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+s"(s0), \
                              "+s"(s1), "+s"(s2), "+s"(s3) : "v"(b), "s"(m), "v"(la) : "vcc", "scc", "memory", "s20", "s21", "s22", "v28", "v29", "v30", "v31", "s24", "s25", "s26", "s27");)               \
        }                                                                                                                  \
        uint32_t r = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7 ^ s0 ^ s1 ^ s2 ^ s3;                                             \
        if (r == 0x12345678u) out[0] = r + lds[r & 1023];                                                                  \
    }

#define DEF_KERNEL(NAME, BODY) DEF_KERNEL_PRE(NAME, "", BODY)

// eight independent streams: I(dst, next) once per vector register
#define S8(I) I("%0", "%1") I("%1", "%2") I("%2", "%3") I("%3", "%4") I("%4", "%5") I("%5", "%6") I("%6", "%7") I("%7", "%0")
// scalar streams over s0..s3, twice
#define SS8(I) I("%8") I("%9") I("%10") I("%11") I("%8") I("%9") I("%10") I("%11")

#define I_ADD(d, n) "v_add_u32 " d ", " d ", %12\n"
#define I_SUB(d, n) "v_sub_u32 " d ", " d ", %12\n"
#define I_AND(d, n) "v_and_b32 " d ", " d ", %12\n"
#define I_XOR(d, n) "v_xor_b32 " d ", " d ", %12\n"
#define I_MAX(d, n) "v_max_i32 " d ", " d ", %12\n"
#define I_MINU(d, n) "v_min_u32 " d ", " d ", %12\n"
#define I_LSHL(d, n) "v_lshlrev_b32 " d ", %12, " d "\n"
#define I_LSHR_LIT(d, n) "v_lshrrev_b32 " d ", 3, " d "\n"
#define I_MUL24(d, n) "v_mul_i32_i24 " d ", " d ", %12\n"
#define I_MOV(d, n) "v_mov_b32 " d ", " n "\n"
#define I_FFBH(d, n) "v_ffbh_u32 " d ", " n "\n"
#define I_NOT(d, n) "v_not_b32 " d ", " d "\n"
#define I_BFREV(d, n) "v_bfrev_b32 " d ", " d "\n"
#define I_CNDMASK_VCC(d, n) "v_cndmask_b32 " d ", " d ", %12, vcc\n"
#define I_CNDMASK_SGPR(d, n) "v_cndmask_b32 " d ", " d ", %12, %13\n"
#define I_CMP_VCC(d, n) "v_cmp_lt_u32 vcc, " d ", %12\n"
#define I_CMP_SGPR(d, n) "v_cmp_lt_u32 s[20:21], " d ", %12\n"
#define I_ALIGNBIT(d, n) "v_alignbit_b32 " d ", " d ", " n ", %12\n"
#define I_ALIGNBIT_LIT(d, n) "v_alignbit_b32 " d ", " d ", " n ", 6\n"
#define I_MIN3(d, n) "v_min3_u32 " d ", " d ", " n ", %12\n"
#define I_MAX3(d, n) "v_max3_i32 " d ", " d ", " n ", %12\n"
#define I_MED3(d, n) "v_med3_i32 " d ", " d ", " n ", %12\n"
#define I_ADD3(d, n) "v_add3_u32 " d ", " d ", " n ", %12\n"
#define I_LSHL_ADD(d, n) "v_lshl_add_u32 " d ", " d ", 2, %12\n"
#define I_LSHL_OR(d, n) "v_lshl_or_b32 " d ", " d ", 2, %12\n"
#define I_AND_OR(d, n) "v_and_or_b32 " d ", " d ", " n ", %12\n"
#define I_BFE(d, n) "v_bfe_u32 " d ", " d ", 3, 5\n"
#define I_BFE_V(d, n) "v_bfe_u32 " d ", " d ", %12, 5\n"
#define I_MAD24(d, n) "v_mad_u32_u24 " d ", " d ", %12, " n "\n"
#define I_MULLO(d, n) "v_mul_lo_u32 " d ", " d ", %12\n"
#define I_PERM(d, n) "v_perm_b32 " d ", " d ", " n ", %12\n"
#define I_ADD_E64_S(d, n) "v_add_u32_e64 " d ", " d ", %8\n"
#define I_SUB_E64_S(d, n) "v_sub_u32_e64 " d ", " d ", %8\n"
#define I_ADD_CO(d, n) "v_add_co_u32 " d ", vcc, " d ", %12\n"
#define I_PK_ADD_U16(d, n) "v_pk_add_u16 " d ", " d ", %12\n"
#define I_PK_SUB_I16(d, n) "v_pk_sub_i16 " d ", " d ", %12\n"
#define I_PK_MAX_I16(d, n) "v_pk_max_i16 " d ", " d ", %12\n"
#define I_PK_MIN_U16(d, n) "v_pk_min_u16 " d ", " d ", %12\n"
#define I_PK_LSHL_B16(d, n) "v_pk_lshlrev_b16 " d ", %12, " d "\n"
#define I_DPP_MAX(d, n) "v_max_i32_dpp " d ", " n ", " d " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_MOV(d, n) "v_mov_b32_dpp " d ", " n " row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define I_DPP_ADD_BCAST(d, n) "v_add_u32_dpp " d ", " n ", " d " row_bcast:15 row_mask:0xa bank_mask:0xf\n"
#define I_SDWA_ADD(d, n) "v_add_u32_sdwa " d ", " d ", %12 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n"
#define I_READLANE(d, n) "v_readlane_b32 s22, " d ", 5\n"
#define I_READFIRST(d, n) "v_readfirstlane_b32 s22, " d "\n"
#define I_PERMLANE16_SWAP(d, n) "v_permlane16_swap_b32 " d ", " n "\n"
#define I_PERMLANE32_SWAP(d, n) "v_permlane32_swap_b32 " d ", " n "\n"
#define I_MBCNT(d, n) "v_mbcnt_lo_u32_b32 " d ", %8, " d "\n"
#define I_DS_READ(d, n) "ds_read_b32 " d ", %14\n"
#define I_DS_READ2(d, n) "ds_read2_b32 v[30:31], %14 offset1:1\n"
#define I_DS_READ_U16(d, n) "ds_read_u16 " d ", %14\n"
#define I_DS_WRITE(d, n) "ds_write_b32 %14, " d "\n"
#define I_DS_WRITE16(d, n) "ds_write_b16 %14, " d "\n"
#define I_DS_BPERMUTE(d, n) "ds_bpermute_b32 " d ", %14, " n "\n"
#define I_DS_SWIZZLE(d, n) "ds_swizzle_b32 " d ", " n " offset:swizzle(SWAP,1)\n"
#define I_S_ADD(s) "s_add_u32 " s ", " s ", 1\n"
#define I_S_AND(s) "s_and_b32 " s ", " s ", 0x7fffffff\n"
#define I_S_CSEL(s) "s_cmp_lg_u32 " s ", 0\n s_cselect_b32 " s ", " s ", 7\n"
#define I_S_BCNT(s) "s_bcnt1_i32_b64 " s ", %13\n"
#define I_S_FF1(s) "s_ff1_i32_b64 " s ", %13\n"
#define I_S_ANDN2_64(s) "s_andn2_b64 s[20:21], %13, exec\n"
#define I_S_NOP(s) "s_nop 0\n"
#define I_MIX_VS(d, n) "v_add_u32 " d ", " d ", %12\n s_add_u32 %8, %8, 1\n"
#define I_MIX_FULL_HALF(d, n) "v_add_u32 " d ", " d ", %12\n v_alignbit_b32 " d ", " d ", " n ", %12\n"
#define I_MIX_HALF_SALU(d, n) "v_alignbit_b32 " d ", " d ", " n ", %12\n s_add_u32 %8, %8, 1\n"
#define I_DEP_ADD(d, n) "v_add_u32 %0, %0, %12\n"
#define I_DEP_ALIGNBIT(d, n) "v_alignbit_b32 %0, %0, %1, %12\n"
#define I_CNDMASK_VALU_SGPR(d, n) "v_cndmask_b32 " d ", " d ", %12, s[24:25]\n"
#define I_CNDMASK_SALU_SGPR(d, n) "v_cndmask_b32 " d ", " d ", %12, s[26:27]\n"
#define I_CMP_CNDMASK(d, n) "v_cmp_lt_u32 vcc, " n ", %12\n v_cndmask_b32 " d ", " d ", %12, vcc\n"
#define I_CMP_CNDMASK_SGPR(d, n) "v_cmp_lt_u32 s[20:21], " n ", %12\n v_cndmask_b32 " d ", " d ", %12, s[20:21]\n"
#define I_LSHR(d, n) "v_lshrrev_b32 " d ", %12, " d "\n"
#define I_LSHL_LIT(d, n) "v_lshlrev_b32 " d ", 3, " d "\n"
#define I_ASHR_LIT(d, n) "v_ashrrev_i32 " d ", 3, " d "\n"
#define I_OR(d, n) "v_or_b32 " d ", " d ", %12\n"
#define I_SUBREV(d, n) "v_subrev_u32 " d ", " d ", %12\n"
#define I_AND_LIT32(d, n) "v_and_b32 " d ", 0x12345, " d "\n"
#define I_AND_INL(d, n) "v_and_b32 " d ", 15, " d "\n"
#define I_ADD_SGPR_E32(d, n) "v_add_u32 " d ", %8, " d "\n"
#define I_ADD_INL(d, n) "v_add_u32 " d ", 1, " d "\n"
#define I_MAXU(d, n) "v_max_u32 " d ", " d ", %12\n"
#define I_MINI(d, n) "v_min_i32 " d ", " d ", %12\n"
#define I_MAX_I16(d, n) "v_max_i16 " d ", " d ", %12\n"
#define I_ADD_U16(d, n) "v_add_u16 " d ", " d ", %12\n"
#define I_XNOR(d, n) "v_xnor_b32 " d ", " d ", %12\n"
#define I_BCNT(d, n) "v_bcnt_u32_b32 " d ", " d ", %12\n"
#define I_FFBL(d, n) "v_ffbl_b32 " d ", " n "\n"
#define I_CVT(d, n) "v_cvt_f32_u32 " d ", " d "\n"
#define I_FADD(d, n) "v_add_f32 " d ", " d ", %12\n"
#define I_FMAX(d, n) "v_max_f32 " d ", " d ", %12\n"
#define I_FMA(d, n) "v_fma_f32 " d ", " d ", %12, " n "\n"
#define I_SAD(d, n) "v_sad_u32 " d ", " d ", " n ", %12\n"
#define I_MOV_SGPR(d, n) "v_mov_b32 " d ", %8\n"
#define I_ACCVGPR(d, n) "v_accvgpr_write_b32 a0, " d "\n"
#define I_DS_READ_B64(d, n) "ds_read_b64 v[30:31], %14\n"
#define I_DS_READ_B128(d, n) "ds_read_b128 v[28:31], %14\n"
#define I_DS_READ_U8(d, n) "ds_read_u8 " d ", %14\n"
#define I_MIX_DS_VALU(d, n) "ds_read_b32 " d ", %14\n v_add_u32 %12, %12, %12\n v_add_u32 %12, %12, %12\n v_add_u32 %12, %12, %12\n"
#define I_CMP_3CNDMASK(d, n) "v_cmp_lt_u32 vcc, " n ", %12\n v_cndmask_b32 " d ", " d ", %12, vcc\n v_cndmask_b32 " d ", " d ", " n ", vcc\n v_cndmask_b32 " d ", %12, " d ", vcc\n"
#define I_CMP_3CNDMASK_SGPR(d, n) "v_cmp_lt_u32 s[20:21], " n ", %12\n v_cndmask_b32 " d ", " d ", %12, s[20:21]\n v_cndmask_b32 " d ", " d ", " n ", s[20:21]\n v_cndmask_b32 " d ", %12, " d ", s[20:21]\n"
#define I_SMOV_VCC_CNDMASK(d, n) "s_mov_b64 vcc, %13\n v_cndmask_b32 " d ", " d ", %12, vcc\n"
#define I_MAX_I16_PAIR(d, n) "v_max_i16 " d ", " d ", %12\n v_add_u16 " d ", " d ", %12\n"
#define I_MIX_ADD_MAXI32(d, n) "v_add_u32 " d ", " d ", %12\n v_add_u32 " d ", " d ", %12\n v_add_u32 " d ", " d ", %12\n v_max_i32 " d ", " d ", %12\n"
// the d-path row body's own mix (tools/isa_mix.py on dw_extend2: 64 % of the VALU instructions in the 4-cycle class, 0.36 SALU
// and 0.09 LDS instructions per VALU instruction), independent streams: 7 four-cycle + 4 two-cycle VALU, 4 SALU, 1 LDS read
#define I_MIX_DW(d, n) "v_alignbit_b32 " d ", " d ", " n ", %12\n s_add_u32 %8, %8, 1\n v_add_u32 " d ", " d ", %12\n v_min_i32 " d ", " d ", %12\n" \
                       "v_cmp_lt_u32 s[20:21], " n ", %12\n s_and_b64 s[24:25], s[20:21], %13\n v_cndmask_b32 " d ", " d ", %12, s[24:25]\n v_xor_b32 " d ", " d ", %12\n" \
                       "v_lshl_add_u32 " d ", " d ", 1, %12\n s_add_u32 %9, %9, 3\n v_sub_u32 " d ", " d ", %12\n v_max_i32 " d ", " d ", %12\n" \
                       "ds_read_b32 v30, %14\n v_ffbh_u32 " d ", " d "\n s_bcnt1_i32_b64 %10, %13\n v_add_u32 " d ", " d ", %12\n"
#define I_LSHR_B64(d, n) "v_lshrrev_b64 v[28:29], " d ", v[30:31]\n"
#define I_LSHR_B64_S(d, n) "v_lshrrev_b64 v[28:29], " d ", %13\n"
#define I_DS_WRITE_B64(d, n) "ds_write_b64 %14, v[30:31]\n"
#define I_BFM(s) "s_bfm_b64 s[20:21], " s ", 0\n"
#define WAIT "s_waitcnt lgkmcnt(0)\n"

#define KERNELS(X)                                                                                                      \
    X(v_add_u32, S8(I_ADD), 8) X(v_sub_u32, S8(I_SUB), 8) X(v_and_b32, S8(I_AND), 8) X(v_xor_b32, S8(I_XOR), 8)                 \
    X(v_max_i32, S8(I_MAX), 8) X(v_min_u32, S8(I_MINU), 8) X(v_lshlrev_b32, S8(I_LSHL), 8) X(v_lshrrev_b32_lit, S8(I_LSHR_LIT), 8) \
    X(v_mul_i32_i24, S8(I_MUL24), 8) X(v_mov_b32, S8(I_MOV), 8) X(v_ffbh_u32, S8(I_FFBH), 8) X(v_not_b32, S8(I_NOT), 8)         \
    X(v_bfrev_b32, S8(I_BFREV), 8) X(v_cndmask_vcc, S8(I_CNDMASK_VCC), 8) X(v_cndmask_sgpr, S8(I_CNDMASK_SGPR), 8)              \
    X(v_cmp_vcc, S8(I_CMP_VCC), 8) X(v_cmp_sgpr, S8(I_CMP_SGPR), 8) X(v_alignbit_b32, S8(I_ALIGNBIT), 8)                        \
    X(v_alignbit_b32_lit, S8(I_ALIGNBIT_LIT), 8) X(v_min3_u32, S8(I_MIN3), 8) X(v_max3_i32, S8(I_MAX3), 8) X(v_med3_i32, S8(I_MED3), 8) \
    X(v_add3_u32, S8(I_ADD3), 8) X(v_lshl_add_u32, S8(I_LSHL_ADD), 8) X(v_lshl_or_b32, S8(I_LSHL_OR), 8) X(v_and_or_b32, S8(I_AND_OR), 8) \
    X(v_bfe_u32_lit, S8(I_BFE), 8) X(v_bfe_u32, S8(I_BFE_V), 8) X(v_mad_u32_u24, S8(I_MAD24), 8) X(v_mul_lo_u32, S8(I_MULLO), 8)    \
    X(v_perm_b32, S8(I_PERM), 8) X(v_add_u32_e64_sgpr, S8(I_ADD_E64_S), 8) X(v_sub_u32_e64_sgpr, S8(I_SUB_E64_S), 8)             \
    X(v_add_co_u32, S8(I_ADD_CO), 8) X(v_pk_add_u16, S8(I_PK_ADD_U16), 8) X(v_pk_sub_i16, S8(I_PK_SUB_I16), 8)                   \
    X(v_pk_max_i16, S8(I_PK_MAX_I16), 8) X(v_pk_min_u16, S8(I_PK_MIN_U16), 8) X(v_pk_lshlrev_b16, S8(I_PK_LSHL_B16), 8)          \
    X(v_max_i32_dpp, "s_nop 1\n" S8(I_DPP_MAX), 8) X(v_mov_b32_dpp, "s_nop 1\n" S8(I_DPP_MOV), 8)                                \
    X(v_add_u32_dpp_bcast, "s_nop 1\n" S8(I_DPP_ADD_BCAST), 8) X(v_add_u32_sdwa, S8(I_SDWA_ADD), 8)                              \
    X(v_readlane_b32, S8(I_READLANE), 8) X(v_readfirstlane_b32, S8(I_READFIRST), 8)                                              \
    X(v_permlane16_swap_b32, S8(I_PERMLANE16_SWAP), 8) X(v_permlane32_swap_b32, S8(I_PERMLANE32_SWAP), 8)                        \
    X(v_mbcnt_lo, S8(I_MBCNT), 8) X(ds_read_b32, S8(I_DS_READ) WAIT, 8) X(ds_read2_b32, S8(I_DS_READ2) WAIT, 8)                  \
    X(ds_read_u16, S8(I_DS_READ_U16) WAIT, 8) X(ds_write_b32, S8(I_DS_WRITE) WAIT, 8) X(ds_write_b16, S8(I_DS_WRITE16) WAIT, 8)  \
    X(ds_bpermute_b32, S8(I_DS_BPERMUTE) WAIT, 8) X(ds_swizzle_b32, S8(I_DS_SWIZZLE) WAIT, 8)                                    \
    X(s_add_u32, SS8(I_S_ADD), 8) X(s_and_b32, SS8(I_S_AND), 8) X(s_cmp_cselect, SS8(I_S_CSEL), 16) X(s_bcnt1_i32_b64, SS8(I_S_BCNT), 8) \
    X(s_ff1_i32_b64, SS8(I_S_FF1), 8) X(s_andn2_b64, SS8(I_S_ANDN2_64), 8) X(s_nop_0, SS8(I_S_NOP), 8)                           \
    X(mix_vadd_sadd, S8(I_MIX_VS), 16) X(mix_vadd_valignbit, S8(I_MIX_FULL_HALF), 16) X(mix_valignbit_sadd, S8(I_MIX_HALF_SALU), 16) \
    X(dep_v_add_u32, S8(I_DEP_ADD), 8) X(dep_v_alignbit_b32, S8(I_DEP_ALIGNBIT), 8)                                              \
    X(v_cndmask_valu_written_sgpr, S8(I_CNDMASK_VALU_SGPR), 8) X(v_cndmask_salu_written_sgpr, S8(I_CNDMASK_SALU_SGPR), 8)           \
    X(pair_v_cmp_vcc_v_cndmask, S8(I_CMP_CNDMASK), 16) X(pair_v_cmp_sgpr_v_cndmask, S8(I_CMP_CNDMASK_SGPR), 16)                   \
    X(v_lshrrev_b32, S8(I_LSHR), 8) X(v_lshlrev_b32_lit, S8(I_LSHL_LIT), 8) X(v_ashrrev_i32_lit, S8(I_ASHR_LIT), 8)               \
    X(v_or_b32, S8(I_OR), 8) X(v_subrev_u32, S8(I_SUBREV), 8) X(v_and_b32_lit32, S8(I_AND_LIT32), 8) X(v_and_b32_inline, S8(I_AND_INL), 8) \
    X(v_add_u32_sgpr_e32, S8(I_ADD_SGPR_E32), 8) X(v_add_u32_inline, S8(I_ADD_INL), 8) X(v_max_u32, S8(I_MAXU), 8)               \
    X(v_min_i32, S8(I_MINI), 8) X(v_max_i16, S8(I_MAX_I16), 8) X(v_add_u16, S8(I_ADD_U16), 8) X(v_xnor_b32, S8(I_XNOR), 8)        \
    X(v_bcnt_u32_b32, S8(I_BCNT), 8) X(v_ffbl_b32, S8(I_FFBL), 8) X(v_cvt_f32_u32, S8(I_CVT), 8) X(v_add_f32, S8(I_FADD), 8)      \
    X(v_max_f32, S8(I_FMAX), 8) X(v_fma_f32, S8(I_FMA), 8) X(v_sad_u32, S8(I_SAD), 8) X(v_mov_b32_sgpr, S8(I_MOV_SGPR), 8)       \
    X(ds_read_b64, S8(I_DS_READ_B64) WAIT, 8) X(ds_read_b128, S8(I_DS_READ_B128) WAIT, 8) X(ds_read_u8, S8(I_DS_READ_U8) WAIT, 8) \
    X(mix_ds_read_3vadd, S8(I_MIX_DS_VALU) WAIT, 32) X(seq_v_cmp_vcc_3cndmask, S8(I_CMP_3CNDMASK), 32)                            \
    X(seq_v_cmp_sgpr_3cndmask, S8(I_CMP_3CNDMASK_SGPR), 32) X(seq_s_mov_vcc_cndmask, S8(I_SMOV_VCC_CNDMASK), 16)                 \
    X(mix_max_i16_add_u16, S8(I_MAX_I16_PAIR), 16) X(mix_3vadd_1vmax, S8(I_MIX_ADD_MAXI32), 32)                                   \
    X(v_lshrrev_b64, S8(I_LSHR_B64), 8) X(v_lshrrev_b64_sgpr, S8(I_LSHR_B64_S), 8) X(ds_write_b64, S8(I_DS_WRITE_B64) WAIT, 8)          \
    X(s_bfm_b64, SS8(I_BFM), 8)                                                                                                   \
    X(mix_dw_rowbody, S8(I_MIX_DW) WAIT, 128)

#define X_DEF(NAME, BODY, N) DEF_KERNEL(NAME, BODY)
KERNELS(X_DEF)

typedef void (*kernel_t)(int, uint32_t*);
struct Entry { const char* name; kernel_t k; int per_block; };
#define X_ENT(NAME, BODY, N) {#NAME, k_##NAME, N},
static const Entry kEntries[] = {KERNELS(X_ENT)};

static int run_one(const Entry& e, int num_cus, int waves_per_simd, int iters, uint32_t* d_out, double* ginstr_per_s) {
    // 256-thread workgroups = one wave on each of the CU's 4 SIMDs; waves_per_simd workgroups per CU
    const int grid = num_cus * waves_per_simd;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0));
    CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, iters / 8 + 1, d_out);          // warm-up
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    hipLaunchKernelGGL(e.k, dim3(grid), dim3(256), 0, 0, iters, d_out);
    CHK(hipEventRecord(e1));
    CHK(hipEventSynchronize(e1));
    float ms = 0;
    CHK(hipEventElapsedTime(&ms, e0, e1));
    const double instr = (double)grid * 4.0 * (double)iters * 8.0 * e.per_block;
    *ginstr_per_s = instr / (ms * 1e-3) / 1e9;
    CHK(hipEventDestroy(e0));
    CHK(hipEventDestroy(e1));
    return 0;
}

int main(int argc, char** argv) {
    int iters = 4000;
    const char* only = NULL;
    if (argc > 1) iters = atoi(argv[1]);
    if (argc > 2) only = argv[2];
    hipDeviceProp_t prop;
    CHK(hipSetDevice(0));
    CHK(hipGetDeviceProperties(&prop, 0));
    uint32_t* d_out;
    CHK(hipMalloc(&d_out, 64));
    const int cus = prop.multiProcessorCount;
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"iters\": %d, \"unit\": \"G wave-instr/s\", \"waves_per_simd\": [1, 2, 4, 8], \"rates\": {",
           prop.gcnArchName, cus, prop.clockRate / 1000, iters);
    const int occ[4] = {1, 2, 4, 8};
    bool first = true;
    for (const Entry& e : kEntries) {
        if (only) {                                      // comma-separated exact names
            bool hit = false;
            const size_t nl = strlen(e.name);
            for (const char* q = only; q && *q; q = strchr(q, ',') ? strchr(q, ',') + 1 : NULL)
                if (!strncmp(q, e.name, nl) && (q[nl] == ',' || q[nl] == 0)) { hit = true; break; }
            if (!hit) continue;
        }
        printf("%s\n \"%s\": [", first ? "" : ",", e.name);
        first = false;
        for (int o = 0; o < 4; ++o) {
            double g = 0;
            if (run_one(e, cus, occ[o], iters, d_out, &g)) return 2;
            printf("%s%.1f", o ? ", " : "", g);
        }
        printf("]");
    }
    printf("\n}}\n");
    (void)hipFree(d_out);
    return 0;
}
