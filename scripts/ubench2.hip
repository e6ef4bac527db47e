// ubench2.hip — which VALU encodings issue at 2 cycles/wave and which at 4 on gfx950.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)
#define REP8(x) x x x x x x x x
// 8 independent in-place ops on a..h; extra inputs: %8 = VGPR x, %9 = VGPR y, %10 = SGPR s
#define BODY(T) REP8(asm volatile(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(x), "v"(y), "s"(s0) : "vcc");)

#define T_ADD_VV(i)   "v_add_u32 %" #i ", %8, %" #i "\n"
#define T_ADD_IMM(i)  "v_add_u32 %" #i ", 17, %" #i "\n"
#define T_ADD_LIT(i)  "v_add_u32 %" #i ", 0x12345, %" #i "\n"
#define T_ADD_SV(i)   "v_add_u32 %" #i ", %10, %" #i "\n"
#define T_AND_VV(i)   "v_and_b32 %" #i ", %8, %" #i "\n"
#define T_XOR_VV(i)   "v_xor_b32 %" #i ", %8, %" #i "\n"
#define T_OR_LIT(i)   "v_or_b32 %" #i ", 0x10001, %" #i "\n"
#define T_MOV(i)      "v_mov_b32 %" #i ", %8\n"
#define T_SHR_IMM(i)  "v_lshrrev_b32 %" #i ", 24, %" #i "\n"
#define T_SHR_VV(i)   "v_lshrrev_b32 %" #i ", %8, %" #i "\n"
#define T_SHL_VV(i)   "v_lshlrev_b32 %" #i ", %" #i ", %8\n"
#define T_ASHR(i)     "v_ashrrev_i32 %" #i ", 31, %" #i "\n"
#define T_SUB_VV(i)   "v_sub_u32 %" #i ", %8, %" #i "\n"
#define T_SUBREV(i)   "v_subrev_u32 %" #i ", %8, %" #i "\n"
#define T_ADDCO(i)    "v_add_co_u32 %" #i ", vcc, %8, %" #i "\n"
#define T_ADDC(i)     "v_addc_co_u32 %" #i ", vcc, %8, %" #i ", vcc\n"
#define T_CNDMASK(i)  "v_cndmask_b32 %" #i ", %8, %" #i ", vcc\n"
#define T_CMP(i)      "v_cmp_gt_u32 vcc, %8, %" #i "\n"
#define T_MIN(i)      "v_min_u32 %" #i ", %8, %" #i "\n"
#define T_FMA(i)      "v_fma_f32 %" #i ", %8, %9, %" #i "\n"
#define T_FMAC(i)     "v_fmac_f32 %" #i ", %8, %9\n"
#define T_MULF(i)     "v_mul_f32 %" #i ", %8, %" #i "\n"
#define T_ADDF(i)     "v_add_f32 %" #i ", %8, %" #i "\n"
#define T_BFE_U(i)    "v_bfe_u32 %" #i ", %" #i ", 18, 9\n"
#define T_BFI(i)      "v_bfi_b32 %" #i ", %8, %9, %" #i "\n"
#define T_ALIGNBIT(i) "v_alignbit_b32 %" #i ", %8, %" #i ", 7\n"
#define T_ANDOR(i)    "v_and_or_b32 %" #i ", %" #i ", %8, %9\n"
#define T_LSHLOR(i)   "v_lshl_or_b32 %" #i ", %" #i ", 2, %8\n"
#define T_ADD3(i)     "v_add3_u32 %" #i ", %" #i ", %8, %9\n"
#define T_PERM(i)     "v_perm_b32 %" #i ", %" #i ", %8, %9\n"
#define T_MADU24(i)   "v_mad_u32_u24 %" #i ", %8, %9, %" #i "\n"
#define T_MULU24(i)   "v_mul_u32_u24 %" #i ", %8, %" #i "\n"
#define T_MULLO(i)    "v_mul_lo_u32 %" #i ", %8, %" #i "\n"
#define T_ADD_E64(i)  "v_add_u32_e64 %" #i ", %8, %" #i "\n"
#define T_XOR_E64(i)  "v_xor_b32_e64 %" #i ", %8, %" #i "\n"
#define T_ADD_DPP(i)  "v_add_u32_dpp %" #i ", %8, %" #i " quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define T_PKADD_F(i)  "v_pk_add_f32 %" #i ", %" #i ", %" #i "\n"
#define T_CVT(i)      "v_cvt_f32_u32 %" #i ", %" #i "\n"
#define T_SDWA_ADD(i) "v_add_u32_sdwa %" #i ", %8, %" #i " dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n"
#define T_XOR_SV(i)   "v_xor_b32 %" #i ", %10, %" #i "\n"
#define T_SHL_SV(i)   "v_lshlrev_b32 %" #i ", %" #i ", %10\n"
#define T_SHR_SW(i)   "v_lshrrev_b32 %" #i ", %" #i ", %10\n"
#define T_READLANE(i) "v_readfirstlane_b32 s20, %" #i "\n"
#define T_ADD_U16(i)  "v_add_u16 %" #i ", %8, %" #i "\n"
#define T_ADD_I32(i)  "v_add_i32 %" #i ", %8, %" #i "\n"
#define T_MAD_I24(i)  "v_mad_i32_i24 %" #i ", %8, %9, %" #i "\n"
#define T_DOT2(i)     "v_dot2_i32_i16 %" #i ", %8, %9, %" #i "\n"
#define T_PKSUB(i)    "v_pk_sub_i16 %" #i ", %" #i ", %8\n"
#define T_PKASHR(i)   "v_pk_ashrrev_i16 %" #i ", 15, %" #i "\n"
#define T_PKLSHL(i)   "v_pk_lshlrev_b16 %" #i ", %8, %" #i "\n"

template <int OP>
__global__ __launch_bounds__(256) void ub(uint32_t *out, int iters, uint32_t s0)
{
    uint32_t a = threadIdx.x, b = a * 3u, c = 5u, d = 7u, e = 11u, f = 13u, g = 17u, h = 19u;
    uint32_t x = a ^ 0x55u, y = a + 99u;
    for (int it = 0; it < iters; ++it) {
#define CASE(n, T) if (OP == n) { BODY(T) }
        CASE(0, T_ADD_VV) CASE(1, T_ADD_IMM) CASE(2, T_ADD_LIT) CASE(3, T_ADD_SV) CASE(4, T_AND_VV) CASE(5, T_XOR_VV)
        CASE(6, T_OR_LIT) CASE(7, T_MOV) CASE(8, T_SHR_IMM) CASE(9, T_SHR_VV) CASE(10, T_SHL_VV) CASE(11, T_ASHR)
        CASE(12, T_SUB_VV) CASE(13, T_SUBREV) CASE(14, T_ADDCO) CASE(15, T_ADDC) CASE(16, T_CNDMASK) CASE(17, T_CMP)
        CASE(18, T_MIN) CASE(19, T_FMA) CASE(20, T_FMAC) CASE(21, T_MULF) CASE(22, T_ADDF) CASE(23, T_BFE_U)
        CASE(24, T_BFI) CASE(25, T_ALIGNBIT) CASE(26, T_ANDOR) CASE(27, T_LSHLOR) CASE(28, T_ADD3) CASE(29, T_PERM)
        CASE(30, T_MADU24) CASE(31, T_MULU24) CASE(32, T_MULLO) CASE(33, T_ADD_E64) CASE(34, T_XOR_E64) CASE(35, T_ADD_DPP)
        CASE(36, T_CVT) CASE(37, T_SDWA_ADD) CASE(38, T_XOR_SV) CASE(39, T_SHL_SV) CASE(40, T_SHR_SW)
        CASE(41, T_ADD_U16) CASE(42, T_MAD_I24) CASE(43, T_DOT2) CASE(44, T_PKSUB) CASE(45, T_PKASHR) CASE(46, T_PKLSHL)
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h;
}

template <int OP>
void run(const char *name)
{
    const int w = 4, blocks = 256 * w, iters = 2000;
    uint32_t *out;
    HIPCHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    ub<OP><<<blocks, 256>>>(out, 10, 3);
    HIPCHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HIPCHK(hipEventRecord(e0));
        ub<OP><<<blocks, 256>>>(out, iters, 3);
        HIPCHK(hipEventRecord(e1));
        HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    printf("%-34s %7.3f ms  %5.2f cyc/wave-instr/SIMD @2.4GHz\n", name, best, best * 1e-3 * 2.4e9 / ((double) iters * 64 * w));
    HIPCHK(hipFree(out));
}

int main()
{
#define R(n, s) run<n>(s);
    R(0, "v_add_u32 v,v") R(1, "v_add_u32 inline-imm") R(2, "v_add_u32 literal") R(3, "v_add_u32 sgpr") R(4, "v_and_b32 v,v") R(5, "v_xor_b32 v,v")
    R(6, "v_or_b32 literal") R(7, "v_mov_b32") R(8, "v_lshrrev_b32 imm") R(9, "v_lshrrev_b32 v,v") R(10, "v_lshlrev_b32 v,v") R(11, "v_ashrrev_i32 31")
    R(12, "v_sub_u32") R(13, "v_subrev_u32") R(14, "v_add_co_u32") R(15, "v_addc_co_u32") R(16, "v_cndmask_b32 vcc") R(17, "v_cmp_gt_u32")
    R(18, "v_min_u32") R(19, "v_fma_f32") R(20, "v_fmac_f32") R(21, "v_mul_f32") R(22, "v_add_f32") R(23, "v_bfe_u32 imm")
    R(24, "v_bfi_b32") R(25, "v_alignbit_b32") R(26, "v_and_or_b32") R(27, "v_lshl_or_b32") R(28, "v_add3_u32") R(29, "v_perm_b32")
    R(30, "v_mad_u32_u24") R(31, "v_mul_u32_u24") R(32, "v_mul_lo_u32") R(33, "v_add_u32_e64") R(34, "v_xor_b32_e64") R(35, "v_add_u32_dpp")
    R(36, "v_cvt_f32_u32") R(37, "v_add_u32_sdwa") R(38, "v_xor_b32 sgpr") R(39, "v_lshlrev_b32 v, sgpr") R(40, "v_lshrrev_b32 v(shift), sgpr")
    R(41, "v_add_u16") R(42, "v_mad_i32_i24") R(43, "v_dot2_i32_i16") R(44, "v_pk_sub_i16") R(45, "v_pk_ashrrev_i16") R(46, "v_pk_lshlrev_b16")
    return 0;
}
