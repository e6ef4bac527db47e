// ubench4.hip — round 6, the round-5 verdict's last question about the high-rate kernel: if the chip signs of a (channel, row) came
// as ONE 64-bit lane mask built in the same wave (no scalar-cache traffic, r02_segm's cost), would the row loop win?
//
// A "channel step" = what a wave issues per (channel, row of 64 samples), LDS read left out as in ubench3.hip:
//   plain-add   today's core (seg): code NCO, window >> chip byte, sign into the phase, address, carrier NCO, half an add3
//   mask, free  carrier NCO, address, + accumulate; the negative lanes accumulate a second time under EXEC = mask
//               (result = pos - 2 neg): the loop r02_segm reached with masks that cost nothing -- the ceiling
//   mask, SGPR  the same, the mask given to v_cndmask as an SGPR pair (no EXEC switching)
//   + readlane  either, with the mask pair DELIVERED: two v_readlane_b32 per channel step from the VGPR pair of the lane that
//               built it (the cheapest in-wave route from a builder lane to a wave-uniform value; ds_read_b64 + 2 x
//               v_readfirstlane costs the same two VALU slots plus an LDS slot the row loop does not have: lds_busy 0.77)
// The builder itself (per lane: <= 4 chip edges per 64 samples at 25 Msps, a 6-bit quotient and a check per edge, ~40
// lane-instructions = 0.6 wave-instructions per channel step) is NOT in these numbers: they are lower bounds for the mask forms.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void ub(uint32_t *out, int iters, uint64_t s0, uint64_t s1, uint32_t mask, uint64_t lanes)
{
    uint64_t P0 = threadIdx.x * 0x9e3779b97f4a7c15ull, Q0 = P0 * 3, P1 = P0 * 5, Q1 = P0 * 7, P2 = P0 * 9, Q2 = P0 * 11, P3 = P0 * 13, Q3 = P0 * 15;
    uint32_t acc = 0, neg = 0, w = threadIdx.x * 2654435761u, a = 0, k = 0;
    uint32_t mlo = w * 3u, mhi = w * 7u;            // "the masks the builder lanes hold": one pair per lane
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#define PA(Pn, Qn, T) asm volatile( \
            "v_lshrrev_b32_sdwa " T ", %[qh], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
            "v_lshl_add_u32 " T ", " T ", 26, %[ph]\n" \
            "v_and_b32_sdwa " T ", " T ", %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            "v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" \
            : [p] "+v"(Pn), [q] "+v"(Qn), [a] "+v"(a), [k] "+v"(k) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [mask] "s"(mask), [sp] "s"(s0), [sq] "s"(s1));
#define PA2(Pa, Qa, Pb, Qb) PA(Pa, Qa, "%[a]") PA(Pb, Qb, "%[k]") asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(k));
            PA2(P0, Q0, P1, Q1) PA2(P2, Q2, P3, Q3) PA2(P0, Q0, P1, Q1) PA2(P2, Q2, P3, Q3)
        } else if (MODE == 1 || MODE == 2) {
            // EXEC form.  MODE 1: the mask pair is a kernel argument (free).  MODE 2: read from lane L of (mlo, mhi) first.
#define EX(Pn, T, L) asm volatile( \
            "v_and_b32_sdwa " T ", %[ph], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            : [p] "+v"(Pn), [a] "+v"(a), [k] "+v"(k) : [ph] "v"((uint32_t) (Pn >> 32)), [mask] "s"(mask), [sp] "s"(s0));
#define EXM_FREE(T) asm volatile( \
            "s_mov_b64 exec, %[m]\n v_add_u32 %[neg], %[neg], " T "\n s_mov_b64 exec, -1\n" \
            : [neg] "+v"(neg) : [m] "s"(lanes), [a] "v"(a), [k] "v"(k));
#define EXM_LANE(T, L) asm volatile( \
            "v_readlane_b32 s20, %[mlo], " L "\n v_readlane_b32 s21, %[mhi], " L "\n" \
            "s_nop 0\n s_mov_b64 exec, s[20:21]\n v_add_u32 %[neg], %[neg], " T "\n s_mov_b64 exec, -1\n" \
            : [neg] "+v"(neg) : [mlo] "v"(mlo), [mhi] "v"(mhi), [a] "v"(a), [k] "v"(k) : "s20", "s21");
#define EX2(Pa, Pb, La, Lb) EX(Pa, "%[a]", La) EX(Pb, "%[k]", Lb) \
            if (MODE == 1) { EXM_FREE("%[a]") EXM_FREE("%[k]") } else { EXM_LANE("%[a]", La) EXM_LANE("%[k]", Lb) } \
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(k));
            EX2(P0, P1, "0", "1") EX2(P2, P3, "2", "3") EX2(Q0, Q1, "4", "5") EX2(Q2, Q3, "6", "7")
        } else if (MODE == 3 || MODE == 4) {
            // v_cndmask form: the sign enters the ADDRESS (a ^ half a cycle), chosen by an SGPR-pair mask.  MODE 4 delivers it.
#define CN_FREE(Pn, T) asm volatile( \
            "v_and_b32_sdwa " T ", %[ph], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_xor_b32 %[x], 0x400, " T "\n" \
            "v_cndmask_b32_e64 " T ", " T ", %[x], %[m]\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            : [p] "+v"(Pn), [a] "+v"(a), [k] "+v"(k), [x] "=&v"(neg) : [ph] "v"((uint32_t) (Pn >> 32)), [mask] "s"(mask), [sp] "s"(s0), [m] "s"(lanes));
#define CN_LANE(Pn, T, L) asm volatile( \
            "v_readlane_b32 s20, %[mlo], " L "\n v_readlane_b32 s21, %[mhi], " L "\n" \
            "v_and_b32_sdwa " T ", %[ph], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_xor_b32 %[x], 0x400, " T "\n" \
            "v_cndmask_b32_e64 " T ", " T ", %[x], s[20:21]\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            : [p] "+v"(Pn), [a] "+v"(a), [k] "+v"(k), [x] "=&v"(neg) : [ph] "v"((uint32_t) (Pn >> 32)), [mask] "s"(mask), [sp] "s"(s0), [mlo] "v"(mlo), [mhi] "v"(mhi) : "s20", "s21");
#define CN2(Pa, Pb, La, Lb) \
            if (MODE == 3) { CN_FREE(Pa, "%[a]") CN_FREE(Pb, "%[k]") } else { CN_LANE(Pa, "%[a]", La) CN_LANE(Pb, "%[k]", Lb) } \
            asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(k));
            CN2(P0, P1, "0", "1") CN2(P2, P3, "2", "3") CN2(Q0, Q1, "4", "5") CN2(Q2, Q3, "6", "7")
        } else if (MODE == 5) {    // two v_readlane_b32 alone
#define RL(L) asm volatile("v_readlane_b32 s20, %[mlo], " L "\n v_readlane_b32 s21, %[mhi], " L "\n s_add_u32 %[o], %[o], s20\n s_add_u32 %[o], %[o], s21\n" \
            : [o] "+s"(mask) : [mlo] "v"(mlo), [mhi] "v"(mhi) : "s20", "s21", "scc");
            RL("0") RL("1") RL("2") RL("3") RL("4") RL("5") RL("6") RL("7")
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t) (P0 + Q0 + P1 + Q1 + P2 + Q2 + P3 + Q3) + acc - 2u * neg + mask;
}

template <int MODE>
double run(const char *name, int w, int per_trip)
{
    const int blocks = 256 * w, iters = 2000;       // 256 CUs x w blocks of 4 waves = w waves per SIMD
    uint32_t *out;
    HIPCHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) ub<MODE><<<blocks, 256>>>(out, iters, 0x123456789abcdull, 0x23456789abcdeull, 0x7fc, 0x5a5a33cc0ff0aa55ull);
    HIPCHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        HIPCHK(hipEventRecord(e0));
        ub<MODE><<<blocks, 256>>>(out, iters, 0x123456789abcdull, 0x23456789abcdeull, 0x7fc, 0x5a5a33cc0ff0aa55ull);
        HIPCHK(hipEventRecord(e1));
        HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    double cyc = best * 1e-3 * 2.4e9 / ((double) iters * 8 * w);
    printf("%-78s w/SIMD=%d %7.3f ms  %6.2f cyc per channel step per SIMD @2.4GHz (%d VALU)\n", name, w, best, cyc, per_trip);
    HIPCHK(hipFree(out));
    return cyc;
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        double base = run<0>("plain-add core (today's seg row loop): 2 NCO adds, sdwa lshr, lshl_add, sdwa and, 1/2 add3", w, 5);
        double a = run<1>("mask as EXEC, free: carrier NCO, sdwa and, add under EXEC, 1/2 add3", w, 3);
        double b = run<2>("mask as EXEC, delivered by 2 x v_readlane_b32", w, 5);
        double c = run<3>("mask as SGPR pair to v_cndmask on the address, free", w, 4);
        double d = run<4>("mask to v_cndmask, delivered by 2 x v_readlane_b32", w, 6);
        run<5>("2 x v_readlane_b32 alone", w, 2);
        printf("    w/SIMD=%d: against today's core  EXEC free %+.1f %%  EXEC delivered %+.1f %%  cndmask free %+.1f %%  cndmask delivered %+.1f %%  (negative = fewer cycles)\n",
               w, 100 * (a / base - 1), 100 * (b / base - 1), 100 * (c / base - 1), 100 * (d / base - 1));
    }
    return 0;
}
