// ubench3.hip — does the row kernel's per-channel core issue at the sum of its parts?
// One "channel step" = the seven VALU instructions of the core (no LDS): address (SDWA and),
// chip byte (lshr), sign (bfe_i32), +-1 pair (or), accumulate (pk_mad_u16), two 64-bit NCO adds
// with SGPR-pair steps.  8 independent channels per trip, 1..8 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

#define CH(P, Q, SP, SQ) \
    "v_and_b32_sdwa %[a], " P "h, %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
    "v_lshrrev_b32 %[k], 24, " Q "h\n" \
    "v_bfe_i32 %[m], %[w], %[k], 1\n" \
    "v_or_b32 %[m], 0x10001, %[m]\n" \
    "v_pk_mad_u16 %[acc], %[a], %[m], %[acc]\n" \
    "v_lshl_add_u64 " P ", " P ", 0, " SP "\n" \
    "v_lshl_add_u64 " Q ", " Q ", 0, " SQ "\n"

template <int MODE>
__global__ __launch_bounds__(256) void ub(uint32_t *out, int iters, uint64_t s0, uint64_t s1, uint32_t mask)
{
    uint64_t P0 = threadIdx.x * 0x9e3779b97f4a7c15ull, Q0 = P0 * 3, P1 = P0 * 5, Q1 = P0 * 7, P2 = P0 * 9, Q2 = P0 * 11, P3 = P0 * 13, Q3 = P0 * 15;
    uint32_t acc = 0, w = threadIdx.x * 2654435761u, a = 0, k = 0, m;
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {           // the full core, 4 channels x 2
#define ONE(Pn, Qn) asm volatile( \
            "v_and_b32_sdwa %[a], %[ph], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_lshrrev_b32 %[k], 24, %[qh]\n" \
            "v_bfe_i32 %[m], %[w], %[k], 1\n" \
            "v_or_b32 %[m], 0x10001, %[m]\n" \
            "v_pk_mad_u16 %[acc], %[a], %[m], %[acc]\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            "v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" \
            : [p] "+v"(Pn), [q] "+v"(Qn), [acc] "+v"(acc), [a] "=&v"(a), [k] "=&v"(k), [m] "=&v"(m) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [mask] "s"(mask), [sp] "s"(s0), [sq] "s"(s1));
            ONE(P0, Q0) ONE(P1, Q1) ONE(P2, Q2) ONE(P3, Q3) ONE(P0, Q0) ONE(P1, Q1) ONE(P2, Q2) ONE(P3, Q3)
        } else if (MODE == 1) {    // only the two NCO adds
#define TWO(Pn, Qn) asm volatile("v_lshl_add_u64 %[p], %[p], 0, %[sp]\n v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" : [p] "+v"(Pn), [q] "+v"(Qn) : [sp] "s"(s0), [sq] "s"(s1));
            TWO(P0, Q0) TWO(P1, Q1) TWO(P2, Q2) TWO(P3, Q3) TWO(P0, Q0) TWO(P1, Q1) TWO(P2, Q2) TWO(P3, Q3)
        } else if (MODE == 2) {    // the five 32-bit instructions only
#define FIVE(Pn, Qn) asm volatile( \
            "v_and_b32_sdwa %[a], %[ph], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_lshrrev_b32 %[k], 24, %[qh]\n" \
            "v_bfe_i32 %[m], %[w], %[k], 1\n" \
            "v_or_b32 %[m], 0x10001, %[m]\n" \
            "v_pk_mad_u16 %[acc], %[a], %[m], %[acc]\n" \
            : [acc] "+v"(acc), [a] "=&v"(a), [k] "=&v"(k), [m] "=&v"(m) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [mask] "s"(mask));
            FIVE(P0, Q0) FIVE(P1, Q1) FIVE(P2, Q2) FIVE(P3, Q3) FIVE(P0, Q0) FIVE(P1, Q1) FIVE(P2, Q2) FIVE(P3, Q3)
            P0 += acc;             // keep the inputs loop-carried
        } else if (MODE == 4) {    // the plain-add core: window >> chip byte, sign into the phase, address, add3 per two channels
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
        } else if (MODE == 5) {    // lane = channel (north_star's "wavefront-level sum over visible channels"): a wave holds
            // 4 samples x 16 channels; the same plain-add core per lane, but the NCO steps and the LUT base are per-lane
            // VGPRs, and the channel sum is a 16-lane DPP reduction (row_shr 8,4,2,1) per 4 samples.  One trip below =
            // 8 wave-steps = 8 x 64 channel-samples, the same count the lane = sample modes process per trip.
            uint64_t v0 = s0 + threadIdx.x, v1 = s1 + threadIdx.x;
            uint32_t base = (threadIdx.x & 15u) * 2048u;
#define LC(Pn, Qn) asm volatile( \
            "v_lshrrev_b32_sdwa %[a], %[qh], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
            "v_lshl_add_u32 %[a], %[a], 26, %[ph]\n" \
            "v_and_b32_sdwa %[a], %[a], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_add_u32 %[a], %[a], %[base]\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            "v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" \
            "v_add_u32_dpp %[k], %[a], %[a] row_shr:8 row_mask:0xf bank_mask:0xf\n" \
            "v_add_u32_dpp %[k], %[k], %[k] row_shr:4 row_mask:0xf bank_mask:0xf\n" \
            "v_add_u32_dpp %[k], %[k], %[k] row_shr:2 row_mask:0xf bank_mask:0xf\n" \
            "v_add_u32_dpp %[k], %[k], %[k] row_shr:1 row_mask:0xf bank_mask:0xf\n" \
            "v_add_u32 %[acc], %[acc], %[k]\n" \
            : [p] "+v"(Pn), [q] "+v"(Qn), [a] "+v"(a), [k] "+v"(k), [acc] "+v"(acc) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [mask] "s"(mask), [sp] "v"(v0), [sq] "v"(v1), [base] "v"(base));
            LC(P0, Q0) LC(P1, Q1) LC(P2, Q2) LC(P3, Q3) LC(P0, Q0) LC(P1, Q1) LC(P2, Q2) LC(P3, Q3)
        } else if (MODE == 6) {    // round-3 experiment, the row loop: the chip sign of channel c is bit c of a per-lane 16-bit word S
            // (one ds_read_u16 per row, not counted here): constant shift (plain VOP2), sign into the phase, address, carrier
            // NCO, add3 per two channels -- no code NCO, no SDWA shift
#define SW(Pn, T, C) asm volatile( \
            "v_lshrrev_b32 " T ", " C ", %[w]\n" \
            "v_lshl_add_u32 " T ", " T ", 26, %[ph]\n" \
            "v_and_b32_sdwa " T ", " T ", %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            : [p] "+v"(Pn), [a] "+v"(a), [k] "+v"(k) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [w] "v"(w), [mask] "s"(mask), [sp] "s"(s0));
#define SW2(Pa, Pb, Ca, Cb) SW(Pa, "%[a]", Ca) SW(Pb, "%[k]", Cb) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(acc) : "v"(a), "v"(k));
            SW2(P0, P1, "0", "1") SW2(P2, P3, "2", "3") SW2(Q0, Q1, "4", "5") SW2(Q2, Q3, "6", "7")
        } else if (MODE == 7) {    // ... and what building those words costs per channel-sample, lane = sample: the code NCO, the
            // window shifted by the chip byte (as today), and the bit shifted into the word (v_alignbit takes bit 0 of the
            // shifted window whatever its upper bits are)
#define SB(Qn) asm volatile( \
            "v_lshrrev_b32_sdwa %[k], %[qh], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
            "v_alignbit_b32 %[a], %[k], %[a], 1\n" \
            "v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" \
            : [q] "+v"(Qn), [a] "+v"(a), [k] "+v"(k) \
            : [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [sq] "s"(s1));
            SB(Q0) SB(Q1) SB(Q2) SB(Q3) SB(P0) SB(P1) SB(P2) SB(P3)
            acc += a;
        } else if (MODE == 8) {    // round-4 verdict's experiment: TWO rows per trip; the even row is the plain-add core with its two
            // 64-bit NCO adds (by two rows' worth), the odd row takes its phase words from hi + dhi with plain 32-bit VOP2 adds
            // (per-channel steps held in VGPRs: an SGPR operand makes the add a 4.3-cycle form) and has no NCO add -- the carry
            // out of the low words is dropped, so a sample whose field sits under a run of ones would have to go to a patch
            // list.  One trip = 4 channels x 2 rows = 8 channel steps, as in the other modes.
            uint32_t dph = (uint32_t) (s0 >> 32) + (threadIdx.x & 0u), dqh = (uint32_t) (s1 >> 32) + (threadIdx.x & 0u), tp, tq;
#define ROW2(Pn, Qn, T) asm volatile( \
            "v_lshrrev_b32_sdwa " T ", %[qh], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
            "v_lshl_add_u32 " T ", " T ", 26, %[ph]\n" \
            "v_and_b32_sdwa " T ", " T ", %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_add_u32 %[tp], %[ph], %[dph]\n" \
            "v_add_u32 %[tq], %[qh], %[dqh]\n" \
            "v_lshl_add_u64 %[p], %[p], 0, %[sp]\n" \
            "v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" \
            "v_lshrrev_b32_sdwa %[tq], %[tq], %[w] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n" \
            "v_lshl_add_u32 %[tq], %[tq], 26, %[tp]\n" \
            "v_and_b32_sdwa %[tq], %[tq], %[mask] dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n" \
            "v_add3_u32 %[acc], %[acc], " T ", %[tq]\n" \
            : [p] "+v"(Pn), [q] "+v"(Qn), [a] "+v"(a), [k] "+v"(k), [acc] "+v"(acc), [tp] "=&v"(tp), [tq] "=&v"(tq) \
            : [ph] "v"((uint32_t) (Pn >> 32)), [qh] "v"((uint32_t) (Qn >> 32)), [w] "v"(w), [mask] "s"(mask), [sp] "s"(s0), [sq] "s"(s1), [dph] "v"(dph), [dqh] "v"(dqh));
            ROW2(P0, Q0, "%[a]") ROW2(P1, Q1, "%[k]") ROW2(P2, Q2, "%[a]") ROW2(P3, Q3, "%[k]")
        } else if (MODE == 3) {    // NCO adds with VGPR steps instead of SGPR pairs
            uint64_t v0 = s0 + threadIdx.x, v1 = s1 + threadIdx.x;
#define TWOV(Pn, Qn) asm volatile("v_lshl_add_u64 %[p], %[p], 0, %[sp]\n v_lshl_add_u64 %[q], %[q], 0, %[sq]\n" : [p] "+v"(Pn), [q] "+v"(Qn) : [sp] "v"(v0), [sq] "v"(v1));
            TWOV(P0, Q0) TWOV(P1, Q1) TWOV(P2, Q2) TWOV(P3, Q3) TWOV(P0, Q0) TWOV(P1, Q1) TWOV(P2, Q2) TWOV(P3, Q3)
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t) (P0 + Q0 + P1 + Q1 + P2 + Q2 + P3 + Q3) + acc;
}

template <int MODE>
void run(const char *name, int w, int per_trip)
{
    const int blocks = 256 * w, iters = 2000;       // 256 CUs x w blocks of 4 waves = w waves per SIMD
    uint32_t *out;
    HIPCHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    for (int r = 0; r < 3; ++r) ub<MODE><<<blocks, 256>>>(out, iters, 0x123456789abcdull, 0x23456789abcdeull, 0x7fc);
    HIPCHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; ++r) {
        HIPCHK(hipEventRecord(e0));
        ub<MODE><<<blocks, 256>>>(out, iters, 0x123456789abcdull, 0x23456789abcdeull, 0x7fc);
        HIPCHK(hipEventRecord(e1));
        HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    // cycles one SIMD spends per channel step (8 steps per trip, w waves per SIMD)
    printf("%-44s w/SIMD=%d %7.3f ms  %6.2f cyc per channel step per SIMD @2.4GHz (%d instr)\n", name, w, best,
           best * 1e-3 * 2.4e9 / ((double) iters * 8 * w), per_trip);
    HIPCHK(hipFree(out));
}

int main()
{
    for (int w : {1, 2, 4, 8}) {
        run<0>("core: sdwa,lshr,bfe,or,pk_mad,2x lshl_add_u64", w, 7);
        run<1>("2x v_lshl_add_u64 (SGPR-pair step)", w, 2);
        run<3>("2x v_lshl_add_u64 (VGPR-pair step)", w, 2);
        run<2>("sdwa,lshr,bfe,or,pk_mad", w, 5);
        run<4>("plain-add core: sdwa lshr,lshl_add,sdwa and,1/2 add3,2x lshl_add_u64", w, 5);
        run<5>("lane = channel: the same core + LUT base + 4 DPP row adds per 4 samples", w, 11);
        run<6>("sign-word row loop: lshr const,lshl_add,sdwa and,1/2 add3,1x lshl_add_u64", w, 4);
        run<7>("sign-word builder: lshl_add_u64, sdwa lshr, alignbit", w, 3);
        run<8>("two rows per trip: 64-bit NCO adds on even rows, hi + dhi (VOP2) on odd rows", w, 5);
    }
    return 0;
}
