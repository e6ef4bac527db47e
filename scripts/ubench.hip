// ubench.hip — instruction-rate microbenchmarks for the ops the row kernel is made of.
// Build+run on the GPU box:  hipcc --offload-arch=gfx950 -O3 scripts/ubench.hip -o /tmp/ubench && /tmp/ubench
// Reports wave-instructions per cycle per CU relative to an assumed 2.4 GHz clock and,
// more usefully, time relative to v_add_u32 (= 1.00).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define REP8(x) x x x x x x x x
#define HIPCHK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { printf("%s: %s\n", #e, hipGetErrorString(r_)); exit(1);} } while (0)

constexpr int kInner = 64;   // asm statements per loop trip (8 x 8)

template <int OP>
__global__ __launch_bounds__(256) void ub(uint32_t *out, int iters, uint32_t s0, uint32_t s1)
{
    __shared__ uint32_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i * 2654435761u;
    __syncthreads();
    uint32_t a = threadIdx.x, b = threadIdx.x * 3u, c = 5u, d = 7u, e = 11u, f = 13u, g = 17u, h = 19u;
    uint64_t A = a, B = b, C = c, D = d;
    const uint64_t S = ((uint64_t) s1 << 32) | s0;
    uint32_t la = (threadIdx.x & 63) * 4u, lb = 128u;
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    u4 r4 = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) { REP8(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s0));) }
        if (OP == 1) { REP8(asm volatile("v_lshl_add_u64 %0, %4, 0, %0\n v_lshl_add_u64 %1, %4, 0, %1\n v_lshl_add_u64 %2, %4, 0, %2\n v_lshl_add_u64 %3, %4, 0, %3\n v_lshl_add_u64 %0, %4, 0, %0\n v_lshl_add_u64 %1, %4, 0, %1\n v_lshl_add_u64 %2, %4, 0, %2\n v_lshl_add_u64 %3, %4, 0, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "s"(S));) }
        if (OP == 2) { REP8(asm volatile("v_add_co_u32 %0, vcc, %8, %0\n v_addc_co_u32 %1, vcc, %9, %1, vcc\n v_add_co_u32 %2, vcc, %8, %2\n v_addc_co_u32 %3, vcc, %9, %3, vcc\n v_add_co_u32 %4, vcc, %8, %4\n v_addc_co_u32 %5, vcc, %9, %5, vcc\n v_add_co_u32 %6, vcc, %8, %6\n v_addc_co_u32 %7, vcc, %9, %7, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "v"(s0), "v"(s1) : "vcc");) }
        if (OP == 3) { REP8(asm volatile("v_pk_mad_u16 %0, %4, %5, %0\n v_pk_mad_u16 %1, %4, %5, %1\n v_pk_mad_u16 %2, %4, %5, %2\n v_pk_mad_u16 %3, %4, %5, %3\n v_pk_mad_u16 %0, %4, %5, %0\n v_pk_mad_u16 %1, %4, %5, %1\n v_pk_mad_u16 %2, %4, %5, %2\n v_pk_mad_u16 %3, %4, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (OP == 4) { REP8(asm volatile("v_bfe_i32 %0, %4, %0, 1\n v_bfe_i32 %1, %4, %1, 1\n v_bfe_i32 %2, %4, %2, 1\n v_bfe_i32 %3, %4, %3, 1\n v_bfe_i32 %0, %5, %0, 1\n v_bfe_i32 %1, %5, %1, 1\n v_bfe_i32 %2, %5, %2, 1\n v_bfe_i32 %3, %5, %3, 1" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (OP == 5) { REP8(asm volatile("v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n v_and_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "s"(s0));) }
        if (OP == 6) { REP8(asm volatile("v_or_b32 %0, 0x10001, %0\n v_or_b32 %1, 0x10001, %1\n v_or_b32 %2, 0x10001, %2\n v_or_b32 %3, 0x10001, %3\n v_or_b32 %4, 0x10001, %4\n v_or_b32 %5, 0x10001, %5\n v_or_b32 %6, 0x10001, %6\n v_or_b32 %7, 0x10001, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 7) { REP8(asm volatile("v_lshrrev_b32 %0, 24, %0\n v_lshrrev_b32 %1, 24, %1\n v_lshrrev_b32 %2, 24, %2\n v_lshrrev_b32 %3, 24, %3\n v_lshrrev_b32 %4, 24, %4\n v_lshrrev_b32 %5, 24, %5\n v_lshrrev_b32 %6, 24, %6\n v_lshrrev_b32 %7, 24, %7" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));) }
        if (OP == 8) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:256\n ds_read_b32 %2, %8 offset:512\n ds_read_b32 %3, %8 offset:768\n ds_read_b32 %4, %8 offset:1024\n ds_read_b32 %5, %8 offset:1280\n ds_read_b32 %6, %8 offset:1536\n ds_read_b32 %7, %8 offset:1792\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(la));) }
        if (OP == 9) { REP8(asm volatile("ds_read_b128 %0, %1\n ds_read_b128 %0, %1 offset:16\n ds_read_b128 %0, %1 offset:32\n ds_read_b128 %0, %1 offset:48\n ds_read_b128 %0, %1 offset:64\n ds_read_b128 %0, %1 offset:80\n ds_read_b128 %0, %1 offset:96\n ds_read_b128 %0, %1 offset:112\n s_waitcnt lgkmcnt(0)" : "=v"(r4) : "v"(lb));) }
        if (OP == 10) { REP8(asm volatile("ds_read_b32 %0, %8\n ds_read_b32 %1, %8 offset:4\n ds_read_b32 %2, %8 offset:8\n ds_read_b32 %3, %8 offset:12\n ds_read_b32 %4, %8 offset:16\n ds_read_b32 %5, %8 offset:20\n ds_read_b32 %6, %8 offset:24\n ds_read_b32 %7, %8 offset:28\n s_waitcnt lgkmcnt(0)" : "=v"(a), "=v"(b), "=v"(c), "=v"(d), "=v"(e), "=v"(f), "=v"(g), "=v"(h) : "v"(lb));) }
        if (OP == 11) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3\n v_mad_u64_u32 %0, vcc, %4, %5, %0\n v_mad_u64_u32 %1, vcc, %4, %5, %1\n v_mad_u64_u32 %2, vcc, %4, %5, %2\n v_mad_u64_u32 %3, vcc, %4, %5, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 12) { REP8(asm volatile("v_pk_add_u16 %0, %4, %0\n v_pk_add_u16 %1, %4, %1\n v_pk_add_u16 %2, %4, %2\n v_pk_add_u16 %3, %4, %3\n v_pk_add_u16 %0, %5, %0\n v_pk_add_u16 %1, %5, %1\n v_pk_add_u16 %2, %5, %2\n v_pk_add_u16 %3, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (OP == 13) { REP8(asm volatile("v_lshl_add_u32 %0, %4, 31, %0\n v_lshl_add_u32 %1, %4, 31, %1\n v_lshl_add_u32 %2, %4, 31, %2\n v_lshl_add_u32 %3, %4, 31, %3\n v_lshl_add_u32 %0, %5, 31, %0\n v_lshl_add_u32 %1, %5, 31, %1\n v_lshl_add_u32 %2, %5, 31, %2\n v_lshl_add_u32 %3, %5, 31, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (OP == 14) { REP8(asm volatile("v_cmp_gt_u32 vcc, %4, %0\n v_addc_co_u32 %0, vcc, %5, %0, vcc\n v_cmp_gt_u32 vcc, %4, %1\n v_addc_co_u32 %1, vcc, %5, %1, vcc\n v_cmp_gt_u32 vcc, %4, %2\n v_addc_co_u32 %2, vcc, %5, %2, vcc\n v_cmp_gt_u32 vcc, %4, %3\n v_addc_co_u32 %3, vcc, %5, %3, vcc" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f) : "vcc");) }
        if (OP == 15) { REP8(asm volatile("v_lshrrev_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD\n v_lshrrev_b32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:BYTE_3 src1_sel:DWORD" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (OP == 16) { REP8(asm volatile("v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3\n v_add_f64 %0, %4, %0\n v_add_f64 %1, %4, %1\n v_add_f64 %2, %4, %2\n v_add_f64 %3, %4, %3" : "+v"(A), "+v"(B), "+v"(C), "+v"(D) : "v"(S));) }
        if (OP == 17) { REP8(asm volatile("v_pk_mul_lo_u16 %0, %4, %0\n v_pk_mul_lo_u16 %1, %4, %1\n v_pk_mul_lo_u16 %2, %4, %2\n v_pk_mul_lo_u16 %3, %4, %3\n v_pk_mul_lo_u16 %0, %5, %0\n v_pk_mul_lo_u16 %1, %5, %1\n v_pk_mul_lo_u16 %2, %5, %2\n v_pk_mul_lo_u16 %3, %5, %3" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e), "v"(f));) }
        if (OP == 18) { REP8(asm volatile("v_xor_b32 %0, %4, %0\n v_pk_sub_i16 %0, %0, %4\n v_xor_b32 %1, %4, %1\n v_pk_sub_i16 %1, %1, %4\n v_xor_b32 %2, %4, %2\n v_pk_sub_i16 %2, %2, %4\n v_xor_b32 %3, %4, %3\n v_pk_sub_i16 %3, %3, %4" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "v"(e));) }
        if (OP == 19) { REP8(asm volatile("s_add_u32 %0, %0, %2\n s_addc_u32 %1, %1, %3\n s_add_u32 %0, %0, %2\n s_addc_u32 %1, %1, %3\n s_add_u32 %0, %0, %2\n s_addc_u32 %1, %1, %3\n s_add_u32 %0, %0, %2\n s_addc_u32 %1, %1, %3" : "+s"(s0), "+s"(s1) : "s"(iters), "s"(it) : "scc");) }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d + e + f + g + h + (uint32_t) (A + B + C + D) + r4.x + r4.y + r4.z + r4.w + s0 + s1;
}

template <int OP>
double run(const char *name, int waves_per_simd, double base)
{
    uint32_t *out;
    const int blocks = 256 * waves_per_simd;     // 256 CUs x (4 waves/block) -> waves_per_simd per SIMD
    HIPCHK(hipMalloc(&out, (size_t) blocks * 256 * 4));
    const int iters = 2000;
    hipEvent_t e0, e1;
    HIPCHK(hipEventCreate(&e0)); HIPCHK(hipEventCreate(&e1));
    ub<OP><<<blocks, 256>>>(out, 10, 3, 5);
    HIPCHK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        HIPCHK(hipEventRecord(e0));
        ub<OP><<<blocks, 256>>>(out, iters, 3, 5);
        HIPCHK(hipEventRecord(e1));
        HIPCHK(hipEventSynchronize(e1));
        float ms; HIPCHK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) best = ms;
    }
    const double instr_per_wave = (double) iters * kInner;
    // cycles (at 2.4 GHz) per wave-instruction per SIMD
    const double cyc = best * 1e-3 * 2.4e9 / (instr_per_wave * waves_per_simd);
    printf("%-28s w/SIMD=%d  %8.3f ms  %6.2f cyc/wave-instr/SIMD @2.4GHz  rel=%.2f\n", name, waves_per_simd, best, cyc,
           base > 0 ? best / base : 1.0);
    HIPCHK(hipFree(out));
    return best;
}

int main()
{
    for (int w : {2, 4, 8}) {
        double b = run<0>("v_add_u32", w, 0);
        run<1>("v_lshl_add_u64", w, b);
        run<2>("v_add_co+v_addc (per instr)", w, b);
        run<3>("v_pk_mad_u16", w, b);
        run<17>("v_pk_mul_lo_u16", w, b);
        run<12>("v_pk_add_u16", w, b);
        run<4>("v_bfe_i32", w, b);
        run<5>("v_and_b32_sdwa", w, b);
        run<15>("v_lshrrev_b32_sdwa", w, b);
        run<6>("v_or_b32 (literal)", w, b);
        run<7>("v_lshrrev_b32", w, b);
        run<13>("v_lshl_add_u32", w, b);
        run<14>("v_cmp+v_addc (per instr)", w, b);
        run<18>("v_xor+v_pk_sub (per instr)", w, b);
        run<11>("v_mad_u64_u32", w, b);
        run<16>("v_add_f64", w, b);
        run<19>("s_add/s_addc", w, b);
        run<8>("ds_read_b32 lane-consec", w, b);
        run<10>("ds_read_b32 broadcast", w, b);
        run<9>("ds_read_b128 broadcast", w, b);
    }
    return 0;
}
