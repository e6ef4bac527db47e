// batch_walk.cpp -- NcoWalk::walk8_up / walk8_down (csrc/gpsiq_exact.cpp: eight carrier cycles walked at once with AVX-512)
// against the scalar walk they vectorise (climb<true> / descend<true>), lane for lane: where a lane says ok, the cycle's end
// state, its number of samples and the range of start states it holds for are the scalar walk's; a lane may only ever be
// MORE cautious (not ok where the scalar walk is).  Random and adversarial addends (exact-tie binades), random and edge start
// states.  TEST INFRASTRUCTURE.   prints  cases=.. lanes_ok=.. scalar_ok=.. bad=..   (skipped=1 without AVX-512)
#include "gpsiq_exact.cpp"
#include <random>
using namespace gpsiq;

int main(int argc, char **argv)
{
#if defined(__x86_64__)
    if (!(__builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512f"))) { std::printf("skipped=1\n"); return 0; }
    std::mt19937_64 rng(argc > 1 ? (unsigned long) atol(argv[1]) : 1);
    std::uniform_real_distribution<double> up(0.0, 1.0);
    long cases = 0, lanes_ok = 0, scalar_ok = 0, bad = 0;
    for (int it = 0; it < 60000; ++it) {
        const double fs = (it % 4 == 0) ? 25e6 : (it % 4 == 1) ? 10e6 : (it % 4 == 2 ? 2.6e6 : 3e6);
        double f = (up(rng) * 2 - 1) * 9000;
        if (std::fabs(f) < 100) f = f < 0 ? -100 : 100;
        double c = f / fs;
        const int mode = it % 7;      // 3..6: trailing zero mantissa bits (exact-tie binades), 5, 6: a lone one above them
        if (mode >= 3) { uint64_t b = bits_of(c); const int z = 8 + (int) (rng() % 40); b &= ~((UINT64_C(1) << z) - 1); if (mode >= 5) b |= UINT64_C(1) << z; c = from_bits(b); }
        NcoWalk w;
        w.setup(c, 1);
        if (w.general) continue;
        const bool neg = c < 0.0;
        const int uexp = neg ? 1022 : 1023;
        const double scale = std::ldexp(1.0, 1075 - uexp), unit = std::ldexp(1.0, uexp - 1075);
        const int64_t m_max = neg ? ((int64_t) 1 << 53) - 1 : ((int64_t) 1 << 52) - 1;
        const int64_t W = (int64_t) (std::fabs(c) * scale) + 4, base = neg ? ((int64_t) 1 << 53) - W : 0;
        NcoWalk::Batch bt;
        for (int q = 0; q < 8; ++q) {
            int64_t m = base + (int64_t) (up(rng) * (double) W);
            if (it % 9 == 0 && q < 4) m = base + (int64_t) (rng() % 5);                 // the range's ends
            if (it % 9 == 1 && q < 4) m = base + W - 1 - (int64_t) (rng() % 5);
            if (it % 9 == 2 && q < 4) m &= ~(((int64_t) 1 << (rng() % 40)) - 1);        // round states
            if (m < 0) m = 0;
            if (m > m_max) m = m_max;
            bt.m[q] = m;
        }
        if (neg) w.walk8_down(m_max, &bt); else w.walk8_up(m_max, &bt);
        ++cases;
        for (int q = 0; q < 8; ++q) {
            double x = (double) bt.m[q] * unit;
            NcoWalk::Slack sl = {-bt.m[q], m_max - bt.m[q], uexp, true};
            long ne = 0;
            const bool wrapped = neg ? w.descend<true>(x, ne, 1L << 40, &sl) : w.climb<true>(x, ne, 1L << 40, &sl);
            const int64_t m2 = (int64_t) (x * scale);
            const bool s_ok = wrapped && sl.ok && x < 1.0 && m2 <= m_max;
            scalar_ok += s_ok;
            if (!bt.ok[q]) continue;
            ++lanes_ok;
            if (!s_ok || bt.m2[q] != m2 || bt.steps[q] != ne || bt.lo[q] != sl.lo || bt.hi[q] != sl.hi) {
                if (++bad < 10)
                    std::fprintf(stderr, "LANE DIFFERS c %a m %lld: batch m2 %lld n %lld [%lld, %lld]; scalar ok %d m2 %lld n %ld [%lld, %lld]\n", c, (long long) bt.m[q],
                                 (long long) bt.m2[q], (long long) bt.steps[q], (long long) bt.lo[q], (long long) bt.hi[q], (int) s_ok, (long long) m2, ne, (long long) sl.lo, (long long) sl.hi);
            }
        }
    }
    std::printf("cases=%ld lanes_ok=%ld scalar_ok=%ld bad=%ld\n", cases, lanes_ok, scalar_ok, bad);
    return bad != 0;
#else
    std::printf("skipped=1\n");
    return 0;
#endif
}
