// ubench_walk.cpp -- cost of the wrap-to-wrap carrier walk (csrc/gpsiq_exact.cpp, NcoWalk) per block and channel on this host,
// one thread: blocks of 26 000 .. 1 040 000 samples at +-2750 Hz (28 .. 1100 carrier cycles), best of 5 passes over 20 000
// addends.  Build:  g++ -O3 -std=c++17 -ffp-contract=off [-DKLOW=n] -I multi-sdr-gps-sim_amd/csrc scripts/ubench_walk.cpp
//                   multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread   (KLOW: binades walked by plain additions, BUCKETS: buckets of the wrap-to-wrap table, A/B)
#ifdef KLOW
#define GPSIQ_WALK_KLOW KLOW
#endif
#ifdef BUCKETS
#define GPSIQ_WALK_BUCKETS BUCKETS
#endif
#include "gpsiq_exact.cpp"
#include <chrono>
#include <random>
int main()
{
    std::mt19937_64 rng(7);
    std::uniform_real_distribution<double> up(0.0, 1.0);
    const double fs = 2.6e6, delt = 1.0 / fs;
    const int N = 20000;
    std::vector<double> x(N);
    for (auto &v : x) v = up(rng);
    for (double f : {2750.0, -2750.0})
        for (long ns : {26000L, 260000L, 1040000L}) {
            double best = 1e30, acc = 0;
            for (int pass = 0; pass < 5; ++pass) {
                gpsiq::NcoWalk w;
                auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; ++i) { w.setup((f + i * 0.01) * delt, 1); acc += w.run(x[i], ns); }
                const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
                if (us < best) best = us;
            }
            std::printf("f %+6.0f Hz, %7ld samples (%4.0f cycles): %6.2f us per block and channel  (%g)\n", f, ns, std::fabs(f) * ns / fs, best, acc);
        }
    return 0;
}
