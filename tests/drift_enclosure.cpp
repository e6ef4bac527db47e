// drift_enclosure.cpp -- Drift (csrc/gpsiq_exact.cpp): the enclosure of the reference's double accumulator at sample n, taken
// from the block's start state alone, against the accumulator walked exactly (Nco::advance).  TEST INFRASTRUCTURE.
//   usage: drift_enclosure [seed]   prints  checked=.. bad=.. max_use=.. mean_width=.. worst_width=..
// max_use: largest |truth - centre| / half-width seen; widths relative to the a-priori window 2 (n+1) 2^-54 (carrier) / 2^-44 (code).
#include "gpsiq_exact.cpp"
#include <random>
using namespace gpsiq;
typedef __int128 i128;

// the enclosure itself, as cell_at computes it (kept in step with Drift::cell_at)
static bool enclose(const Drift &D, double x0, long n, i128 *lo, i128 *hi)
{
    if (!D.valid || !(x0 >= 0.0 && x0 < D.L)) return false;
    const i128 R = D.neg ? D.units(x0) - (i128) n * D.mc : D.units(x0) + (i128) n * D.mc;
    i128 q = R / D.Lint, r = R % D.Lint;
    if (r < 0) { r += D.Lint; q -= 1; }
    const double core = (double) q * D.GL + D.Gof((double) r * D.ulp_c) - D.Gof(x0);
    const double I = ((double) (q < 0 ? -q : q) + 3.0) * D.Acyc;
    const double eta = 4.0 * D.gmax * (std::fabs(core) + I) + 1e-12 * (std::fabs(core) + std::fabs((double) q * D.GL)) + 4.0 * D.gmax * D.L * 0x1p-52;
    *lo = R + (i128) std::floor((core - I - eta) / D.ulp_c) - 2;
    *hi = R + (i128) std::ceil((core + I + eta) / D.ulp_c) + 2;
    return true;
}

int main(int argc, char **argv)
{
    std::mt19937_64 rng(argc > 1 ? (unsigned long) atol(argv[1]) : 1);
    std::uniform_real_distribution<double> up(0.0, 1.0);
    long bad = 0, tot = 0, nratio = 0, cells_ok = 0, cells = 0;
    double maxu = 0, worst = 0, sum = 0;
    for (int kind = 1; kind >= 0; --kind)
        for (int it = 0; it < 200000; ++it) {
            const double fs = (it % 4 == 0) ? 25e6 : (it % 4 == 1) ? 10e6 : (it % 4 == 2 ? 2.6e6 : 3e6);
            double f = (up(rng) * 2 - 1) * 6000;
            if (kind == 1 && std::fabs(f) < 50) f = 50;
            double c = kind == 1 ? f / fs : (1.023e6 + f / 1540) / fs;
            const int mode = it % 7;      // 3..6: trailing zero mantissa bits (exact-tie binades), 5, 6: a lone one above them
            if (mode >= 3) { uint64_t b = bits_of(c); const int z = 8 + (int) (rng() % 40); b &= ~((UINT64_C(1) << z) - 1); if (mode >= 5) b |= UINT64_C(1) << z; c = from_bits(b); }
            const double Lw = kind == 0 ? 1023.0 : 1.0;
            double x0 = up(rng) * Lw;
            if (it % 11 == 0) x0 = std::ldexp(up(rng), -(int) (rng() % 60));
            if (it % 13 == 0) { uint64_t b = bits_of(x0); b &= ~((UINT64_C(1) << (rng() % 50)) - 1); x0 = from_bits(b); }
            if (it % 17 == 0 && kind == 1) x0 = 1.0 - std::ldexp(1.0 + (double) (rng() % 1000), -53);         // just below the wrap
            if (!(x0 >= 0 && x0 < Lw)) continue;
            const long ns = (long) (fs / 10);
            long n = (long) (up(rng) * ns);
            if (it % 5 == 0) n = ns;
            Drift D; D.setup(c, kind);
            if (!D.valid) continue;
            Nco a = {x0, c, 0, 0, kind};
            a.advance(n);
            const i128 U = D.neg ? D.units(a.x) - (i128) a.wraps * D.Lint : D.units(a.x) + (i128) a.wraps * D.Lint;   // the unwrapped double phase, cut below one unit
            i128 lo, hi;
            if (!enclose(D, x0, n, &lo, &hi)) continue;
            ++tot;
            if (!(lo <= U && U + 1 <= hi)) { ++bad; if (bad < 10) std::fprintf(stderr, "OUTSIDE kind %d c %a x0 %a n %ld: lo-U %g hi-U %g\n", kind, c, x0, n, (double) (lo - U), (double) (hi - U)); }
            const double ctr = (double) (hi + lo - 2 * U) / 2.0, hw = (double) (hi - lo) / 2.0;
            if (std::fabs(ctr) / hw > maxu) maxu = std::fabs(ctr) / hw;
            if (n > 1000) { const double r = (double) (hi - lo) * D.ulp_c / (2.0 * (double) (n + 1) * (kind == 0 ? 0x1p-44 : 0x1p-54)); sum += r; ++nratio; if (r > worst) worst = r; }
            // and the decision taken from it: the cell, when one is named, is the walked accumulator's
            i128 cell;
            ++cells;
            if (D.cell_at(x0, n, &cell)) {
                ++cells_ok;
                const i128 truth = kind == 1 ? (i128) std::floor(a.x * 512.0) : (i128) (long) a.x + (i128) a.wraps * GPSIQ_CA_SEQ_LEN;
                const bool same = kind == 1 ? (long) (cell & 511) == (long) truth && a.x < 1.0 : cell == truth;
                if (!same) { ++bad; if (bad < 10) std::fprintf(stderr, "WRONG CELL kind %d c %a x0 %a n %ld\n", kind, c, x0, n); }
            }
        }
    std::printf("checked=%ld bad=%ld max_use=%.3f mean_width=%.5f worst_width=%.5f decided=%.5f\n", tot, bad, maxu, sum / (double) nratio, worst, (double) cells_ok / (double) cells);
    return bad != 0;
}
