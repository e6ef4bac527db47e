// chain_parallel.cpp -- the time-parallel carrier chain (csrc/gpsiq_lane.h, gpsiq_chain.cpp: maps + link) against the serial
// chain gpsiq_reference_chain (NcoWalk, itself pinned to the plain loop of gps.c:2821-2826 by tests/soak_carrier_walk.py), on
// random and adversarial timelines: Doppler ramps through zero, slow blocks, exact-tie addends (a power-of-two sample rate),
// slots re-allocated and unused, a timeline continued from an earlier call, ranges summarised and folded as the ranks of a
// time-sharded run do.  Every start state, end state and last_prn must be equal, bit for bit.  TEST INFRASTRUCTURE.
//   usage: chain_parallel [seed] [cases]       prints  cases=.. blocks=.. linked=.. walked=.. bad=..
#define GPSIQ_JOIN_CHECK 1
#include "gpsiq_exact.cpp"          // (with its internals: the general walker Nco is what the top-tie walk is held against)
#include "gpsiq_chain.cpp"
#include <random>

using namespace gpsiq;

// FpWalk (the lanes' walker: table steps in doubles) against WalkCore (integer mantissas), state for state and note for note
static long walkers_agree(std::mt19937_64 &rng, int cases)
{
    std::uniform_real_distribution<double> up(0.0, 1.0);
    long bad = 0;
    for (int it = 0; it < cases; ++it) {
        const double fs = (it % 4 == 0) ? 25e6 : (it % 4 == 1) ? 10e6 : (it % 4 == 2 ? 2.6e6 : 2097152.0);
        double f = (up(rng) * 2 - 1) * 9000;
        if (it % 11 == 0) f = (up(rng) * 2 - 1) * 40;
        double c = f / fs;
        if (it % 7 >= 4) { uint64_t b = bits_of(c); const int z = 8 + (int) (rng() % 40); b &= ~((UINT64_C(1) << z) - 1); if (it % 7 >= 5) b |= UINT64_C(1) << z; c = from_bits(b); }
        WalkCore<lane::kTab> a;
        FpWalk<lane::kTab> b;
        a.setup(c, 1);
        b.setup(c);
        if (a.general && !b.general && b.top_tie) {
            // an exact tie in the top binade of a descending carrier: the integer walker leaves it to the general one (its table
            // of cycles would not hold for odd offsets); the lanes' walker walks it (for even offsets): against Nco::advance
            for (int rep = 0; rep < 4; ++rep) {
                double x0 = rep == 0 ? up(rng) : rep == 1 ? std::ldexp((double) (rng() >> 11), -53) : rep == 2 ? 1.0 - std::fabs(c) * up(rng) : std::fabs(c) * up(rng);
                if (!(x0 >= 0.0 && x0 < 1.0)) continue;
                const long ns = 1 + (long) (rng() % 600000);
                Nco g = {x0, c, 0, 0, 1};
                g.advance(ns);
                double y = x0;
                long m = 0;
                while (m < ns && b.cycle(y, m, ns)) {}
                if (bits_of(y) != bits_of(g.x)) { if (bad++ < 5) std::printf("top-tie walk differs: c = %a, x0 = %a, ns = %ld: %a / %a\n", c, x0, ns, y, g.x); }
            }
            continue;
        }
        if (a.general != b.general) { if (bad++ < 5) std::printf("walkers: general differs for c = %a\n", c); continue; }
        if (a.general) continue;
        for (int rep = 0; rep < 4; ++rep) {
            double x0 = up(rng);
            if (rep == 1) x0 = std::ldexp((double) (rng() >> 11), -53);                       // any state of the top grid
            if (rep == 2) x0 = std::ldexp(1.0, -(int) (rng() % 30)) * (1.0 + std::ldexp((double) (rng() % 5), -52));   // binade edges
            if (rep == 3) x0 = std::fabs(c) * up(rng) * 1.5;                                    // just after a wrap
            if (!(x0 >= 0.0 && x0 < 1.0)) continue;
            const long ns = 1 + (long) (rng() % 3000000);
            double xa = x0, xb = x0;
            long na = 0, nb = 0;
            WalkCore<lane::kTab>::FastSlack sa, sb;
            sa.init(c < 0 ? 1022 : 1023); sb.init(c < 0 ? 1022 : 1023);
            if (x0 >= a.thr) { sa.note(x0); sb.note(x0); }
            bool same = true;
            for (int cyc = 0; cyc < 400 && na < ns && same; ++cyc) {
                const bool wa = a.neg ? a.descend<true>(xa, na, ns, &sa) : a.climb<true>(xa, na, ns, &sa);
                const bool wb = b.neg ? b.descend<true>(xb, nb, ns, &sb) : b.climb<true>(xb, nb, ns, &sb);
                same = wa == wb && na == nb && bits_of(xa) == bits_of(xb);
                if (!wa) break;
            }
            int64_t la = 0, ha = 0, lb = 0, hb = 0;
            const bool oa = sa.finish(&la, &ha), ob = sb.finish(&lb, &hb);
            if (!same || oa != ob || sa.sigma != sb.sigma || (oa && (la != lb || ha != hb))) {
                if (bad++ < 5) std::printf("walkers differ: c = %a, x0 = %a, ns = %ld: x %a / %a, n %ld / %ld, slack %d [%ld, %ld] / %d [%ld, %ld]\n",
                                           c, x0, ns, xa, xb, na, nb, (int) oa, (long) la, (long) ha, (int) ob, (long) lb, (long) hb);
            }
            // and without notes (the tail of the block before)
            double ya = x0, yb = x0;
            long ma = 0, mb = 0;
            while (ma < ns && a.cycle(ya, ma, ns)) {}
            while (mb < ns && b.cycle(yb, mb, ns)) {}
            if (bits_of(ya) != bits_of(yb) || (bits_of(ya) != bits_of(xa) && na == ns)) { if (bad++ < 5) std::printf("walkers differ without notes: c = %a, x0 = %a\n", c, x0); }
        }
    }
    return bad;
}

int main(int argc, char **argv)
{
    std::mt19937_64 rng(argc > 1 ? (unsigned long) atol(argv[1]) : 1);
    const int cases = argc > 2 ? atoi(argv[2]) : 40;
    std::uniform_real_distribution<double> up(0.0, 1.0);
    long blocks = 0, bad = walkers_agree(rng, 200 * cases);
    uint64_t s0[2], s1[2];
    gpsiq_chain_stats(s0);
    for (int it = 0; it < cases; ++it) {
        static const double rates[] = {2.6e6, 3.0e6, 10.0e6, 25.0e6, 2097152.0, 2.6e6};
        const double fs = rates[it % 6];
        int nsamp = (int) (fs / 10.0);
        if (it % 5 == 3) nsamp = 20000 + (int) (rng() % 50000);                // short blocks: more blocks per second of test
        const int nchan = 1 + (int) (rng() % 16), nblocks = 20 + (int) (rng() % 60);
        std::vector<gpsiq_chain_in_t> in((size_t) nblocks * nchan);
        for (int i = 0; i < nchan; ++i) {
            const int mode = (int) (rng() % 8);
            double f = (up(rng) * 2 - 1) * 6000.0, df = (up(rng) * 2 - 1) * 0.9;
            if (mode == 1) { f = (up(rng) * 2 - 1) * 30.0; df = (up(rng) * 2 - 1) * 3.0; }        // through zero Doppler, slow blocks
            if (mode == 2) { f = (up(rng) * 2 - 1) * 300.0; df = (up(rng) * 2 - 1) * 20.0; }
            int prn = 1 + (int) (rng() % 32);
            for (int b = 0; b < nblocks; ++b) {
                gpsiq_chain_in_t &d = in[(size_t) b * nchan + i];
                if (rng() % 97 == 0) prn = 1 + (int) (rng() % 32);                                  // the slot gets another satellite
                d.prn = (mode == 3 && (b / 7) % 3 == 1) ? 0 : prn;                                  // unused for a while
                d.carr_phase = up(rng);
                if (rng() % 13 == 0) d.carr_phase = std::ldexp((double) (rng() % 1024), -10);      // round phases
                d.f_carr = f + df * b + (up(rng) - 0.5) * 0.05;
                if (mode == 4) {                                                                    // addends with trailing zeros: exact-tie binades
                    uint64_t bb; std::memcpy(&bb, &d.f_carr, 8);
                    const int z = 20 + (int) (rng() % 28);
                    bb &= ~((UINT64_C(1) << z) - 1); if (rng() & 1) bb |= UINT64_C(1) << z;
                    std::memcpy(&d.f_carr, &bb, 8);
                }
                if (mode == 5 && b % 11 == 5) d.f_carr = 0.0;
                d.reserved = 0;
            }
        }
        const int max_seg = 1 + (int) (rng() % 32);
        std::vector<double> want((size_t) nblocks * nchan), got((size_t) nblocks * nchan);
        double want_end[16], got_end[16];
        int32_t want_prn[16], got_prn[16];
        if (gpsiq_reference_chain(in.data(), nblocks, nchan, fs, nsamp, nullptr, nullptr, want.data(), want_end, want_prn)) { std::printf("serial chain failed: %s\n", gpsiq_last_error()); return 2; }
        std::vector<gpsiq_chain_map_t> maps((size_t) nblocks * nchan);
        // (a) the whole timeline at once; (b) two calls, the second continuing the first; (c) three ranges, summarised and folded
        const int variant = it % 3;
        if (variant == 0) {
            if (gpsiq_chain_maps(in.data(), nblocks, nchan, fs, nsamp, nullptr, max_seg, maps.data(), nullptr) ||
                gpsiq_chain_link(in.data(), maps.data(), nblocks, nchan, fs, nsamp, nullptr, nullptr, got.data(), got_end, got_prn)) { std::printf("parallel chain failed: %s\n", gpsiq_last_error()); return 2; }
        } else if (variant == 1) {
            const int cut = 1 + (int) (rng() % (nblocks - 1));
            double mid_end[16]; int32_t mid_prn[16];
            gpsiq_chain_est_t st[16];
            if (gpsiq_chain_maps(in.data(), cut, nchan, fs, nsamp, nullptr, max_seg, maps.data(), nullptr) ||
                gpsiq_chain_link(in.data(), maps.data(), cut, nchan, fs, nsamp, nullptr, nullptr, got.data(), mid_end, mid_prn)) { std::printf("parallel chain failed: %s\n", gpsiq_last_error()); return 2; }
            for (int i = 0; i < nchan; ++i) {
                std::memset(&st[i], 0, sizeof st[i]);
                st[i].carr = mid_end[i]; st[i].prn = mid_prn[i]; st[i].flags = GPSIQ_CHAIN_EXACT;
                st[i].f_carr = in[(size_t) (cut - 1) * nchan + i].f_carr;
            }
            if (gpsiq_chain_maps(in.data() + (size_t) cut * nchan, nblocks - cut, nchan, fs, nsamp, st, max_seg, maps.data() + (size_t) cut * nchan, nullptr) ||
                gpsiq_chain_link(in.data() + (size_t) cut * nchan, maps.data() + (size_t) cut * nchan, nblocks - cut, nchan, fs, nsamp, mid_end, mid_prn,
                                 got.data() + (size_t) cut * nchan, got_end, got_prn)) { std::printf("parallel chain (continued) failed: %s\n", gpsiq_last_error()); return 2; }
        } else {
            int cuts[4] = {0, (int) (rng() % (nblocks + 1)), (int) (rng() % (nblocks + 1)), nblocks};
            if (cuts[1] > cuts[2]) std::swap(cuts[1], cuts[2]);
            std::vector<gpsiq_chain_est_t> sums((size_t) 3 * nchan), sums2((size_t) 3 * nchan);
            gpsiq_chain_est_t st[16];
            for (int r = 0; r < 3; ++r) gpsiq_chain_summary(in.data() + (size_t) cuts[r] * nchan, cuts[r + 1] - cuts[r], nchan, fs, nsamp, nullptr, &sums[(size_t) r * nchan]);
            for (int r = 0; r < 3; ++r) {
                gpsiq_chain_fold(sums.data(), r, nchan, st);
                gpsiq_chain_summary(in.data() + (size_t) cuts[r] * nchan, cuts[r + 1] - cuts[r], nchan, fs, nsamp, st, &sums2[(size_t) r * nchan]);
            }
            double carr[16]; int32_t prn[16];
            for (int r = 0; r < 3; ++r) {
                gpsiq_chain_fold(sums2.data(), r, nchan, st);
                const size_t off = (size_t) cuts[r] * nchan;
                const int nb = cuts[r + 1] - cuts[r];
                if (gpsiq_chain_maps(in.data() + off, nb, nchan, fs, nsamp, st, max_seg, maps.data() + off, nullptr) ||
                    gpsiq_chain_link(in.data() + off, maps.data() + off, nb, nchan, fs, nsamp, r ? carr : nullptr, r ? prn : nullptr,
                                     got.data() + off, carr, prn)) { std::printf("parallel chain (range %d) failed: %s\n", r, gpsiq_last_error()); return 2; }
                if (r == 0 && nb == 0) for (int i = 0; i < nchan; ++i) { carr[i] = 0.0; prn[i] = 0; }
            }
            for (int i = 0; i < nchan; ++i) { got_end[i] = carr[i]; got_prn[i] = prn[i]; }
        }
        blocks += (long) nblocks * nchan;
        for (size_t k = 0; k < want.size(); ++k)
            if (std::memcmp(&want[k], &got[k], 8)) { if (bad < 10) std::printf("case %d (variant %d, fs %g, nsamp %d, seg %d): start of block %zu slot %zu: %.17g != %.17g\n", it, variant, fs, nsamp, max_seg, k / nchan, k % nchan, got[k], want[k]); ++bad; }
        for (int i = 0; i < nchan; ++i)
            if (std::memcmp(&want_end[i], &got_end[i], 8) || want_prn[i] != got_prn[i]) { if (bad < 10) std::printf("case %d (variant %d): end of slot %d: %.17g (prn %d) != %.17g (prn %d)\n", it, variant, i, got_end[i], got_prn[i], want_end[i], want_prn[i]); ++bad; }
    }
    gpsiq_chain_stats(s1);
    // the join of a block's stretches as a scan (the device's form, gpsiq_lane.h join_stretches_scan) == the loop, every map of every case
    if (g_join_mismatch.load()) { std::printf("join as a scan differs from the loop in %ld of %ld maps\n", g_join_mismatch.load(), g_join_checked.load()); bad += g_join_mismatch.load(); }
    if (!g_join_checked.load()) { std::printf("the join check did not run\n"); ++bad; }
    std::printf("cases=%d blocks=%ld linked=%lu walked=%lu joins_checked=%ld bad=%ld\n", cases, blocks, (unsigned long) (s1[0] - s0[0]), (unsigned long) (s1[1] - s0[1]), g_join_checked.load(), bad);
    return bad ? 1 : 0;
}
