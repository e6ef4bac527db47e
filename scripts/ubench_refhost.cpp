// ubench_refhost.cpp -- the host half of GPSIQ_NCO_REFERENCE per block and channel on this host, one thread, by part: the
// quantiser, the candidate search (two Euclid-style descents), the serial carrier chain (NcoWalk: wrap-to-wrap table) and the
// whole evaluation of a block from its start state (quantiser + candidates + drift enclosure, walks only where it is
// undecided), at 2.6 and 25 Msps, Doppler uniform in +-5 kHz (bench.py's synthetic channels).
// Build:  g++ -O3 -std=c++17 -ffp-contract=off -I multi-sdr-gps-sim_amd/csrc scripts/ubench_refhost.cpp multi-sdr-gps-sim_amd/csrc/gpsiq_host.cpp -lpthread
#include "gpsiq_exact.cpp"
#include <chrono>
#include <random>
using namespace gpsiq;
static double now() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main()
{
    for (double fs : {2.6e6, 10e6, 25e6}) {
        const double delt = 1.0 / fs; const int ns = (int) (fs / 10);
        const int N = 4000;
        std::mt19937_64 rng(7);
        std::uniform_real_distribution<double> up(0.0, 1.0), uf(-5000, 5000);
        std::vector<gpsiq_chan_t> ch(N);
        for (auto &e : ch) {
            std::memset(&e, 0, sizeof e);
            e.prn = 7 /* one satellite: an evaluation task stays with its channel, the C/A code is generated once */; e.f_carr = uf(rng); e.f_code = 1.023e6 + e.f_carr / 1540.0;
            e.carr_phase = up(rng); e.code_phase = up(rng) * 1023; e.gain = 0.5;
            e.iword = (int) (rng() % 50); e.ibit = (int) (rng() % 30); e.icode = (int) (rng() % 20);
            for (int k = 0; k < GPSIQ_N_DWRD; ++k) e.dwrd[k] = (uint32_t) rng() & 0x3fffffffu;
        }
        std::vector<gpsiq_qchan_t> q(N);
        double tq = 1e30, tc = 1e30, tw = 1e30, te = 1e30, acc = 0; size_t ncand = 0, npatch = 0;
        uint64_t s0[4], s1[4];
        gpsiq_reference_stats(s0);
        for (int pass = 0; pass < 3; ++pass) {
            double t0 = now();
            for (int i = 0; i < N; ++i) quantize_one(ch[i], delt, ns, nullptr, &q[i], nullptr);
            tq = std::min(tq, (now() - t0) / N);
            std::vector<long> a, b; ncand = 0;
            const uint64_t w_carr = ((uint64_t) ns << (GPSIQ_CARR_FRAC_BITS - 54)) + (uint64_t) ns / 2 + 4;
            const uint64_t w_code = ((uint64_t) ns << (GPSIQ_CODE_FRAC_BITS - 44)) + (uint64_t) ns / 2 + 4;
            t0 = now();
            for (int i = 0; i < N; ++i) {
                a.clear(); b.clear();
                candidates(q[i].carr_phase, (uint64_t) q[i].carr_step, GPSIQ_CARR_FRAC_BITS - 9, w_carr, ns, 256, &a);
                candidates(q[i].code_frac, q[i].code_step, GPSIQ_CODE_FRAC_BITS, w_code, ns, 256, &b);
                ncand += a.size() + b.size();
            }
            tc = std::min(tc, (now() - t0) / N);
            t0 = now();
            for (int i = 0; i < N; ++i) acc += chain_block(ch[i].f_carr, delt, ns, ch[i].carr_phase);
            tw = std::min(tw, (now() - t0) / N);
            CodeCache codes; std::vector<gpsiq_patch_t> out;
            t0 = now();
            for (int i = 0; i < N; ++i) (void) eval_block(ch[i], ch[i].carr_phase, delt, ns, 0, 0, &codes, &q[i], &out);
            te = std::min(te, (now() - t0) / N); npatch = out.size();
        }
        gpsiq_reference_stats(s1);
        std::printf("fs %4.1f Msps: quantiser %.3f us, candidate search %.3f us (%.3f candidates per block and channel), carrier chain %.3f us, "
                    "evaluation from the start state %.3f us (%zu patches in %d blocks; %.4f of the candidate states decided without a walk)  [%g]\n",
                    fs / 1e6, tq, tc, (double) ncand / N, tw, te, npatch, N, (double) (s1[1] - s0[1]) / (double) std::max<uint64_t>(1, s1[0] - s0[0]), acc);
    }
    return 0;
}
