// eval_twin.cpp -- the device evaluation of GPSIQ_NCO_REFERENCE (csrc/gpsiq_eval.h: the code the kernels of
// gpsiq_eval_kernels.hip run, compiled here for the host) against the host path it replaces, on random and adversarial
// descriptors.  TEST INFRASTRUCTURE.
//   A  pack_chan + quantize_dchan            == quantize_one (gpsiq_host.cpp): same descriptor bytes, same error class
//   B  next_candidate, one at a time         == candidates() (gpsiq_exact.cpp), the recursive Euclid descent
//   C  eval_chan from the block's true start == eval_block: same descriptor, same patches; what it hands to the host walker
//                                               (an undecided candidate) is counted and must stay rare
//   D  the chain as a scan (Link)            == gpsiq_chain_link: every start state the scan calls known is the serial chain's,
//                                               whatever the association of the scan and wherever the timeline is cut in pieces
//   usage: eval_twin [seed] [cases]       prints  cases=.. chans=.. evals=.. host=.. evals_near=.. host_near=.. patches=.. known=.. unknown=.. bad=..
//   (host: evaluations handed to the host walker; _near: those whose descriptor was seeded within 1e-9 cycle of the start state)
#include "gpsiq_exact.cpp"
#include "gpsiq_chain.cpp"
#include "gpsiq_eval.h"
#include <random>

using namespace gpsiq;

static std::mt19937_64 rng;
static double up() { return std::uniform_real_distribution<double>(0.0, 1.0)(rng); }

static gpsiq_chan_t random_chan(double fs, int mode)
{
    gpsiq_chan_t c;
    std::memset(&c, 0, sizeof c);
    c.prn = 1 + (int) (rng() % 32);
    c.f_carr = (up() * 2 - 1) * 6000.0;
    if (mode == 1) c.f_carr = (up() * 2 - 1) * 30.0;
    if (mode == 2) { uint64_t b; std::memcpy(&b, &c.f_carr, 8); const int z = 20 + (int) (rng() % 28); b &= ~((UINT64_C(1) << z) - 1); std::memcpy(&c.f_carr, &b, 8); }
    if (mode == 3 && rng() % 4 == 0) c.f_carr = 0.0;
    c.f_code = 1.023e6 + c.f_carr / 1540.0;
    c.carr_phase = up();
    c.code_phase = up() * 1023.0;
    if (mode == 4) {                                               // phases on and next to boundaries
        c.carr_phase = std::ldexp((double) (rng() % 512), -9) + (rng() % 3 == 0 ? 0.0 : (up() - 0.3) * 1e-11);
        if (!(c.carr_phase >= 0.0 && c.carr_phase < 1.0)) c.carr_phase = 0.0;
        c.code_phase = (double) (rng() % 1023) + (rng() % 3 == 0 ? 0.0 : up() * 1e-9);
    }
    c.iword = (int) (rng() % 59); c.ibit = (int) (rng() % 30); c.icode = (int) (rng() % 20);
    if (mode == 5) c.iword = 57 + (int) (rng() % 3);                  // the end of dwrd
    c.gain = 0.3 + 0.7 * up();
    for (int w = 0; w < GPSIQ_N_DWRD; ++w) c.dwrd[w] = (uint32_t) (rng() & 0x3fffffffu) | (rng() % 5 == 0 ? 0xc0000000u : 0u);     // bits 31..30 are not data
    if (mode == 6) {                                               // one field out of range
        switch (rng() % 9) {
        case 0: c.prn = 33 + (int) (rng() % 5); break;
        case 1: c.f_carr = fs * 0.6; break;
        case 2: c.f_code = fs * 2.5; break;
        case 3: c.code_phase = 1023.0; break;
        case 4: c.iword = 60; break;
        case 5: c.ibit = 30; break;
        case 6: c.icode = -1; break;
        case 7: c.gain = std::nan(""); break;
        default: c.f_code = -1.0; break;
        }
    }
    return c;
}

struct HostChips {
    mutable CodeCache cc;
    unsigned operator()(int prn, unsigned chip) const { return cc.get(prn)[chip]; }
};
struct Collect {
    std::vector<gpsiq_patch_t> *out; uint32_t block; uint8_t slot;
    void operator()(uint32_t sample, uint16_t lut, uint8_t neg) { gpsiq_patch_t p; p.block = block; p.sample = sample; p.slot = slot; p.neg = neg; p.lut = lut; out->push_back(p); }
};

static long test_quantiser(int cases, long *chans)
{
    long bad = 0;
    for (int it = 0; it < cases * 200; ++it) {
        static const double rates[] = {2.6e6, 3.0e6, 10.0e6, 25.0e6, 1.1e6};
        const double fs = rates[it % 5], delt = 1.0 / fs;
        const int nsamp = it % 7 == 0 ? 1 + (int) (rng() % 3000000) : (int) (fs / 10.0);
        gpsiq_chan_t c = random_chan(fs, it % 8);
        if (it % 31 == 0) c.prn = 0;
        ev::DChan d;
        ev::pack_chan(c, &d);
        gpsiq_qchan_t want, got;
        const bool seeded = it % 3 == 0;
        const uint64_t seed = rng();
        const int rc = quantize_one(c, delt, nsamp, seeded ? &seed : nullptr, &want, nullptr);
        if (seeded && !(c.carr_phase >= 0.0 && c.carr_phase < 1.0)) continue;       // (eval_block's own business: it hands the quantiser 0.0 there)
        const int st = ev::quantize_dchan(d, delt, nsamp, seeded ? &seed : nullptr, &got);
        ++*chans;
        if (ev::qstatus_code(st) != rc || (rc == GPSIQ_OK && std::memcmp(&want, &got, sizeof want))) {
            if (bad++ < 5) std::printf("quantiser differs: rc %d, status %d (prn %d, f_carr %g, f_code %g, code_phase %.17g, pos %d/%d/%d, nsamp %d)\n", rc, st, c.prn, c.f_carr, c.f_code, c.code_phase, c.iword, c.ibit, c.icode, nsamp);
        }
        // the window of data bits == nav_bit() of gpsiq_exact.cpp
        if (c.prn > 0 && !(d.pos & ev::kBadPos))
            for (long b = 0; b < 45; ++b)
                if (ev::dchan_nav_bit(d, b) != nav_bit(c, b) && b < ev::kNavWindow) { if (bad++ < 5) std::printf("nav bit %ld differs (iword %d ibit %d)\n", b, c.iword, c.ibit); break; }
    }
    return bad;
}

static long test_candidates(int cases)
{
    long bad = 0;
    for (int it = 0; it < cases * 60; ++it) {
        const int k = it % 2 ? GPSIQ_CODE_FRAC_BITS : GPSIQ_CARR_FRAC_BITS - 9;
        const uint64_t M = UINT64_C(1) << k;
        const long nsamp = it % 5 == 0 ? 1 + (long) (rng() % 3000000) : 260000;
        uint64_t a = rng() & (M - 1), b = rng() & (M - 1), w = ((uint64_t) nsamp << (it % 2 ? 12 : 5)) + (uint64_t) nsamp / 2 + 4;
        if (it % 7 == 0) b = (uint64_t) ((double) M * (it % 2 ? 0.39 : 0.001) * (1.0 + up() * 1e-3));      // the addends of the real signal
        if (it % 11 == 0) w = 1 + rng() % (M / 16);                                                     // wide windows: many hits
        if (it % 13 == 0) b = M / (2 + rng() % 7) + (rng() % 3);                                        // near-rational steps
        if (it % 17 == 0) a = (M - w / 2) & (M - 1);
        if (2 * w + 1 >= M) continue;
        std::vector<long> want;
        const bool listed = candidates(a, b, k, w, nsamp, 4096, &want);
        if (!listed) continue;
        std::vector<long> got;
        long base = 0;
        for (;;) {
            const uint64_t n = ev::next_candidate(a, b, k, w, nsamp, base);
            if (n == ev::kNoHit) break;
            got.push_back((long) n);
            base = (long) n + 1;
            if (got.size() > 5000) break;
        }
        if (got != want) { if (bad++ < 5) std::printf("candidates differ: a %llx b %llx k %d w %llu nsamp %ld: %zu / %zu\n", (unsigned long long) a, (unsigned long long) b, k, (unsigned long long) w, nsamp, got.size(), want.size()); }
    }
    return bad;
}

static bool patch_less(const gpsiq_patch_t &a, const gpsiq_patch_t &b) { return a.sample < b.sample; }

static long test_evaluation(int cases, long *chans, long *host, long *npatch, long *near, long *host_near)
{
    long bad = 0;
    HostChips chips;
    CodeCache codes;
    for (int it = 0; it < cases * 400; ++it) {
        static const double rates[] = {2.6e6, 3.0e6, 10.0e6, 25.0e6, 2.6e6, 25.0e6};
        const double fs = rates[it % 6], delt = 1.0 / fs;
        int nsamp = it % 9 == 0 ? 1000 + (int) (rng() % 400000) : (int) (fs / 10.0);
        // what the descriptor is seeded from: the start state itself (the host path), or an estimate of it some way off -- the
        // device renders from chain_prepare's estimate (good to ~1e-11 cycle; here also far worse ones) and learns the truth later.
        // Estimates 1e-6 .. 0.3 cycle off make every sample a candidate (the host walks them all): few, and on short blocks
        int em = (int) (rng() % 6);
        if (it % 16 == 5) { em = 6 + (int) (rng() % 2); nsamp = 2000 + (int) (rng() % 30000); }
        gpsiq_chan_t c = random_chan(fs, it % 6);
        double start = up();
        if (it % 6 == 4) start = std::ldexp((double) (rng() % 512), -9) + (rng() % 3 == 0 ? 0.0 : (up() - 0.3) * 2e-11);      // a start next to a LUT step
        if (it % 97 == 0) start = 1.0;
        if (!(start >= 0.0 && start <= 1.0)) start = 0.25;
        std::vector<gpsiq_patch_t> want, got;
        gpsiq_qchan_t qw, qg;
        double est = start;
        if (em >= 3) est = start + (up() - 0.5) * (em == 3 ? 1e-13 : em == 4 ? 1e-11 : em == 5 ? 1e-9 : em == 6 ? 1e-6 : 0.3);
        if (est < 0.0) est += 1.0;
        if (est >= 1.0) est -= 1.0;
        if (!(est >= 0.0 && est < 1.0)) est = 0.0;
        const uint64_t seed = carr_phase_to_fixed(est);
        const bool own = est == start;
        const int rc = eval_block(c, start, delt, nsamp, 7, 3, &codes, &qw, &want, false, nullptr, 0, own ? nullptr : &seed);
        ev::DChan d;
        ev::pack_chan(c, &d);
        Collect col = {&got, 7, 3};
        const int st = ev::eval_chan(d, start, est, delt, nsamp, chips, col, &qg);
        ++*chans;
        if (em < 6) ++*near;                           // an estimate as chain_prepare gives them, or somewhat worse
        if (em < 6 && st == ev::kEvalHost) ++*host_near;
        if (st < 0) { if (rc != ev::qstatus_code(-st)) { if (bad++ < 5) std::printf("evaluation: status %d against rc %d\n", st, rc); } continue; }
        if (rc != GPSIQ_OK) { if (bad++ < 5) std::printf("evaluation: ok against rc %d (%s)\n", rc, gpsiq_last_error()); continue; }
        if (std::memcmp(&qw, &qg, sizeof qw)) { if (bad++ < 5) std::printf("evaluation: descriptors differ\n"); continue; }
        if (st == ev::kEvalHost) { ++*host; if (std::getenv("TWIN_VERBOSE")) std::printf("host: mode %d fs %g nsamp %d f_carr %.6g start %.17g\n", it % 6, fs, nsamp, c.f_carr, start); continue; }
        std::sort(want.begin(), want.end(), patch_less);
        *npatch += (long) want.size();
        bool same = want.size() == got.size();
        for (size_t k = 0; same && k < want.size(); ++k) same = !std::memcmp(&want[k], &got[k], sizeof(gpsiq_patch_t));
        if (!same) { if (bad++ < 5) std::printf("evaluation: patches differ (%zu / %zu), fs %g nsamp %d f_carr %.17g start %.17g code_phase %.17g\n", got.size(), want.size(), fs, nsamp, c.f_carr, start, c.code_phase); }
    }
    return bad;
}

// the scan over blocks [b0, b1) of slot i in a random association; carry: the offset and the state the blocks before left
struct Carry { int64_t d; double y; int prn; bool known; };
static void scan_piece(const ev::DChan *in, const lane::Rec *rec, int nchan, int i, int b0, int b1, double delt, Carry *cy,
                       double *start, char *known, long *unknown)
{
    const int n = b1 - b0;
    if (n <= 0) return;
    std::vector<ev::Link> el((size_t) n);
    std::vector<char> el_ok((size_t) n), seed((size_t) n), active((size_t) n);
    for (int k = 0; k < n; ++k) {
        const int b = b0 + k;
        const ev::DChan &d = in[(size_t) b * nchan + i];
        const int prev_prn = b > 0 ? (in[(size_t) (b - 1) * nchan + i].prn > 0 ? in[(size_t) (b - 1) * nchan + i].prn : 0) : 0;
        active[k] = d.prn > 0;
        seed[k] = active[k] && (b == 0 || prev_prn != d.prn);
        bool ok = true;
        int64_t d0 = 0;
        const lane::Rec &r = rec[(size_t) b * nchan + i];
        if (!active[k]) el[k] = ev::link_element(true, 0, r, 0.0, &ok);
        else if (seed[k]) { ok = lane::exact_units(d.carr_phase, r.xs, &d0); el[k] = ev::link_element(true, ok ? d0 : 0, r, 0.0, &ok); if (!lane::exact_units(d.carr_phase, r.xs, &d0)) ok = false; }
        else el[k] = ev::link_element(false, 0, rec[(size_t) (b - 1) * nchan + i], r.xs, &ok);
        el_ok[k] = ok;
    }
    // inclusive scan, pairs combined in a random tree order
    std::vector<ev::Link> inc(el);
    std::vector<int> lo((size_t) n), hi((size_t) n);
    // simple: recursive halving at random split points
    struct Rec2 { static void go(std::vector<ev::Link> &v, int a, int b) {
        if (b - a <= 1) return;
        const int m = a + 1 + (int) (rng() % (unsigned) (b - a - 1));
        go(v, a, m); go(v, m, b);
        const ev::Link left = v[m - 1];
        for (int k = m; k < b; ++k) v[k] = ev::link_compose(left, v[k]);
    } };
    Rec2::go(inc, 0, n);
    bool kn = cy->known;
    double y = cy->y;
    int64_t d_last = cy->d;
    for (int k = 0; k < n; ++k) {
        const int b = b0 + k;
        const size_t at = (size_t) b * nchan + i;
        const ev::DChan &d = in[at];
        const int64_t dd = ev::link_apply(inc[k], cy->d);
        if (k == n - 1) d_last = dd;
        if (!active[k]) { start[at] = 0.0; known[at] = 1; kn = false; cy->prn = 0; continue; }
        if (seed[k]) { kn = true; y = d.carr_phase; }
        known[at] = kn;
        start[at] = kn ? y : -1.0;
        if (!kn) ++*unknown;
        double next = 0.0;
        const bool linked = kn && el_ok[k] && std::fabs(d.f_carr * delt) < 0.5 && ev::link_check(rec[at], dd, &next);
        kn = linked; y = next;
        cy->prn = d.prn;
    }
    cy->known = kn; cy->y = y; cy->d = d_last;
}

static long test_link(int cases, long *nknown, long *nunknown)
{
    long bad = 0;
    for (int it = 0; it < cases; ++it) {
        static const double rates[] = {2.6e6, 3.0e6, 10.0e6, 25.0e6, 2097152.0, 2.6e6};
        const double fs = rates[it % 6];
        int nsamp = (int) (fs / 10.0);
        if (it % 5 == 3) nsamp = 20000 + (int) (rng() % 50000);
        const int nchan = 1 + (int) (rng() % 16), nblocks = 20 + (int) (rng() % 80);
        std::vector<gpsiq_chain_in_t> in((size_t) nblocks * nchan);
        std::vector<ev::DChan> din((size_t) nblocks * nchan);
        for (int i = 0; i < nchan; ++i) {
            const int mode = (int) (rng() % 8);
            double f = (up() * 2 - 1) * 6000.0, df = (up() * 2 - 1) * 0.9;
            if (mode == 1) { f = (up() * 2 - 1) * 30.0; df = (up() * 2 - 1) * 3.0; }
            if (mode == 2) { f = (up() * 2 - 1) * 300.0; df = (up() * 2 - 1) * 20.0; }
            int prn = 1 + (int) (rng() % 32);
            for (int b = 0; b < nblocks; ++b) {
                gpsiq_chain_in_t &d = in[(size_t) b * nchan + i];
                if (rng() % 97 == 0) prn = 1 + (int) (rng() % 32);
                d.prn = (mode == 3 && (b / 7) % 3 == 1) ? 0 : prn;
                d.carr_phase = up();
                if (rng() % 13 == 0) d.carr_phase = std::ldexp((double) (rng() % 1024), -10);
                d.f_carr = f + df * b + (up() - 0.5) * 0.05;
                if (mode == 4) { uint64_t bb; std::memcpy(&bb, &d.f_carr, 8); const int z = 20 + (int) (rng() % 28); bb &= ~((UINT64_C(1) << z) - 1); if (rng() & 1) bb |= UINT64_C(1) << z; std::memcpy(&d.f_carr, &bb, 8); }
                if (mode == 5 && b % 11 == 5) d.f_carr = 0.0;
                d.reserved = 0;
                ev::DChan &e = din[(size_t) b * nchan + i];
                std::memset(&e, 0, sizeof e);
                e.f_carr = d.f_carr; e.carr_phase = d.carr_phase; e.prn = d.prn;
            }
        }
        std::vector<double> want((size_t) nblocks * nchan), got((size_t) nblocks * nchan, -2.0);
        std::vector<char> known((size_t) nblocks * nchan, 0);
        double want_end[16]; int32_t want_prn[16];
        if (gpsiq_reference_chain(in.data(), nblocks, nchan, fs, nsamp, nullptr, nullptr, want.data(), want_end, want_prn)) { std::printf("serial chain failed: %s\n", gpsiq_last_error()); return 1000; }
        std::vector<gpsiq_chain_map_t> maps((size_t) nblocks * nchan);
        if (gpsiq_chain_maps(in.data(), nblocks, nchan, fs, nsamp, nullptr, 1 + (int) (rng() % 32), maps.data(), nullptr)) { std::printf("maps failed\n"); return 1000; }
        const lane::Rec *rec = reinterpret_cast<const lane::Rec *>(maps.data());
        // two or three pieces, the carry handed from one to the next
        int cuts[4] = {0, (int) (rng() % (nblocks + 1)), (int) (rng() % (nblocks + 1)), nblocks};
        if (cuts[1] > cuts[2]) std::swap(cuts[1], cuts[2]);
        for (int i = 0; i < nchan; ++i) {
            Carry cy = {0, 0.0, 0, false};
            for (int p = 0; p < 3; ++p) scan_piece(din.data(), rec, nchan, i, cuts[p], cuts[p + 1], 1.0 / fs, &cy, got.data(), known.data(), nunknown);
            // the end of the slot: the carry's state is the accumulator after the last block when that one linked
            if (cy.known && in[(size_t) (nblocks - 1) * nchan + i].prn > 0 && std::memcmp(&cy.y, &want_end[i], 8)) { if (bad++ < 10) std::printf("case %d: end of slot %d: %.17g != %.17g\n", it, i, cy.y, want_end[i]); }
        }
        for (size_t k = 0; k < want.size(); ++k) {
            if (!known[k]) continue;
            ++*nknown;
            if (std::memcmp(&want[k], &got[k], 8)) { if (bad++ < 10) std::printf("case %d (fs %g, nsamp %d): start of block %zu slot %zu: %.17g != %.17g\n", it, fs, nsamp, k / nchan, k % nchan, got[k], want[k]); }
        }
    }
    return bad;
}

int main(int argc, char **argv)
{
    rng.seed(argc > 1 ? (unsigned long) atol(argv[1]) : 1);
    const int cases = argc > 2 ? atoi(argv[2]) : 20;
    long chans = 0, host = 0, npatch = 0, nknown = 0, nunknown = 0;
    long bad = test_quantiser(cases, &chans);
    bad += test_candidates(cases);
    long evals = 0;
    long near = 0, host_near = 0;
    bad += test_evaluation(cases, &evals, &host, &npatch, &near, &host_near);
    bad += test_link(cases * 2, &nknown, &nunknown);
    std::printf("cases=%d chans=%ld evals=%ld host=%ld evals_near=%ld host_near=%ld patches=%ld known=%ld unknown=%ld bad=%ld\n", cases, chans, evals, host, near, host_near,
                npatch, nknown, nunknown, bad);
    return bad ? 1 : 0;
}
