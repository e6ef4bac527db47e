// sanitize_refwalk.cpp -- the one-pass reference walker consumed piece by piece while it runs (what generate_reference and
// gpsiq_generate_batch_multi do on the device side), host only, for ThreadSanitizer / ASan + UBSan builds: the pieces taken
// while the walkers run must add up to what one call over the whole timeline gives; two timelines walked at once on two host
// threads; the chain and the evaluation called on their own (channel subsets, time ranges).  Run once with the default pool
// and once under GPSIQ_THREADS=2 (fewer threads than channels: piece-major order).  TEST INFRASTRUCTURE.
#include "gpsiq_internal.h"

#include <cstdio>
#include <cstring>
#include <random>
#include <time.h>

using namespace gpsiq;

static void *run_walk(void *w) { static_cast<RefWalk *>(w)->run(); return nullptr; }

int main()
{
    const int nb = 72, nc = 12, ns = 30000;
    const double fs = 10.0e6, delt = 1.0 / fs;
    std::mt19937_64 rng(11);
    std::uniform_real_distribution<double> uf(-5000.0, 5000.0), up(0.0, 1.0);
    std::vector<gpsiq_chan_t> ch((size_t) nb * nc);
    for (int b = 0; b < nb; ++b)
        for (int c = 0; c < nc; ++c) {
            gpsiq_chan_t &e = ch[(size_t) b * nc + c];
            std::memset(&e, 0, sizeof e);
            e.prn = (c == 3 && b >= 40 && b < 55) ? 0 : (c == 5 && b >= 70) ? 29 : 1 + c;      // a slot going out of view, one re-allocated
            e.f_carr = b == 0 ? uf(rng) : ch[(size_t) (b - 1) * nc + c].f_carr + 0.3;
            e.f_code = 1.023e6 + e.f_carr / 1540.0;
            // phases a hair off LUT / chip boundaries: plenty of candidates, some patches
            e.carr_phase = ((double) (rng() % 512) + 1e-12) / 512.0;
            e.code_phase = (double) (rng() % 1023) + 1e-9;
            if (c % 3 == 0) {       // a whole number of samples per LUT step and per chip, phases a hair below a boundary: patches
                e.f_carr = fs / 512.0 / (double) (5 + c);
                e.f_code = fs / 25.0;
                e.carr_phase = (double) (rng() % 512) / 512.0 + 0x1p-12 - 0x1p-50;
                e.code_phase = (double) (1 + rng() % 6) - 0x1p-41;
            }
            e.gain = 0.5; e.iword = (int) (rng() % 50); e.ibit = (int) (rng() % 30); e.icode = (int) (rng() % 20);
            for (int k = 0; k < GPSIQ_N_DWRD; ++k) e.dwrd[k] = (uint32_t) rng() & 0x3fffffffu;
        }
    // one call over the whole timeline
    std::vector<gpsiq_qchan_t> q0((size_t) nb * nc), q1((size_t) nb * nc);
    std::vector<gpsiq_patch_t> p0;
    double carr0[GPSIQ_MAX_CHAN];
    int prn0[GPSIQ_MAX_CHAN];
    if (reference_timeline(ch.data(), nb, nc, delt, ns, q0.data(), &p0, carr0, prn0) != GPSIQ_OK) { std::fprintf(stderr, "timeline: %s\n", gpsiq_last_error()); return 1; }
    // the same, consumed in ragged pieces while the walkers run
    for (int rep = 0; rep < 2; ++rep) {
        std::vector<int> ends = {7, 8, 31, 64, 65, nb};      // ragged, incl. a piece of one block
        RefWalk w(ch.data(), nb, nc, delt, ns, q1.data(), nullptr, nullptr, ends);
        pthread_t th;
        if (pthread_create(&th, nullptr, run_walk, &w) != 0) return 1;
        std::vector<gpsiq_patch_t> all, piece;
        for (size_t k = 0; k < w.npieces(); ++k) {
            if (w.wait_piece(k) != GPSIQ_OK) { std::fprintf(stderr, "piece %zu: %s\n", k, w.err); return 1; }
            w.take_patches(k, &piece, true);
            const int b0 = k ? w.ends[k - 1] : 0;
            for (gpsiq_patch_t p : piece) { p.block += (uint32_t) b0; all.push_back(p); }
            // the piece's descriptors are final as soon as the piece is complete
            if (std::memcmp(&q1[(size_t) b0 * nc], &q0[(size_t) b0 * nc], (size_t) (w.ends[k] - b0) * nc * sizeof(gpsiq_qchan_t)) != 0) { std::fprintf(stderr, "piece %zu: descriptors differ\n", k); return 1; }
        }
        pthread_join(th, nullptr);
        if (all.size() != p0.size() || (all.size() && std::memcmp(all.data(), p0.data(), all.size() * sizeof(gpsiq_patch_t)) != 0)) { std::fprintf(stderr, "patches differ: %zu vs %zu\n", all.size(), p0.size()); return 1; }
        for (int c = 0; c < nc; ++c)
            if (w.carr_end[c] != carr0[c] || w.last_prn[c] != prn0[c]) { std::fprintf(stderr, "end state differs in slot %d\n", c); return 1; }
    }
    // two timelines walked at the same time on two host threads (two contexts of one process): they share the pool, neither
    // falls back to one thread, both equal the single call
    {
        std::vector<gpsiq_qchan_t> qa((size_t) nb * nc), qb((size_t) nb * nc);
        std::vector<int> ends = {16, 40, nb};
        RefWalk wa(ch.data(), nb, nc, delt, ns, qa.data(), nullptr, nullptr, ends), wb(ch.data(), nb, nc, delt, ns, qb.data(), nullptr, nullptr, ends);
        pthread_t ta, tb;
        if (pthread_create(&ta, nullptr, run_walk, &wa) != 0 || pthread_create(&tb, nullptr, run_walk, &wb) != 0) return 1;
        pthread_join(ta, nullptr); pthread_join(tb, nullptr);
        if (wa.rc != GPSIQ_OK || wb.rc != GPSIQ_OK) { std::fprintf(stderr, "concurrent walks: %s %s\n", wa.err, wb.err); return 1; }
        if (std::memcmp(qa.data(), q0.data(), qa.size() * sizeof(gpsiq_qchan_t)) != 0 || std::memcmp(qb.data(), q0.data(), qb.size() * sizeof(gpsiq_qchan_t)) != 0) { std::fprintf(stderr, "concurrent walks: descriptors differ\n"); return 1; }
        for (int c = 0; c < nc; ++c)
            if (wa.carr_end[c] != carr0[c] || wb.carr_end[c] != carr0[c]) { std::fprintf(stderr, "concurrent walks: end state differs in slot %d\n", c); return 1; }
    }
    // the two halves on their own: the chain over channel subsets (as ranks of a sharded run walk them), then the blocks
    // evaluated from their start states in two time ranges == the single call
    {
        std::vector<gpsiq_chain_in_t> cin((size_t) nb * nc);
        gpsiq_chain_inputs(ch.data(), nb * nc, cin.data());
        std::vector<double> start((size_t) nb * nc);
        const int split = 5;
        for (int part = 0; part < 2; ++part) {
            const int c0 = part ? split : 0, c1 = part ? nc : split, w = c1 - c0;
            std::vector<gpsiq_chain_in_t> cols((size_t) nb * w);
            std::vector<double> st((size_t) nb * w), end((size_t) w);
            std::vector<int32_t> last((size_t) w);
            for (int b = 0; b < nb; ++b) for (int c = 0; c < w; ++c) cols[(size_t) b * w + c] = cin[(size_t) b * nc + c0 + c];
            if (gpsiq_reference_chain(cols.data(), nb, w, fs, ns, nullptr, nullptr, st.data(), end.data(), last.data()) != GPSIQ_OK) { std::fprintf(stderr, "chain: %s\n", gpsiq_last_error()); return 1; }
            for (int b = 0; b < nb; ++b) for (int c = 0; c < w; ++c) start[(size_t) b * nc + c0 + c] = st[(size_t) b * w + c];
            for (int c = 0; c < w; ++c)
                if (end[(size_t) c] != carr0[c0 + c] || last[(size_t) c] != prn0[c0 + c]) { std::fprintf(stderr, "chain: end state differs in slot %d\n", c0 + c); return 1; }
        }
        std::vector<gpsiq_qchan_t> qs((size_t) nb * nc);
        std::vector<gpsiq_patch_t> ps(p0.size() + 8), all;
        const int cut = 29;
        for (int part = 0; part < 2; ++part) {
            const int b0 = part ? cut : 0, b1 = part ? nb : cut;
            int np = 0;
            if (gpsiq_reference_seeded(&ch[(size_t) b0 * nc], b1 - b0, nc, fs, ns, &start[(size_t) b0 * nc], &qs[(size_t) b0 * nc], ps.data(), (int) ps.size(), &np) != GPSIQ_OK) { std::fprintf(stderr, "seeded: %s\n", gpsiq_last_error()); return 1; }
            for (int k = 0; k < np; ++k) { gpsiq_patch_t p = ps[(size_t) k]; p.block += (uint32_t) b0; all.push_back(p); }
        }
        if (std::memcmp(qs.data(), q0.data(), qs.size() * sizeof(gpsiq_qchan_t)) != 0) { std::fprintf(stderr, "seeded: descriptors differ\n"); return 1; }
        if (all.size() != p0.size() || std::memcmp(all.data(), p0.data(), all.size() * sizeof(gpsiq_patch_t)) != 0) { std::fprintf(stderr, "seeded: patches differ: %zu vs %zu\n", all.size(), p0.size()); return 1; }
    }
    // the chain linked through certified maps that LAND LATE (generate_reference with level 1 on the device: a thread of the HIP
    // runtime releases a range of blocks when its maps are in host memory): the walkers may not touch a map before it is
    // released, the released ranges grow while they run, and the result is the single call's.  Then a walk that is aborted
    // while maps are still pending: every task still finishes once the ranges are released (what the error path relies on).
    {
        std::vector<gpsiq_chain_in_t> cin((size_t) nb * nc);
        gpsiq_chain_inputs(ch.data(), nb * nc, cin.data());
        std::vector<gpsiq_chain_map_t> maps((size_t) nb * nc), late((size_t) nb * nc);
        if (gpsiq_chain_maps(cin.data(), nb, nc, fs, ns, nullptr, 5, maps.data(), nullptr) != GPSIQ_OK) { std::fprintf(stderr, "maps: %s\n", gpsiq_last_error()); return 1; }
        struct Lander { RefWalk *w; const gpsiq_chain_map_t *from; gpsiq_chain_map_t *to; int nc, cut, nb; };
        auto land = [](void *arg) -> void * {
            Lander &l = *static_cast<Lander *>(arg);
            struct timespec ts = {0, 300000};
            nanosleep(&ts, nullptr);
            std::memcpy(l.to, l.from, (size_t) l.cut * l.nc * sizeof(gpsiq_chain_map_t));                  // the head's maps arrive ...
            l.w->release_maps(l.cut);
            nanosleep(&ts, nullptr);
            std::memcpy(l.to + (size_t) l.cut * l.nc, l.from + (size_t) l.cut * l.nc, (size_t) (l.nb - l.cut) * l.nc * sizeof(gpsiq_chain_map_t));
            l.w->release_maps(l.nb);                                                                       // ... then the rest
            return nullptr;
        };
        uint64_t stats0[2], stats1[2];
        gpsiq_chain_stats(stats0);
        for (int rep = 0; rep < 3; ++rep) {
            std::memset(late.data(), 0xff, late.size() * sizeof(gpsiq_chain_map_t));      // garbage until landed: must not be read early
            std::vector<gpsiq_qchan_t> qm((size_t) nb * nc);
            std::vector<double> st((size_t) nb * nc);
            std::vector<int> ends = {7, 8, 31, 64, 65, nb};
            RefWalk w(ch.data(), nb, nc, delt, ns, qm.data(), nullptr, nullptr, ends);
            w.in = cin.data(); w.maps = late.data(); w.maps_upto.store(0); w.start_out = st.data();
            Lander l = {&w, maps.data(), late.data(), nc, 31 - rep, nb};                   // the head ends on / before a piece boundary
            pthread_t th, tl;
            if (pthread_create(&th, nullptr, run_walk, &w) != 0 || pthread_create(&tl, nullptr, land, &l) != 0) return 1;
            const bool aborted = rep == 2;
            if (aborted) w.abort();
            for (size_t k = 0; k < w.npieces(); ++k) {
                if (w.wait_piece(k) != GPSIQ_OK) { std::fprintf(stderr, "late maps, piece %zu: %s\n", k, w.err); return 1; }
                const int b0 = k ? w.ends[k - 1] : 0;
                if (!aborted && std::memcmp(&qm[(size_t) b0 * nc], &q0[(size_t) b0 * nc], (size_t) (w.ends[k] - b0) * nc * sizeof(gpsiq_qchan_t)) != 0) { std::fprintf(stderr, "late maps, piece %zu: descriptors differ\n", k); return 1; }
            }
            pthread_join(tl, nullptr); pthread_join(th, nullptr);
            if (aborted) continue;
            for (int c = 0; c < nc; ++c)
                if (w.carr_end[c] != carr0[c] || w.last_prn[c] != prn0[c]) { std::fprintf(stderr, "late maps: end state differs in slot %d\n", c); return 1; }
        }
        gpsiq_chain_stats(stats1);
        if (stats1[0] - stats0[0] < (uint64_t) nb * nc) { std::fprintf(stderr, "late maps: only %llu blocks were linked through their map\n", (unsigned long long) (stats1[0] - stats0[0])); return 1; }
    }
    // the pool itself: thousands of tiny jobs from two submitting threads at once (a job wakes only as many workers as it wants
    // helpers and polls for the ones still inside before it sleeps; a job's record lives on its submitter's stack): every index
    // of every job is visited exactly once
    {
        struct Sub { int jobs; long sum; bool ok; };
        auto body = [](void *arg) -> void * {
            Sub &s = *static_cast<Sub *>(arg);
            std::vector<unsigned char> seen;
            for (int j = 0; j < s.jobs; ++j) {
                const int n = 1 + (j * 37) % 300, want = (j % 5) + 1, grain = 1 + j % 7;
                seen.assign((size_t) n, 0);
                gpsiq::parallel_for(n, want, grain, [](void *p, int b0, int b1) {
                    unsigned char *v = static_cast<unsigned char *>(p);
                    for (int b = b0; b < b1; ++b) __atomic_add_fetch(&v[b], 1, __ATOMIC_RELAXED);
                }, seen.data());
                for (int b = 0; b < n; ++b) { if (seen[(size_t) b] != 1) s.ok = false; s.sum += seen[(size_t) b]; }
            }
            return nullptr;
        };
        Sub a = {1500, 0, true}, b = {1500, 0, true};
        pthread_t ta, tb;
        if (pthread_create(&ta, nullptr, body, &a) != 0 || pthread_create(&tb, nullptr, body, &b) != 0) return 1;
        pthread_join(ta, nullptr); pthread_join(tb, nullptr);
        if (!a.ok || !b.ok || a.sum != b.sum || a.sum <= 0) { std::fprintf(stderr, "pool: an index was visited %s\n", a.ok && b.ok ? "a wrong number of times in total" : "not exactly once"); return 1; }
    }
    if (p0.empty()) { std::fprintf(stderr, "the scenario should have patches\n"); return 1; }
    std::printf("ok\n");
    return 0;
}
