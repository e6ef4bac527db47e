// gpsiq_chain.cpp -- host side of the time-parallel carrier chain of GPSIQ_NCO_REFERENCE (gpsiq_lane.h has the method):
//   prepare   one scan down every slot: exact real phase (128-bit) + modelled drift -> an estimate of every block's start;
//   maps      level 1, every block on its own: the certified map of the block (host threads here; the device runs the same
//             lane code in gpsiq_chain_kernels.hip, one lane per stretch);
//   link      level 2, the chain itself: one exact subtraction / range check / addition per block, a true walk of the rare
//             block whose map does not apply (reference: gps.c:2821-2826, re-seeding gps.c:2208-2214).
// The results are gpsiq_reference_chain's, bit for bit (tests/test_chain_parallel.py, tests/soak_chain_parallel.py).
#include "gpsiq_internal.h"
#include "gpsiq_lane.h"
#include "gpsiq_eval.h"

#include <cmath>
#include <cstdlib>
#include <cstring>

namespace gpsiq {

using lane::Prep;
using lane::Rec;
using lane::u128;

static_assert(sizeof(gpsiq_chain_map_t) == sizeof(Rec), "gpsiq_chain_map_t is lane::Rec");
static_assert(sizeof(Prep) == 32 && sizeof(Rec) == 56 && sizeof(gpsiq_chain_est_t) == 56, "layouts shared with the device");

// One slot's estimator between two blocks, unpacked.
struct EstState {
    u128   R = 0;
    double D = 0.0, carr = 0.0, f_prev = 0.0;
    int    prn = 0;              // -1: unknown (a range summarised on its own: its first block is taken to continue the slot)
    bool   exact = false;
    int    first_prn = 0;        // the satellite of block 0 where prn was unknown there, and its carr_phase: the fold decides
    double first_phase = 0.0;
    void load(const gpsiq_chain_est_t &e)
    {
        R = ((u128) e.r_hi << 64) | e.r_lo; D = e.drift; carr = e.carr; f_prev = e.f_carr; prn = e.prn; exact = (e.flags & GPSIQ_CHAIN_EXACT) != 0;
        first_prn = 0; first_phase = 0.0;
    }
    void store(gpsiq_chain_est_t *e, int flags) const
    {
        e->r_hi = (uint64_t) (R >> 64); e->r_lo = (uint64_t) R; e->drift = D; e->carr = first_prn ? first_phase : carr; e->f_carr = f_prev; e->prn = prn;
        e->flags = (exact ? GPSIQ_CHAIN_EXACT : 0) | flags;
        e->first_prn = first_prn; e->reserved = 0;
    }
};

// The scan down slot i.  prep may be null (summaries only); with_drift false leaves the drift at zero (the phase pass of a
// time-sharded run: the drift model wants the absolute phase, which the ranks only know after the first exchange).
// Returns whether the slot was re-seeded inside the range (its end state then does not depend on the state before).
bool chain_scan_slot(const gpsiq_chain_in_t *in, int nblocks, int nchan, int i, double delt, int nsamp, EstState *st, bool with_drift,
                     Prep *prep, double *c_before)
{
    bool reseeded = false;
    if (c_before) *c_before = st->prn > 0 ? st->f_prev * delt : 0.0;
    for (int b = 0; b < nblocks; ++b) {
        const gpsiq_chain_in_t &d = in[(size_t) b * nchan + i];
        Prep p = {0.0, 0.0, 0.0, lane::kSkip, 0};
        if (d.prn <= 0) {
            st->prn = 0; st->exact = false;
            if (prep) prep[(size_t) b * nchan + i] = p;
            continue;
        }
        p.c = d.f_carr * delt;
        p.flags = 0;
        if (st->prn < 0) {                                        // a summary: whether this block seeds itself is the fold's to say
            st->first_prn = d.prn; st->first_phase = d.carr_phase;
            st->R = 0; st->D = 0.0;
        } else if (st->prn != d.prn) {                            // the slot got this satellite here: gps.c:2208-2214
            st->R = lane::phase_units(d.carr_phase); st->D = 0.0;
            p.est = d.carr_phase; p.flags = lane::kSeed;
            reseeded = true;
        } else if (st->exact) {                                   // block 0 of a timeline that continues a walked one
            st->R = lane::phase_units(st->carr); st->D = 0.0;
            p.est = st->carr; p.flags = lane::kSeed;
        } else {
            p.est = lane::wrap01(lane::units_to_double(st->R) + st->D);
        }
        st->exact = false;
        if (with_drift && std::fabs(p.c) < 0.5) p.core = lane::drift_core(p.c, p.est < 1.0 ? p.est : 0.0, nsamp);
        if (prep) prep[(size_t) b * nchan + i] = p;
        st->R += lane::advance_units(p.c, nsamp);
        st->D += p.core;
        st->prn = d.prn; st->f_prev = d.f_carr;
    }
    return reseeded;
}

#ifdef GPSIQ_JOIN_CHECK
std::atomic<long> g_join_mismatch{0}, g_join_checked{0};
#endif
// level 1 on the host: the map of block b of slot i.  W / Wp: scratch walkers (Wp holds the previous block's addend when
// wp_c says so: a thread walks the blocks of a slot in order and sets every walker up once)
static void map_block(const Prep *prep, int nchan, int i, int b, double c_before, int nsamp, int max_seg,
                      lane::Walker *W, lane::Walker *Wp, double *w_c, double *wp_c, Rec *rec)
{
    const Prep &p = prep[(size_t) b * nchan + i];
    Rec r;
    std::memset(&r, 0, sizeof r);
    *rec = r;
    if ((p.flags & lane::kSkip) || !(std::fabs(p.c) < 0.5) || p.c == 0.0) return;
    if (!(*w_c == p.c)) { W->setup(p.c); *w_c = p.c; }
    const lane::Walker *prev = nullptr;
    if (!(p.flags & lane::kSeed)) {
        const double cp = b > 0 ? prep[(size_t) (b - 1) * nchan + i].c : c_before;
        if (cp == 0.0 || !(std::fabs(cp) < 0.5)) return;
        if (!(*wp_c == cp)) { Wp->setup(cp); *wp_c = cp; }
        prev = Wp;
    }
    if (W->general) return;
    const int nseg = lane::stretches(p.c, nsamp, max_seg);
    lane::Stretch st[lane::kMaxSeg];
    // the block's table of whole cycles, when the block has cycles enough to pay for it (the device builds one entry per lane)
    lane::Cycle tab[lane::kEntriesHost];
    const bool use_tab = std::fabs(p.c) * (double) nsamp > 2.0 * lane::kEntriesHost;
    if (use_tab)
        for (int k = 0; k < lane::kEntriesHost; ++k) lane::build_cycle(*W, k, lane::kEntriesHost, &tab[k]);
    for (int t = 0; t < nseg; ++t) lane::walk_stretch(*W, prev, p, nsamp, t, nseg, use_tab ? tab : nullptr, lane::kEntriesHost, &st[t]);
    lane::join_stretches(st, nseg, W->neg, rec);
#ifdef GPSIQ_JOIN_CHECK            // tests/chain_parallel.cpp: the join as a scan (what the device runs) against the loop, block for block
    Rec scanned;
    std::memset(&scanned, 0, sizeof scanned);
    lane::join_stretches_scan(st, nseg, W->neg, &scanned);
    if (std::memcmp(&scanned, rec, sizeof scanned)) g_join_mismatch.fetch_add(1);
    g_join_checked.fetch_add(1);
#endif
}

struct MapsJob {
    const Prep *prep; const double *c_before; int nblocks, nchan, nsamp, max_seg; Rec *rec;
};

void chain_maps_host(const Prep *prep, const double *c_before, int nblocks, int nchan, int nsamp, int max_seg, Rec *rec)
{
    MapsJob job = {prep, c_before, nblocks, nchan, nsamp, max_seg, rec};
    // slot-major: consecutive items are consecutive blocks of one slot, the next block's "previous" walker is this block's
    parallel_for(nblocks * nchan, 0, 8, [](void *ctx, int k0, int k1) {
        const MapsJob &j = *static_cast<MapsJob *>(ctx);
        lane::Walker wa, wb;
        lane::Walker *W = &wa, *Wp = &wb;
        double w_c = 0.0, wp_c = 0.0;
        for (int k = k0; k < k1; ++k) {
            const int i = k / j.nblocks, b = k % j.nblocks;
            map_block(j.prep, j.nchan, i, b, j.c_before[i], j.nsamp, j.max_seg, W, Wp, &w_c, &wp_c, &j.rec[(size_t) b * j.nchan + i]);
            std::swap(W, Wp); std::swap(w_c, wp_c);              // this block's walker is the next block's "previous"
        }
    }, &job);
}

double chain_block_true(double f_carr, double delt, int nsamp, double start);      // gpsiq_exact.cpp: the block walked (NcoWalk)

struct LinkJob {
    const gpsiq_chain_in_t *in; const Rec *rec; int nblocks, nchan, nsamp; double delt;
    const double *carr_in; const int32_t *prn_in;
    double *carr_start, *carr_end; int32_t *last_prn;
    std::atomic<long> walked{0}, linked{0};
    std::atomic<int> rc{GPSIQ_OK}; std::atomic<int> bad_block{-1};
};

static void link_slot(LinkJob &j, int i)
{
    double x = 0.0;
    int pv = 0;
    const bool have_in = j.carr_in && j.prn_in;
    if (have_in) { x = j.carr_in[i]; pv = j.prn_in[i]; }
    long walked = 0, linked = 0;
    for (int b = 0; b < j.nblocks; ++b) {
        const size_t at = (size_t) b * j.nchan + i;
        if (b + 8 < j.nblocks) {                     // a slot's rows lie nchan rows apart: every block a cache miss unless asked for early
            __builtin_prefetch(&j.in[at + (size_t) 8 * j.nchan]);
            __builtin_prefetch(&j.rec[at + (size_t) 8 * j.nchan]);
        }
        const gpsiq_chain_in_t &d = j.in[at];
        if (d.prn <= 0) { pv = 0; x = 0.0; j.carr_start[at] = 0.0; continue; }
        if ((b == 0 && !have_in) || pv != d.prn) x = d.carr_phase;
        pv = d.prn;
        j.carr_start[at] = x;
        if (j.rc.load(std::memory_order_relaxed) != GPSIQ_OK) continue;
        const double inc = d.f_carr * j.delt;
        if (!(x >= 0.0 && x <= 1.0) || !(std::fabs(inc) < 0.5)) {       // what gpsiq_reference_chain refuses
            int ok = GPSIQ_OK;
            if (j.rc.compare_exchange_strong(ok, GPSIQ_E_RANGE)) j.bad_block.store(b);
            continue;
        }
        double y;
        if (lane::link_block(j.rec[at], x, &y)) { x = y; ++linked; }
        else { x = chain_block_true(d.f_carr, j.delt, j.nsamp, x); ++walked; }
    }
    if (j.carr_end) j.carr_end[i] = x;
    if (j.last_prn) j.last_prn[i] = pv;
    j.walked.fetch_add(walked, std::memory_order_relaxed);
    j.linked.fetch_add(linked, std::memory_order_relaxed);
}

static std::atomic<uint64_t> g_chain_stats[2];          // blocks linked through their map / walked from their true start

bool chain_step_mapped(const void *maps, size_t at, double x, double *next)
{
#ifdef GPSIQ_TEST_HOOKS          // fault injection: GPSIQ_TEST_CORRUPT_MAP_AT=<flat index> gives that block's map another end offset
    if (const char *e = std::getenv("GPSIQ_TEST_CORRUPT_MAP_AT"))
        if ((size_t) std::atol(e) == at) { Rec r = static_cast<const Rec *>(maps)[at]; r.cum[0] += 4; r.cum[1] += 4; return lane::link_block(r, x, next); }
#endif
    return lane::link_block(static_cast<const Rec *>(maps)[at], x, next);
}
void chain_prefetch_map(const void *maps, size_t at) { __builtin_prefetch(&static_cast<const Rec *>(maps)[at]); }
void chain_count(long linked, long walked)
{
    g_chain_stats[0].fetch_add((uint64_t) linked, std::memory_order_relaxed);
    g_chain_stats[1].fetch_add((uint64_t) walked, std::memory_order_relaxed);
}

int chain_link(const gpsiq_chain_in_t *in, const void *maps, int nblocks, int nchan, double delt, int nsamp,
               const double *carr_in, const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn)
{
    LinkJob job;
    job.in = in; job.rec = static_cast<const Rec *>(maps); job.nblocks = nblocks; job.nchan = nchan; job.nsamp = nsamp; job.delt = delt;
    job.carr_in = carr_in; job.prn_in = prn_in; job.carr_start = carr_start; job.carr_end = carr_end; job.last_prn = last_prn;
    // a slot is a microsecond per hundred blocks: short timelines are not worth waking the pool for
    parallel_for(nchan, nblocks >= 512 ? 0 : 1, 1, [](void *ctx, int i0, int i1) {
        for (int i = i0; i < i1; ++i) link_slot(*static_cast<LinkJob *>(ctx), i);
    }, &job);
    g_chain_stats[0].fetch_add((uint64_t) job.linked.load(), std::memory_order_relaxed);
    g_chain_stats[1].fetch_add((uint64_t) job.walked.load(), std::memory_order_relaxed);
    if (job.rc.load() != GPSIQ_OK) return fail(job.rc.load(), "block %d: carrier phase or Doppler outside the NCO format", job.bad_block.load());
    return GPSIQ_OK;
}

static int check_chain_args(const void *in, int nblocks, int nchan, double fs, int nsamp)
{
    if (!in && nblocks) return fail(GPSIQ_E_ARG, "null pointer");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0)) return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g", nsamp, fs);
    if ((long) nblocks * nchan > 0x7fffffffL) return fail(GPSIQ_E_ARG, "timeline too long");
    return GPSIQ_OK;
}

// prepare every slot (threads over slots), from `start` (null: the timeline begins here)
void chain_prepare_host(const gpsiq_chain_in_t *in, int nblocks, int nchan, double delt, int nsamp, const gpsiq_chain_est_t *start,
                        Prep *prep, double *c_before, gpsiq_chain_est_t *end)
{
    struct Job { const gpsiq_chain_in_t *in; int nblocks, nchan, nsamp; double delt; const gpsiq_chain_est_t *start; Prep *prep; double *c_before; gpsiq_chain_est_t *end; };
    Job job = {in, nblocks, nchan, nsamp, delt, start, prep, c_before, end};
    parallel_for(nchan, nblocks >= 256 ? 0 : 1, 1, [](void *ctx, int i0, int i1) {
        const Job &j = *static_cast<Job *>(ctx);
        for (int i = i0; i < i1; ++i) {
            EstState st;
            if (j.start) st.load(j.start[i]);
            const bool rs = chain_scan_slot(j.in, j.nblocks, j.nchan, i, j.delt, j.nsamp, &st, true, j.prep, &j.c_before[i]);
            if (j.end) st.store(&j.end[i], rs ? GPSIQ_CHAIN_RESEEDED : 0);
        }
    }, &job);
}

}  // namespace gpsiq

using namespace gpsiq;

extern "C" int gpsiq_chain_summary(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                                   const gpsiq_chain_est_t *start, gpsiq_chain_est_t *sum)
{
    int rc = check_chain_args(in, nblocks, nchan, fs, nsamp);
    if (rc) return rc;
    if (!sum) return fail(GPSIQ_E_ARG, "null pointer");
    struct Job { const gpsiq_chain_in_t *in; int nblocks, nchan, nsamp; double delt; const gpsiq_chain_est_t *start; gpsiq_chain_est_t *sum; };
    Job job = {in, nblocks, nchan, nsamp, 1.0 / fs, start, sum};
    // a slot per thread (the drift pass is ~0.1 us per block and slot)
    parallel_for(nchan, nblocks >= 256 ? 0 : 1, 1, [](void *ctx, int i0, int i1) {
        const Job &j = *static_cast<Job *>(ctx);
        for (int i = i0; i < i1; ++i) {
            EstState st;
            u128 r0 = 0;
            if (j.start) { st.load(j.start[i]); r0 = st.R; st.D = 0.0; }
            else st.prn = -1;                                    // the phase pass: what came before is not known yet
            const bool rs = chain_scan_slot(j.in, j.nblocks, j.nchan, i, j.delt, j.nsamp, &st, j.start != nullptr, nullptr, nullptr);
            if (!rs && j.start) st.R -= r0;                      // what the range adds; absolute from its last re-seed otherwise
            if (st.prn < 0) st.prn = 0;
            st.store(&j.sum[i], (rs ? GPSIQ_CHAIN_RESEEDED : 0) | (j.nblocks == 0 ? GPSIQ_CHAIN_EMPTY : 0));
        }
    }, &job);
    return GPSIQ_OK;
}

extern "C" int gpsiq_chain_fold(const gpsiq_chain_est_t *sums, int nranges, int nchan, gpsiq_chain_est_t *out)
{
    if ((!sums && nranges) || !out || nranges < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad argument");
    for (int i = 0; i < nchan; ++i) {
        EstState acc;                                            // nothing before the first range: its first block seeds itself
        for (int r = 0; r < nranges; ++r) {
            const gpsiq_chain_est_t &s = sums[(size_t) r * nchan + i];
            if (s.flags & GPSIQ_CHAIN_EMPTY) continue;           // a rank without blocks is transparent
            EstState v;
            v.load(s);
            if (s.flags & GPSIQ_CHAIN_RESEEDED) acc = v;         // absolute from its last re-seed on
            else if (s.first_prn > 0 && acc.prn != s.first_prn) {           // its first block seeds itself after all
                acc = v;
                acc.R = lane::phase_units(s.carr) + v.R;
            } else { const u128 R = acc.R + v.R; const double D = acc.D + v.D; acc = v; acc.R = R; acc.D = D; }
            acc.exact = false;
        }
        acc.store(&out[i], 0);
    }
    return GPSIQ_OK;
}

extern "C" int gpsiq_chain_maps(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                                const gpsiq_chain_est_t *start, int max_stretches, gpsiq_chain_map_t *maps, gpsiq_chain_est_t *end)
{
    int rc = check_chain_args(in, nblocks, nchan, fs, nsamp);
    if (rc) return rc;
    if (!maps && nblocks) return fail(GPSIQ_E_ARG, "null pointer");
    if (max_stretches <= 0) max_stretches = 1;                  // the host walks a block in one piece unless told otherwise
    if (max_stretches > lane::kMaxSeg) max_stretches = lane::kMaxSeg;
    std::vector<Prep> prep((size_t) nblocks * nchan + 1);
    double c_before[GPSIQ_MAX_CHAN];
    chain_prepare_host(in, nblocks, nchan, 1.0 / fs, nsamp, start, prep.data(), c_before, end);
    chain_maps_host(prep.data(), c_before, nblocks, nchan, nsamp, max_stretches, reinterpret_cast<Rec *>(maps));
    return GPSIQ_OK;
}

extern "C" int gpsiq_chain_link(const gpsiq_chain_in_t *in, const gpsiq_chain_map_t *maps, int nblocks, int nchan, double fs, int nsamp,
                                const double *carr_in, const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn)
{
    int rc = check_chain_args(in, nblocks, nchan, fs, nsamp);
    if (rc) return rc;
    if ((!maps || !carr_start) && nblocks) return fail(GPSIQ_E_ARG, "null pointer");
    if (!carr_in != !prn_in) return fail(GPSIQ_E_ARG, "continuation state: both or neither");
    return chain_link(in, maps, nblocks, nchan, 1.0 / fs, nsamp, carr_in, prn_in, carr_start, carr_end, last_prn);
}

// ---- a range's maps composed: the relay of a time-sharded chain in one exchange (gpsiq_plumbing.h) ------------------------------
extern "C" int gpsiq_chain_range(const gpsiq_chain_in_t *in, const gpsiq_chain_map_t *maps, int nblocks, int nchan, double fs, int nsamp,
                                 gpsiq_chain_range_t *out)
{
    int rc = check_chain_args(in, nblocks, nchan, fs, nsamp);
    if (rc) return rc;
    if ((!maps && nblocks) || !out) return fail(GPSIQ_E_ARG, "null pointer");
    const Rec *rec = reinterpret_cast<const Rec *>(maps);
    // the state after the range when its first block seeds itself: the ordinary link (also what a restart inside leaves behind)
    std::vector<double> start((size_t) nblocks * nchan + 1), end_abs((size_t) nchan);
    std::vector<int32_t> last_prn((size_t) nchan);
    if (nblocks) {
        rc = chain_link(in, maps, nblocks, nchan, 1.0 / fs, nsamp, nullptr, nullptr, start.data(), end_abs.data(), last_prn.data());
        if (rc) return rc;
    }
    for (int i = 0; i < nchan; ++i) {
        gpsiq_chain_range_t r;
        std::memset(&r, 0, sizeof r);
        r.nblocks = nblocks;
        if (nblocks == 0) { out[i] = r; continue; }
        r.abs_end = end_abs[i]; r.last_prn = last_prn[i];
        const gpsiq_chain_in_t &d0 = in[i];
        r.first_prn = d0.prn > 0 ? d0.prn : 0;
        // the blocks that continue the state the range is entered with: all of them, unless the slot restarts inside
        int n = 0;
        for (int b = 0; b < nblocks; ++b) {
            const int prn = in[(size_t) b * nchan + i].prn;
            if (prn <= 0 || (b > 0 && prn != in[(size_t) (b - 1) * nchan + i].prn)) break;
            ++n;
        }
        r.restart = n < nblocks;
        if (r.first_prn == 0 || r.restart) { out[i] = r; continue; }          // (nothing of the end depends on the entry state)
        // compose the maps of blocks 0 .. n-1; per residue of the entry offset, the offsets at every block and what they must satisfy
        r.xs = rec[i].xs;
        r.e = rec[(size_t) (n - 1) * nchan + i].e;
        int64_t pre[4] = {0, 0, 0, 0};                   // offset at block b, less the entry offset, per residue r
        int ok = 15;
        for (int q = 0; q < 4; ++q) { r.lo[q] = INT64_MIN / 4; r.hi[q] = INT64_MAX / 4; }
        for (int b = 0; b < n && ok; ++b) {
            const Rec &m = rec[(size_t) b * nchan + i];
            const int64_t grid = m.info & 0xff;
            if (!m.ok || grid < 1 || grid > 2 || !(std::fabs(in[(size_t) b * nchan + i].f_carr / fs) < 0.5)) { ok = 0; break; }
            int64_t J = 0;
            if (b > 0 && !lane::exact_units(rec[(size_t) (b - 1) * nchan + i].e, m.xs, &J)) { ok = 0; break; }
            for (int q = 0; q < 4; ++q) {
                if (!(ok & (1 << q))) continue;
                if (b > 0) {
                    const Rec &pm = rec[(size_t) (b - 1) * nchan + i];
                    const int64_t pg = pm.info & 0xff;
                    const int64_t dprev = (int64_t) q + pre[q];                              // mod 4 is all that matters for the parity
                    pre[q] += pm.cum[(dprev >> (pg - 1)) & 1] + J;
                }
                const int64_t dres = (int64_t) q + pre[q];
                if ((dres & (grid - 1)) || !(m.ok & (1 << ((dres >> (grid - 1)) & 1)))) { ok &= ~(1 << q); continue; }
                int64_t l = m.lo - pre[q], h = m.hi - pre[q];
                // ... and the state after the block, e + (d + cum)*U, has to be a double in [0, 1) (link_block's exact_shift): below
                // 1.0, and below the first binade whose ulp would drop e's lowest bit
                const int64_t cum = m.cum[(dres >> (grid - 1)) & 1];
                const uint64_t be = bits_of(m.e);
                double ymax = 1.0;
                if (m.e > 0.0) {
                    const int tz = __builtin_ctzll((be & kMant) | (kMant + 1));               // trailing zeros of the 53-bit mantissa
                    const double lowbit = from_bits((uint64_t) ((int64_t) (be >> 52) - 52 + tz) << 52);
                    if (lowbit * 0x1p53 < ymax) ymax = lowbit * 0x1p53;
                } else if (m.e < 0.0) { ok &= ~(1 << q); continue; }
                const double E = m.e * 0x1p53, Y = ymax * 0x1p53;
                const int64_t kmin = (int64_t) std::ceil(-E), kmax = (int64_t) std::ceil(Y - E) - 1;      // kmin <= d + cum <= kmax
                if (kmin - cum - pre[q] > l) l = kmin - cum - pre[q];
                if (kmax - cum - pre[q] < h) h = kmax - cum - pre[q];
                if (l > r.lo[q]) r.lo[q] = l;
                if (h < r.hi[q]) r.hi[q] = h;
            }
        }
        for (int q = 0; q < 4; ++q) { r.t[q] = pre[q]; if (r.lo[q] > r.hi[q]) ok &= ~(1 << q); }
        const Rec &ml = rec[(size_t) (n - 1) * nchan + i];
        r.cum_last[0] = ml.cum[0]; r.cum_last[1] = ml.cum[1]; r.grid_last = (int32_t) (ml.info & 0xff);
        r.ok = ok;
        out[i] = r;
    }
    return GPSIQ_OK;
}

extern "C" int gpsiq_chain_range_fold(const gpsiq_chain_range_t *ranges, int nranges, int nchan, const double *true_end, const int32_t *true_prn,
                                      const uint8_t *true_known, double *carr, int32_t *prn, uint8_t *known)
{
    if ((!ranges && nranges) || !carr || !prn || !known || nranges < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad argument");
    int all = 1;
    for (int i = 0; i < nchan; ++i) {
        double x = 0.0;
        int pv = 0;
        bool kn = true;
        carr[i] = 0.0; prn[i] = 0; known[i] = 1;
        for (int r = 0; r < nranges; ++r) {
            const gpsiq_chain_range_t &g = ranges[(size_t) r * nchan + i];
            const size_t at = (size_t) r * nchan + i;
            if (g.nblocks > 0) {
                if (true_known && true_known[r]) { x = true_end[at]; pv = true_prn[at]; kn = true; }        // the rank has linked its range: the state itself
                else {
                    // (which satellite the slot holds never depends on the accumulator: pv is always known)
                    const bool continues = g.first_prn > 0 && pv == g.first_prn;
                    if (!continues || g.restart) { x = g.abs_end; kn = true; }      // its first block seeds the slot, or the slot restarts inside: whatever came before
                    else if (kn) {
                        // the range continues the state it is entered with, all the way: through its composed maps
                        int64_t d = 0;
                        bool ok = lane::exact_units(x, g.xs, &d);
                        const int q = (int) (d & 3);
                        ok = ok && (g.ok & (1 << q)) && d >= g.lo[q] && d <= g.hi[q] && g.grid_last >= 1 && g.grid_last <= 2;
                        double y = 0.0;
                        if (ok) {
                            const int64_t dl = d + g.t[q];
                            ok = lane::exact_shift(g.e, dl + g.cum_last[(dl >> (g.grid_last - 1)) & 1], &y) && y >= 0.0 && y < 1.0;
                        }
                        kn = ok;
                        x = y;
                    }                                                                // (else: unknown in, unknown out)
                    pv = g.last_prn;
                }
                if (pv <= 0) { pv = 0; x = 0.0; }
            }
            carr[(size_t) (r + 1) * nchan + i] = kn ? x : 0.0; prn[(size_t) (r + 1) * nchan + i] = kn ? pv : 0;
            known[(size_t) (r + 1) * nchan + i] = kn ? 1 : 0;
            if (!kn) all = 0;
        }
    }
    return all;
}

extern "C" void gpsiq_chain_stats(uint64_t out[2])
{
    if (!out) return;
    out[0] = g_chain_stats[0].load(std::memory_order_relaxed);
    out[1] = g_chain_stats[1].load(std::memory_order_relaxed);
}
