// gpsiq_lane.h -- the reference's carrier chain (gps.c:2821-2826) made parallel in time.  Shared by the host
// (gpsiq_chain.cpp) and the device (gpsiq_chain_kernels.hip): same source, same IEEE double additions.
//
// The chain x_{b+1} = F_b(x_b) (F_b: the block's nsamp double additions of c_b = f_carr*delt with the wrap rule) is serial
// because every rounding depends on the low bits the one before left.  But F_b is a TRANSLATION on a whole residue class of
// start states (gpsiq_walk.h / NcoWalk: a walk started d*U higher makes the same roundings as long as every result stays in
// the binade it had; U = 2^-53, the coarsest ulp below 1).  So, per block and independently of every other block:
//   1. an ESTIMATE of the accumulator at the block's start (exact real arithmetic R_b = x_0 + sum nsamp*c_j mod 1 in 128-bit
//      integers + the modelled rounding drift, drift_core) places the last wrap before the block; right after a wrap the
//      state is a multiple of U, so a REPRESENTATIVE post-wrap state there (the estimate rounded to the grid) differs from
//      the true one by d*U with an integer d nobody knows yet;
//   2. the block is walked from the representative (the previous block's tail first: that gives the residue class of the
//      state at the block's first sample), noting how far the start may move (Slack): a certified map
//          start xs + d*U, lo <= d <= hi   ->   end e + (d + cum)*U.
//      A block is cut into up to kMaxSeg stretches walked by separate lanes, each from its own representative wrap; the
//      stretches are joined at the wraps they share (same sample index, difference of the two representatives = an integer).
//   3. what is left of the chain is one exact subtraction, one range check and one exact addition per block (link_block):
//      d_b = (x_b - xs_b)/U, x_{b+1} = e_b + (d_b + cum_b)*U.  A block whose map does not apply (a start outside [lo, hi] or
//      in another residue class: the estimate put the wrap one sample off; a slow or irregular addend; a phase of exactly
//      1.0) is walked from its true start state as before (chain_block) -- rare by construction: the estimate is good to
//      ~1e-13 cycle and the ranges are ~1e-4 cycle wide.
// Correctness rests on the walk (exact), the slack (sufficient conditions, gpsiq_walk.h) and the two exactness checks in
// link_block; a bad estimate can only cost a fallback, never a wrong state.
#ifndef GPSIQ_LANE_H
#define GPSIQ_LANE_H

#include "gpsiq_walk.h"

namespace gpsiq {
namespace lane {

typedef unsigned __int128 u128;

constexpr int kTab = 22;                 // table binades of a lane's walker: |c| >= 2^-22 cycle per sample (slower blocks hold no wrap to start from)
constexpr int kMaxSeg = 32;              // stretches per block at most
typedef FpWalk<kTab> Walker;                 // (WalkCore<kTab> is the same walk on integer mantissas: twice the device time)
constexpr double kU = 0x1p-53;           // the unit of every offset here

// what a block needs besides its own addend: 32 bytes per channel and block, written by prepare (scan over the timeline)
struct Prep {
    double  c;           // addend f_carr*delt of this block (0: unused slot, or a block the lanes leave to the true walk)
    double  est;         // estimate of the accumulator at the block's first sample, [0, 1)
    double  core;        // modelled drift over this block (interpolated inside it)
    int32_t flags;       // kSeed, kSkip
    int32_t reserved;
};
enum : int32_t {
    kSeed = 1,           // the block's start state is known exactly (first block of a slot's satellite): est IS the state
    kSkip = 2,           // no map for this block (unused slot, an addend the fast walk does not take): true walk, or nothing
};

// the certified map of one block (level 1 result), 56 bytes
struct Rec {
    double  xs;          // representative state at the block's first sample
    double  e;           // state after the block when the last stretch starts from its own representative
    int64_t cum[2];      // the true end is e + (d + cum[p])*U for a true start xs + d*U, p = parity of d counted in steps of the wrap's
                         // grid (1 unit for a negative addend, 2 for a positive one): an exact tie on a wrap sends odd offsets one step aside
    int64_t lo, hi;      // ... if lo <= d <= hi (units of U = 2^-53)
    int32_t ok;          // 0: no map (level 2 walks the block from its true start); bit 0: the map holds for even p, bit 1: for odd p
    int32_t info;        // bits 0-7: units per grid step (d must be a multiple), bits 8-15: why not ok (the statistics' business)
};

// ---- exact real arithmetic of the phase: 2^-128 cycle units, the wrap is the overflow -----------------------------------
GPSIQ_HD inline u128 phase_units(double x)             // x in [0, 1]; bits below 2^-128 are cut (an estimate does not care)
{
    const uint64_t b = bits_of(x);
    const int e = (int) (b >> 52) & 0x7ff;
    if (e == 0) return 0;
    const u128 m = (u128) ((b & kMant) | (kMant + 1));
    const int sh = e - 947;                              // m * 2^(e - 1075) / 2^-128
    return sh >= 0 ? (sh < 128 ? m << sh : (u128) 0) : (sh > -64 ? m >> -sh : (u128) 0);
}
GPSIQ_HD inline u128 advance_units(double c, long nsamp)   // nsamp * c mod 1, exact for |c| >= 2^-75
{
    const uint64_t b = bits_of(c);
    const int e = (int) (b >> 52) & 0x7ff;
    if (e == 0 || nsamp <= 0) return 0;
    const u128 p = (u128) ((b & kMant) | (kMant + 1)) * (u128) (uint64_t) nsamp;
    const int sh = e - 947;
    const u128 v = sh >= 0 ? (sh < 128 ? p << sh : (u128) 0) : (sh > -64 ? p >> -sh : (u128) 0);
    return (b >> 63) ? (u128) 0 - v : v;
}
GPSIQ_HD inline double units_to_double(u128 r) { return (double) (uint64_t) (r >> 64) * 0x1p-64; }

GPSIQ_HD inline double wrap01(double x)                 // into [0, 1)
{
    x -= __builtin_floor(x);
    return x < 1.0 ? x : 0.0;
}

// ---- the modelled rounding drift of one block (Drift in gpsiq_exact.cpp, its centre only) --------------------------------
// While state and sum share binade b the addition adds S_b = rnd(|c|/u_b)*u_b instead of |c|: with G piecewise linear of
// slope (S_b - |c|)/S_b in binade b, continued over the wraps, the accumulator after n additions is x0 + n*c + G^(end) -
// G(x0) up to a dozen irregular roundings per cycle.  x0: where the block starts (an estimate does).
GPSIQ_HD inline double drift_core(double c, double x0, long nsamp)
{
    const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
    const int64_t ec = (int64_t) (bc >> 52), mc = (int64_t) ((bc & kMant) | (kMant + 1));
    if (ec > 1022 - 6 || ec < 1022 - 40) return 0.0;
    const double end = x0 + (double) nsamp * c, q = __builtin_floor(end), r = end - q;
    double GL = 0.0, Gr = 0.0, G0 = 0.0;
    for (int b = (int) ec + 3; b <= 1022; ++b) {
        const int s = b - (int) ec;
        int64_t dm = mc >> s;
        const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
        if (rem > half) ++dm;
        else if (rem == half) dm += dm & 1;
        const double delta = (double) ((dm << s) - mc), Sb = (double) dm * (double) ((int64_t) 1 << s);
        const double gb = delta / Sb, lo = from_bits((uint64_t) b << 52);        // binade [lo, 2 lo)
        GL += gb * lo;
        const double ar = r - lo, a0 = x0 - lo;
        Gr += gb * (ar < 0.0 ? 0.0 : ar > lo ? lo : ar);
        G0 += gb * (a0 < 0.0 ? 0.0 : a0 > lo ? lo : a0);
    }
    return q * GL + Gr - G0;
}

// ---- placing the last wrap at or before a sample -----------------------------------------------------------------------
// s: estimate of the accumulator at some sample, c: the addend in force before it.  The accumulator was last wrapped k
// additions earlier; *r = the estimate of that post-wrap state on its grid (a multiple of 2^-52 in [0, c] for a positive
// addend, of 2^-53 in [1 + c, 1) for a negative one).  false: no usable answer.
GPSIQ_HD inline bool last_wrap(double s, double c, long *k, double *r)
{
    if (!(s >= 0.0 && s < 1.0)) return false;
    if (c > 0.0) {
        double kk = __builtin_floor(s / c);
        double w = __builtin_fma(-kk, c, s);
        if (w < 0.0) { kk -= 1.0; w += c; }
        else if (w >= c) { kk += 1.0; w -= c; }
        if (!(kk >= 0.0 && kk < 0x1p40)) return false;
        double g = __builtin_rint(w * 0x1p52) * 0x1p-52;
        if (g < 0.0) g = 0.0;
        *k = (long) kk; *r = g;
        return true;
    }
    if (c < 0.0) {
        const double a = -c;
        double kk = __builtin_floor((1.0 - s) / a);
        double w = __builtin_fma(kk, a, s);
        if (w >= 1.0) { kk -= 1.0; w -= a; }
        else if (w < 1.0 - a) { kk += 1.0; w += a; }
        if (!(kk >= 0.0 && kk < 0x1p40)) return false;
        double g = __builtin_rint(w * 0x1p53) * 0x1p-53;
        if (g >= 1.0) g = 1.0 - 0x1p-53;
        *k = (long) kk; *r = g;
        return true;
    }
    return false;
}

// a - b as a multiple of U = 2^-53, when it is exactly that
GPSIQ_HD inline bool exact_units(double a, double b, int64_t *d)
{
    const double s = a - b;
    const double bb = s - a, err = (a - (s - bb)) + (-b - bb);      // TwoSum of a + (-b): err == 0 <=> s is the real difference
    if (err != 0.0) return false;
    const double u = s * 0x1p53;                                     // exact (a power of two; |s| <= 1)
    if (!(__builtin_fabs(u) < 0x1p62)) return false;
    const int64_t i = (int64_t) u;
    if ((double) i != u) return false;
    *d = i;
    return true;
}

// e + k*U when that is exact
GPSIQ_HD inline bool exact_shift(double e, int64_t k, double *out)
{
    if (k > ((int64_t) 1 << 53) || k < -((int64_t) 1 << 53)) return false;
    const double t = (double) k * kU, s = e + t;
    const double bb = s - e, err = (e - (s - bb)) + (t - bb);
    if (err != 0.0) return false;
    *out = s;
    return true;
}

// ---- one stretch of one block ----------------------------------------------------------------------------------------
struct Stretch {
    double  r_in;        // representative state the stretch starts from: at the wrap n_in (stretch 0: at the block's first sample)
    double  x_out;       // the state at the last wrap at or before the stretch's end sample (n_out; n_in itself if there was none)
    double  x_end;       // state at the stretch's end sample
    int64_t lo, hi;      // units of U = 2^-53, relative to the stretch's own start
    int32_t n_in, n_out; // sample indices inside the block
    int16_t ok, why;     // ok == 0: what stood in the way (kWhy*), for the statistics; ok == 2: for even offsets of the wrap's grid only
    int16_t sigma, pad;  // the stretch's first exact tie on a wrap: odd offsets go on sigma grid steps aside (0: none)
};
enum { kWhyNone = 0, kWhyAddend = 1, kWhyPrev = 2, kWhyWrap = 3, kWhyState = 4, kWhyAnchor = 5, kWhySlack = 6, kWhyEdge = 7, kWhyJoin = 8, kWhyUnits = 9, kWhyRange = 10, kWhyParity = 11 };

// how many stretches a block gets: at least ~4 wraps in each
GPSIQ_HD inline int stretches(double c, long nsamp, int max_seg)
{
    const double cyc = __builtin_fabs(c) * (double) nsamp;
    int s = (int) (cyc * 0.25);
    if (s > max_seg) s = max_seg;
    if (s < 1) s = 1;
    return s;
}

// ---- the block's table of whole cycles -----------------------------------------------------------------------------------
// From one wrap to the next the accumulator runs through the same dozen binades whatever the post-wrap state was, and a cycle
// walked from one post-wrap state holds, translated, for a whole range of them (the walk's own range): a block's ~260 cycles are
// a dozen distinct ones.  So the lanes of a block first walk ONE cycle each, from post-wrap states spread evenly over their
// range [0, c] or [1 + c, 1) (kEntries of them: an entry {range of start states, state increment, samples}), and a stretch then
// LOOKS its cycles UP: a comparison and an exact addition instead of ~20 binade steps, with the distance of the state to the
// entry's edges noted like the distance of a result to its binade's edges.  A state no entry covers (a sliver between two
// sampled cycles, a cycle with an exact tie on its wrap) is walked as before.  This is NcoWalk's wrap-to-wrap table (gpsiq_exact.cpp)
// built from an even sample instead of on demand: the device has the lanes for it.
struct Cycle { double first, last, inc; int32_t steps, ok; };       // 32 bytes; first / last / inc are multiples of the wrap grid

constexpr int kEntriesHost = 32;

GPSIQ_HD inline void build_cycle(const Walker &W, int k, int nent, Cycle *out)
{
    Cycle e;
    e.first = 0.0; e.last = 0.0; e.inc = 0.0; e.steps = 0; e.ok = 0;
    *out = e;
    if (W.general) return;
    const double a = __builtin_fabs(W.c);
    const double grid = W.neg ? 0x1p-53 : 0x1p-52, inv = W.neg ? 0x1p53 : 0x1p52;
    const double x0 = __builtin_rint(((W.neg ? 1.0 - a : 0.0) + a * (((double) k + 0.5) / (double) nent)) * inv) * grid;
    if (!(x0 >= 0.0 && x0 < 1.0)) return;
    typename WalkCore<kTab>::FastSlack sl;
    sl.init(W.neg ? 1022 : 1023);
    if (x0 >= W.thr) sl.note(x0);
    double x = x0;
    long n = 0;
    const long cap = (long) (2.0 / a) + 8;                          // a cycle is 1/|c| samples give or take one
    const bool wrapped = W.neg ? W.template descend<true>(x, n, cap, &sl) : W.template climb<true>(x, n, cap, &sl);
    int64_t lo, hi;
    if (!wrapped || sl.sigma || W.top_tie || !sl.finish(&lo, &hi) || lo > 0 || hi < 0 || !(x >= 0.0 && x < 1.0)) return;
    // post-wrap states lie in [0, 1): what the range allows beyond that is never asked for
    double first = x0 + (lo < -((int64_t) 1 << 53) ? -1.0 : (double) lo * grid), last = x0 + (hi > ((int64_t) 1 << 53) ? 1.0 : (double) hi * grid);
    if (first < 0.0) first = 0.0;
    if (last > 1.0 - grid) last = 1.0 - grid;
    e.first = first; e.last = last; e.inc = x - x0; e.steps = (int32_t) n; e.ok = 1;
    *out = e;
}

// the entry that holds post-wrap state x, or null: the entry sampled nearest to x first, then its neighbours
// (scale = nent / |c|, base = the lower end of the post-wrap states: once per stretch, a division costs the device thirty instructions)
GPSIQ_HD inline const Cycle *find_cycle(const Cycle *tab, int nent, double base, double scale, double x)
{
    int k0 = (int) ((x - base) * scale);
    if (k0 < 0) k0 = 0;
    if (k0 >= nent) k0 = nent - 1;
    for (int d = 0; d < nent; ++d) {
        const int k = k0 + ((d & 1) ? (d + 1) / 2 : -(d / 2));      // k0, k0 + 1, k0 - 1, k0 + 2, ...
        if (k < 0 || k >= nent) continue;
        const Cycle &e = tab[k];
        if (e.ok && x >= e.first && x <= e.last) return &e;
        if (d >= 8) break;                                          // farther away than four samples either side: a sliver, walk it
    }
    return nullptr;
}

// the walk of stretch t of nseg of a block: W the walker of the block's addend, Wp of the previous block's (stretch 0 of a
// block that is no seed only); p: the block's Prep; tab[nent]: the block's table of cycles (null: every cycle walked)
GPSIQ_HD inline void walk_stretch(const Walker &W, const Walker *Wp, const Prep &p, long nsamp, int t, int nseg, const Cycle *tab, int nent, Stretch *out)
{
    Stretch o;
    o.n_in = -1; o.r_in = 0.0; o.n_out = -1; o.x_out = 0.0; o.x_end = 0.0; o.lo = 0; o.hi = 0; o.ok = 0; o.why = kWhyAddend; o.sigma = 0; o.pad = 0;
    *out = o;
    if ((p.flags & kSkip) || W.general || !(p.c != 0.0)) return;
    const long a0 = (long) (((int64_t) t * nsamp) / nseg), a1 = (long) (((int64_t) (t + 1) * nsamp) / nseg);
    typename WalkCore<kTab>::FastSlack sl;
    sl.init(W.neg ? 1022 : 1023);
    double x;
    long n;
    bool at_wrap;                                    // x is the state right after a wrap: the table may know its cycle
    if (t == 0) {
        if (p.flags & kSeed) x = p.est;
        else {
            // the tail of the block before, from its last wrap: the residue class of the state this block starts from
            out->why = kWhyPrev;
            if (!Wp || Wp->general || Wp->neg != W.neg) return;
            long k;
            out->why = kWhyWrap;
            if (!last_wrap(p.est, Wp->c, &k, &x) || k > nsamp) return;
            long m = 0;
            while (m < k && Wp->cycle(x, m, k)) {}
        }
        out->why = kWhyState;
        if (!(x >= 0.0 && x < 1.0)) return;
        o.n_in = 0; o.r_in = x;
        n = 0;
        at_wrap = false;
    } else {
        // the accumulator at sample a0: the block's start + a0 additions + the share of the block's drift
        const double fa = (double) a0, pr = fa * p.c, pe = __builtin_fma(fa, p.c, -pr);
        const double s = wrap01(p.est + (pr - __builtin_floor(pr)) + pe + p.core * (fa / (double) nsamp));
        long k;
        out->why = kWhyAnchor;
        if (!last_wrap(s, p.c, &k, &x) || k > a0) return;
        n = a0 - k;
        o.n_in = (int32_t) n; o.r_in = x;
        at_wrap = true;
    }
    o.n_out = o.n_in; o.x_out = x;
    if (x >= W.thr) sl.note(x);                      // a start inside a table binade has to stay in it (results are noted by the walk)
    const bool need_end = t == nseg - 1;             // only the block's last stretch is asked for the state at its end sample
    // Two loops, for the device's sake: inside, a lane looks up cycle after cycle for as long as the table knows them (a dozen
    // instructions each); outside, ONE cycle is walked (a thousand instructions) by the lanes that met a state no entry covers
    // -- 2 % of the look-ups, so with one loop every round of a 64-lane wave would have a lane walking and 63 waiting for it.
    const double tab_a = __builtin_fabs(W.c), tab_base = W.neg ? 1.0 - tab_a : 0.0, tab_scale = tab ? (double) nent / tab_a : 0.0;
    bool done = false;
    while (!done) {
        while (at_wrap && tab && n < a1) {
            const Cycle *e = find_cycle(tab, nent, tab_base, tab_scale, x);
            if (!e) break;
            sl.room_below(x - e->first);             // this state is one the entry holds for; how far it may move
            sl.room_above(e->last - x);
            if (n + e->steps > a1) {                 // the next wrap lies beyond the stretch
                done = !need_end;                    // nothing more to find out here, unless the state at the end sample is wanted
                break;
            }
            x += e->inc;                             // the whole cycle lies inside the stretch: the entry IS the cycle, translated
            n += e->steps;                           // (exact: both are multiples of the wrap grid, and so is the state after the wrap)
            o.n_out = (int32_t) n; o.x_out = x;
        }
        if (done || n >= a1) break;
        const bool wrapped = W.neg ? W.template descend<true>(x, n, a1, &sl) : W.template climb<true>(x, n, a1, &sl);
        if (!wrapped) break;
        o.n_out = (int32_t) n; o.x_out = x;
        at_wrap = true;
    }
    o.x_end = x;
    // the slack counts units of the wrap grid: 2^-52 for a positive addend
    const bool slack_ok = sl.finish(&o.lo, &o.hi);
    if (!W.neg) { o.lo *= 2; o.hi *= 2; }
    o.ok = slack_ok && o.lo <= 0 && o.hi >= 0 && x >= 0.0 && x < 1.0;
    if (o.ok && W.top_tie) o.ok = 2;                 // (every cycle's first addition is a tie on the unit's own grid: even offsets only)
    o.sigma = (int16_t) sl.sigma;
    o.why = o.ok ? kWhyNone : !slack_ok ? kWhySlack : kWhyEdge;
    *out = o;
}

// the stretches of a block joined into its map
GPSIQ_HD inline void join_stretches(const Stretch *st, int nseg, bool neg, Rec *rec)
{
    Rec r;
    const int64_t grid = neg ? 1 : 2;                 // units of U per step of the wrap's grid
    const int gsh = neg ? 0 : 1;                      // (shifts and masks below: a 64-bit division costs the device a hundred instructions)
    r.xs = st[0].r_in; r.e = st[nseg - 1].x_end; r.cum[0] = 0; r.cum[1] = 0; r.lo = st[0].lo; r.hi = st[0].hi; r.ok = st[0].ok ? 3 : 0;
    int why = st[0].why;
    // Branch p: the block's true start lies d = (2k + p) grid steps above xs.  cum[p]: how far the stretch at hand starts above
    // ITS representative in that branch, less d.  A stretch with a tie sends an odd offset sigma steps aside (FastSlack::tie),
    // a stretch that translates for even offsets only (ok == 2) closes the branch in which its offset is odd.
    for (int t = 0; t < nseg && r.ok; ++t) {
        if (t > 0) {
            int64_t d;
            if (!st[t].ok) { r.ok = 0; why = st[t].why; break; }
            if (st[t - 1].n_out != st[t].n_in) { r.ok = 0; why = kWhyJoin; break; }
            if (!exact_units(st[t - 1].x_out, st[t].r_in, &d) || (d & (grid - 1))) { r.ok = 0; why = kWhyUnits; break; }
            r.cum[0] += d; r.cum[1] += d;
            // the stretch's range holds for its own offset, before and after a step aside: a grid step short either side
            const int64_t cmin = r.cum[0] < r.cum[1] ? r.cum[0] : r.cum[1], cmax = r.cum[0] < r.cum[1] ? r.cum[1] : r.cum[0];
            const int64_t l = st[t].lo - cmin + grid, h = st[t].hi - cmax - grid;
            if (l > r.lo) r.lo = l;
            if (h < r.hi) r.hi = h;
        }
        for (int p = 0; p < 2; ++p) {
            const bool odd = ((p + (r.cum[p] >> gsh)) & 1) != 0;            // parity of this stretch's offset, in grid steps
            if (odd && st[t].ok == 2) r.ok &= ~(1 << p);
            if (odd && st[t].sigma) r.cum[p] += st[t].sigma > 0 ? grid : -grid;
        }
        if (!r.ok) why = kWhyParity;
    }
    if (r.ok && st[0].sigma) { r.lo += grid; r.hi -= grid; }
    if (r.ok && r.lo > r.hi) { r.ok = 0; why = kWhyRange; }
    if (!r.ok) { r.cum[0] = 0; r.cum[1] = 0; r.lo = 0; r.hi = 0; }      // (nobody reads them: the same bytes whoever joined)
    r.info = (int32_t) grid | (r.ok ? 0 : why << 8);
    *rec = r;
}

// ---- the join as a scan: what join_stretches does stretch after stretch, for one lane per stretch ---------------------------
// In steps of the wrap's grid a stretch moves the offset o of its start to that of the next stretch's: o -> o + D, and an odd
// offset goes sigma steps aside where the stretch met an exact tie -- a translation that depends on the parity of o only, so the
// stretches compose as pairs (the translation for an even / an odd offset before them) and every stretch learns both branches'
// offsets from an exclusive scan.  Ranges, closed branches and the first thing that stood in the way are reductions over the
// stretches.  The device runs scan and reductions with cross-lane moves (gpsiq_chain_kernels.hip); join_stretches_scan below is
// the same arithmetic in a loop, held against join_stretches block for block by tests/chain_parallel.cpp.
struct JoinMap { int64_t t0, t1; };
GPSIQ_HD inline JoinMap join_identity() { JoinMap m; m.t0 = 0; m.t1 = 0; return m; }
GPSIQ_HD inline JoinMap join_compose(const JoinMap &a, const JoinMap &b)      // first a, then b
{
    JoinMap o;
    o.t0 = a.t0 + ((a.t0 & 1) ? b.t1 : b.t0);
    o.t1 = a.t1 + (((1 + a.t1) & 1) ? b.t1 : b.t0);
    return o;
}
struct JoinLane { JoinMap el; int64_t D; int32_t fail, pad; };
// stretch t (cur) after stretch t - 1 (prev; unused for t == 0); grid: units of U per step of the wrap's grid
GPSIQ_HD inline JoinLane join_lane(const Stretch &prev, const Stretch &cur, int t, int64_t grid)
{
    JoinLane j;
    j.D = 0; j.fail = 0; j.pad = 0;
    if (!cur.ok) j.fail = cur.why ? (int32_t) cur.why : (int32_t) kWhyEdge;
    else if (t > 0) {
        int64_t d = 0;
        if (prev.n_out != cur.n_in) j.fail = kWhyJoin;
        else if (!exact_units(prev.x_out, cur.r_in, &d) || (d & (grid - 1))) j.fail = kWhyUnits;
        else j.D = grid == 2 ? d >> 1 : d;              // (d is a multiple of the grid)
    }
    const int64_t sg = cur.sigma > 0 ? 1 : cur.sigma < 0 ? -1 : 0;
    j.el.t0 = j.D + ((j.D & 1) ? sg : 0);
    j.el.t1 = j.D + (((1 + j.D) & 1) ? sg : 0);
    return j;
}
struct JoinTerm { int64_t l, h; int32_t mask, pad; };
// excl: the stretches before t composed (identity for t == 0)
GPSIQ_HD inline JoinTerm join_term(const JoinMap &excl, const JoinLane &j, const Stretch &cur, int t, int64_t grid)
{
    JoinTerm o;
    const int64_t m0 = excl.t0 + j.D, m1 = 1 + excl.t1 + j.D;                   // the stretch's offset in either branch, in grid steps
    o.mask = 3; o.pad = 0;
    if (cur.ok == 2) { if (m0 & 1) o.mask &= ~1; if (m1 & 1) o.mask &= ~2; }
    if (t == 0) { o.l = cur.lo; o.h = cur.hi; }
    else {
        const int64_t c0 = m0 * grid, c1 = (m1 - 1) * grid;                     // cum[0], cum[1] of join_stretches at this point
        const int64_t cmin = c0 < c1 ? c0 : c1, cmax = c0 < c1 ? c1 : c0;
        o.l = cur.lo - cmin + grid; o.h = cur.hi - cmax - grid;
    }
    return o;
}
// the reductions' results -> the map.  incl: all stretches composed; first_fail: the smallest (t << 8 | fail) of a stretch with
// fail != 0 (or INT32_MAX); first_closed: the first stretch at which both branches are closed (or INT32_MAX)
GPSIQ_HD inline void join_finish(const Stretch &st0, double x_end, bool neg, const JoinMap &incl, int64_t lo, int64_t hi, int mask,
                                 int32_t first_fail, int32_t first_closed, Rec *rec)
{
    Rec r;
    const int64_t grid = neg ? 1 : 2;
    r.xs = st0.r_in; r.e = x_end; r.cum[0] = incl.t0 * grid; r.cum[1] = incl.t1 * grid; r.lo = lo; r.hi = hi; r.ok = mask;
    int why = kWhyNone;
    const int32_t t_fail = first_fail == INT32_MAX ? INT32_MAX : first_fail >> 8;
    if (t_fail != INT32_MAX && t_fail <= first_closed) { r.ok = 0; why = first_fail & 0xff; }     // (a stretch is looked at before its ties are)
    else if (first_closed != INT32_MAX) { r.ok = 0; why = kWhyParity; }
    if (r.ok && st0.sigma) { r.lo += grid; r.hi -= grid; }
    if (r.ok && r.lo > r.hi) { r.ok = 0; why = kWhyRange; }
    if (!r.ok) { r.cum[0] = 0; r.cum[1] = 0; r.lo = 0; r.hi = 0; }
    r.info = (int32_t) grid | (r.ok ? 0 : why << 8);
    *rec = r;
}
GPSIQ_HD inline void join_stretches_scan(const Stretch *st, int nseg, bool neg, Rec *rec)
{
    const int64_t grid = neg ? 1 : 2;
    JoinMap pre = join_identity();
    int64_t lo = INT64_MIN, hi = INT64_MAX;
    int mask = 3;
    int32_t first_fail = INT32_MAX, first_closed = INT32_MAX;
    for (int t = 0; t < nseg; ++t) {
        const JoinLane j = join_lane(st[t > 0 ? t - 1 : 0], st[t], t, grid);
        const JoinTerm q = join_term(pre, j, st[t], t, grid);
        if (j.fail && first_fail == INT32_MAX) first_fail = t << 8 | j.fail;
        lo = q.l > lo ? q.l : lo; hi = q.h < hi ? q.h : hi;
        mask &= q.mask;
        if (!mask && first_closed == INT32_MAX) first_closed = t;
        pre = join_compose(pre, j.el);
    }
    join_finish(st[0], st[nseg - 1].x_end, neg, pre, lo, hi, mask, first_fail, first_closed, rec);
}

// level 2, one block: the accumulator after the block from its true start state x through the block's map; false: the map
// does not apply (walk the block)
GPSIQ_HD inline bool link_block(const Rec &r, double x, double *next)
{
    int64_t d;
    if (!r.ok || !exact_units(x, r.xs, &d)) return false;
    const int64_t grid = r.info & 0xff;                          // 1 or 2
    if (d < r.lo || d > r.hi || grid < 1 || grid > 2 || (d & (grid - 1))) return false;
    const int p = (int) ((d >> (grid - 1)) & 1);
    if (!(r.ok & (1 << p))) return false;
    double y;
    if (!exact_shift(r.e, d + r.cum[p], &y) || !(y >= 0.0 && y < 1.0)) return false;
    *next = y;
    return true;
}

}  // namespace lane
}  // namespace gpsiq
#endif
