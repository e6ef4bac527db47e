// gpsiq_walk.h -- the reference's double carrier / code accumulator walked from wrap to wrap: the scalar core, shared between the
// host (gpsiq_exact.cpp: NcoWalk adds the per-block wrap-to-wrap table and its AVX-512 build) and the device
// (gpsiq_chain_kernels.hip: one lane per stretch of a block, gpsiq_lane.h).  Same source, same IEEE double additions
// (-ffp-contract=off), so what the host tests pin is what the device runs.
#ifndef GPSIQ_WALK_H
#define GPSIQ_WALK_H

#include <stdint.h>

#if defined(__HIPCC__)
#define GPSIQ_HD __host__ __device__
#else
#define GPSIQ_HD
#endif

namespace gpsiq {

GPSIQ_HD inline uint64_t bits_of(double x) { return __builtin_bit_cast(uint64_t, x); }
GPSIQ_HD inline double from_bits(uint64_t b) { return __builtin_bit_cast(double, b); }
constexpr uint64_t kMant = (UINT64_C(1) << 52) - 1;
constexpr int kCaSeqLen = 1023;                           // GPSIQ_CA_SEQ_LEN (include/gpsiq.h), checked in gpsiq_exact.cpp

// The accumulators walked from wrap to wrap (the comment on NcoWalk in gpsiq_exact.cpp has the whole story).  kTab: binades
// above the addend's own that get a table piece; an addend that needs more takes the general walker.
template <int kTab>
struct WalkCore {
    struct Piece { int64_t dm, k, rem, kdm, span; bool tie; };
#ifndef GPSIQ_WALK_KLOW
#define GPSIQ_WALK_KLOW 2
#endif
    static constexpr int kLow = GPSIQ_WALK_KLOW;          // binades ec .. ec + kLow: plain additions (scripts/ubench_walk.cpp A/Bs it)
    int     kind = 1;                                     // 0: code phase (wrap at 1023 chips), 1: carrier phase (wrap into [0,1))
    double  c = 0.0, thr = 0.0, wrap = 1.0;
    int64_t ec = 0;
    int     top_exp = 1022;                               // biased exponent of the binade that holds the wrap limit
    bool    neg = false, general = true;
    Piece   T[kTab];

    // distance of the cycle's results to their binade edges, in units of the state grid U
    struct Slack {
        int64_t lo, hi;        // the cycle holds for start states x0 + d*U, lo <= d <= hi
        int     unit_exp;      // biased exponent of the binade whose ulp is U
        bool    ok;
        GPSIQ_HD inline void note(double v)
        {
            const uint64_t b = bits_of(v);
            const int e = (int) (b >> 52);                                        // sign bit set -> e >= 2048 -> sh < 0
            const int sh = unit_exp - e;                                          // U / ulp(v) = 2^sh
            if (sh < 0 || sh > 62 || e == 0) { ok = false; return; }
            const int64_t mx = (int64_t) ((b & kMant) | (kMant + 1));
            const int64_t l = 2 - ((mx - ((int64_t) 1 << 52)) >> sh), h = ((((int64_t) 1 << 53) - mx) >> sh) - 2;
            if (l > lo) lo = l;
            if (h < hi) hi = h;
        }
        // the code phase's wrap limit lies inside its top binade: last value before the wrap, first value after the addition
        GPSIQ_HD inline void note_limit(double below, double at_or_above, double limit, double scale)
        {
            const double a = (limit - at_or_above) * scale, t = (limit - below) * scale;      // a <= 0 < t, exact (same binade, times 2^k)
            const int64_t l = 2 - (int64_t) (-a), h = (int64_t) t - 2;
            if (l > lo) lo = l;
            if (h < hi) hi = h;
        }
        // a value that may move up by less than `dist` / down by at most `dist` (descending carrier: U = 2^-53), two units short
        GPSIQ_HD inline void room_above(double dist) { const int64_t h = (int64_t) (dist * 0x1p53) - 2; if (h < hi) hi = h; }
        GPSIQ_HD inline void room_below(double dist) { const int64_t l = 2 - (int64_t) (dist * 0x1p53); if (l > lo) lo = l; }
        GPSIQ_HD inline void fail() { ok = false; }
        GPSIQ_HD inline void tie(double) { ok = false; }       // an exact tie on a wrap: not translation invariant for odd offsets
    };

    // The same bookkeeping in doubles, for the lanes of the time-parallel chain (gpsiq_lane.h; on the device 64-bit shifts and
    // compares per noted value were a third of the walk): the smallest distance of a noted value to the lower / upper edge of
    // its binade, exactly (v - 2^e and 2^(e+1) - v are exact), turned into units once at the end -- floor is monotone, so
    // lo / hi are the ones Slack would have found.  Carrier only (note_limit is the code phase's).
    struct FastSlack {
        double dlo, dhi;       // room below / above, in cycles
        double vmin, vmax;     // range of the noted values: they must be normal and inside the accumulator's range
        int    unit_exp;
        bool   ok;
        int    sigma;          // 0, or the first exact tie met on a wrap: +1 the sum went down to the even value, -1 up (tie())
        GPSIQ_HD inline void init(int uexp) { dlo = 4.0; dhi = 4.0; vmin = 1.0; vmax = 0.0; unit_exp = uexp; ok = true; sigma = 0; }
        GPSIQ_HD inline void note(double v)
        {
            const double edge = from_bits(bits_of(v) & ~kMant), dl = v - edge, dh = edge - dl;
            dlo = dl < dlo ? dl : dlo;
            dhi = dh < dhi ? dh : dhi;
            vmin = v < vmin ? v : vmin;
            vmax = v > vmax ? v : vmax;
        }
        GPSIQ_HD inline void note_limit(double, double, double, double) { ok = false; }
        GPSIQ_HD inline void room_above(double dist) { dhi = dist < dhi ? dist : dhi; }
        GPSIQ_HD inline void room_below(double dist) { dlo = dist < dlo ? dist : dlo; }
        GPSIQ_HD inline void fail() { ok = false; }
        // A sum exactly half way between two values of the wrap's grid goes to the even one.  Moved by an EVEN number of grid
        // units it is half way again between values of the same parities and goes the same way: the walk translates.  Moved by
        // an ODD number d it goes the other way: where this walk went down (err = exact - rounded > 0) to y, that one goes up to
        // y + (d + 1) units, where this one went up, down to y + (d - 1): from there on its offset is d + sigma, an even number
        // -- and even offsets translate through every later tie.  So only the FIRST tie of a walk matters, and its direction.
        GPSIQ_HD inline void tie(double err) { if (!sigma) sigma = err > 0.0 ? 1 : -1; }
        // -> the range of start offsets in units of U (Slack's lo / hi); false: not usable
        GPSIQ_HD inline bool finish(int64_t *lo, int64_t *hi) const
        {
            const double unit = from_bits((uint64_t) (unit_exp - 52) << 52);
            // Slack::note refuses values outside [2^(unit_exp - 62), 2^(unit_exp + 1)) (biased exponents) and zero / subnormals
            if (!ok || (vmax >= vmin && (!(vmin >= from_bits((uint64_t) (unit_exp - 62) << 52)) || !(vmax < from_bits((uint64_t) (unit_exp + 1) << 52))))) return false;
            const double inv = 1.0 / unit;                                 // a power of two: exact
            *lo = 2 - (int64_t) __builtin_floor(dlo * inv);
            *hi = (int64_t) __builtin_floor(dhi * inv) - 2;
            return true;
        }
    };

    // setup = setup_head (the scalars, whether the fast walk takes this addend at all) + setup_piece for every table binade
    // kLow < s <= top_exp - ec: the pieces are independent, so a device workgroup spreads their divisions over its threads.
    GPSIQ_HD void setup_head(double addend, int kind_)
    {
        kind = kind_;
        c = addend;
        wrap = kind == 0 ? (double) kCaSeqLen : 1.0;
        top_exp = kind == 0 ? 1023 + 9 : 1022;
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        ec = (int64_t) (bc >> 52);
        const int64_t mc = (int64_t) ((bc & kMant) | (kMant + 1));
        neg = c < 0.0;
        // the addend at least 2^6 below the top binade's start and not absurdly small; the code phase only ever climbs
        general = ec > top_exp - 6 || ec < top_exp - 40 || top_exp - ec >= kTab || (kind == 0 && neg);
        if (!general && (neg || kind == 0)) {
            // c / ulp of the TOP binade ending in exactly one half: a translated cycle (the wrap-to-wrap table) keeps its
            // parities as long as the state unit is an even multiple of the tie binade's ulp -- not so in the top binade of
            // a descending carrier or of the code phase, whose ulp IS the unit: the probing walk there (see setup_piece).
            const int top = (int) (top_exp - ec);
            if ((mc & (((int64_t) 1 << top) - 1)) == (int64_t) 1 << (top - 1)) general = true;
        }
        thr = general ? 0.0 : from_bits((uint64_t) (ec + kLow + 1) << 52);   // first value the table handles; |c| < thr / 2^kLow
    }

    GPSIQ_HD void setup_piece(int s)
    {
        const int64_t mc = (int64_t) ((bits_of(c) & kMant) | (kMant + 1));
        const int top = (int) (top_exp - ec);
        int64_t dm = mc >> s;                         // rnd(c / ulp) in ulps of the binade, see Nco::build_piece
        const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
        Piece &p = T[s];
        p.tie = false;
        if (rem > half) ++dm;
        else if (rem == half) {
            // c / ulp ends in exactly one half (one addend in 2^s): the sum goes to the even neighbour.  From an even
            // mantissa that is a constant even step, and one addition makes the mantissa even: the walks below take that
            // one addition for real and the run after it from the table.
            p.tie = true;
            dm += dm & 1;
        }
        p.dm = dm;
        // the run ends on the last mantissa of the binade -- or, in the code phase's top binade [512, 1024), short of
        // 1023 = 1023 * 2^43 ulps; downwards it must stay strictly above the binade's first value (Nco::build_piece)
        if (neg) p.span = ((int64_t) 1 << 52) - 2;
        else if (kind == 0 && s == top) p.span = ((int64_t) kCaSeqLen << 43) - 1 - ((int64_t) 1 << 52);
        else p.span = ((int64_t) 1 << 52) - 1;
        p.k = p.span / dm; p.kdm = p.k * dm; p.rem = p.span - p.kdm;
    }

    GPSIQ_HD void setup(double addend, int kind_)
    {
        setup_head(addend, kind_);
        if (general) return;
        const int top = (int) (top_exp - ec);
        for (int s = kLow + 1; s <= top; ++s) setup_piece(s);
    }

    // Positive addend: from x (sample n) up to the next wrap.  true: wrapped, x is the state after the wrap (sample n);
    // false: sample ns was reached first, x is its state.
    template <bool kNote, class S>
    GPSIQ_HD inline bool climb(double &x, long &n, long ns, S *sl) const
    {
        constexpr int64_t one52 = (int64_t) 1 << 52;
        while (x < thr) {                                             // cannot wrap: x + c < thr + thr / 2^kLow
            x += c;
            if (kNote) sl->note(x);
            if (++n == ns) return false;
        }
        for (;;) {                                                    // x in [thr, wrap)
            const uint64_t bx = bits_of(x);
            const Piece &p = T[(int64_t) (bx >> 52) - ec];
            const int64_t mx = (int64_t) ((bx & kMant) | (kMant + 1)), off = mx - one52;
            int64_t run, moved;
            if (p.tie && (mx & 1)) { run = 0; moved = 0; }            // an odd mantissa in a tie binade: one real addition first
            else if (off <= p.rem) { run = p.k; moved = p.kdm; }
            else if (off <= p.rem + p.dm) { run = p.k - 1; moved = p.kdm - p.dm; }
            else if (off > p.span) { run = 0; moved = 0; }            // code phase within one step of 1023
            else { run = (p.span - off) / p.dm; moved = run * p.dm; }
            if (run >= ns - n) { x = from_bits((bx & ~kMant) | ((uint64_t) (mx + (ns - n) * p.dm) & kMant)); n = ns; if (kNote) sl->note(x); return false; }
            x = from_bits((bx & ~kMant) | ((uint64_t) (mx + moved) & kMant));
            n += run;
            if (kNote && run) sl->note(x);
            const double y = x + c;                                   // leaves the binade, or wraps
            ++n;
            if (y >= wrap) {
                if (kNote) {
                    sl->note(y);
                    if (kind == 0) sl->note_limit(x, y, wrap, 0x1p43);
                    const double bb = y - x, err = (x - (y - bb)) + (c - bb);     // the rounding error of x + c, exactly
                    if (__builtin_fabs(err) == (kind == 0 ? 0x1p-44 : 0x1p-53)) sl->tie(err);   // a tie on the grid the wrap is taken on
                }
                x = y - wrap;
                return true;
            }
            x = y;
            if (kNote) sl->note(x);
            if (n == ns) return false;
        }
    }

    // Negative addend (carrier only), the same downwards.
    template <bool kNote, class S>
    GPSIQ_HD inline bool descend(double &x, long &n, long ns, S *sl) const
    {
        constexpr int64_t one52 = (int64_t) 1 << 52;
        while (x >= thr) {
            if (x >= 1.0) {                                           // a wrap that rounded to exactly 1.0 (see evaluate_block)
                if (kNote) sl->fail();
                x += c;
                if (++n == ns) return false;
                continue;
            }
            const uint64_t bx = bits_of(x);
            const Piece &p = T[(int64_t) (bx >> 52) - ec];
            const int64_t mx = (int64_t) ((bx & kMant) | (kMant + 1)), off = (2 * one52 - 1) - mx;
            int64_t run, moved;
            if (p.tie && (mx & 1)) { run = 0; moved = 0; }            // an odd mantissa in a tie binade: one real addition first
            else if (off <= p.rem) { run = p.k; moved = p.kdm; }
            else if (off <= p.rem + p.dm) { run = p.k - 1; moved = p.kdm - p.dm; }
            else if (off > p.span) { run = 0; moved = 0; }            // on the binade's first value: the next sum is rounded underneath
            else { run = (p.span - off) / p.dm; moved = run * p.dm; }
            if (run >= ns - n) { x = from_bits((bx & ~kMant) | ((uint64_t) (mx - (ns - n) * p.dm) & kMant)); n = ns; if (kNote) sl->note(x); return false; }
            x = from_bits((bx & ~kMant) | ((uint64_t) (mx - moved) & kMant));
            n += run;
            if (kNote && run) sl->note(x);
            x += c;                                                   // into the binade underneath; x >= thr >= 2^kLow |c|: still positive
            if (kNote) sl->note(x);
            if (++n == ns) return false;
        }
        for (;;) {                                                    // x < thr: plain additions until the sum turns negative
            const double y = x + c;
            ++n;
            if (y < 0.0) {
                const double r = y + 1.0;
                if (kNote) {
                    // y's own rounding depends on its binade -- unless nothing is rounded: with x in the addend's own binade
                    // both operands are multiples of that binade's ulp and |x + c| < 2^E, so the sum is exact for this x and
                    // for every translated one (which stays in x's binade, see the note on x); only its sign has to hold
                    if ((int64_t) (bits_of(x) >> 52) == ec) {
                        sl->room_above(-y);                                        // y + d*U < 0, two units short
                    } else {
                        sl->note(-y);
                    }
                    const double bb = r - y, err = (y - (r - bb)) + (1.0 - bb);
                    if (r >= 1.0) sl->fail();                                     // rounded up to 1.0
                    else { if (__builtin_fabs(err) == 0x1p-54) sl->tie(err); sl->note(r); }  // (a tie on the grid of [0.5, 1))
                }
                x = r;
                return true;
            }
            if (kNote) {
                // x <= 2|c| (and x >= |c|, the sum is not negative): x + c is exact (Sterbenz), for this x and for every
                // translated one, whatever binade the small difference falls into -- it only has to stay non-negative
                if (x <= -2.0 * c) {
                    sl->room_below(y);
                } else {
                    sl->note(y);
                }
            }
            x = y;
            if (n == ns) return false;
        }
    }

    GPSIQ_HD inline bool cycle(double &x, long &n, long ns) const { return neg ? descend<false>(x, n, ns, (Slack *) nullptr) : climb<false>(x, n, ns, (Slack *) nullptr); }
};


// The carrier walk once more with the table steps in FLOATING POINT, for the lanes of the time-parallel chain (gpsiq_lane.h).
// WalkCore moves a state inside a binade on its 53-bit integer mantissa (64-bit integer adds, compares and a multiply per
// step: pairs of 32-bit instructions on the device, a third of them quarter rate); but every quantity there is an integer
// below 2^53 times the binade's ulp u, so the same step is EXACT in doubles: the distance to the binade's edge x - 2^e, the
// comparisons with rem*u and (rem + dm)*u, the jump x + k*dm*u (the result is a state of the binade: representable).  The
// plain additions at the binade crossings, the wraps and the low binades are WalkCore's own lines.  The one step that needs a
// division (a run entered in the middle of a binade: the first step of a walk) goes back to integers.  Same results, same
// notes: tests/chain_parallel.cpp holds FpWalk against WalkCore state for state, and the chain built on it against the
// serial chain (NcoWalk) bit for bit.  Carrier only (wrap into [0, 1)).
template <int kTab>
struct FpWalk {
    struct Piece { double S, kS, rem_t, span_t; int32_t k, tie; };     // step, k steps, the two thresholds on the distance (see dist())
    static constexpr int kLow = GPSIQ_WALK_KLOW;
    double  c, thr;
    int64_t ec;
    bool    neg, general;
    bool    top_tie;            // descending carrier with an exact tie in [0.5, 1), whose ulp IS the unit: even offsets only
    Piece   F[kTab];

    GPSIQ_HD void setup_head(double addend)
    {
        c = addend;
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        ec = (int64_t) (bc >> 52);
        const int64_t mc = (int64_t) ((bc & kMant) | (kMant + 1));
        neg = c < 0.0;
        general = ec > 1022 - 6 || ec < 1022 - 40 || 1022 - ec >= kTab;
        top_tie = false;
        if (!general && neg) {                                   // (WalkCore::setup_head hands such an addend to the general walker)
            const int top = (int) (1022 - ec);
            top_tie = (mc & (((int64_t) 1 << top) - 1)) == (int64_t) 1 << (top - 1);
        }
        thr = general ? 0.0 : from_bits((uint64_t) (ec + kLow + 1) << 52);
    }

    // the integer piece of binade ec + s (WalkCore::setup_piece)
    GPSIQ_HD inline void int_piece(int s, int64_t *dm_, int64_t *span_, bool *tie_) const
    {
        const int64_t mc = (int64_t) ((bits_of(c) & kMant) | (kMant + 1));
        int64_t dm = mc >> s;
        const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
        bool tie = false;
        if (rem > half) ++dm;
        else if (rem == half) { tie = true; dm += dm & 1; }
        *dm_ = dm; *tie_ = tie;
        *span_ = neg ? ((int64_t) 1 << 52) - 2 : ((int64_t) 1 << 52) - 1;
    }

    GPSIQ_HD void setup_piece(int s)
    {
        int64_t dm, span;
        bool tie;
        int_piece(s, &dm, &span, &tie);
        const int64_t k = span / dm, kdm = k * dm, rem = span - kdm;
        const double u = from_bits((uint64_t) (ec + s - 52) << 52);            // the binade's ulp
        Piece &p = F[s];
        p.S = (double) dm * u; p.kS = (double) kdm * u; p.k = (int32_t) k; p.tie = tie ? 1 : 0;
        // upwards the distance is x - 2^e = off*u; downwards 2^(e+1) - x = (off + 1)*u with off = 2^53 - 1 - mx
        p.rem_t = (double) (neg ? rem + 1 : rem) * u;
        p.span_t = (double) (neg ? span + 1 : span) * u;
    }

    GPSIQ_HD void setup(double addend)
    {
        setup_head(addend);
        if (general) return;
        for (int s = kLow + 1; s <= (int) (1022 - ec); ++s) setup_piece(s);
    }

    // the run from x inside a table binade: steps and distance moved.  dist: the distance the thresholds are written for.
    GPSIQ_HD inline void run_of(uint64_t bx, double dist, int32_t *run, double *moved) const
    {
        const int s = (int) ((int64_t) (bx >> 52) - ec);
        const Piece &p = F[s];
        if (p.tie && (bx & 1)) { *run = 0; *moved = 0.0; }              // an odd mantissa in a tie binade: one real addition first
        else if (dist <= p.rem_t) { *run = p.k; *moved = p.kS; }
        else if (dist <= p.rem_t + p.S) { *run = p.k - 1; *moved = p.kS - p.S; }
        else if (dist > p.span_t) { *run = 0; *moved = 0.0; }
        else {                                                          // entered in the middle of the binade: the division, on integers
            int64_t dm, span;
            bool tie;
            int_piece(s, &dm, &span, &tie);
            const int64_t mx = (int64_t) ((bx & kMant) | (kMant + 1));
            const int64_t off = neg ? (((int64_t) 1 << 53) - 1) - mx : mx - ((int64_t) 1 << 52);
            const int64_t r = (span - off) / dm;
            *run = (int32_t) r;
            *moved = (double) (r * dm) * from_bits((uint64_t) ((bx >> 52) - 52) << 52);
        }
    }

    template <bool kNote, class Sl>
    GPSIQ_HD inline bool climb(double &x, long &n, long ns, Sl *sl) const
    {
        while (x < thr) {                                             // cannot wrap: x + c < thr + thr / 2^kLow
            x += c;
            if (kNote) sl->note(x);
            if (++n == ns) return false;
        }
        for (;;) {                                                    // x in [thr, 1)
            const uint64_t bx = bits_of(x);
            const double edge = from_bits(bx & ~kMant);
            int32_t run;
            double moved;
            run_of(bx, x - edge, &run, &moved);
            if ((long) run >= ns - n) { x += (double) (ns - n) * F[(int64_t) (bx >> 52) - ec].S; n = ns; if (kNote) sl->note(x); return false; }
            x += moved;
            n += run;
            if (kNote && run) sl->note(x);
            const double y = x + c;                                   // leaves the binade, or wraps
            ++n;
            if (y >= 1.0) {
                if (kNote) {
                    sl->note(y);
                    const double bb = y - x, err = (x - (y - bb)) + (c - bb);     // the rounding error of x + c, exactly
                    if (__builtin_fabs(err) == 0x1p-53) sl->tie(err);             // a tie on the grid the wrap is taken on
                }
                x = y - 1.0;
                return true;
            }
            x = y;
            if (kNote) sl->note(x);
            if (n == ns) return false;
        }
    }

    template <bool kNote, class Sl>
    GPSIQ_HD inline bool descend(double &x, long &n, long ns, Sl *sl) const
    {
        while (x >= thr) {
            if (x >= 1.0) {                                           // a wrap that rounded to exactly 1.0
                if (kNote) sl->fail();
                x += c;
                if (++n == ns) return false;
                continue;
            }
            const uint64_t bx = bits_of(x);
            const double edge = from_bits(bx & ~kMant);
            int32_t run;
            double moved;
            run_of(bx, (edge + edge) - x, &run, &moved);
            if ((long) run >= ns - n) { x -= (double) (ns - n) * F[(int64_t) (bx >> 52) - ec].S; n = ns; if (kNote) sl->note(x); return false; }
            x -= moved;
            n += run;
            if (kNote && run) sl->note(x);
            x += c;                                                   // into the binade underneath; x >= thr >= 2^kLow |c|: still positive
            if (kNote) sl->note(x);
            if (++n == ns) return false;
        }
        for (;;) {                                                    // x < thr: plain additions until the sum turns negative (WalkCore::descend)
            const double y = x + c;
            ++n;
            if (y < 0.0) {
                const double r = y + 1.0;
                if (kNote) {
                    if ((int64_t) (bits_of(x) >> 52) == ec) sl->room_above(-y);
                    else sl->note(-y);
                    const double bb = r - y, err = (y - (r - bb)) + (1.0 - bb);
                    if (r >= 1.0) sl->fail();
                    else { if (__builtin_fabs(err) == 0x1p-54) sl->tie(err); sl->note(r); }
                }
                x = r;
                return true;
            }
            if (kNote) {
                if (x <= -2.0 * c) sl->room_below(y);
                else sl->note(y);
            }
            x = y;
            if (n == ns) return false;
        }
    }

    struct NoSlack { GPSIQ_HD void note(double) {} GPSIQ_HD void room_above(double) {} GPSIQ_HD void room_below(double) {} GPSIQ_HD void fail() {} GPSIQ_HD void tie(double) {} };
    GPSIQ_HD inline bool cycle(double &x, long &n, long ns) const
    {
        return neg ? descend<false>(x, n, ns, (NoSlack *) nullptr) : climb<false>(x, n, ns, (NoSlack *) nullptr);
    }
};

}  // namespace gpsiq
#endif
