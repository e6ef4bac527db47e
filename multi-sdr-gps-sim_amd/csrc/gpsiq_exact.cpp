// gpsiq_exact.cpp — GPSIQ_NCO_REFERENCE: the reference's double-precision NCOs, reproduced exactly.
//
// The reference advances code_phase and carr_phase by sequential double additions
// (gps.c:2789-2792, gps.c:2821-2826).  The library's default model is the closed form on integers of
// include/gpsiq.h; given the same block-start state the two differ in a few samples per 10^7, where a
// phase sits within the double path's accumulated rounding drift of a chip or LUT boundary.  This file
// finds exactly those samples on the host and describes them as patches for the device, and it
// carries carr_phase from block to block as the reference's own accumulator would leave it, so that a
// whole run equals the reference element for element.
//
// How, without stepping 260 000 additions per channel and block:
//   * x += c in double is piecewise linear.  While x stays inside one binade it is a multiple of the
//     binade's ulp u and the addition adds the constant S = rnd(c/u)*u (ties to even: constant from the
//     second addition inside the binade on, when the parity has settled).  Nco::advance() does the
//     reference's own operation (a real double addition, then the wrap rule) at binade crossings and
//     wraps and jumps over the steady run in between with one integer division (S straight from the addend's
//     mantissa; the rare exact-tie binades are probed with two real additions): about a dozen pieces
//     per carrier cycle or code period.
//   * Only samples whose fixed-point phase lies within the drift bound of a boundary can differ:
//     |double path - real arithmetic| <= n * 2^-54 cycle (carrier, values < 1) or n * 2^-44 chip (code,
//     values < 1024), the fixed-point path is within (n/2 + 1) units of its last place of real arithmetic.
//     Those candidates are the n with (a + n*b) mod 2^k inside a narrow window: found with a
//     Euclid-style descent in O(log) per hit (first_in_window), typically none or one per channel and block.
//   * At every candidate both paths are evaluated exactly; where (LUT index, sign) differ, a patch
//     {block, sample, slot, index, sign} is emitted.  The device runs the fixed-point kernels unchanged
//     and then recomputes the patched samples (apply_patches in gpsiq_kernels.hip).
// The carrier chain is serial per channel (block k starts where the double accumulator left block k-1);
// channels run on host threads, and everything but the carrier walk is parallel over blocks.
#include "gpsiq_internal.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>

namespace gpsiq {

namespace {

typedef unsigned __int128 u128;

inline uint64_t bits_of(double x) { uint64_t b; std::memcpy(&b, &x, 8); return b; }
inline double from_bits(uint64_t b) { double x; std::memcpy(&x, &b, 8); return x; }
constexpr uint64_t kMant = (UINT64_C(1) << 52) - 1;

// One double accumulator of the reference loop.  kind 0: code phase (wrap at 1023 chips,
// gps.c:2789-2792); kind 1: carrier phase (wrap into [0,1), gps.c:2821-2826).
struct Nco {
    double x, c;      // state used by sample n; the addend
    long   n, wraps;  // wraps: number of wraps on the way to sample n (code: code periods completed)
    int    kind;

    // Per binade of x (indexed by the exponent difference to the addend), filled on first use: the settled step dm in
    // ulps (state 2: not usable -- exact tie, or step zero), and the run length from the binade's entry edge written as
    // span = k*dm + rem, so that a run entered within one step of the edge needs a comparison instead of a division.
    struct Piece { int64_t dm, span, k, rem, kdm; int state; };       // state 0: not built, 1: usable, 2: take the probing path
    Piece piece[55] = {};

    inline double step(double v, int *wrapped) const
    {
        double y = v + c;
        *wrapped = 0;
        if (kind == 0) {
            if (y >= (double) GPSIQ_CA_SEQ_LEN) { y -= (double) GPSIQ_CA_SEQ_LEN; *wrapped = 1; }
        } else {
            if (y >= 1.0) { y -= 1.0; *wrapped = 1; }
            else if (y < 0.0) { y += 1.0; *wrapped = 1; }
        }
        return y;
    }

    // Largest j such that x, x + S, ..., x + j*S (S = dm ulps of x's binade) all stay inside x's binade and short of
    // the wrap, capped at `cap`.  mx = x's 53-bit mantissa.
    inline long run_length(int64_t mx, int64_t dm, uint64_t expo, long cap) const
    {
        if (dm > 0) {
            int64_t lim = (int64_t) 1 << 53;            // first mantissa of the next binade
            // the wrap comes before the binade ends only in the code phase's top binade [512, 1024):
            // 1023 = 1023 * 2^43 ulps there (the carrier's top binade [0.5, 1) ends exactly at its wrap)
            if (kind == 0 && expo == 1023 + 9) lim = (int64_t) GPSIQ_CA_SEQ_LEN << 43;
            const int64_t j = (lim - 1 - mx) / dm;
            return j < cap ? (long) j : cap;
        }
        // downwards the run must stay strictly above the binade's first value 2^52 ulps: a sum that falls below it
        // is rounded on the finer grid of the binade underneath, so the last step onto (or from) 2^52 is a real addition
        if (mx <= ((int64_t) 1 << 52) && c < 0.0) return 0;
        if (dm < 0) {
            const int64_t j = (mx - ((int64_t) 1 << 52) - 1) / -dm;
            return j < cap ? (long) j : cap;
        }
        return cap;                                      // the addend is below half an ulp: the phase stands still
    }

    void build_piece(Piece &p, int64_t shift, int64_t ex, int64_t mc, bool neg) const
    {
        // x + c inside x's binade is x + S with S = rnd(c/u) ulps, whatever x is -- unless c/u ends in exactly one
        // half, where the rounding goes to even and depends on x's parity (left to the probing path of advance())
        int64_t dm = shift == 0 ? mc : (shift >= 53 ? 0 : mc >> shift);
        bool tie = false;
        if (shift > 0 && shift <= 53) {
            const int64_t rem = mc & (((int64_t) 1 << shift) - 1), half = (int64_t) 1 << (shift - 1);
            if (rem > half) ++dm;
            else if (rem == half) tie = true;
        }
        p.state = (tie || dm == 0) ? 2 : 1;
        if (p.state == 2) return;
        p.dm = dm;
        if (!neg) {
            // the run ends on the last mantissa of the binade -- or of the code period: the wrap comes before the binade
            // ends only in the code phase's top binade [512, 1024), 1023 = 1023 * 2^43 ulps there (the carrier's top
            // binade [0.5, 1) ends exactly at its wrap)
            int64_t lim = (int64_t) 1 << 53;
            if (kind == 0 && ex == 1023 + 9) lim = (int64_t) GPSIQ_CA_SEQ_LEN << 43;
            p.span = lim - 1 - ((int64_t) 1 << 52);                       // from the first mantissa 2^52
        } else {
            // downwards the run must stay strictly above the binade's first value 2^52 ulps: a sum that falls below it is
            // rounded on the finer grid of the binade underneath, so the last step onto (or from) 2^52 is a real addition
            p.span = ((int64_t) 1 << 52) - 2;                             // from the last mantissa 2^53 - 1 down to 2^52 + 1
        }
        p.k = p.span / dm;
        p.kdm = p.k * dm;
        p.rem = p.span - p.kdm;
    }

    // Move to sample `target` (>= n).
    void advance(long target)
    {
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        const int64_t ec = (int64_t) (bc >> 52), mc = (int64_t) ((bc & kMant) | (kMant + 1));
        const bool neg = c < 0.0;
        while (n < target) {
            const uint64_t bx = bits_of(x);
            const int64_t ex = (int64_t) (bx >> 52);
            if (ex != 0 && ec != 0 && ex >= ec && ex - ec <= 54) {
                Piece &p = piece[ex - ec];
                if (p.state == 0) build_piece(p, ex - ec, ex, mc, neg);
                if (p.state == 1) {
                    const int64_t mx = (int64_t) ((bx & kMant) | (kMant + 1));
                    // distance of x from the edge the run was measured from; the run is (span - off) / dm steps long
                    const int64_t off = neg ? (((int64_t) 1 << 53) - 1) - mx : mx - ((int64_t) 1 << 52);
                    int64_t run, moved;
                    if (off <= p.rem) { run = p.k; moved = p.kdm; }
                    else if (off <= p.rem + p.dm) { run = p.k - 1; moved = p.kdm - p.dm; }
                    else if (off > p.span) { run = 0; moved = 0; }
                    else { run = (p.span - off) / p.dm; moved = run * p.dm; }
                    const long cap = target - n;
                    if (run >= cap) { run = cap; moved = run * p.dm; }
                    if (run > 0) {
                        x = from_bits((bx & ~kMant) | ((uint64_t) (neg ? mx - moved : mx + moved) & kMant));
                        n += run;
                        if (run == cap) return;
                    }
                    // the next addition leaves the binade or wraps: the reference's own operation
                    int w;
                    const double y = step(x, &w);
                    if (y == x && !w) { n = target; return; }    // e.g. exactly 2^k with a negative addend below a quarter ulp: stuck for good
                    x = y; ++n; wraps += w;
                    continue;
                }
            }
            // zero / subnormal, a binade below the addend's, or the tie case: real additions, and a jump once two
            // consecutive additions inside one binade have shown the settled step
            int wy;
            const double y = step(x, &wy);
            if (y == x && !wy) { n = target; return; }           // the addend no longer moves x: it never will again
            const uint64_t by = bits_of(y);
            if (wy || (bx >> 52) != (by >> 52) || (bx >> 52) == 0) {
                x = y; ++n; wraps += wy;
                continue;
            }
            if (n + 1 == target) { x = y; ++n; return; }
            int wz;
            const double z = step(y, &wz);
            const uint64_t bz = bits_of(z);
            if (wz || (bz >> 52) != (by >> 52)) { x = z; n += 2; wraps += wz; continue; }
            const int64_t my = (int64_t) ((by & kMant) | (kMant + 1)), mz = (int64_t) ((bz & kMant) | (kMant + 1));
            const long run = run_length(my, mz - my, by >> 52, target - (n + 1));    // samples n+1 .. n+1+run are y + j*S
            x = from_bits((by & ~kMant) | ((uint64_t) (my + (int64_t) run * (mz - my)) & kMant));
            n += 1 + run;
        }
    }
};

// Smallest x >= 0 with L <= (A*x mod M) <= R, for 0 <= L <= R < M and 0 <= A < M; kNone if there is none.
constexpr u128 kNone = ~(u128) 0;
u128 first_in_range(uint64_t A, uint64_t M, uint64_t L, uint64_t R)
{
    if (L == 0) return 0;
    if (A == 0) return kNone;
    if (A > M - A) {                 // 2A > M: count downwards instead
        A = M - A;
        const uint64_t l = M - R, r = M - L;
        L = l; R = r;
    }
    const uint64_t k = (L + A - 1) / A;
    if ((u128) k * A <= R) return k;
    // no multiple of A inside [L, R]: after y wraps of M the window is at M*y + [L, R]; it holds a
    // multiple of A iff ((-M mod A) * y) mod A lies in [L mod A, R mod A]
    const u128 y = first_in_range((A - M % A) % A, A, L % A, R % A);
    if (y == kNone) return kNone;
    return ((u128) M * y + L + A - 1) / A;
}

// All n in [0, nsamp) with (a + n*b) mod 2^k within w of 0 (either side), ascending; false if there
// are more than `cap` (the caller then looks at every sample).
bool candidates(uint64_t a, uint64_t b, int k, uint64_t w, long nsamp, size_t cap, std::vector<long> *out)
{
    const uint64_t M = UINT64_C(1) << k, mask = M - 1;
    if (2 * w + 1 >= M) return false;
    const uint64_t W = 2 * w + 1;                    // shifted by w the window is [0, W)
    b &= mask;
    long base = 0;
    while (base < nsamp) {
        const uint64_t cur = (uint64_t) (((u128) a + w + (u128) b * (uint64_t) base) & mask);
        long hit;
        if (cur < W) hit = base;
        else {
            const u128 x = first_in_range(b, M, M - cur, M - cur + W - 1);
            if (x == kNone || x >= (u128) (nsamp - base)) break;
            hit = base + (long) x;
        }
        if (out->size() >= cap) return false;
        out->push_back(hit);
        base = hit + 1;
    }
    return true;
}

inline unsigned nav_bit(const gpsiq_chan_t &ch, long bit)     // the data bit `bit` nav-bit periods after the block's first
{
    const long pos = (long) ch.ibit + bit;
    const long w = (long) ch.iword + pos / 30;
    if (w >= GPSIQ_N_DWRD) return 0;                            // the float loop would run past dwrd[] too; quantize_one rejects it
    return (ch.dwrd[w] >> (29 - (int) (pos % 30))) & 1u;
}

}  // namespace

// One C/A code, generated when a block first has a candidate to look at (most have none).
struct CodeCache {
    int     prn = 0;
    uint8_t ca[GPSIQ_CA_SEQ_LEN];
    const uint8_t *get(int p) { if (p != prn) { ca_code(p, ca); prn = p; } return ca; }
};

// The accumulators walked from wrap to wrap.
//
// Nco::advance costs a dozen dependent pieces per carrier cycle / code period, ~70 ns.  For the common case (a normal
// addend well below the accumulator's range, no exact-tie binade) NcoWalk has the same walk in a leaner form -- between
// two wraps the phase climbs (or, with a negative carrier addend, descends) through the binades from the addend's own to
// the top one; the lowest three hold 1, 2, 4 steps: plain additions there cost less than a table piece each (five binades
// measured 5 % slower on the GPU box's host, scripts/ubench_walk.cpp); above them one piece per binade, with the run length from a per-binade table -- and, on top of it, a map from wrap to wrap.
// Right after a wrap the state is a multiple of U = the ulp of the binade the wrap is taken in (carrier: 2^-52 for
// y - 1.0 with y in [1, 2), 2^-53 for y + 1.0 in [0.5, 1); code: 2^-43 for y - 1023.0 with y in [512, 1024)).  Start the
// same cycle from x0 + d*U instead of x0: as long as every RESULT of the cycle's additions stays inside the binade it had
// (and on its side of the wrap limit), every rounding drops the same bits -- d*U is a multiple of every ulp involved, an
// even one wherever a tie could occur, since addends with an exact-tie binade above the plain additions take the general
// walker -- and the whole cycle is the first one translated: same number of samples, final state moved by d*U.  The cycle
// walk therefore also returns the range of d for which that holds (the distance of every result to the edges of its
// binade, two units short on either side because a sum that crosses an edge is rounded on the other grid), and the cycle
// becomes an entry {first state, last state, state increment, samples} of a small table.  As the start state sweeps the
// addend's width every binade edge on the way is crossed one step earlier exactly once, so the table has about as many
// entries as a cycle has binades (a dozen or two); a block of 260 000 samples at 2.6 kHz Doppler is 260 carrier cycles, of
// which the first ~17 are walked and the rest are look-ups on integers.  The one rounding that is not translation
// invariant is the wrap itself when it is an exact tie (the kept bit's parity moves with d): such a cycle is never tabled.
// The addends change with every block (gps.c:2042-2043), so a table lives for one block.
struct NcoWalk {
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
    Piece   T[64];

    // distance of the cycle's results to their binade edges, in units of the state grid U
    struct Slack {
        int64_t lo, hi;        // the cycle holds for start states x0 + d*U, lo <= d <= hi
        int     unit_exp;      // biased exponent of the binade whose ulp is U
        bool    ok;
        inline void note(double v)
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
        inline void note_limit(double below, double at_or_above, double limit, double scale)
        {
            const double a = (limit - at_or_above) * scale, t = (limit - below) * scale;      // a <= 0 < t, exact (same binade, times 2^k)
            const int64_t l = 2 - (int64_t) (-a), h = (int64_t) t - 2;
            if (l > lo) lo = l;
            if (h < hi) hi = h;
        }
    };

    void setup(double addend, int kind_)
    {
        kind = kind_;
        c = addend;
        wrap = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
        top_exp = kind == 0 ? 1023 + 9 : 1022;
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        ec = (int64_t) (bc >> 52);
        const int64_t mc = (int64_t) ((bc & kMant) | (kMant + 1));
        neg = c < 0.0;
        // the addend at least 2^6 below the top binade's start and not absurdly small; the code phase only ever climbs
        general = ec > top_exp - 6 || ec < top_exp - 40 || (kind == 0 && neg);
        const int top = (int) (top_exp - ec);             // exponent difference of the top binade
        for (int s = kLow + 1; s <= top && !general; ++s) {
            int64_t dm = mc >> s;                         // rnd(c / ulp) in ulps of the binade, see Nco::build_piece
            const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
            Piece &p = T[s];
            p.tie = false;
            if (rem > half) ++dm;
            else if (rem == half) {
                // c / ulp ends in exactly one half (one addend in 2^s): the sum goes to the even neighbour.  From an even
                // mantissa that is a constant even step, and one addition makes the mantissa even: the walks below take that
                // one addition for real and the run after it from the table.  A translated cycle (the wrap-to-wrap table)
                // keeps its parities as long as the state unit is an even multiple of this binade's ulp -- not so in the top
                // binade of a descending carrier or of the code phase, whose ulp IS the unit: the probing walk there.
                if (s == top && (neg || kind == 0)) general = true;
                p.tie = true;
                dm += dm & 1;
            }
            p.dm = dm;
            // the run ends on the last mantissa of the binade -- or, in the code phase's top binade [512, 1024), short of
            // 1023 = 1023 * 2^43 ulps; downwards it must stay strictly above the binade's first value (Nco::build_piece)
            if (neg) p.span = ((int64_t) 1 << 52) - 2;
            else if (kind == 0 && s == top) p.span = ((int64_t) GPSIQ_CA_SEQ_LEN << 43) - 1 - ((int64_t) 1 << 52);
            else p.span = ((int64_t) 1 << 52) - 1;
            p.k = p.span / dm; p.kdm = p.k * dm; p.rem = p.span - p.kdm;
        }
        if (!general) thr = from_bits((uint64_t) (ec + kLow + 1) << 52);   // first value the table handles; |c| < thr / 2^kLow
    }

    // Positive addend: from x (sample n) up to the next wrap.  true: wrapped, x is the state after the wrap (sample n);
    // false: sample ns was reached first, x is its state.
    template <bool kNote>
    inline bool climb(double &x, long &n, long ns, Slack *sl) const
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
            if (run >= ns - n) { x = from_bits((bx & ~kMant) | ((uint64_t) (mx + (ns - n) * p.dm) & kMant)); n = ns; return false; }
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
                    if (std::fabs(err) == (kind == 0 ? 0x1p-44 : 0x1p-53)) sl->ok = false;   // a tie on the grid the wrap is taken on
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
    template <bool kNote>
    inline bool descend(double &x, long &n, long ns, Slack *sl) const
    {
        constexpr int64_t one52 = (int64_t) 1 << 52;
        while (x >= thr) {
            if (x >= 1.0) {                                           // a wrap that rounded to exactly 1.0 (see evaluate_block)
                if (kNote) sl->ok = false;
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
            if (run >= ns - n) { x = from_bits((bx & ~kMant) | ((uint64_t) (mx - (ns - n) * p.dm) & kMant)); n = ns; return false; }
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
                        const int64_t h = (int64_t) (-y * 0x1p53) - 2;            // y + d*U < 0, two units short
                        if (h < sl->hi) sl->hi = h;
                    } else {
                        sl->note(-y);
                    }
                    const double bb = r - y, err = (y - (r - bb)) + (1.0 - bb);
                    if (r >= 1.0 || std::fabs(err) == 0x1p-54) sl->ok = false;   // rounded up to 1.0, or a tie on the grid of [0.5, 1)
                    else sl->note(r);
                }
                x = r;
                return true;
            }
            if (kNote) {
                // x <= 2|c| (and x >= |c|, the sum is not negative): x + c is exact (Sterbenz), for this x and for every
                // translated one, whatever binade the small difference falls into -- it only has to stay non-negative
                if (x <= -2.0 * c) {
                    const int64_t l = 2 - (int64_t) (y * 0x1p53);
                    if (l > sl->lo) sl->lo = l;
                } else {
                    sl->note(y);
                }
            }
            x = y;
            if (n == ns) return false;
        }
    }

    inline bool cycle(double &x, long &n, long ns) const { return neg ? descend<false>(x, n, ns, nullptr) : climb<false>(x, n, ns, nullptr); }

    // the state at sample `target` (>= n) of a walk that is at sample n in state x; counts the wraps on the way
    inline double state_at(double x, long n, long target, long *wraps) const
    {
        while (n < target && cycle(x, n, target)) ++*wraps;
        return x;
    }

    // wrap-to-wrap table.  Entries are valid each on its own (overlaps are harmless) and are found through 1024 buckets
    // over the range of post-wrap states: a bucket wholly inside an entry carries its increment and sample count (one shift,
    // one load and an add per cycle, no search: the states are as good as random, a binary search would mispredict at every
    // level); the few buckets that straddle an edge fall back to a scan of the entries.
    struct Entry { int64_t first, last, inc; long steps; };
    struct Bucket { int64_t inc; long steps; };
    static constexpr int kMaxEntries = 64, kBuckets = 1024;

    // A function of its own (and not inlined): in the middle of run() the compiler keeps the state in memory, and the store
    // forwarding on it would cost more than the look-up itself.
    __attribute__((noinline)) static void hits(const Bucket *bkt, int64_t base, int64_t W, int bshift, long limit,
                                               int64_t *m_io, long *n_io, long *wraps_io)
    {
        int64_t m = *m_io;
        long n = *n_io, wraps = *wraps_io;
        for (;;) {
            const uint64_t rel = (uint64_t) (m - base);
            if (rel >= (uint64_t) W) break;
            const Bucket &bk = bkt[rel >> bshift];
            if (bk.steps == 0 || n + bk.steps > limit) break;
            m += bk.inc; n += bk.steps; ++wraps;
        }
        *m_io = m; *n_io = n; *wraps_io = wraps;
    }

    // The state after ns samples from x0.  targets[nt] (ascending, < ns): samples whose state is wanted as well -> xs[nt]
    // and, if wraps_at is given, the number of wraps before each (code: code periods completed, gps.c:2791-2793).
    double run(double x0, long ns, const long *targets = nullptr, int nt = 0, double *xs = nullptr, long *wraps_at = nullptr) const
    {
        int tk = 0;
        if (ns <= 0) return x0;
        if (general || !(x0 >= 0.0 && x0 < wrap)) {
            Nco a = {x0, c, 0, 0, kind};
            for (; tk < nt; ++tk) { a.advance(targets[tk]); xs[tk] = a.x; if (wraps_at) wraps_at[tk] = a.wraps; }
            a.advance(ns);
            return a.x;
        }
        double x = x0;
        long n = 0, wraps = 0;
        // the targets that lie in [n, upto), from a walk that is at sample n in state x (not disturbed)
        auto visit = [&](double xv, long nv, long upto) {
            long w = wraps;
            for (; tk < nt && targets[tk] < upto; ++tk) {      // each from the one before: a block whose every sample is a target stays linear
                xv = state_at(xv, nv, targets[tk], &w);
                nv = targets[tk];
                xs[tk] = xv;
                if (wraps_at) wraps_at[tk] = w;
            }
        };
        // few cycles in the block: the table would never be read
        static const bool no_map = std::getenv("GPSIQ_WALK_NOMAP") != nullptr;      // A/B knob: every cycle walked
        const double span = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
        const bool use_map = !no_map && std::fabs(c) * (double) ns > 24.0 * span;
        if (!use_map) {
            visit(x, n, ns);
            while (n < ns && cycle(x, n, ns)) {}
            return x;
        }
        {   // to the first wrap
            const double xs0 = x; const long n0 = n;
            const bool wrapped = cycle(x, n, ns);
            visit(xs0, n0, wrapped ? n : ns + 1);
            if (!wrapped || n == ns) return x;
            ++wraps;
        }
        const int uexp = kind == 0 ? 1023 + 9 : neg ? 1022 : 1023;   // biased exponent of the binade whose ulp is the state grid
        const double scale = std::ldexp(1.0, 1075 - uexp), unit = std::ldexp(1.0, uexp - 1075);
        const int64_t m_max = kind == 0 ? ((int64_t) GPSIQ_CA_SEQ_LEN << 43) - 1 : neg ? ((int64_t) 1 << 53) - 1 : ((int64_t) 1 << 52) - 1;
        // post-wrap states: [0, c] (positive addend) or [1 + c, 1) (negative), W units wide
        const int64_t W = (int64_t) (std::fabs(c) * scale) + 4;
        const int64_t base = neg ? ((int64_t) 1 << 53) - W : 0;
        int bshift = 0;
        while ((W >> bshift) >= kBuckets) ++bshift;
        Entry tab[kMaxEntries];
        Bucket bkt[kBuckets];                                         // steps == 0: no entry covers the whole bucket
        for (int i = 0; i < kBuckets; ++i) bkt[i].steps = 0;
        int ntab = 0;
        long min_steps = ns;                                          // shortest cycle seen
        int64_t m = (int64_t) (x * scale);                            // exact: the state is a multiple of the unit
        for (;;) {
            // the common case, cycle after cycle: the state's bucket lies wholly inside one entry and names its increment.
            // Stops before a cycle that would pass the end of the block or hold the next target.
            hits(bkt, base, W, bshift, tk < nt && targets[tk] < ns ? targets[tk] : ns, &m, &n, &wraps);
            if (n >= ns) return (double) m * unit;
            const int64_t rel = m - base;
            const Entry *e = nullptr;
            if (rel >= 0 && rel < W)                                  // a bucket that straddles an edge, a target inside the cycle, the block's end
                for (int i = 0; i < ntab; ++i)
                    if (tab[i].first <= m && m <= tab[i].last) { e = &tab[i]; break; }
            if (e) {
                if (e->steps > ns - n) break;                         // the block ends inside this cycle
                if (tk < nt && targets[tk] < n + e->steps) visit((double) m * unit, n, n + e->steps);
                m += e->inc; n += e->steps; ++wraps;
                continue;
            }
            // not in the table (or 1.0, outside its domain): walk the cycle, noting how far the start state may move
            x = (double) m * unit;
            const double xc = x; const long nc = n;
            const bool memo = rel >= 0 && rel < W && ntab < kMaxEntries && (ntab == 0 || ns - n >= 2 * min_steps);
            if (!memo) {
                const bool wrapped = cycle(x, n, ns);
                visit(xc, nc, wrapped ? n : ns + 1);
                if (!wrapped) return x;
                ++wraps;
                m = (int64_t) (x * scale);
                continue;
            }
            Slack sl = {-m, m_max - m, uexp, true};                   // the start state itself stays inside the accumulator's range
            long ne = 0;
            const bool wrapped = neg ? descend<true>(x, ne, ns - n, &sl) : climb<true>(x, ne, ns - n, &sl);
            n += ne;
            visit(xc, nc, wrapped ? n : ns + 1);
            if (!wrapped) return x;                                   // the block ended inside this cycle
            ++wraps;
            const int64_t m2 = (int64_t) (x * scale);
            if (sl.ok && sl.lo <= 0 && sl.hi >= 0 && x < wrap && m2 <= m_max) {
                Entry &t = tab[ntab++];
                t.first = m + sl.lo; t.last = m + sl.hi; t.inc = m2 - m; t.steps = ne;
                if (ne < min_steps) min_steps = ne;
                // buckets wholly inside [first, last]
                const int64_t f = t.first - base, l = t.last - base;
                int64_t k0 = f <= 0 ? 0 : ((f - 1) >> bshift) + 1;    // first bucket starting at or after `first`
                int64_t k1 = l >= W ? kBuckets - 1 : ((l + 1) >> bshift) - 1;  // last bucket ending at or before `last`
                if (k1 > kBuckets - 1) k1 = kBuckets - 1;
                for (int64_t k = k0; k <= k1; ++k) { bkt[k].inc = t.inc; bkt[k].steps = ne; }
            }
            m = m2;
        }
        // the last, partial cycle
        x = (double) m * unit;
        visit(x, n, ns);
        while (n < ns && cycle(x, n, ns)) {}
        return x;
    }
};

// One channel of one block.  ch.carr_phase is the double the block starts from (1.0 included, see below); q its quantised
// form (quantize_one without carry_in, i.e. seeded from that double).  Returns the carrier phase the reference's accumulator
// holds after the block (gps.c:2821-2826 nsamp times) and appends the samples where the double path takes another LUT entry or
// sign than the closed form.  One walk of each accumulator serves both: the candidates are found first (without visiting
// samples), and the walk that carries the phase to the end of the block reports the state at the candidate samples on its way.
static double evaluate_block(const gpsiq_chan_t &ch, const gpsiq_qchan_t &q, double delt, int nsamp, int block, int slot,
                             CodeCache *codes, std::vector<gpsiq_patch_t> *out)
{
    const long ns = nsamp;
    const double carr_inc = ch.f_carr * delt, code_inc = ch.f_code * delt;
    NcoWalk cw;
    cw.setup(carr_inc, 1);
    if (ns <= 0) return ch.carr_phase;
    // drift bounds at the end of the block, in units of the fixed-point formats (see the header)
    const uint64_t w_carr = ((uint64_t) ns << (GPSIQ_CARR_FRAC_BITS - 54)) + (uint64_t) ns / 2 + 4;
    const uint64_t w_code = ((uint64_t) ns << (GPSIQ_CODE_FRAC_BITS - 44)) + (uint64_t) ns / 2 + 4;
    std::vector<long> t_carr, t_code, targets;
    constexpr size_t kCap = 256;
    bool every = false;
    if (carr_inc != 0.0)          // a zero addend leaves both paths constant and equal
        every |= !candidates(q.carr_phase, (uint64_t) q.carr_step, GPSIQ_CARR_FRAC_BITS - 9, w_carr, ns, kCap, &t_carr);
    every |= !candidates(q.code_frac, q.code_step, GPSIQ_CODE_FRAC_BITS, w_code, ns, kCap, &t_code);
    if (every) {
        targets.resize((size_t) ns);
        for (long n = 0; n < ns; ++n) targets[(size_t) n] = n;
    } else {
        targets.resize(t_carr.size() + t_code.size());
        std::merge(t_carr.begin(), t_carr.end(), t_code.begin(), t_code.end(), targets.begin());
        targets.erase(std::unique(targets.begin(), targets.end()), targets.end());
    }
    if (targets.empty()) return cw.run(ch.carr_phase, ns);
    const int nt = (int) targets.size();
    std::vector<double> xc((size_t) nt), xk;
    std::vector<long> periods;
    const double carr_end = cw.run(ch.carr_phase, ns, targets.data(), nt, xc.data());
    const bool walk_code = every || !t_code.empty();
    if (walk_code) {
        NcoWalk kw;
        kw.setup(code_inc, 0);
        xk.resize((size_t) nt); periods.resize((size_t) nt);
        (void) kw.run(ch.code_phase, targets[(size_t) nt - 1] + 1, targets.data(), nt, xk.data(), periods.data());
    }
    const uint8_t *ca = codes->get(ch.prn);
    for (int k = 0; k < nt; ++k) {
        const long n = targets[(size_t) k];
        // fixed-point path (include/gpsiq.h)
        const uint64_t P = (q.carr_phase + (uint64_t) q.carr_step * (uint64_t) n) & ((UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1);
        const unsigned idx_f = (unsigned) (P >> (GPSIQ_CARR_FRAC_BITS - 9));
        const u128 T = (u128) q.code_frac + (u128) q.code_step * (uint64_t) n;
        const uint64_t A = (uint64_t) q.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
        const unsigned chip_f = (unsigned) (A % GPSIQ_CA_SEQ_LEN);
        const long per_f = (long) (A / GPSIQ_CA_SEQ_LEN);
        const unsigned neg_f = ca[chip_f] ^ nav_bit(ch, ((long) ch.icode + per_f) / 20);
        // double path
        unsigned idx_d = (unsigned) (int) std::floor(xc[(size_t) k] * 512.0);           // gps.c:2775
        if (idx_d > 511u) idx_d = 511u;   // carr_phase == 1.0 (a negative phase within 2^-54 of zero): the reference indexes past its table there
        unsigned neg_d = neg_f;
        if (walk_code) {
            const unsigned chip_d = (unsigned) (int) xk[(size_t) k];                     // gps.c:2817
            neg_d = ca[chip_d] ^ nav_bit(ch, ((long) ch.icode + periods[(size_t) k]) / 20);   // gps.c:2791-2811
        }
        if (idx_d != idx_f || neg_d != neg_f) {
            gpsiq_patch_t p;
            p.block = (uint32_t) block; p.sample = (uint32_t) n;
            p.slot = (uint8_t) slot; p.neg = (uint8_t) neg_d; p.lut = (uint16_t) idx_d;
            out->push_back(p);
        }
    }
    return carr_end;
}

// ---- the host side of GPSIQ_NCO_REFERENCE as ONE pass per channel -------------------------------------------------
// Everything a channel needs for block b -- the accumulator the reference holds at its start, the quantised descriptor
// seeded from it, the samples where the double path leaves the closed form -- depends on that channel's own history only
// (the patch slot is the count of active channels before it in the block, read off the descriptors).  So one host thread
// per channel walks the whole timeline once, block after block, and counts the pieces it has finished; whoever renders
// (the calling thread of a batch, the device queues of a multi-device batch) waits for a piece to be complete in all
// channels and takes its descriptors and patches while the walkers are already in the pieces behind it.  No barrier
// between the carrier chain and the candidate search, no thread woken per piece, and the work per thread is the same for
// every channel whatever the piece length.
RefWalk::RefWalk(const gpsiq_chan_t *ch_, int nblocks_, int nchan_, double delt_, int nsamp_, gpsiq_qchan_t *q_,
                 const double *carr_in_, const int *prn_in_, const std::vector<int> &piece_ends)
    : ch(ch_), nblocks(nblocks_), nchan(nchan_), nsamp(nsamp_), delt(delt_), q(q_), ends(piece_ends), have_in(carr_in_ && prn_in_)
{
    if (ends.empty() || ends.back() != nblocks) ends.push_back(nblocks);
    for (int i = 0; i < GPSIQ_MAX_CHAN; ++i) {
        carr_in[i] = have_in && i < nchan ? carr_in_[i] : 0.0;
        prn_in[i] = have_in && i < nchan ? prn_in_[i] : 0;
        carr_end[i] = 0.0; last_prn[i] = 0; taken[i] = 0;
        pthread_mutex_init(&pmu[i], nullptr);
    }
    done.reset(new std::atomic<int>[ends.size()]);
    for (size_t k = 0; k < ends.size(); ++k) done[k].store(0, std::memory_order_relaxed);
    pthread_mutex_init(&mu, nullptr);
    pthread_cond_init(&cv, nullptr);
}

RefWalk::~RefWalk()
{
    for (int i = 0; i < GPSIQ_MAX_CHAN; ++i) pthread_mutex_destroy(&pmu[i]);
    pthread_mutex_destroy(&mu);
    pthread_cond_destroy(&cv);
}

void RefWalk::finish_piece(size_t k)
{
    if (done[k].fetch_add(1, std::memory_order_acq_rel) + 1 == nchan) {
        pthread_mutex_lock(&mu);
        pthread_cond_broadcast(&cv);
        pthread_mutex_unlock(&mu);
    }
}

void RefWalk::run_channel(int i)
{
    CodeCache codes;
    std::vector<gpsiq_patch_t> mine;
    // carr_in / prn_in: this timeline goes on where another call stopped: the slot's accumulator and satellite after that
    // call's last block
    double carr = carr_in[i];
    int prev = prn_in[i];
    size_t k = 0;
    while (k < ends.size() && ends[k] == 0) finish_piece(k++);          // empty pieces in front
    for (int b = 0; b < nblocks; ++b) {
        gpsiq_chan_t d = ch[(size_t) b * nchan + i];
        gpsiq_qchan_t &qq = q[(size_t) b * nchan + i];
        if (d.prn <= 0) {
            prev = 0; carr = 0.0;
            (void) quantize_one(d, delt, nsamp, nullptr, &qq, nullptr);  // an unused slot: zeroes
        } else {
            // the carrier chain (gps.c:2821 carries chan[i].carr_phase from block to block; allocateChannel re-initialises
            // it when the slot gets another satellite, gps.c:2208-2214)
            if ((b == 0 && !have_in) || prev != d.prn) carr = d.carr_phase;
            const double start = carr;
            prev = d.prn;
            // a start of exactly 1.0 (a wrap of the block before that rounded up to one) is phase 0 of the closed form
            // (mod 1); the walk goes on from 1.0 as the reference does and sample 0, where the reference indexes its table
            // at 512, is patched
            d.carr_phase = start == 1.0 ? 0.0 : start;
            const int qrc = quantize_one(d, delt, nsamp, nullptr, &qq, nullptr);
            d.carr_phase = start;
            if (qrc != GPSIQ_OK) {
                pthread_mutex_lock(&mu);
                if (rc == GPSIQ_OK) { rc = qrc; std::snprintf(err, sizeof err, "block %d: %.280s", b, gpsiq_last_error()); }
                pthread_mutex_unlock(&mu);
            } else {
                int slot = 0;                                            // device order: active channels first (gpsiq_set_descriptors)
                for (int j = 0; j < i; ++j) slot += ch[(size_t) b * nchan + j].prn > 0;
                mine.clear();
                carr = evaluate_block(d, qq, delt, nsamp, b, slot, &codes, &mine);
                if (!mine.empty()) {
                    pthread_mutex_lock(&pmu[i]);
                    patches[i].insert(patches[i].end(), mine.begin(), mine.end());
                    pthread_mutex_unlock(&pmu[i]);
                }
            }
        }
        while (k < ends.size() && ends[k] == b + 1) finish_piece(k++);
    }
    carr_end[i] = carr; last_prn[i] = prev;
}

void RefWalk::run()
{
    // a block or two (the drop-in block call): 16 channels x ~6 us here cost less than waking 15 pool threads
    parallel_for(nchan, nblocks <= 2 ? 1 : nchan, 1, [](void *p, int i0, int i1) {
        RefWalk &w = *static_cast<RefWalk *>(p);
        for (int i = i0; i < i1; ++i) w.run_channel(i);
    }, this);
}

int RefWalk::wait_piece(size_t k)
{
    pthread_mutex_lock(&mu);
    while (done[k].load(std::memory_order_acquire) < nchan) pthread_cond_wait(&cv, &mu);
    const int r = rc;
    pthread_mutex_unlock(&mu);
    return r;
}

void RefWalk::take_patches(size_t k, std::vector<gpsiq_patch_t> *out, bool relative)
{
    const uint32_t b0 = k ? (uint32_t) ends[k - 1] : 0u, b1 = (uint32_t) ends[k];
    out->clear();
    for (int i = 0; i < nchan; ++i) {
        pthread_mutex_lock(&pmu[i]);
        size_t t = taken[i];
        while (t < patches[i].size() && patches[i][t].block < b1) out->push_back(patches[i][t++]);
        taken[i] = t;
        pthread_mutex_unlock(&pmu[i]);
    }
    std::sort(out->begin(), out->end(), [](const gpsiq_patch_t &a, const gpsiq_patch_t &b) {
        if (a.block != b.block) return a.block < b.block;
        if (a.sample != b.sample) return a.sample < b.sample;
        return a.slot < b.slot;
    });
    if (relative)
        for (gpsiq_patch_t &p : *out) p.block -= b0;
}

int reference_timeline(const gpsiq_chan_t *ch, int nblocks, int nchan, double delt, int nsamp,
                       gpsiq_qchan_t *q, std::vector<gpsiq_patch_t> *patches, double *carr_end, int *last_prn,
                       const double *carr_in, const int *prn_in)
{
    RefWalk w(ch, nblocks, nchan, delt, nsamp, q, carr_in, prn_in, std::vector<int>());
    w.run();
    if (w.rc != GPSIQ_OK) return fail(w.rc, "%s", w.err);
    w.take_patches(0, patches, false);
    for (int i = 0; i < nchan; ++i) {
        if (carr_end) carr_end[i] = w.carr_end[i];
        if (last_prn) last_prn[i] = w.last_prn[i];
    }
    return GPSIQ_OK;
}

}  // namespace gpsiq

using namespace gpsiq;

extern "C" int gpsiq_reference_batch(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp,
                                     gpsiq_qchan_t *q, gpsiq_patch_t *patches, int max_patches, int *npatches,
                                     double *carr_phase_out)
{
    if ((!ch || !q) && nblocks) return fail(GPSIQ_E_ARG, "null descriptor pointer");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0) || max_patches < 0 || !npatches || (max_patches && !patches))
        return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g / patch buffer", nsamp, fs);
    std::vector<gpsiq_patch_t> v;
    int rc = reference_timeline(ch, nblocks, nchan, 1.0 / fs, nsamp, q, &v, carr_phase_out, nullptr, nullptr, nullptr);
    if (rc) return rc;
    *npatches = (int) v.size();
    if (v.size() > (size_t) max_patches)
        return fail(GPSIQ_E_RANGE, "%zu patches, room for %d", v.size(), max_patches);
    if (!v.empty()) std::memcpy(patches, v.data(), v.size() * sizeof(gpsiq_patch_t));
    return GPSIQ_OK;
}
