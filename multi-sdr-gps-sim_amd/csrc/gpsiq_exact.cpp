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
//   * At every candidate the double path's LUT index / chip is decided from the block's START state by a rigorous enclosure of
//     the accumulated rounding (Drift: 99.5+ % of the candidates), else by walking the accumulator (NcoWalk); where (LUT index,
//     sign) differ from the closed form a patch {block, sample, slot, index, sign} is emitted.  The device runs the fixed-point
//     kernels unchanged and then recomputes the patched samples (apply_patches in gpsiq_kernels.hip).
// Only the carrier chain is serial (block k of a channel starts where the double accumulator left block k-1): NcoWalk walks it
// from wrap to wrap through a per-block table of its dozen distinct cycles (built eight cycles at a time where the host has
// AVX-512), and RefWalk runs chains and evaluations as tasks on the shared thread pool, piece by piece, so that devices render
// behind them; gpsiq_reference_chain / gpsiq_reference_seeded export the two halves for sharding over processes.
#include "gpsiq_internal.h"
#include "gpsiq_walk.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace gpsiq {

namespace {

typedef unsigned __int128 u128;

static_assert(kCaSeqLen == GPSIQ_CA_SEQ_LEN, "gpsiq_walk.h");

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

// Smallest x in [0, bound) with L <= (A*x mod M) <= R, for 0 <= L <= R < M and 0 <= A < M; kNone if there is none.
// The bound goes down the descent with the problem (x < bound needs y <= (A*(bound-1) - L) / M wraps of M), so a search
// that cannot succeed inside the block stops after ~log2(bound) levels instead of ~log(M).
constexpr u128 kNone = ~(u128) 0;
u128 first_in_range(uint64_t A, uint64_t M, uint64_t L, uint64_t R, u128 bound)
{
    if (bound == 0) return kNone;
    if (L == 0) return 0;
    if (A == 0) return kNone;
    if (A > M - A) {                 // 2A > M: count downwards instead
        A = M - A;
        const uint64_t l = M - R, r = M - L;
        L = l; R = r;
    }
    const uint64_t lq = L / A, lr = L - lq * A;                   // one division serves the first multiple and L mod A
    const uint64_t k = lq + (lr != 0);
    if ((u128) k * A <= R) return k < bound ? (u128) k : kNone;
    // no multiple of A inside [L, R]: after y wraps of M the window is at M*y + [L, R]; it holds a
    // multiple of A iff ((-M mod A) * y) mod A lies in [L mod A, R mod A]
    const u128 reach = (u128) A * (bound - 1);                    // A * x for the largest x allowed
    if (reach < L) return kNone;
    // the wraps that can matter, a little generously (a double quotient instead of a 128-bit division; the result is checked)
    const u128 ybound = (u128) ((double) (reach - L) / (double) M * (1.0 + 0x1p-40)) + 2;
    const uint64_t mr = M % A;
    // no multiple of A lies in [L, R] here (k * A > R and lr != 0): L and R have the same quotient, so R mod A needs no division
    const u128 y = first_in_range(mr ? A - mr : 0, A, lr, lr + (R - L), ybound);
    if (y == kNone) return kNone;
    const u128 x = ((u128) M * y + L + A - 1) / A;
    return x < bound ? x : kNone;
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
            const u128 x = first_in_range(b, M, M - cur, M - cur + W - 1, (u128) (nsamp - base));
            if (x == kNone) break;
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
struct NcoWalk : WalkCore<64> {

#if defined(__x86_64__)
    // ---- eight carrier cycles walked at once (AVX-512) -------------------------------------------------------------------
    // A walked cycle is ~10 dependent binade steps, and a block needs a dozen of them before its wrap-to-wrap table is
    // complete.  Discovered one at a time as the chain reaches them they are a third of the chain's time; but a cycle depends
    // only on its start state, every lane goes through the same binades in the same order, and the per-binade piece
    // (step, run length) is the same for all of them: eight start states take the vector form of climb<true> / descend<true>
    // -- the same additions, the same slack notes, lane for lane -- in about the time of one.  Anything that is not the plain
    // case in some lane (a run that needs the division, a state of exactly 1.0, a tie on the wrap, a lane that leaves the
    // binade sequence) marks that lane not-ok; such a start state is walked by the scalar code when the chain gets there.
    struct Batch { int64_t m[8], lo[8], hi[8], m2[8], steps[8]; bool ok[8]; };

#define GPSIQ_AVX512 __attribute__((target("avx512f,avx512dq")))
#define GPSIQ_NOTE(v, act)                                                                                               \
    do {                                                                                                                 \
        const __m512i b_ = _mm512_castpd_si512(v);                                                                       \
        const __m512i e_ = _mm512_srli_epi64(b_, 52);                                                                    \
        const __m512i sh_ = _mm512_sub_epi64(uexp_v, e_);                                                                \
        const __mmask8 bad_ = _mm512_cmplt_epi64_mask(sh_, zero) | _mm512_cmpgt_epi64_mask(sh_, sixty2) | _mm512_cmpeq_epi64_mask(e_, zero); \
        okm &= (__mmask8) ~(bad_ & (act));                                                                               \
        const __m512i mx_ = _mm512_or_si512(_mm512_and_si512(b_, mant), one52);                                          \
        const __m512i l_ = _mm512_sub_epi64(two, _mm512_srlv_epi64(_mm512_sub_epi64(mx_, one52), sh_));                  \
        const __m512i h_ = _mm512_sub_epi64(_mm512_srlv_epi64(_mm512_sub_epi64(one53, mx_), sh_), two);                  \
        lo = _mm512_mask_max_epi64(lo, (act), lo, l_);                                                                   \
        hi = _mm512_mask_min_epi64(hi, (act), hi, h_);                                                                   \
    } while (0)

    // One table binade for all lanes (run + the addition that leaves it): x in binade ec + s on entry.  kDown: negative addend.
    // Returns with x = the last value inside the binade (after the run), n advanced by the run; the caller adds c.
    template <bool kDown>
    GPSIQ_AVX512 inline void batch_level(int s, __m512d &x, __m512i &n, __m512i &lo, __m512i &hi, __mmask8 &okm, const __m512i uexp_v) const
    {
        const __m512i zero = _mm512_setzero_si512(), two = _mm512_set1_epi64(2), sixty2 = _mm512_set1_epi64(62);
        const __m512i mant = _mm512_set1_epi64((int64_t) kMant), one52 = _mm512_set1_epi64((int64_t) 1 << 52), one53 = _mm512_set1_epi64((int64_t) 1 << 53);
        const Piece &p = T[s];
        const __m512d cv = _mm512_set1_pd(c);
        __m512i b = _mm512_castpd_si512(x);
        okm &= _mm512_cmpeq_epi64_mask(_mm512_srli_epi64(b, 52), _mm512_set1_epi64(ec + s));          // still on the common path
        __m512i mx = _mm512_or_si512(_mm512_and_si512(b, mant), one52);
        if (p.tie) {                                                   // an odd mantissa in a tie binade: one real addition first
            const __mmask8 odd = _mm512_test_epi64_mask(mx, _mm512_set1_epi64(1));
            if (odd) {
                x = _mm512_mask_add_pd(x, odd, x, cv);
                n = _mm512_mask_add_epi64(n, odd, n, _mm512_set1_epi64(1));
                GPSIQ_NOTE(x, odd);
                b = _mm512_castpd_si512(x);
                okm &= _mm512_cmpeq_epi64_mask(_mm512_srli_epi64(b, 52), _mm512_set1_epi64(ec + s));
                mx = _mm512_or_si512(_mm512_and_si512(b, mant), one52);
            }
        }
        const __m512i off = kDown ? _mm512_sub_epi64(_mm512_set1_epi64(((int64_t) 1 << 53) - 1), mx) : _mm512_sub_epi64(mx, one52);
        const __mmask8 c1 = _mm512_cmple_epi64_mask(off, _mm512_set1_epi64(p.rem));
        const __mmask8 c2 = (__mmask8) (~c1 & _mm512_cmple_epi64_mask(off, _mm512_set1_epi64(p.rem + p.dm)));
        const __mmask8 c3 = _mm512_cmpgt_epi64_mask(off, _mm512_set1_epi64(p.span));
        okm &= (__mmask8) (c1 | c2 | c3);                              // the run that needs a division: scalar code
        __m512i run = _mm512_maskz_mov_epi64(c1, _mm512_set1_epi64(p.k));
        run = _mm512_mask_mov_epi64(run, c2, _mm512_set1_epi64(p.k - 1));
        __m512i moved = _mm512_maskz_mov_epi64(c1, _mm512_set1_epi64(p.kdm));
        moved = _mm512_mask_mov_epi64(moved, c2, _mm512_set1_epi64(p.kdm - p.dm));
        const __m512i mx2 = kDown ? _mm512_sub_epi64(mx, moved) : _mm512_add_epi64(mx, moved);
        x = _mm512_castsi512_pd(_mm512_or_si512(_mm512_andnot_si512(mant, b), _mm512_and_si512(mx2, mant)));
        n = _mm512_add_epi64(n, run);
        const __mmask8 ran = _mm512_cmpgt_epi64_mask(run, zero);
        GPSIQ_NOTE(x, ran);
    }

#define GPSIQ_VCONSTS(UEXP)                                                                                               \
    const __m512i zero = _mm512_setzero_si512(), two = _mm512_set1_epi64(2), sixty2 = _mm512_set1_epi64(62);              \
    const __m512i mant = _mm512_set1_epi64((int64_t) kMant), one52 = _mm512_set1_epi64((int64_t) 1 << 52), one53 = _mm512_set1_epi64((int64_t) 1 << 53); \
    const __m512i uexp_v = _mm512_set1_epi64(UEXP);                                                                       \
    const __m512d cv = _mm512_set1_pd(c), thr_v = _mm512_set1_pd(thr), one = _mm512_set1_pd(1.0);                         \
    (void) zero; (void) two; (void) sixty2; (void) mant; (void) one52; (void) one53; (void) uexp_v; (void) cv; (void) thr_v; (void) one

    // positive addend, the binades below the table: plain additions (at most 2^(kLow + 1) + 1 of them)
    GPSIQ_AVX512 inline void batch_low_up(__m512d &x, __m512i &n, __m512i &lo, __m512i &hi, __mmask8 &okm) const
    {
        GPSIQ_VCONSTS(1023);
        for (int it = 0; it < (2 << kLow) + 2; ++it) {
            const __mmask8 act = _mm512_cmp_pd_mask(x, thr_v, _CMP_LT_OQ);
            if (!act) break;
            x = _mm512_mask_add_pd(x, act, x, cv);
            n = _mm512_mask_add_epi64(n, act, n, _mm512_set1_epi64(1));
            GPSIQ_NOTE(x, act);
        }
        okm &= (__mmask8) ~_mm512_cmp_pd_mask(x, thr_v, _CMP_LT_OQ);
    }

    // positive addend, table binade ec + s: the run and the addition that leaves it.  true: that was the wrap (s == top), x is
    // the state after it
    GPSIQ_AVX512 inline bool batch_level_up(int s, int top, __m512d &x, __m512i &n, __m512i &lo, __m512i &hi, __mmask8 &okm) const
    {
        GPSIQ_VCONSTS(1023);
        batch_level<false>(s, x, n, lo, hi, okm, uexp_v);
        const __m512d y = _mm512_add_pd(x, cv);                        // leaves the binade; at the top: wraps
        n = _mm512_add_epi64(n, _mm512_set1_epi64(1));
        if (s < top) { x = y; GPSIQ_NOTE(x, (__mmask8) 0xff); return false; }
        okm &= _mm512_cmp_pd_mask(y, one, _CMP_GE_OQ);
        GPSIQ_NOTE(y, (__mmask8) 0xff);
        const __m512d bb = _mm512_sub_pd(y, x);
        const __m512d err = _mm512_add_pd(_mm512_sub_pd(x, _mm512_sub_pd(y, bb)), _mm512_sub_pd(cv, bb));    // the rounding error of x + c, exactly
        okm &= (__mmask8) ~_mm512_cmp_pd_mask(_mm512_abs_pd(err), _mm512_set1_pd(0x1p-53), _CMP_EQ_OQ);      // a tie on the grid the wrap is taken on
        x = _mm512_sub_pd(y, one);
        return true;
    }

    // negative addend, table binade ec + s: the run and the addition into the binade underneath
    GPSIQ_AVX512 inline void batch_level_down(int s, __m512d &x, __m512i &n, __m512i &lo, __m512i &hi, __mmask8 &okm) const
    {
        GPSIQ_VCONSTS(1022);
        batch_level<true>(s, x, n, lo, hi, okm, uexp_v);
        x = _mm512_add_pd(x, cv);
        n = _mm512_add_epi64(n, _mm512_set1_epi64(1));
        GPSIQ_NOTE(x, (__mmask8) 0xff);
    }

    // negative addend, below the table: plain additions until the sum turns negative, then + 1.0; x: the state after the wrap
    GPSIQ_AVX512 inline void batch_tail_down(__m512d &x, __m512i &n, __m512i &lo, __m512i &hi, __mmask8 &okm) const
    {
        GPSIQ_VCONSTS(1022);
        const __m512d zd = _mm512_setzero_pd();
        okm &= _mm512_cmp_pd_mask(x, thr_v, _CMP_LT_OQ);
        const __m512d c2 = _mm512_mul_pd(cv, _mm512_set1_pd(-2.0));
        __mmask8 open = 0xff;
        __m512d r_end = zd;
        for (int it = 0; it < (4 << kLow) + 4 && open; ++it) {
            const __m512d y = _mm512_add_pd(x, cv);
            n = _mm512_mask_add_epi64(n, open, n, _mm512_set1_epi64(1));
            const __mmask8 ng = (__mmask8) (open & _mm512_cmp_pd_mask(y, zd, _CMP_LT_OQ)), ps = (__mmask8) (open & ~ng);
            if (ng) {
                const __m512d r = _mm512_add_pd(y, one);
                // y's own rounding: exact when x lies in the addend's binade (only its sign has to hold), else noted
                const __mmask8 same = (__mmask8) (ng & _mm512_cmpeq_epi64_mask(_mm512_srli_epi64(_mm512_castpd_si512(x), 52), _mm512_set1_epi64(ec)));
                const __m512i h = _mm512_sub_epi64(_mm512_cvttpd_epi64(_mm512_mul_pd(_mm512_sub_pd(zd, y), _mm512_set1_pd(0x1p53))), two);
                hi = _mm512_mask_min_epi64(hi, same, hi, h);
                const __m512d ny = _mm512_sub_pd(zd, y);
                GPSIQ_NOTE(ny, (__mmask8) (ng & ~same));
                const __m512d bb = _mm512_sub_pd(r, y);
                const __m512d err = _mm512_add_pd(_mm512_sub_pd(y, _mm512_sub_pd(r, bb)), _mm512_sub_pd(one, bb));
                const __mmask8 badr = (__mmask8) (_mm512_cmp_pd_mask(r, one, _CMP_GE_OQ) | _mm512_cmp_pd_mask(_mm512_abs_pd(err), _mm512_set1_pd(0x1p-54), _CMP_EQ_OQ));
                okm &= (__mmask8) ~(badr & ng);
                GPSIQ_NOTE(r, (__mmask8) (ng & ~badr));
                r_end = _mm512_mask_mov_pd(r_end, ng, r);
                open &= (__mmask8) ~ng;
            }
            if (ps) {
                // x <= 2|c|: x + c is exact (Sterbenz) for this x and every translated one; it only has to stay non-negative
                const __mmask8 st = (__mmask8) (ps & _mm512_cmp_pd_mask(x, c2, _CMP_LE_OQ));
                const __m512i l = _mm512_sub_epi64(two, _mm512_cvttpd_epi64(_mm512_mul_pd(y, _mm512_set1_pd(0x1p53))));
                lo = _mm512_mask_max_epi64(lo, st, lo, l);
                GPSIQ_NOTE(y, (__mmask8) (ps & ~st));
                x = _mm512_mask_mov_pd(x, ps, y);
            }
        }
        okm &= (__mmask8) ~open;
        x = r_end;
    }

    // positive addend: climb<true> for eight post-wrap states m (units of 2^-52)
    GPSIQ_AVX512 void walk8_up(int64_t m_max, Batch *io) const
    {
        const __m512i m = _mm512_loadu_si512(io->m);
        __m512d x = _mm512_mul_pd(_mm512_cvtepi64_pd(m), _mm512_set1_pd(0x1p-52));
        __m512i n = _mm512_setzero_si512(), lo = _mm512_sub_epi64(_mm512_setzero_si512(), m), hi = _mm512_sub_epi64(_mm512_set1_epi64(m_max), m);
        __mmask8 okm = 0xff;
        batch_low_up(x, n, lo, hi, okm);
        const int top = (int) (top_exp - ec);
        for (int s = kLow + 1; s <= top; ++s)
            if (batch_level_up(s, top, x, n, lo, hi, okm)) break;
        okm &= _mm512_cmp_pd_mask(x, _mm512_set1_pd(1.0), _CMP_LT_OQ);
        const __m512i m2 = _mm512_cvttpd_epi64(_mm512_mul_pd(x, _mm512_set1_pd(0x1p52)));
        _mm512_storeu_si512(io->lo, lo); _mm512_storeu_si512(io->hi, hi); _mm512_storeu_si512(io->m2, m2); _mm512_storeu_si512(io->steps, n);
        for (int q = 0; q < 8; ++q) io->ok[q] = (okm >> q) & 1;
    }

    // negative addend: descend<true> for eight post-wrap states m (units of 2^-53)
    GPSIQ_AVX512 void walk8_down(int64_t m_max, Batch *io) const
    {
        const __m512i m = _mm512_loadu_si512(io->m);
        __m512d x = _mm512_mul_pd(_mm512_cvtepi64_pd(m), _mm512_set1_pd(0x1p-53));
        __m512i n = _mm512_setzero_si512(), lo = _mm512_sub_epi64(_mm512_setzero_si512(), m), hi = _mm512_sub_epi64(_mm512_set1_epi64(m_max), m);
        __mmask8 okm = (__mmask8) (0xff & _mm512_cmp_pd_mask(x, _mm512_set1_pd(1.0), _CMP_LT_OQ));        // a state of exactly 1.0: scalar code
        const int top = (int) (top_exp - ec);
        for (int s = top; s > kLow; --s) batch_level_down(s, x, n, lo, hi, okm);
        batch_tail_down(x, n, lo, hi, okm);
        const __m512i m2 = _mm512_cvttpd_epi64(_mm512_mul_pd(x, _mm512_set1_pd(0x1p53)));
        _mm512_storeu_si512(io->lo, lo); _mm512_storeu_si512(io->hi, hi); _mm512_storeu_si512(io->m2, m2); _mm512_storeu_si512(io->steps, n);
        for (int q = 0; q < 8; ++q) io->ok[q] = (okm >> q) & 1;
    }
    // the buckets [k0, k1] of one entry, eight per store
    GPSIQ_AVX512 static void fill8(int64_t *inc, uint64_t *tag, int64_t k0, int64_t k1, int64_t vinc, uint64_t vtag)
    {
        const __m512i vi = _mm512_set1_epi64(vinc), vt = _mm512_set1_epi64((int64_t) vtag);
        int64_t k = k0;
        for (; k + 7 <= k1; k += 8) { _mm512_storeu_si512(inc + k, vi); _mm512_storeu_si512(tag + k, vt); }
        for (; k <= k1; ++k) { inc[k] = vinc; tag[k] = vtag; }
    }
#undef GPSIQ_NOTE
#undef GPSIQ_VCONSTS
#endif   // __x86_64__

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
#ifndef GPSIQ_WALK_BUCKETS
#define GPSIQ_WALK_BUCKETS 1024
#endif
    static constexpr int kMaxEntries = 64, kBuckets = GPSIQ_WALK_BUCKETS;     // scripts/ubench_walk.cpp A/Bs the bucket count
    // The buckets live per thread and are never cleared: a bucket belongs to the table of the run (= block) whose stamp it
    // carries (clearing 1024 of them per block cost as much as a hundred look-ups).  Increment and sample count sit in two
    // arrays, so that the chain state -> shift -> load -> add that every cycle waits for is those three instructions and
    // nothing else; stamp and count are read beside it.
    struct Buckets {
        int64_t  inc[kBuckets];
        uint64_t tag[kBuckets];              // (stamp << 32) | samples of the cycle
        uint32_t stamp = 0;
        Buckets() { for (int i = 0; i < kBuckets; ++i) tag[i] = 0; }
        uint32_t next_stamp()
        {
            if (++stamp == 0) { for (int i = 0; i < kBuckets; ++i) tag[i] = 0; stamp = 1; }
            return stamp;
        }
    };

    // A function of its own (and not inlined): in the middle of run() the compiler keeps the state in memory, and the store
    // forwarding on it would cost more than the look-up itself.  rel = state - base.
    __attribute__((noinline)) static void hits(const Buckets *bk, uint32_t stamp, int64_t W, int bshift, long limit,
                                               int64_t *rel_io, long *n_io, long *wraps_io)
    {
        uint64_t rel = (uint64_t) *rel_io;
        long n = *n_io, wraps = *wraps_io;
        for (;;) {
            if (rel >= (uint64_t) W) break;
            const uint64_t idx = rel >> bshift, tag = bk->tag[idx];
            const long steps = (long) (uint32_t) tag;
            if ((uint32_t) (tag >> 32) != stamp || n + steps > limit) break;
            rel += (uint64_t) bk->inc[idx]; n += steps; ++wraps;
        }
        *rel_io = (int64_t) rel; *n_io = n; *wraps_io = wraps;
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
        const double span = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
        const bool use_map = std::fabs(c) * (double) ns > 24.0 * span;
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
        const double scale = from_bits((uint64_t) (1023 + 1075 - uexp) << 52), unit = from_bits((uint64_t) (1023 + uexp - 1075) << 52);   // 2^(1075 - uexp) and its inverse
        const int64_t m_max = kind == 0 ? ((int64_t) GPSIQ_CA_SEQ_LEN << 43) - 1 : neg ? ((int64_t) 1 << 53) - 1 : ((int64_t) 1 << 52) - 1;
        // post-wrap states: [0, c] (positive addend) or [1 + c, 1) (negative), W units wide
        const int64_t W = (int64_t) (std::fabs(c) * scale) + 4;
        const int64_t base = neg ? ((int64_t) 1 << 53) - W : 0;
        int bshift = 0;
        while ((W >> bshift) >= kBuckets) ++bshift;
        Entry tab[kMaxEntries];
        static thread_local Buckets bkt;
        const uint32_t stamp = bkt.next_stamp();
        int ntab = 0;
        long min_steps = ns;                                          // shortest cycle seen
        int64_t m = (int64_t) (x * scale);                            // exact: the state is a multiple of the unit
#if defined(__x86_64__)
        static const bool wide = __builtin_cpu_supports("avx512dq") && __builtin_cpu_supports("avx512f");
#endif
        // an entry for the cycle walked from state ms to state m2 in ne samples, valid for start states ms + [lo, hi]
        auto add_entry = [&](int64_t ms, int64_t m2, long ne, int64_t elo, int64_t ehi) {
            Entry &t = tab[ntab++];
            t.first = ms + elo; t.last = ms + ehi; t.inc = m2 - ms; t.steps = ne;
            if (ne < min_steps) min_steps = ne;
            // buckets wholly inside [first, last]
            const int64_t f = t.first - base, l = t.last - base;
            int64_t k0 = f <= 0 ? 0 : ((f - 1) >> bshift) + 1;        // first bucket starting at or after `first`
            int64_t k1 = l >= W ? kBuckets - 1 : ((l + 1) >> bshift) - 1;  // last bucket ending at or before `last`
            if (k1 > kBuckets - 1) k1 = kBuckets - 1;
            if (ne <= 0xffffffffL) {
                const uint64_t tag = ((uint64_t) stamp << 32) | (uint64_t) ne;
#if defined(__x86_64__)
                if (wide) { fill8(bkt.inc, bkt.tag, k0, k1, t.inc, tag); return; }
#endif
                for (int64_t k = k0; k <= k1; ++k) { bkt.inc[k] = t.inc; bkt.tag[k] = tag; }
            }
        };
#if defined(__x86_64__)
        // The table built ahead of the chain, eight cycles at a time (walk8_up / walk8_down).  The post-wrap states of a block
        // fill their range [base, base + W) evenly (a rotation by an irrational step), so with enough cycles in the block every
        // entry will be needed -- and which start states to walk is known without the chain: first eight states spread over the
        // range, then the first uncovered state of every gap those eight entries leave, then once more.  What is still
        // uncovered after that (slivers; cycles that are not the plain case) is found on demand below, as before.
        if (wide && kind == 1 && std::fabs(c) * (double) ns > 64.0 * span) {
            const int64_t d_lo = base < 0 ? 0 : base, d_hi = base + W - 1 > m_max ? m_max : base + W - 1;      // the range, inside the accumulator's
            const double range = (double) (d_hi - d_lo);
            Batch bt;
            auto take = [&]() {                                        // the lanes of a round that are new, valid entries
                if (neg) walk8_down(m_max, &bt); else walk8_up(m_max, &bt);
                for (int q = 0; q < 8 && ntab < kMaxEntries; ++q) {
                    if (!(bt.ok[q] && bt.lo[q] <= 0 && bt.hi[q] >= 0 && bt.m2[q] >= 0 && bt.m2[q] <= m_max && bt.steps[q] > 0)) continue;
                    bool dup = false;                                  // a lane that repeats a start state, or fell into an entry already there
                    for (int i = 0; i < ntab && !dup; ++i) dup = tab[i].first <= bt.m[q] && bt.m[q] <= tab[i].last;
                    if (!dup) add_entry(bt.m[q], bt.m2[q], (long) bt.steps[q], bt.lo[q], bt.hi[q]);
                }
            };
            // Where the entries of the table this thread built last lay, if that was for nearly the same addend (a thread walks the
            // blocks of one channel one after the other; the Doppler moves by a fraction of a hertz per block, and the entries with
            // it): one lane per entry, two rounds, no search.  Else: eight states spread over the range, then twice the gaps the
            // entries leave, lanes by gap width (a gap's first uncovered state, and states spread over its inside when it gets
            // more than one lane: an entry reaches down from its start state as well as up).
            struct Hint { double c; int n; double at[16]; };
            static thread_local Hint hint = {0.0, 0, {}};
            if (hint.n >= 6 && (hint.c < 0.0) == neg && std::fabs(c - hint.c) <= std::fabs(c) * 0x1p-10) {
                for (int r0 = 0; r0 < hint.n; r0 += 8) {
                    for (int q = 0; q < 8; ++q) bt.m[q] = d_lo + (int64_t) (range * hint.at[r0 + q < hint.n ? r0 + q : r0]);
                    take();
                }
            } else {
                for (int q = 0; q < 8; ++q) bt.m[q] = d_lo + ((d_hi - d_lo) * (2 * q + 1)) / 16;
                for (int round = 0; round < 3 && ntab <= kMaxEntries - 8; ++round) {
                    take();
                    int order[kMaxEntries];
                    for (int i = 0; i < ntab; ++i) {                   // insertion sort by `first`
                        int j = i;
                        for (; j > 0 && tab[order[j - 1]].first > tab[i].first; --j) order[j] = order[j - 1];
                        order[j] = i;
                    }
                    int64_t gs[kMaxEntries + 1], gl[kMaxEntries + 1];
                    int ng = 0;
                    int64_t at = d_lo, open = 0;
                    for (int i = 0; i < ntab; ++i) {
                        const Entry &e = tab[order[i]];
                        if (e.first > at) { gs[ng] = at; gl[ng] = e.first - at; open += gl[ng]; ++ng; }
                        if (e.last + 1 > at) at = e.last + 1;
                    }
                    if (at <= d_hi) { gs[ng] = at; gl[ng] = d_hi - at + 1; open += gl[ng]; ++ng; }
                    if (ng == 0 || open * 64 < (d_hi - d_lo)) break;    // covered but for slivers
                    int q = 0;
                    while (q < 8) {
                        int w = -1;
                        for (int g = 0; g < ng; ++g) if (gl[g] > 0 && (w < 0 || gl[g] > gl[w])) w = g;
                        if (w < 0) break;
                        int k = (int) ((8 * gl[w] + open / 2) / open);
                        if (k < 1) k = 1;
                        if (k > 8 - q) k = 8 - q;
                        for (int i = 0; i < k; ++i) bt.m[q++] = gs[w] + (gl[w] * i) / k;
                        gl[w] = 0;
                    }
                    for (; q < 8; ++q) bt.m[q] = bt.m[0];                // fewer gaps than lanes: repeats (dropped as duplicates)
                }
            }
            // for the next block of this thread: the middle of every entry, as a fraction of the range
            hint.n = ntab < 16 ? ntab : 16;
            hint.c = c;
            const double inv = 1.0 / range;
            for (int i = 0; i < hint.n; ++i) {
                const int64_t lo_i = tab[i].first < d_lo ? d_lo : tab[i].first, hi_i = tab[i].last > d_hi ? d_hi : tab[i].last;
                hint.at[i] = ((double) (lo_i - d_lo) + (double) (hi_i - lo_i) * 0.5) * inv;
            }
        }
#endif
        for (;;) {
            // the common case, cycle after cycle: the state's bucket lies wholly inside one entry and names its increment.
            // Stops before a cycle that would pass the end of the block or hold the next target.
            int64_t rel = m - base;
            hits(&bkt, stamp, W, bshift, tk < nt && targets[tk] < ns ? targets[tk] : ns, &rel, &n, &wraps);
            m = rel + base;
            if (n >= ns) return (double) m * unit;
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
            if (sl.ok && sl.lo <= 0 && sl.hi >= 0 && x < wrap && m2 <= m_max) add_entry(m, m2, ne, sl.lo, sl.hi);
            m = m2;
        }
        // the last, partial cycle
        x = (double) m * unit;
        visit(x, n, ns);
        while (n < ns && cycle(x, n, ns)) {}
        return x;
    }
};

// The double accumulator at sample n WITHOUT walking it: a rigorous enclosure.
//
// Every addition of the reference loop is the exact sum plus a rounding error e_j, so the unwrapped phase after n samples is
// U_n = x_0 + n*c + d_n with d_n = e_0 + ... + e_{n-1} (wraps are exact; the carrier's y + 1.0 is one more rounding).  While
// state and sum lie in one binade b (ulp u_b) the state is a multiple of u_b and the sum is rounded on that grid, so the
// addition adds S_b = rnd(|c|/u_b)*u_b and e_j = +-(S_b - |c|) =: +-eps_b, a constant of the binade (ties to even: constant
// from the second addition inside the binade on).  Let G be the piecewise linear function on [0, L] with slope eps_b/S_b
// in binade b (0 below the lowest binade counted), continued over the wraps as G^(U) = floor(U/L)*G(L) + G(U mod L).  A steady
// step moves the phase by +-S_b and G^ by +-eps_b = e_j exactly, so the steady steps telescope:
//     d_n = G^(U_n) - G(x_0) + sum over the IRREGULAR steps j of (e_j - [G^(U_{j+1}) - G^(U_j)]),
// and the irregular steps of one cycle are few and each is bounded by the ulp of its binade: the step that enters a binade
// (rounded on the new, coarser grid from a state on the finer one: <= 1.001 u_b; twice that where |c|/u_b ends in exactly one
// half and the first addition from an odd mantissa rounds the other way), the wrap (carrier: the sum in [1, 2) or y + 1.0,
// <= 1.51 u_top; code: an ordinary addition in [512, 1024) followed by an exact subtraction, 0.51 u_top for the piece of G it
// skips) and the handful of additions below 8|c| (G is flat there; <= 3 u of the lowest binade counted).  Since G^ is
// Lipschitz with a constant of ~2^-44, U_n on the right may be replaced by R_n = x_0 + n*c (exact integer arithmetic in
// units of c's ulp) at a cost far below one unit.  The enclosure is ~0.3 % as wide as the a-priori window n * 2^-54 the
// candidates are found with (tests/test_reference_nco_host.py holds it against the walked accumulator on adversarial
// addends: ties, binade-edge starts; observed use of the half-width <= 0.6), so nearly every candidate is decided here, from
// the block's start state alone -- which is what lets the blocks of a timeline be evaluated on any thread, device or
// process once the serial carrier chain has given their start states.
struct Drift {
    typedef __int128 i128;
    bool    valid = false;
    int     kind = 1;
    bool    neg = false;
    double  L = 1.0;
    int64_t ec = 0, mc = 0;           // |c| = mc * 2^(ec - 1075)
    int     b_lo = 0, b_top = 0;      // biased exponents of the binades that count
    double  ulp_c = 0.0;
    double  g[64], Gedge[65];         // slope in binade b_lo + i; G at its lower edge
    double  edge_lo = 0.0;            // lower edge of binade b_lo (G = 0 below)
    double  GL = 0.0, gmax = 0.0, Acyc = 0.0;
    int     cell_shift = 0;           // log2(cell / ulp_c): a cell is 1/512 cycle (carrier) or one chip (code)
    i128    Lint = 0;                 // L / ulp_c

    void setup(double c, int kind_)
    {
        kind = kind_;
        L = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
        const int top_exp = kind == 0 ? 1023 + 9 : 1022;
        neg = c < 0.0;
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        ec = (int64_t) (bc >> 52);
        mc = (int64_t) ((bc & kMant) | (kMant + 1));
        // the same addends NcoWalk's fast walk takes: at least 2^6 below the top binade, not absurdly small, the code phase only climbs
        valid = !(ec > top_exp - 6 || ec < top_exp - 40 || (kind == 0 && neg));
        if (!valid) return;
        ulp_c = from_bits((uint64_t) (ec - 52) << 52);
        b_lo = (int) ec + 3; b_top = top_exp;
        edge_lo = from_bits((uint64_t) b_lo << 52);
        double G = 0.0, allow = 0.0;
        gmax = 0.0;
        for (int b = b_lo; b <= b_top; ++b) {
            const int s = b - (int) ec;
            int64_t dm = mc >> s;
            const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
            bool tie = false;
            if (rem > half) ++dm;
            else if (rem == half) { tie = true; dm += dm & 1; }
            const double delta = (double) ((dm << s) - mc);                  // eps_b / ulp_c, exact
            const double Sb = (double) dm * (double) ((int64_t) 1 << s);     // S_b / ulp_c, exact
            const double gb = delta / Sb;
            const double lo = from_bits((uint64_t) b << 52);
            const double width = (kind == 0 && b == b_top) ? L - lo : lo;    // [512, 1023) for the code phase's top binade
            g[b - b_lo] = gb;
            Gedge[b - b_lo] = G;
            G += gb * width;
            if (std::fabs(gb) > gmax) gmax = std::fabs(gb);
            allow += 1.001 * from_bits((uint64_t) (b - 52) << 52) * (tie ? 2.0 : 1.0);
        }
        Gedge[b_top - b_lo + 1] = G;
        GL = G;
        const double u_top = from_bits((uint64_t) (b_top - 52) << 52), u_lo = from_bits((uint64_t) (b_lo - 52) << 52);
        Acyc = allow + 3.0 * u_lo + (kind == 0 ? 0.51 : 1.51) * u_top;
        cell_shift = kind == 0 ? (int) (1075 - ec) : (int) (1075 - ec - 9);
        Lint = kind == 0 ? (i128) GPSIQ_CA_SEQ_LEN << (1075 - ec) : (i128) 1 << (1075 - ec);
    }

    inline double Gof(double x) const            // 0 <= x <= L
    {
        if (x < edge_lo) return 0.0;
        if (x >= L) return GL;
        const int b = (int) (bits_of(x) >> 52);
        return Gedge[b - b_lo] + g[b - b_lo] * (x - from_bits((uint64_t) b << 52));
    }

    // x0 in units of ulp_c, cut below one unit
    inline i128 units(double x0) const
    {
        const uint64_t bx = bits_of(x0);
        const int64_t ex = (int64_t) (bx >> 52);
        if (ex == 0) return 0;
        const i128 mx = (i128) ((bx & kMant) | (kMant + 1));
        const int64_t sh = ex - ec;
        return sh >= 0 ? mx << sh : (sh > -64 ? mx >> -sh : (i128) 0);
    }

    // The cell (LUT step or chip, counted from phase 0 of the block's first cycle / code period, unwrapped) that holds the double
    // path's phase at sample n, when the enclosure lies inside one cell; false: undecided (a boundary inside the enclosure -- also
    // every phase the reference wraps to exactly 1.0 --, a start outside [0, L), an addend the fast walk does not take).
    bool cell_at(double x0, long n, i128 *cell) const
    {
        if (!valid || !(x0 >= 0.0 && x0 < L)) return false;
        const i128 R = neg ? units(x0) - (i128) n * mc : units(x0) + (i128) n * mc;     // x0 + n*c
        // floor(R / L) and the rest, without a 128-bit division: L = 2^sh (carrier) or 1023 * 2^sh units
        const int sh = (int) (1075 - ec);
        const int64_t hi_part = (int64_t) (R >> sh);                                    // floor(R / 2^sh): cycles, or chips (< 2^40)
        int64_t q;
        if (kind == 1) q = hi_part;
        else { q = hi_part / GPSIQ_CA_SEQ_LEN; if (hi_part % GPSIQ_CA_SEQ_LEN < 0) --q; }
        const i128 r = R - (i128) q * Lint;
        const double core = (double) q * GL + Gof((double) r * ulp_c) - Gof(x0);
        const double I = ((double) (q < 0 ? -q : q) + 3.0) * Acyc;
        const double eta = 4.0 * gmax * (std::fabs(core) + I) + 1e-12 * (std::fabs(core) + std::fabs((double) q * GL)) + 4.0 * gmax * L * 0x1p-52;
        const i128 lo = R + (i128) std::floor((core - I - eta) / ulp_c) - 2, hi = R + (i128) std::ceil((core + I + eta) / ulp_c) + 2;
        const i128 c0 = lo >> cell_shift, c1 = hi >> cell_shift;                          // arithmetic shifts: floor for negative phases too
        if (c0 != c1) return false;
        *cell = c0;
        return true;
    }
};

// process-wide counts (gpsiq_reference_stats): candidates seen, decided from the start state alone, walked (carrier / code)
static std::atomic<uint64_t> g_stats[4];

// The serial half: the accumulator after the block's nsamp additions of f_carr*delt (gps.c:2821-2826), from `start`.
static inline double chain_block(double f_carr, double delt, int nsamp, double start)
{
    NcoWalk cw;
    cw.setup(f_carr * delt, 1);
    return cw.run(start, nsamp);
}

// The parallel half: one channel of one block, given the double the reference's accumulator holds at its start (1.0 included,
// see below).  Quantises the descriptor seeded from it (-> q) and appends the samples where the double path takes another LUT
// entry or sign than the closed form.  The candidates are found without visiting samples (candidates()); each is decided by
// the drift enclosure from the block's start state, and only where a boundary lies inside the enclosure are the accumulators
// walked (from the block's start, to the last undecided sample).
// slot_of: the block's descriptors up to this channel (blk[0 .. i]), or null with `slot` given: the device order counts the
// active channels before this one (gpsiq_set_descriptors), which is only looked up when a patch is written.
// seed_override: the fixed-point carrier phase the descriptor is seeded with instead of the one cut from `start` (the device
// renders from an ESTIMATE of the start state and learns the true one later, gpsiq_evaldev.cpp): the closed form then runs
// |seed - fixed(start)| units beside the one seeded from the truth, the candidate window is that much wider, and the patches are
// what the double path from the TRUE start takes against that closed form.
static int eval_block(const gpsiq_chan_t &ch_in, double start, double delt, int nsamp, int block, int slot,
                      CodeCache *codes, gpsiq_qchan_t *qq, std::vector<gpsiq_patch_t> *out, bool no_drift = false,
                      const gpsiq_chan_t *blk = nullptr, int i_in_blk = 0, const uint64_t *seed_override = nullptr)
{
    const gpsiq_chan_t &ch = ch_in;
    // a start of exactly 1.0 (a wrap of the block before that rounded up to one) is phase 0 of the closed form (mod 1); the
    // reference goes on from 1.0, and sample 0, where it indexes its table at 512, is patched.  The start state replaces the
    // descriptor's own carr_phase (handed to the quantiser as the carried phase: the 296-byte descriptor is not copied)
    if (!(start >= 0.0 && start <= 1.0)) return fail(GPSIQ_E_RANGE, "prn %d: start phase %g outside [0, 1]", ch.prn, start);
    const uint64_t seeded_true = carr_phase_to_fixed(start == 1.0 ? 0.0 : start);
    const uint64_t seeded = seed_override ? (*seed_override & ((UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1)) : seeded_true;
    uint64_t seed_off = (seeded_true - seeded) & ((UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1);       // |difference| mod 2^59
    if (seed_off > (UINT64_C(1) << (GPSIQ_CARR_FRAC_BITS - 1))) seed_off = (UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - seed_off;
    int qrc;
    if (ch.carr_phase >= 0.0 && ch.carr_phase < 1.0) qrc = quantize_one(ch, delt, nsamp, &seeded, qq, nullptr);
    else {                         // the descriptor's own phase is the 1.0 the reference handed back (or garbage): not the quantiser's business here
        gpsiq_chan_t tmp = ch;
        tmp.carr_phase = 0.0;
        qrc = quantize_one(tmp, delt, nsamp, &seeded, qq, nullptr);
    }
    if (qrc != GPSIQ_OK) return qrc;
    const gpsiq_qchan_t &q = *qq;
    const long ns = nsamp;
    if (ns <= 0) return GPSIQ_OK;
    const double carr_inc = ch.f_carr * delt, code_inc = ch.f_code * delt;
    // drift bounds at the end of the block, in units of the fixed-point formats (see the header)
    const uint64_t w_carr = ((uint64_t) ns << (GPSIQ_CARR_FRAC_BITS - 54)) + (uint64_t) ns / 2 + 4 + seed_off;
    const uint64_t w_code = ((uint64_t) ns << (GPSIQ_CODE_FRAC_BITS - 44)) + (uint64_t) ns / 2 + 4;
    // scratch that keeps its capacity from block to block (an evaluation task runs thousands of these back to back)
    static thread_local std::vector<long> t_carr, t_code, open_c, open_k;
    static thread_local std::vector<double> xc, xk;
    static thread_local std::vector<long> periods;
    t_carr.clear(); t_code.clear(); open_c.clear(); open_k.clear();
    constexpr size_t kCap = 256;
    bool every = false;
    if (carr_inc != 0.0 || seed_off != 0)          // a zero addend leaves both paths constant, and equal when seeded alike
        every |= !candidates(q.carr_phase, (uint64_t) q.carr_step, GPSIQ_CARR_FRAC_BITS - 9, w_carr, ns, kCap, &t_carr);
    every |= !candidates(q.code_frac, q.code_step, GPSIQ_CODE_FRAC_BITS, w_code, ns, kCap, &t_code);
    if (every) {                  // too many to list (a sample rate far outside the format's design range): every sample, both accumulators
        t_carr.resize((size_t) ns);
        for (long n = 0; n < ns; ++n) t_carr[(size_t) n] = n;
        t_code = t_carr;
    }
    if (t_carr.empty() && t_code.empty()) return GPSIQ_OK;
    // A sample that is no candidate of an accumulator has that accumulator's closed-form index by construction: the carrier is
    // looked at in the carrier's candidates only, the code phase in the code's.
    // 1. decide from the start state alone: cells[k] >= 0 the double path's cell, -1 undecided (walk)
    static thread_local std::vector<long> cell_c, cell_k;
    cell_c.assign(t_carr.size(), -1); cell_k.assign(t_code.size(), -1);
    const double x0c = start;                                                   // 1.0 included: undecided by construction, walked
    if (!every && !no_drift) {
        Drift::i128 cell;
        if (!t_carr.empty()) {
            Drift dc;
            dc.setup(carr_inc, 1);
            for (size_t k = 0; k < t_carr.size(); ++k)
                if (dc.cell_at(x0c, t_carr[k], &cell)) cell_c[k] = (long) (cell & 511);
                else open_c.push_back(t_carr[k]);
        }
        if (!t_code.empty()) {
            Drift dk;
            dk.setup(code_inc, 0);
            for (size_t k = 0; k < t_code.size(); ++k)
                if (dk.cell_at(ch.code_phase, t_code[k], &cell)) cell_k[k] = (long) cell;
                else open_k.push_back(t_code[k]);
        }
    } else {
        open_c = t_carr;
        open_k = t_code;
    }
    // 2. walk what is left (from the block's start: the walk reports the state at its targets on the way)
    if (!open_c.empty()) {
        NcoWalk cw;
        cw.setup(carr_inc, 1);
        xc.resize(open_c.size());
        (void) cw.run(start, open_c.back() + 1, open_c.data(), (int) open_c.size(), xc.data());
    }
    if (!open_k.empty()) {
        NcoWalk kw;
        kw.setup(code_inc, 0);
        xk.resize(open_k.size()); periods.resize(open_k.size());
        (void) kw.run(ch.code_phase, open_k.back() + 1, open_k.data(), (int) open_k.size(), xk.data(), periods.data());
    }
    g_stats[0].fetch_add(t_carr.size() + t_code.size(), std::memory_order_relaxed);
    g_stats[1].fetch_add(t_carr.size() + t_code.size() - open_c.size() - open_k.size(), std::memory_order_relaxed);
    g_stats[2].fetch_add(open_c.size(), std::memory_order_relaxed);
    g_stats[3].fetch_add(open_k.size(), std::memory_order_relaxed);
    const uint8_t *ca = codes->get(ch.prn);
    size_t ic = 0, ik = 0, jc = 0, jk = 0;                     // merge of the two ascending candidate lists
    while (ic < t_carr.size() || ik < t_code.size()) {
        const long n = ik >= t_code.size() || (ic < t_carr.size() && t_carr[ic] <= t_code[ik]) ? t_carr[ic] : t_code[ik];
        const bool in_c = ic < t_carr.size() && t_carr[ic] == n, in_k = ik < t_code.size() && t_code[ik] == n;
        // fixed-point path (include/gpsiq.h)
        const uint64_t P = (q.carr_phase + (uint64_t) q.carr_step * (uint64_t) n) & ((UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1);
        const unsigned idx_f = (unsigned) (P >> (GPSIQ_CARR_FRAC_BITS - 9));
        const u128 T = (u128) q.code_frac + (u128) q.code_step * (uint64_t) n;
        const uint64_t A = (uint64_t) q.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
        const unsigned chip_f = (unsigned) (A % GPSIQ_CA_SEQ_LEN);
        const long per_f = (long) (A / GPSIQ_CA_SEQ_LEN);
        const unsigned neg_f = ca[chip_f] ^ nav_bit(ch, ((long) ch.icode + per_f) / 20);
        // double path
        unsigned idx = idx_f;
        if (in_c) {
            if (cell_c[ic] >= 0) idx = (unsigned) cell_c[ic];
            else {
                idx = (unsigned) (int) std::floor(xc[jc++] * 512.0);                      // gps.c:2775
                if (idx > 511u) idx = 511u;   // carr_phase == 1.0 (a negative phase within 2^-54 of zero): the reference indexes past its table there
            }
            ++ic;
        }
        unsigned neg_d = neg_f;
        if (in_k) {
            unsigned chip_d;
            long per_d;
            if (cell_k[ik] >= 0) { chip_d = (unsigned) (cell_k[ik] % GPSIQ_CA_SEQ_LEN); per_d = cell_k[ik] / GPSIQ_CA_SEQ_LEN; }
            else { chip_d = (unsigned) (int) xk[jk]; per_d = periods[jk]; ++jk; }         // gps.c:2817, 2791-2793
            neg_d = ca[chip_d] ^ nav_bit(ch, ((long) ch.icode + per_d) / 20);             // gps.c:2791-2811
            ++ik;
        }
        if (idx != idx_f || neg_d != neg_f) {
            if (blk) { slot = 0; for (int j = 0; j < i_in_blk; ++j) slot += blk[j].prn > 0; blk = nullptr; }
            gpsiq_patch_t p;
            p.block = (uint32_t) block; p.sample = (uint32_t) n;
            p.slot = (uint8_t) slot; p.neg = (uint8_t) neg_d; p.lut = (uint16_t) idx;
            out->push_back(p);
        }
    }
    return GPSIQ_OK;
}

double chain_block_true(double f_carr, double delt, int nsamp, double start) { return chain_block(f_carr, delt, nsamp, start); }

// GPSIQ_CHAIN_VERIFY=N (read per call): every N-th block linked through its certified map is also walked serially from the same
// start state and must end on the same double; 0 / unset: off
int chain_verify_every()
{
    const char *e = std::getenv("GPSIQ_CHAIN_VERIFY");
    const int n = e ? std::atoi(e) : 0;
    return n > 0 ? n : 0;
}

// one channel of one block for the device path (gpsiq_evaldev.cpp): the (block, channel) pairs its enclosure cannot decide
int eval_block_host(const gpsiq_chan_t &ch, double start, const uint64_t *seed, double delt, int nsamp, int block, int slot, gpsiq_qchan_t *q,
                    std::vector<gpsiq_patch_t> *out)
{
    CodeCache codes;
    return eval_block(ch, start, delt, nsamp, block, slot, &codes, q, out, false, nullptr, 0, seed);
}

// ---- the host side of GPSIQ_NCO_REFERENCE as tasks ---------------------------------------------------------------
// Only the carrier chain is serial: block b of a channel starts where the double accumulator left block b-1 (gps.c:2821
// carries chan[i].carr_phase; allocateChannel re-initialises it when the slot gets another satellite, gps.c:2208-2214).
// Everything else a block needs -- the quantised descriptor seeded from its start state, the samples where the double path
// leaves the closed form (eval_block) -- depends on that start state alone.  So the timeline is cut into pieces and the work
// into 2 * nchan tasks per piece: CHAIN(i, k) carries channel i through piece k and publishes the start state of each of its
// blocks (after CHAIN(i, k-1)); EVAL(i, k) quantises and patches channel i's blocks of piece k (after CHAIN(i, k), on any
// thread).  The threads of the shared pool (never more than GPSIQ_THREADS allows) take the runnable task of the LOWEST
// piece, chain before evaluation: with a thread per channel every thread alternates between its chain and its evaluations,
// with fewer the timeline is worked through piece-major -- all channels through piece k before anyone starts piece k+1 -- so
// whoever renders (the calling thread of a batch, the device queues of a multi-device batch: they wait for a piece to be
// complete in all channels) still gets its pieces in order and early, and a thread that finds nothing at the front runs a
// chain ahead.  With the start states given (RefWalk::seeds: another process or device walked the chain) only the
// evaluations run; with chain_only only the chains (gpsiq_reference_chain).
RefWalk::RefWalk(const gpsiq_chan_t *ch_, int nblocks_, int nchan_, double delt_, int nsamp_, gpsiq_qchan_t *q_,
                 const double *carr_in_, const int *prn_in_, const std::vector<int> &piece_ends)
    : ch(ch_), nblocks(nblocks_), nchan(nchan_), nsamp(nsamp_), delt(delt_), q(q_), ends(piece_ends), have_in(carr_in_ && prn_in_)
{
    if (ends.empty() || ends.back() != nblocks) ends.push_back(nblocks);
    for (int i = 0; i < GPSIQ_MAX_CHAN; ++i) {
        carr_in[i] = have_in && i < nchan ? carr_in_[i] : 0.0;
        prn_in[i] = have_in && i < nchan ? prn_in_[i] : 0;
        carr_end[i] = 0.0; last_prn[i] = 0;
        chain_next[i] = 0; chain_busy[i] = false; eval_next[i] = 0;
        carr[i] = carr_in[i]; prev[i] = prn_in[i];
    }
    patches.resize((size_t) nchan * ends.size());
    done.reset(new std::atomic<int>[ends.size()]);
    for (size_t k = 0; k < ends.size(); ++k) done[k].store(0, std::memory_order_relaxed);
    pthread_mutex_init(&mu, nullptr);
    pthread_cond_init(&cv, nullptr);
    pthread_cond_init(&task_cv, nullptr);
}

RefWalk::~RefWalk()
{
    pthread_mutex_destroy(&mu);
    pthread_cond_destroy(&cv);
    pthread_cond_destroy(&task_cv);
}

void RefWalk::set_error(int code, const char *text, int block)
{
    pthread_mutex_lock(&mu);
    if (rc == GPSIQ_OK) { rc = code; std::snprintf(err, sizeof err, "block %d: %.280s", block, text); }
    pthread_mutex_unlock(&mu);
    abort_flag.store(true, std::memory_order_release);       // the tasks still to come finish at once (their pieces complete, waiters wake)
}

// channel i through piece k: the start state of every block, then the accumulator after it.  With the blocks' certified maps
// at hand (maps: level 1 of the time-parallel chain, gpsiq_lane.h, walked on the device) a block is one exact subtraction,
// range check and addition (chain_step_mapped) and only the rare block whose map does not apply is walked.
void RefWalk::chain_task(int i, size_t k)
{
    const int b0 = k ? ends[k - 1] : 0, b1 = ends[k];
    double c = carr[i];
    int pv = prev[i];
    long walked = 0, linked = 0;
    for (int b = b0; b < b1; ++b) {
        const size_t at = (size_t) b * nchan + i;
        // a channel's descriptors lie nchan * 296 bytes apart: every block is a cache miss unless it is asked for early
        if (b + 6 < b1) {
            if (in) __builtin_prefetch(&in[at + (size_t) 6 * nchan]); else __builtin_prefetch(&ch[at + (size_t) 6 * nchan]);
            if (maps) chain_prefetch_map(maps, at + (size_t) 6 * nchan);
        }
        const double f_carr = in ? in[at].f_carr : ch[at].f_carr, phase0 = in ? in[at].carr_phase : ch[at].carr_phase;
        const int prn = in ? in[at].prn : ch[at].prn;
        if (prn <= 0) { pv = 0; c = 0.0; start_out[at] = 0.0; continue; }
        if ((b == 0 && !have_in) || pv != prn) c = phase0;
        pv = prn;
        start_out[at] = c;
        if (abort_flag.load(std::memory_order_acquire)) continue;
        // what quantize_one checks of the carrier (the evaluation reports the rest): an addend the walk is defined for
        const double inc = f_carr * delt;
        if (!(c >= 0.0 && c <= 1.0) || !(std::fabs(inc) < 0.5)) {
            set_error(GPSIQ_E_RANGE, "carrier phase or Doppler outside the NCO format", b);
            continue;
        }
        double y;
        if (maps && chain_step_mapped(maps, at, c, &y)) {
            // GPSIQ_CHAIN_VERIFY=N: every N-th block that went through its map is also walked from the same start state
            if (verify_every > 0 && (b + i) % verify_every == 0) {
                const double yw = chain_block(f_carr, delt, nsamp, c);
                if (bits_of(yw) != bits_of(y)) {
                    char msg[160];
                    std::snprintf(msg, sizeof msg, "slot %d: the certified map ends on %.17g, the serial walk on %.17g", i, y, yw);
                    set_error(GPSIQ_E_VERIFY, msg, b);
                }
            }
            c = y; ++linked;
        } else { c = chain_block(f_carr, delt, nsamp, c); ++walked; }
    }
    carr[i] = c; prev[i] = pv;
    if (maps) chain_count(linked, walked);
}

// channel i's blocks of piece k: descriptors and patches from their start states
void RefWalk::eval_task(int i, size_t k, CodeCache *codes)
{
    const int b0 = k ? ends[k - 1] : 0, b1 = ends[k];
    // the task's own list: two evaluations of one channel may run at the same time (pieces k and k+1 on two threads), and
    // whoever takes piece k's patches does so after done[k] has counted this task (release / acquire)
    std::vector<gpsiq_patch_t> *mine = &patches[(size_t) i * ends.size() + k];
    for (int b = b0; b < b1; ++b) {
        const size_t at = (size_t) b * nchan + i;
        if (b + 3 < b1) {                                        // the next descriptors of this channel (five cache lines each, all misses)
            const char *nx = reinterpret_cast<const char *>(&ch[at + (size_t) 3 * nchan]);
            __builtin_prefetch(nx); __builtin_prefetch(nx + 64); __builtin_prefetch(nx + 128); __builtin_prefetch(nx + 192); __builtin_prefetch(nx + 256);
        }
        const gpsiq_chan_t &d = ch[at];
        if (d.prn <= 0 || abort_flag.load(std::memory_order_acquire)) {
            gpsiq_chan_t none = d;
            none.prn = 0;
            (void) quantize_one(none, delt, nsamp, nullptr, &q[at], nullptr);           // an unused slot: zeroes
            continue;
        }
        const int erc = eval_block(d, start[at], delt, nsamp, b, 0, codes, &q[at], mine, no_drift, &ch[(size_t) b * nchan], i);
        if (erc != GPSIQ_OK) set_error(erc, gpsiq_last_error(), b);
    }
}

// the maps of blocks [0, upto) have arrived (the device walks level 1 of a long timeline in two launches): their chain tasks may run
void RefWalk::release_maps(int upto)
{
    pthread_mutex_lock(&mu);
    maps_upto.store(upto, std::memory_order_release);
    pthread_cond_broadcast(&task_cv);
    pthread_mutex_unlock(&mu);
}

void RefWalk::finish_piece(size_t k)
{
    if (done[k].fetch_add(1, std::memory_order_acq_rel) + 1 == nchan) {
        pthread_mutex_lock(&mu);
        pthread_cond_broadcast(&cv);
        pthread_mutex_unlock(&mu);
    }
}

// one thread of the job: take tasks until none is left
void RefWalk::work()
{
    CodeCache codes;
    const size_t np = ends.size();
    pthread_mutex_lock(&mu);
    for (;;) {
        // the runnable task of the lowest piece; chain before evaluation, then the lower channel
        int best_i = -1;
        size_t best_k = np;
        bool best_chain = false, pending = false;
        for (int i = 0; i < nchan; ++i) {
            if (!seeds && chain_next[i] < np) {
                pending = true;
                if (!chain_busy[i] && ends[chain_next[i]] <= maps_upto.load(std::memory_order_acquire) && (chain_next[i] < best_k || (chain_next[i] == best_k && !best_chain))) { best_i = i; best_k = chain_next[i]; best_chain = true; }
            }
            if (!chain_only && eval_next[i] < np) {
                pending = true;
                const size_t ready = seeds ? np : chain_next[i];                      // pieces whose start states are known
                if (eval_next[i] < ready && eval_next[i] < best_k) { best_i = i; best_k = eval_next[i]; best_chain = false; }
            }
        }
        if (best_i < 0) {
            if (!pending) { pthread_cond_broadcast(&task_cv); break; }                // everything is taken: the sleepers may leave too
            pthread_cond_wait(&task_cv, &mu);                                         // a chain in flight will make more runnable
            continue;
        }
        const int i = best_i;
        const size_t k = best_k;
        if (best_chain) chain_busy[i] = true; else ++eval_next[i];
        pthread_mutex_unlock(&mu);
        if (best_chain) {
            chain_task(i, k);
            if (chain_only) finish_piece(k);
            pthread_mutex_lock(&mu);
            chain_busy[i] = false;
            ++chain_next[i];
            // EVAL(i, k) and CHAIN(i, k+1) are runnable now: this thread takes one of them, one sleeper the other (a thread only
            // sleeps after it has found nothing runnable under the lock, and tasks only become runnable here: nothing is missed,
            // and nobody is woken to find nothing -- with 16 threads and tasks of ~50 us a broadcast is most of the work)
            pthread_cond_signal(&task_cv);
        } else {
            eval_task(i, k, &codes);
            finish_piece(k);
            pthread_mutex_lock(&mu);
        }
    }
    pthread_mutex_unlock(&mu);
}

void RefWalk::run()
{
    const size_t total = (size_t) nblocks * (size_t) nchan;
    if (seeds) start = seeds;
    else if (!start_out) { own_start.assign(total ? total : 1, 0.0); start_out = own_start.data(); }
    if (!seeds) start = start_out;
    size_t k0 = 0;
    while (k0 < ends.size() && ends[k0] == 0) ++k0;                                   // empty pieces in front
    for (int i = 0; i < nchan; ++i) { chain_next[i] = k0; eval_next[i] = k0; }
    for (size_t k = 0; k < k0; ++k) done[k].store(nchan, std::memory_order_release);
    if (k0) { pthread_mutex_lock(&mu); pthread_cond_broadcast(&cv); pthread_mutex_unlock(&mu); }
    // a block or two (the drop-in block call): 16 channels x a few microseconds cost less than waking the pool
    const int want = nblocks <= 2 ? 1 : (chain_only || seeds ? nchan : 2 * nchan);
    no_drift = std::getenv("GPSIQ_NO_DRIFT") != nullptr;      // A/B + test knob, read per call: every candidate walked
    verify_every = chain_verify_every();
    const bool trace = std::getenv("GPSIQ_TRACE") != nullptr;
    if (trace && want > 1 && host_threads() < nchan)
        std::fprintf(stderr, "[gpsiq trace] reference NCO: %d host threads for %d channels: pieces are worked through piece-major\n", host_threads(), nchan);
    run_job([](void *p) { static_cast<RefWalk *>(p)->work(); }, this, want);
    for (int i = 0; i < nchan; ++i) { carr_end[i] = carr[i]; last_prn[i] = prev[i]; }
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
    const uint32_t b0 = k ? (uint32_t) ends[k - 1] : 0u;
    out->clear();
    for (int i = 0; i < nchan; ++i) {
        const std::vector<gpsiq_patch_t> &v = patches[(size_t) i * ends.size() + k];
        out->insert(out->end(), v.begin(), v.end());
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
    if (nblocks == 1) {
        // the drop-in block call (one 0.1 s block per call, ~100 us all told): chain and evaluation of the sixteen channels right
        // here, without the scheduler's mutexes, condition variables and task state
        CodeCache codes;
        patches->clear();
        int slot = 0;
        for (int i = 0; i < nchan; ++i) {
            const gpsiq_chan_t &d = ch[i];
            if (d.prn <= 0) {
                (void) quantize_one(d, delt, nsamp, nullptr, &q[i], nullptr);
                if (carr_end) carr_end[i] = 0.0;
                if (last_prn) last_prn[i] = 0;
                continue;
            }
            const double start = carr_in && prn_in && prn_in[i] == d.prn ? carr_in[i] : d.carr_phase;
            if (!(start >= 0.0 && start <= 1.0) || !(std::fabs(d.f_carr * delt) < 0.5))
                return fail(GPSIQ_E_RANGE, "block 0: carrier phase or Doppler outside the NCO format");
            const int erc = eval_block(d, start, delt, nsamp, 0, slot++, &codes, &q[i], patches, std::getenv("GPSIQ_NO_DRIFT") != nullptr);
            if (erc != GPSIQ_OK) { char msg[300]; std::snprintf(msg, sizeof msg, "%s", gpsiq_last_error()); return fail(erc, "block 0: %.280s", msg); }
            if (carr_end) carr_end[i] = chain_block(d.f_carr, delt, nsamp, start);
            if (last_prn) last_prn[i] = d.prn;
        }
        std::sort(patches->begin(), patches->end(), [](const gpsiq_patch_t &a, const gpsiq_patch_t &b) {
            if (a.sample != b.sample) return a.sample < b.sample;
            return a.slot < b.slot;
        });
        return GPSIQ_OK;
    }
    // a handful of pieces so that a thread-starved host still overlaps chains and evaluations; a short timeline is one piece
    std::vector<int> ends;
    const int chunk = nblocks > 64 ? (nblocks + 15) / 16 : nblocks;
    for (int b = chunk; b < nblocks; b += chunk) ends.push_back(b);
    RefWalk w(ch, nblocks, nchan, delt, nsamp, q, carr_in, prn_in, ends);
    w.run();
    if (w.rc != GPSIQ_OK) return fail(w.rc, "%s", w.err);
    patches->clear();
    std::vector<gpsiq_patch_t> part;
    for (size_t k = 0; k < w.npieces(); ++k) {
        w.take_patches(k, &part, false);
        patches->insert(patches->end(), part.begin(), part.end());
    }
    for (int i = 0; i < nchan; ++i) {
        if (carr_end) carr_end[i] = w.carr_end[i];
        if (last_prn) last_prn[i] = w.last_prn[i];
    }
    return GPSIQ_OK;
}

}  // namespace gpsiq

using namespace gpsiq;

extern "C" int gpsiq_reference_chain(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                                     const double *carr_in, const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn)
{
    if ((!in || !carr_start) && nblocks) return fail(GPSIQ_E_ARG, "null pointer");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0) || (!carr_in != !prn_in)) return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g / continuation state", nsamp, fs);
    // one task per channel when there is a thread for every channel; else pieces, so that the channels share the threads evenly
    std::vector<int> ends;
    const int chunk = host_threads() >= nchan || nblocks <= 64 ? nblocks : (nblocks + 7) / 8;
    for (int b = chunk; b < nblocks; b += chunk) ends.push_back(b);
    RefWalk w(nullptr, nblocks, nchan, 1.0 / fs, nsamp, nullptr, carr_in, prn_in, ends);
    w.in = in; w.chain_only = true; w.start_out = carr_start;
    w.run();
    if (w.rc != GPSIQ_OK) return fail(w.rc, "%s", w.err);
    for (int i = 0; i < nchan; ++i) {
        if (carr_end) carr_end[i] = w.carr_end[i];
        if (last_prn) last_prn[i] = w.last_prn[i];
    }
    return GPSIQ_OK;
}

extern "C" int gpsiq_reference_seeded(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp, const double *carr_start,
                                      gpsiq_qchan_t *q, gpsiq_patch_t *patches, int max_patches, int *npatches)
{
    if ((!ch || !q || !carr_start) && nblocks) return fail(GPSIQ_E_ARG, "null pointer");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0) || max_patches < 0 || !npatches || (max_patches && !patches))
        return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g / patch buffer", nsamp, fs);
    for (size_t k = 0; k < (size_t) nblocks * (size_t) nchan; ++k)
        if (ch[k].prn > 0 && !(carr_start[k] >= 0.0 && carr_start[k] <= 1.0)) return fail(GPSIQ_E_RANGE, "start phase %zu outside [0, 1]", k);
    std::vector<int> ends;
    const int chunk = nblocks > 64 ? (nblocks + 15) / 16 : nblocks;
    for (int b = chunk; b < nblocks; b += chunk) ends.push_back(b);
    RefWalk w(ch, nblocks, nchan, 1.0 / fs, nsamp, q, nullptr, nullptr, ends);
    w.seeds = carr_start;
    w.run();
    if (w.rc != GPSIQ_OK) return fail(w.rc, "%s", w.err);
    std::vector<gpsiq_patch_t> v, part;
    for (size_t k = 0; k < w.npieces(); ++k) {
        w.take_patches(k, &part, false);
        v.insert(v.end(), part.begin(), part.end());
    }
    *npatches = (int) v.size();
    if (v.size() > (size_t) max_patches) return fail(GPSIQ_E_RANGE, "%zu patches, room for %d", v.size(), max_patches);
    if (!v.empty()) std::memcpy(patches, v.data(), v.size() * sizeof(gpsiq_patch_t));
    return GPSIQ_OK;
}

extern "C" void gpsiq_reference_stats(uint64_t out[4])
{
    if (!out) return;
    for (int k = 0; k < 4; ++k) out[k] = g_stats[k].load(std::memory_order_relaxed);
}

extern "C" void gpsiq_chain_inputs(const gpsiq_chan_t *ch, int n, gpsiq_chain_in_t *out)
{
    if (!ch || !out) return;
    for (int k = 0; k < n; ++k) { out[k].f_carr = ch[k].f_carr; out[k].carr_phase = ch[k].carr_phase; out[k].prn = ch[k].prn; out[k].reserved = 0; }
}

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
