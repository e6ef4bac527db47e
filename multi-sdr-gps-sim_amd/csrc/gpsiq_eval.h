// gpsiq_eval.h -- GPSIQ_NCO_REFERENCE, the per-block EVALUATION as code shared by host and device (GPSIQ_HD, same source, same
// IEEE double operations under -ffp-contract=off): what csrc/gpsiq_exact.cpp::eval_block does on host threads -- quantise the
// descriptor seeded from the double the reference's accumulator holds at the block's first sample, find the samples where the
// double path (gps.c:2775-2782 index + truncation, gps.c:2789-2826 the two accumulators) can leave the closed form, decide
// them from the block's start state by the drift enclosure, emit the patches -- for ONE lane per (block, channel), and
// level 2 of the time-parallel carrier chain (gpsiq_lane.h link_block) as an associative scan instead of a serial loop.
//
//   DChan            what a lane needs of a 296-byte gpsiq_chan_t: 64 bytes (the first 24 are gpsiq_chain_in_t)
//   quantize_dchan   = quantize_one (gpsiq_host.cpp), status codes instead of text
//   next_candidate   = candidates() one at a time (the Euclid-style descent without recursion: a GPU lane has no call stack to spare)
//   DriftLite        = Drift's enclosure (gpsiq_exact.cpp) without its tables: G is summed up to the binade asked for
//   eval_chan        = eval_block; a candidate the enclosure cannot decide, or more candidates than a list holds, hands the
//                      (block, channel) to the host walker (kEvalHost): 1 in 10^4 at the BASELINE rates
//   Link / link_*    the chain x_{b+1} = F_b(x_b) through the certified maps, d_{b+1} = d_b + cum_b[parity(d_b)] + J_{b+1} with
//                      J_{b+1} = (e_b - xs_{b+1}) / U: integer translations that depend on d mod 4 only -- a scan over 4-tuples
// tests/eval_twin.cpp holds all of it against eval_block / chain_link on the CPU; the kernels are in gpsiq_eval_kernels.hip.
#ifndef GPSIQ_EVAL_H
#define GPSIQ_EVAL_H

#include "../../include/gpsiq.h"
#include "gpsiq_lane.h"

namespace gpsiq {
namespace ev {

typedef unsigned __int128 u128;
typedef __int128 i128;

// ---- the packed channel state --------------------------------------------------------------------------------------------
struct DChan {
    double   f_carr;       // gpsiq_chan_t.f_carr        } the layout of gpsiq_chain_in_t: the chain kernels read these three
    double   carr_phase;   // gpsiq_chan_t.carr_phase    }
    int32_t  prn;          // as given; <= 0: unused     }
    uint32_t pos;          // iword | ibit << 6 | icode << 11 | navail << 16; kBadPos: iword / ibit / icode outside dwrd
    double   f_code;
    double   code_phase;
    double   gain;
    uint64_t nav;          // bit k: the data bit k nav-bit periods after (iword, ibit), k < navail (gps.c:2811)
    double   start;        // the accumulator at the block's first sample (written by the link; seeded calls: by the caller)
};
static_assert(sizeof(DChan) == 64, "DChan is one cache line");
constexpr uint32_t kBadPos = UINT32_C(1) << 31;
constexpr int kNavWindow = 40;            // a block touches at most GPSIQ_MAX_NAV_BITS = 32 bits; the double path may sit one further

GPSIQ_HD inline uint32_t reverse_bits(uint32_t v, int n)     // the low n bits of v, first bit last
{
    v = ((v >> 1) & 0x55555555u) | ((v & 0x55555555u) << 1);
    v = ((v >> 2) & 0x33333333u) | ((v & 0x33333333u) << 2);
    v = ((v >> 4) & 0x0f0f0f0fu) | ((v & 0x0f0f0f0fu) << 4);
    v = ((v >> 8) & 0x00ff00ffu) | ((v & 0x00ff00ffu) << 8);
    v = (v >> 16) | (v << 16);
    return n ? v >> (32 - n) : 0u;
}

GPSIQ_HD inline void pack_chan(const gpsiq_chan_t &ch, DChan *out)
{
    DChan d;
    d.f_carr = ch.f_carr; d.carr_phase = ch.carr_phase; d.prn = ch.prn; d.pos = 0;
    d.f_code = ch.f_code; d.code_phase = ch.code_phase; d.gain = ch.gain; d.nav = 0; d.start = 0.0;
    if (ch.prn > 0) {
        if (ch.iword < 0 || ch.iword >= GPSIQ_N_DWRD || ch.ibit < 0 || ch.ibit > 29 || ch.icode < 0 || ch.icode > 19) d.pos = kBadPos;
        else {
            const int p0 = ch.iword * 30 + ch.ibit;
            const int navail = GPSIQ_N_DWRD * 30 - p0 < kNavWindow ? GPSIQ_N_DWRD * 30 - p0 : kNavWindow;
            int got = 0, w = ch.iword, bit = ch.ibit;
            uint64_t nav = 0;
            while (got < navail) {
                const int take = 30 - bit < navail - got ? 30 - bit : navail - got;
                const uint32_t chunk = (ch.dwrd[w] >> (30 - bit - take)) & ((UINT32_C(1) << take) - 1u);   // first data bit on top
                nav |= (uint64_t) reverse_bits(chunk, take) << got;
                got += take; bit = 0; ++w;
            }
            d.nav = nav;
            d.pos = (uint32_t) ch.iword | (uint32_t) ch.ibit << 6 | (uint32_t) ch.icode << 11 | (uint32_t) navail << 16;
        }
    }
    *out = d;
}
GPSIQ_HD inline int dchan_iword(const DChan &d) { return (int) (d.pos & 63u); }
GPSIQ_HD inline int dchan_ibit(const DChan &d) { return (int) ((d.pos >> 6) & 31u); }
GPSIQ_HD inline int dchan_icode(const DChan &d) { return (int) ((d.pos >> 11) & 31u); }
GPSIQ_HD inline int dchan_navail(const DChan &d) { return (int) ((d.pos >> 16) & 63u); }
// the data bit `bit` nav-bit periods after the block's first (nav_bit of gpsiq_exact.cpp: 0 past the end of dwrd)
GPSIQ_HD inline unsigned dchan_nav_bit(const DChan &d, long bit) { return bit >= 0 && bit < dchan_navail(d) ? (unsigned) (d.nav >> bit) & 1u : 0u; }

// ---- the quantiser (quantize_one, gpsiq_host.cpp) ---------------------------------------------------------------------------
enum QStatus { kQOk = 0, kQPrn, kQCarrInc, kQCodeInc, kQCarrPhase, kQCodePhase, kQPos, kQGain, kQNavBits, kQDwrd, kQStart, kQStatusCount };
GPSIQ_HD inline int qstatus_code(int s) { return s == kQOk ? GPSIQ_OK : s == kQPrn ? GPSIQ_E_ARG : GPSIQ_E_RANGE; }
constexpr double kEvMaxGain = 4.0e6;                 // kMaxGain of gpsiq_internal.h
constexpr uint64_t kEvCarrMask = (UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1, kEvCodeMask = (UINT64_C(1) << GPSIQ_CODE_FRAC_BITS) - 1;

GPSIQ_HD inline uint64_t phase_to_fixed(double cycles)        // carr_phase_to_fixed
{
    return (uint64_t) __builtin_floor(cycles * 0x1p59) & kEvCarrMask;
}

// seed: replaces the phase derived from d.carr_phase (which is then not looked at: the caller's business)
GPSIQ_HD inline int quantize_dchan(const DChan &d, double delt, int nsamp, const uint64_t *seed, gpsiq_qchan_t *q)
{
    gpsiq_qchan_t z;
    z.carr_phase = 0; z.carr_step = 0; z.code_frac = 0; z.code_step = 0; z.gain = 0.0; z.nav_bits = 0; z.chip0 = 0; z.icode = 0; z.prn = 0;
    *q = z;
    if (d.prn <= 0) return kQOk;
    if (d.prn > 32) return kQPrn;
    const double carr_inc = d.f_carr * delt;   // the operand of gps.c:2821
    const double code_inc = d.f_code * delt;   // the operand of gps.c:2789
    if (!(__builtin_fabs(carr_inc) < 0.5)) return kQCarrInc;
    if (!(code_inc > 0.0 && code_inc < 2.0)) return kQCodeInc;
    if (!seed && !(d.carr_phase >= 0.0 && d.carr_phase < 1.0)) return kQCarrPhase;
    if (!(d.code_phase >= 0.0 && d.code_phase < (double) GPSIQ_CA_SEQ_LEN)) return kQCodePhase;
    if (d.pos & kBadPos) return kQPos;
    if (!(d.gain > -kEvMaxGain && d.gain < kEvMaxGain)) return kQGain;
    z.prn = (uint8_t) d.prn;
    z.icode = (uint8_t) dchan_icode(d);
    z.gain = d.gain;
    z.carr_step = (int64_t) __builtin_rint(carr_inc * 0x1p59);        // llrint(ldexp(., 59)): the scaling is exact
    z.carr_phase = seed ? (*seed & kEvCarrMask) : phase_to_fixed(d.carr_phase);
    const int whole = (int) d.code_phase;
    z.chip0 = (uint16_t) whole;
    z.code_frac = (uint64_t) __builtin_floor((d.code_phase - (double) whole) * 0x1p56) & kEvCodeMask;
    z.code_step = (uint64_t) (int64_t) __builtin_rint(code_inc * 0x1p56);
    // nav data bits this block can reach (gps.c:2795-2811: 20 code periods per bit)
    const u128 last = (u128) z.code_frac + (u128) z.code_step * (u128) (uint64_t) (nsamp > 0 ? nsamp - 1 : 0);
    const uint64_t chips_end = (uint64_t) whole + (uint64_t) (last >> GPSIQ_CODE_FRAC_BITS);
    const uint64_t nbits = ((uint64_t) dchan_icode(d) + chips_end / GPSIQ_CA_SEQ_LEN) / 20 + 1;
    if (nbits > GPSIQ_MAX_NAV_BITS) return kQNavBits;
    if (nbits > (uint64_t) dchan_navail(d)) return kQDwrd;                // the block runs past dwrd[59]
    z.nav_bits = (uint32_t) (d.nav & ((UINT64_C(1) << nbits) - 1u));
    *q = z;
    return kQOk;
}

// ---- candidates: all n in [0, nsamp) with (a + n*b) mod 2^k within w of 0, one at a time ------------------------------------
constexpr uint64_t kNoHit = ~UINT64_C(0);

GPSIQ_HD inline double u128_to_double(u128 v) { return (double) (uint64_t) (v >> 64) * 0x1p64 + (double) (uint64_t) v; }

// Smallest x in [0, bound) with L <= (A*x mod M) <= R, for 0 <= L <= R < M, 0 <= A < M, M <= 2^56, bound < 2^40; kNoHit if none
// (first_in_range of gpsiq_exact.cpp, the descent on a stack of its own: M halves with every level, 57 levels at most; the
// bound goes down with it, so a search that cannot succeed inside the block stops after ~log2(bound) levels).
GPSIQ_HD inline uint64_t first_in_range(uint64_t A, uint64_t M, uint64_t L, uint64_t R, uint64_t bound)
{
    constexpr int kDepth = 60;
    uint64_t sa[kDepth], sl[kDepth], sb[kDepth];       // per level: A and L after the flip, the bound; its M is the level above's A
    const uint64_t M0 = M;
    int depth = 0;
    uint64_t y;
    for (;;) {
        if (bound == 0) return kNoHit;
        if (L == 0) { y = 0; break; }
        if (A == 0) return kNoHit;
        if (A > M - A) {                 // 2A > M: count downwards instead
            A = M - A;
            const uint64_t l = M - R, r = M - L;
            L = l; R = r;
        }
        const uint64_t lq = L / A, lr = L - lq * A;                   // one division serves the first multiple and L mod A
        const uint64_t k = lq + (lr != 0);
        if ((u128) k * A <= R) { if (k >= bound) return kNoHit; y = k; break; }
        // no multiple of A inside [L, R]: after y wraps of M the window is at M*y + [L, R]; it holds a multiple of A iff
        // ((-M mod A) * y) mod A lies in [L mod A, R mod A]
        const u128 reach = (u128) A * (bound - 1);                    // A * x for the largest x allowed
        if (reach < L) return kNoHit;
        // the wraps that can matter, a little generously (the result is checked against the bound on the way back)
        const double yb = u128_to_double(reach - L) / (double) M * (1.0 + 0x1p-40) + 2.0;
        if (depth >= kDepth) return kNoHit;                           // cannot happen (M halves per level); never index past the stack
        sa[depth] = A; sl[depth] = L; sb[depth] = bound; ++depth;
        const uint64_t mr = M % A;
        const uint64_t span = R - L;                                  // L and R have the same quotient here: R mod A = lr + (R - L)
        M = A; A = mr ? A - mr : 0; L = lr; R = lr + span;
        bound = yb < 0x1p40 ? (uint64_t) yb : (UINT64_C(1) << 40);
    }
    // back up: x = ceil((M*y + L) / A) at every level, refused where it reaches the level's bound
    while (depth > 0) {
        --depth;
        const uint64_t Ai = sa[depth], Li = sl[depth], Mi = depth ? sa[depth - 1] : M0, bi = sb[depth];
        const u128 N = (u128) Mi * y + Li + (Ai - 1);
        // floor(N / Ai) when it is below the bound (< 2^40): a double quotient, corrected on integers
        const double qd = u128_to_double(N) / (double) Ai;
        if (qd >= (double) bi + 2.0) return kNoHit;
        uint64_t q = (uint64_t) qd;
        u128 p = (u128) q * Ai;
        while (p > N) { --q; p -= Ai; }
        while (N - p >= Ai) { ++q; p += Ai; }
        if (q >= bi) return kNoHit;
        y = q;
    }
    return y;
}

// the first candidate at or after sample `base` (kNoHit: none before nsamp).  Window half-width w with 2w + 1 < 2^k.
GPSIQ_HD inline uint64_t next_candidate(uint64_t a, uint64_t b, int k, uint64_t w, long nsamp, long base)
{
    if (base >= nsamp) return kNoHit;
    const uint64_t M = UINT64_C(1) << k, mask = M - 1, W = 2 * w + 1;      // shifted by w the window is [0, W)
    b &= mask;
    const uint64_t cur = (uint64_t) (((u128) a + w + (u128) b * (uint64_t) base) & mask);
    if (cur < W) return (uint64_t) base;
    const uint64_t x = first_in_range(b, M, M - cur, M - cur + W - 1, (uint64_t) (nsamp - base));
    return x == kNoHit ? kNoHit : (uint64_t) base + x;
}

// ---- the drift enclosure (Drift of gpsiq_exact.cpp: the story is there) ---------------------------------------------------
struct DriftLite {
    bool    valid, neg;
    int     kind;
    double  L;
    int64_t ec, mc;
    int     b_lo, b_top;
    double  ulp_c, edge_lo, GL, gmax, Acyc, G_x0, x0;
    int     cell_shift, sh;

    // slope of G in binade b (Drift::setup's loop body)
    GPSIQ_HD inline double slope(int b, bool *tie) const
    {
        const int s = b - (int) ec;
        int64_t dm = mc >> s;
        const int64_t rem = mc & (((int64_t) 1 << s) - 1), half = (int64_t) 1 << (s - 1);
        *tie = false;
        if (rem > half) ++dm;
        else if (rem == half) { *tie = true; dm += dm & 1; }
        const double delta = (double) ((dm << s) - mc);                  // eps_b / ulp_c, exact
        const double Sb = (double) dm * (double) ((int64_t) 1 << s);     // S_b / ulp_c, exact
        return delta / Sb;
    }
    GPSIQ_HD inline double width(int b) const
    {
        const double lo = from_bits((uint64_t) b << 52);
        return (kind == 0 && b == b_top) ? L - lo : lo;                  // [512, 1023) for the code phase's top binade
    }
    // G(x), 0 <= x <= L: the sum of the binades below x's, in Drift's order (same doubles as its Gedge table)
    GPSIQ_HD inline double Gof(double x) const
    {
        if (x < edge_lo) return 0.0;
        if (x >= L) return GL;
        const int bx = (int) (bits_of(x) >> 52);
        double G = 0.0;
        bool tie;
        for (int b = b_lo; b < bx; ++b) G += slope(b, &tie) * width(b);
        return G + slope(bx, &tie) * (x - from_bits((uint64_t) bx << 52));
    }
    GPSIQ_HD inline void setup(double c, int kind_, double x0_)
    {
        kind = kind_;
        L = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
        const int top_exp = kind == 0 ? 1023 + 9 : 1022;
        neg = c < 0.0;
        const uint64_t bc = bits_of(c) & ~(UINT64_C(1) << 63);
        ec = (int64_t) (bc >> 52);
        mc = (int64_t) ((bc & kMant) | (kMant + 1));
        x0 = x0_;
        valid = !(ec > top_exp - 6 || ec < top_exp - 40 || (kind == 0 && neg)) && x0 >= 0.0 && x0 < L;
        if (!valid) return;
        ulp_c = from_bits((uint64_t) (ec - 52) << 52);
        b_lo = (int) ec + 3; b_top = top_exp;
        edge_lo = from_bits((uint64_t) b_lo << 52);
        double G = 0.0, allow = 0.0;
        gmax = 0.0;
        for (int b = b_lo; b <= b_top; ++b) {
            bool tie;
            const double gb = slope(b, &tie);
            G += gb * width(b);
            if (__builtin_fabs(gb) > gmax) gmax = __builtin_fabs(gb);
            allow += 1.001 * from_bits((uint64_t) (b - 52) << 52) * (tie ? 2.0 : 1.0);
        }
        GL = G;
        const double u_top = from_bits((uint64_t) (b_top - 52) << 52), u_lo = from_bits((uint64_t) (b_lo - 52) << 52);
        Acyc = allow + 3.0 * u_lo + (kind == 0 ? 0.51 : 1.51) * u_top;
        sh = (int) (1075 - ec);
        cell_shift = kind == 0 ? sh : sh - 9;
        G_x0 = Gof(x0);
    }
    GPSIQ_HD inline i128 units(double x) const          // x in units of ulp_c, cut below one unit
    {
        const uint64_t bx = bits_of(x);
        const int64_t ex = (int64_t) (bx >> 52);
        if (ex == 0) return 0;
        const i128 mx = (i128) ((bx & kMant) | (kMant + 1));
        const int64_t s = ex - ec;
        return s >= 0 ? mx << s : (s > -64 ? mx >> -s : (i128) 0);
    }
    // the cell (LUT step / chip, unwrapped, counted from phase 0 of the block's first cycle / code period) that holds the double
    // path's phase at sample n, when the enclosure lies inside one cell (Drift::cell_at); false: undecided
    GPSIQ_HD inline bool cell_at(long n, int64_t *cell) const
    {
        if (!valid) return false;
        const i128 R = neg ? units(x0) - (i128) n * mc : units(x0) + (i128) n * mc;     // x0 + n*c
        const int64_t hi_part = (int64_t) (R >> sh);                                    // floor(R / 2^sh): cycles, or chips (< 2^40)
        int64_t q;
        if (kind == 1) q = hi_part;
        else { q = hi_part / GPSIQ_CA_SEQ_LEN; if (hi_part % GPSIQ_CA_SEQ_LEN < 0) --q; }
        const i128 Lint = kind == 0 ? (i128) GPSIQ_CA_SEQ_LEN << sh : (i128) 1 << sh;
        const i128 r = R - (i128) q * Lint;                                             // 0 <= r < Lint
        const double rd = u128_to_double((u128) r) * ulp_c;
        const double core = (double) q * GL + Gof(rd) - G_x0;
        const double I = ((double) (q < 0 ? -q : q) + 3.0) * Acyc;
        // (the last term also covers rd's own rounding: two conversions where Drift has one)
        const double eta = 4.0 * gmax * (__builtin_fabs(core) + I) + 1e-12 * (__builtin_fabs(core) + __builtin_fabs((double) q * GL)) + 8.0 * gmax * L * 0x1p-52;
        const double flo = __builtin_floor((core - I - eta) / ulp_c), fhi = __builtin_ceil((core + I + eta) / ulp_c);
        if (!(flo > -0x1p62 && fhi < 0x1p62)) return false;
        const i128 lo = R + (i128) (int64_t) flo - 2, hi = R + (i128) (int64_t) fhi + 2;
        const i128 c0 = lo >> cell_shift, c1 = hi >> cell_shift;                        // arithmetic shifts: floor for negative phases too
        if (c0 != c1) return false;
        if (c0 > (i128) INT64_MAX / 2 || c0 < -((i128) INT64_MAX / 2)) return false;
        *cell = (int64_t) c0;
        return true;
    }
};

// ---- one channel of one block -----------------------------------------------------------------------------------------
enum { kEvalOk = 0, kEvalHost = 1 };      // kEvalHost: the host walker takes this (block, channel); negative: -QStatus
constexpr int kCandCap = 256;            // candidates per accumulator and block before every sample is looked at (eval_block's kCap)

// chips(prn, i): chip i (0 / 1) of the satellite's C/A code; emit(sample, lut, neg): one patch of this channel.
// q receives the descriptor seeded from `start` (also when the host is asked to redo the patches: the descriptor is the same).
// The descriptor of a channel with the carrier seeded from `phase`: the accumulator at the block's first sample where it is known
// (the host path; the caller's start states), else the ESTIMATE of it that chain_prepare gives -- the device renders from the
// estimate at once and learns the true state when the chain has been linked (eval_candidates then works with the difference).
// kQOk or the QStatus that refuses it.
GPSIQ_HD inline int eval_quantize(const DChan &d, double phase, double delt, int nsamp, gpsiq_qchan_t *q)
{
    // a phase of exactly 1.0 (a wrap that rounded up to one) is phase 0 of the closed form
    if (d.prn > 0 && !(phase >= 0.0 && phase <= 1.0)) { gpsiq_qchan_t z; z.carr_phase = 0; z.carr_step = 0; z.code_frac = 0; z.code_step = 0; z.gain = 0.0; z.nav_bits = 0; z.chip0 = 0; z.icode = 0; z.prn = 0; *q = z; return kQStart; }
    const uint64_t seeded = phase_to_fixed(phase == 1.0 ? 0.0 : phase);
    return quantize_dchan(d, delt, nsamp, &seeded, q);
}

// The patches of a channel whose descriptor qq eval_quantize has accepted, given the TRUE accumulator `start` at the block's first
// sample: the samples where the double path from `start` takes another LUT entry or sign than the closed form of qq -- whose
// carrier may have been seeded from an estimate: it then runs |qq.carr_phase - fixed(start)| units beside the closed form seeded
// from the truth, and the candidate window is that much wider (eval_block, gpsiq_exact.cpp, with a seed override).
// kEvalOk, kEvalHost, or -kQStart (a start outside [0, 1]).  A start of exactly 1.0: the reference goes on from 1.0 and indexes
// its table at 512 for sample 0 -- undecided by construction (the enclosure refuses x0 = 1.0), the host walks it.
template <class Chips, class Emit>
GPSIQ_HD inline int eval_candidates(const DChan &d, const gpsiq_qchan_t &qq, double start, double delt, int nsamp, const Chips &chips, Emit &emit)
{
    const long ns = nsamp;
    if (ns <= 0 || d.prn <= 0) return kEvalOk;
    if (!(start >= 0.0 && start <= 1.0)) return -kQStart;
    const double carr_inc = d.f_carr * delt, code_inc = d.f_code * delt;
    uint64_t seed_off = (phase_to_fixed(start == 1.0 ? 0.0 : start) - qq.carr_phase) & kEvCarrMask;
    if (seed_off > (UINT64_C(1) << (GPSIQ_CARR_FRAC_BITS - 1))) seed_off = (UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - seed_off;
    // drift bounds at the end of the block, in units of the fixed-point formats (gpsiq_exact.cpp)
    const uint64_t w_carr = ((uint64_t) ns << (GPSIQ_CARR_FRAC_BITS - 54)) + (uint64_t) ns / 2 + 4 + seed_off;
    const uint64_t w_code = ((uint64_t) ns << (GPSIQ_CODE_FRAC_BITS - 44)) + (uint64_t) ns / 2 + 4;
    const bool want_c = carr_inc != 0.0 || seed_off != 0;          // a zero addend leaves both paths constant, and equal when seeded alike
    if ((want_c && 2 * w_carr + 1 >= (UINT64_C(1) << (GPSIQ_CARR_FRAC_BITS - 9))) || 2 * w_code + 1 >= (UINT64_C(1) << GPSIQ_CODE_FRAC_BITS)) return kEvalHost;
    uint64_t nc = want_c ? next_candidate(qq.carr_phase, (uint64_t) qq.carr_step, GPSIQ_CARR_FRAC_BITS - 9, w_carr, ns, 0) : kNoHit;
    uint64_t nk = next_candidate(qq.code_frac, qq.code_step, GPSIQ_CODE_FRAC_BITS, w_code, ns, 0);
    if (nc == kNoHit && nk == kNoHit) return kEvalOk;
    DriftLite dc, dk;
    bool have_c = false, have_k = false;
    int count_c = 0, count_k = 0;
    const int icode = dchan_icode(d);
    while (nc != kNoHit || nk != kNoHit) {
        const uint64_t n = nc < nk ? nc : nk;                            // (kNoHit is the largest value)
        const bool in_c = nc == n, in_k = nk == n;
        // fixed-point path (include/gpsiq.h)
        const uint64_t P = (qq.carr_phase + (uint64_t) qq.carr_step * n) & kEvCarrMask;
        const unsigned idx_f = (unsigned) (P >> (GPSIQ_CARR_FRAC_BITS - 9));
        const u128 T = (u128) qq.code_frac + (u128) qq.code_step * n;
        const uint64_t A = (uint64_t) qq.chip0 + (uint64_t) (T >> GPSIQ_CODE_FRAC_BITS);
        const unsigned chip_f = (unsigned) (A % GPSIQ_CA_SEQ_LEN);
        const long per_f = (long) (A / GPSIQ_CA_SEQ_LEN);
        const unsigned neg_f = chips(d.prn, chip_f) ^ dchan_nav_bit(d, ((long) icode + per_f) / 20);
        // double path
        unsigned idx = idx_f, neg_d = neg_f;
        int64_t cell;
        if (in_c) {
            if (++count_c > kCandCap) return kEvalHost;
            if (!have_c) { dc.setup(carr_inc, 1, start); have_c = true; }
            if (!dc.cell_at((long) n, &cell)) return kEvalHost;
            idx = (unsigned) (cell & 511);
            nc = next_candidate(qq.carr_phase, (uint64_t) qq.carr_step, GPSIQ_CARR_FRAC_BITS - 9, w_carr, ns, (long) n + 1);
        }
        if (in_k) {
            if (++count_k > kCandCap) return kEvalHost;
            if (!have_k) { dk.setup(code_inc, 0, d.code_phase); have_k = true; }
            if (!dk.cell_at((long) n, &cell) || cell < 0) return kEvalHost;
            const unsigned chip_d = (unsigned) (cell % GPSIQ_CA_SEQ_LEN);
            const long per_d = (long) (cell / GPSIQ_CA_SEQ_LEN);
            neg_d = chips(d.prn, chip_d) ^ dchan_nav_bit(d, ((long) icode + per_d) / 20);       // gps.c:2791-2811
            nk = next_candidate(qq.code_frac, qq.code_step, GPSIQ_CODE_FRAC_BITS, w_code, ns, (long) n + 1);
        }
        if (idx != idx_f || neg_d != neg_f) emit((uint32_t) n, (uint16_t) idx, (uint8_t) neg_d);
    }
    return kEvalOk;
}

// chips(prn, i): chip i (0 / 1) of the satellite's C/A code; emit(sample, lut, neg): one patch of this channel.
// q receives the descriptor seeded from `start` (also when the host is asked to redo the patches: the descriptor is the same).
// est: what the descriptor is seeded from (== start on the host path)
template <class Chips, class Emit>
GPSIQ_HD inline int eval_chan(const DChan &d, double start, double est, double delt, int nsamp, const Chips &chips, Emit &emit, gpsiq_qchan_t *q)
{
    const int qs = eval_quantize(d, est, delt, nsamp, q);
    if (qs != kQOk) return -qs;
    return eval_candidates(d, *q, start, delt, nsamp, chips, emit);
}

// ---- level 2 of the chain as a scan ------------------------------------------------------------------------------------
// A block's map moves the offset d = (x - xs)/U of its true start to that of the next block's:
//     d' = d + cum[p(d)] + J,   p(d) = (d >> (grid - 1)) & 1,   J = (e - xs')/U  (exact, or the next block does not link),
// a translation that depends on d mod 4 only.  Such maps compose, T_ab[r] = T_a[r] + T_b[(r + T_a[r]) & 3], and a block that
// seeds its slot (gps.c:2208-2214) or is unused is a constant map: one segmented inclusive scan down the timeline gives every
// block's d, after which every block checks on its own what link_block checks (range, grid, parity branch, the two exactness
// tests).  The first block of a slot that fails one leaves the blocks after it (up to the next seed) to the host walker.
struct Link {
    int64_t t[4];      // the offset after this element for an offset r (mod 4) before it; abs: whatever was before
    int32_t abs;       // 1: constant map (a seed, an unused slot)
    int32_t pad;
};
GPSIQ_HD inline Link link_identity() { Link l; l.t[0] = l.t[1] = l.t[2] = l.t[3] = 0; l.abs = 0; l.pad = 0; return l; }
// first a, then b
GPSIQ_HD inline Link link_compose(const Link &a, const Link &b)
{
    if (b.abs) return b;
    Link o;
    o.abs = a.abs; o.pad = 0;
    for (int r = 0; r < 4; ++r) {
        const int64_t v = a.t[r];                                       // a.abs: the offset itself; else the translation
        const int64_t at = a.abs ? v : (int64_t) r + v;
        o.t[r] = v + b.t[at & 3];
    }
    return o;
}
// the offset an element leaves behind, entered with offset d (ignored by a constant map)
GPSIQ_HD inline int64_t link_apply(const Link &a, int64_t d) { return a.abs ? a.t[0] : d + a.t[d & 3]; }

// The element of block b: how the offset at block b follows from the offset at block b - 1.
//   seed / unused: the constant d0 (a seed's own (carr_phase - xs)/U, normally 0).
//   else: through the map of block b - 1 (prev) and the join J to this block's representative.
// ok false: no usable element (the map before does not exist, the join is not a whole number of units): the block's start is
// still the chain's business -- it is the end of the block before -- but its offset is unknown, so it cannot link.
GPSIQ_HD inline Link link_element(bool seed_or_unused, int64_t d0, const lane::Rec &prev, double xs, bool *ok)
{
    Link l = link_identity();
    *ok = true;
    if (seed_or_unused) { l.abs = 1; l.t[0] = l.t[1] = l.t[2] = l.t[3] = d0; return l; }
    int64_t J;
    const int64_t grid = prev.info & 0xff;
    if (!prev.ok || grid < 1 || grid > 2 || !lane::exact_units(prev.e, xs, &J)) { *ok = false; l.abs = 1; return l; }
    for (int r = 0; r < 4; ++r) l.t[r] = prev.cum[(r >> (grid - 1)) & 1] + J;
    return l;
}
// what link_block checks, with the offset known: the block links, *next = the accumulator after it
GPSIQ_HD inline bool link_check(const lane::Rec &r, int64_t d, double *next)
{
    if (!r.ok) return false;
    const int64_t grid = r.info & 0xff;
    if (d < r.lo || d > r.hi || grid < 1 || grid > 2 || (d & (grid - 1))) return false;
    const int p = (int) ((d >> (grid - 1)) & 1);
    if (!(r.ok & (1 << p))) return false;
    double y;
    if (!lane::exact_shift(r.e, d + r.cum[p], &y) || !(y >= 0.0 && y < 1.0)) return false;
    *next = y;
    return true;
}

}  // namespace ev
}  // namespace gpsiq
#endif
