// gpsiq_chain_kernels.hip -- gfx950 kernels of the time-parallel carrier chain of GPSIQ_NCO_REFERENCE (csrc/gpsiq_lane.h has
// the method, csrc/gpsiq_chain.cpp the host twin and level 2).  Reference lines: gps.c:2821-2826 (the accumulator),
// gps.c:2208-2214 (re-seeding a slot).
//
//   chain_prepare   one workgroup per slot: two segmented scans down the timeline -- the exact phase R_b = x_0 + sum nsamp*c_j
//                   (128-bit integers, the wrap is the overflow) and the modelled rounding drift -- give every block an
//                   estimate of the accumulator at its first sample (Prep, 32 bytes per block and slot).
//   chain_lanes     level 1.  One LANE per stretch of a block (up to kSeg stretches per block): the lane places the last wrap
//                   before its stretch from the estimate, starts from a representative post-wrap state there and walks the
//                   reference's double additions -- plain FP64 adds at the binade crossings and wraps, integer jumps over the
//                   steady runs, exactly the host's walker (gpsiq_walk.h) -- noting how far its start may move.  A workgroup
//                   holds 256 / kSeg consecutive blocks of ONE slot: their addends differ by parts per million, so the lanes of
//                   a wave take the same binades in the same order and diverge only at the ends of runs.  The walkers'
//                   per-binade tables (one 64-bit division per binade) are built once per block in LDS, the divisions spread
//                   over the workgroup's threads; the stretches of a block are joined by its first lane.
// No MFMA (no contraction), no LDS staging of anything big: 56 bytes out per block and slot; the kernel is latency-bound on
// dependent FP64 / 64-bit integer VALU chains, which is why the work is cut into as many lanes as there are wraps to spare.
#include <hip/hip_runtime.h>

#include "gpsiq_internal.h"
#include "gpsiq_lane.h"

namespace gpsiq {

using lane::Prep;
using lane::Rec;
using lane::Stretch;
using lane::Walker;
using lane::u128;

constexpr int kPrepThreads = 1024;          // blocks of one slot per round of chain_prepare (one block per thread, sixteen waves)
constexpr int kLaneThreads = 256;
constexpr int kTabMinSeg = 16;            // lanes per block from which a block gets its table of cycles (measured: with 8 lanes the four
                                          // table walks per lane cost more than they save, 0.65 against 0.50 ms per 2 000 x 16 blocks)

struct ScanR { u128 v; int flag; };     // f(R) = flag ? v : R + v
struct ScanD { double v; int flag; };
// Segmented scans over the threads of a workgroup: inside a wave with cross-lane moves, the waves' totals through LDS -- two
// barriers per scan (a Hillis-Steele scan over LDS took twenty, and 256 blocks per round: 0.1-0.2 ms per 4 000 blocks, in front of
// everything else of the chain).  first p, then a: a segment start in a hides p.
__device__ inline ScanR scan_join(const ScanR &p, ScanR a) { if (!a.flag) { a.v = p.v + a.v; a.flag = p.flag; } return a; }
__device__ inline ScanD scan_join(const ScanD &p, ScanD a) { if (!a.flag) { a.v = p.v + a.v; a.flag = p.flag; } return a; }
__device__ inline ScanR scan_up(const ScanR &x, int off)
{
    ScanR o;
    const uint64_t hi = __shfl_up((uint64_t) (x.v >> 64), off, 64), lo = __shfl_up((uint64_t) x.v, off, 64);
    o.v = ((u128) hi << 64) | lo; o.flag = __shfl_up(x.flag, off, 64);
    return o;
}
__device__ inline ScanD scan_up(const ScanD &x, int off) { ScanD o; o.v = __shfl_up(x.v, off, 64); o.flag = __shfl_up(x.flag, off, 64); return o; }
// -> *before: everything before this thread (identity for thread 0), *total: all threads.  s_wave: one entry per wave.
template <class S>
__device__ inline void block_scan(S mine, S *s_wave, S *before, S *total)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nwaves = (int) (blockDim.x >> 6);
    S inc = mine;
    for (int off = 1; off < 64; off <<= 1) {
        const S o = scan_up(inc, off);
        if (lane >= off) inc = scan_join(o, inc);
    }
    S exc = scan_up(inc, 1);
    if (lane == 0) { exc.v = 0; exc.flag = 0; }
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    S acc; acc.v = 0; acc.flag = 0;
    S mine_before = acc;
    for (int w = 0; w < nwaves; ++w) {
        if (w == wave) mine_before = acc;
        acc = scan_join(acc, s_wave[w]);
    }
    *before = scan_join(mine_before, exc);
    *total = acc;
    __syncthreads();                                    // s_wave is free again
}

// ---- prepare ---------------------------------------------------------------------------------------------------------
// est layout as on the host: start[nchan] may be null (the timeline begins here).
// in: rows of `stride` bytes that begin with a gpsiq_chain_in_t (24: the type itself; 64: ev::DChan, gpsiq_eval.h)
__global__ __launch_bounds__(kPrepThreads) void chain_prepare(const char *__restrict__ in_rows, int stride, int nblocks, int nchan, double delt, int nsamp,
                                                              const gpsiq_chain_est_t *__restrict__ start, Prep *__restrict__ prep,
                                                              double *__restrict__ c_before, gpsiq_chain_est_t *__restrict__ end)
{
    auto in_at = [&](size_t k) -> gpsiq_chain_in_t { return *reinterpret_cast<const gpsiq_chain_in_t *>(in_rows + k * (size_t) stride); };
    __shared__ ScanR s_wr[kPrepThreads / 64];
    __shared__ ScanD s_wd[kPrepThreads / 64];
    __shared__ int s_any_seed;
    const int i = blockIdx.x, tid = threadIdx.x;
    // the estimator before block 0
    u128 Rc = 0;
    double Dc = 0.0, carr0 = 0.0, f_prev0 = 0.0;
    int prn0 = 0;
    bool exact0 = false;
    if (start) {
        const gpsiq_chain_est_t e = start[i];
        Rc = ((u128) e.r_hi << 64) | e.r_lo; Dc = e.drift; carr0 = e.carr; f_prev0 = e.f_carr; prn0 = e.prn; exact0 = (e.flags & GPSIQ_CHAIN_EXACT) != 0;
    }
    if (tid == 0) { c_before[i] = prn0 > 0 ? f_prev0 * delt : 0.0; s_any_seed = 0; }
    int last_prn = prn0;
    double last_f = f_prev0;
    __syncthreads();
    for (int base = 0; base < nblocks; base += kPrepThreads) {
        const int b = base + tid;
        const bool live = b < nblocks;
        gpsiq_chain_in_t d = {0.0, 0.0, 0, 0};
        int prev_prn = 0;
        if (live) {
            d = in_at((size_t) b * nchan + i);
            prev_prn = b > 0 ? in_at((size_t) (b - 1) * nchan + i).prn : prn0;
            if (prev_prn < 0) prev_prn = 0;
        }
        const bool active = live && d.prn > 0;
        const bool cont = active && b == 0 && exact0 && prev_prn == d.prn;       // continues a walked timeline: the accumulator itself
        const bool seed = active && (prev_prn != d.prn || cont);
        const double c = active ? d.f_carr * delt : 0.0;
        const double x_seed = cont ? carr0 : d.carr_phase;
        const u128 adv = active ? lane::advance_units(c, nsamp) : (u128) 0;
        // R after this block as a function of R before it
        ScanR mine;
        mine.flag = seed ? 1 : 0;
        mine.v = seed ? lane::phase_units(x_seed) + adv : adv;
        ScanR pr, er;                                   // the blocks before this one inside the round; the whole round
        block_scan(mine, s_wr, &pr, &er);
        // R at this block's first sample
        const u128 R0 = seed ? lane::phase_units(x_seed) : pr.flag ? pr.v : Rc + pr.v;
        const double r0 = lane::units_to_double(R0);
        double core = 0.0;
        if (active && __builtin_fabs(c) < 0.5) core = lane::drift_core(c, seed ? (x_seed < 1.0 ? x_seed : 0.0) : r0, nsamp);
        ScanD md;
        md.flag = seed ? 1 : 0;
        md.v = core;
        ScanD pd, ed;
        block_scan(md, s_wd, &pd, &ed);
        const double D0 = seed ? 0.0 : pd.flag ? pd.v : Dc + pd.v;
        if (live) {
            Prep p = {0.0, 0.0, 0.0, lane::kSkip, 0};
            if (active) {
                p.c = c;
                p.est = seed ? x_seed : lane::wrap01(r0 + D0);
                p.core = core;
                p.flags = seed ? lane::kSeed : 0;
            }
            prep[(size_t) b * nchan + i] = p;
        }
        // the round's carry
        if (seed) atomicOr(&s_any_seed, 1);
        const int n_here = nblocks - base < kPrepThreads ? nblocks - base : kPrepThreads;
        __syncthreads();
        Rc = er.flag ? er.v : Rc + er.v;
        Dc = ed.flag ? ed.v : Dc + ed.v;
        // the satellite and Doppler of the chunk's last block, for `end` (every thread reads them: two loads)
        {
            const gpsiq_chain_in_t l = in_at((size_t) (base + n_here - 1) * nchan + i);
            last_prn = l.prn > 0 ? l.prn : 0;
            last_f = l.f_carr;
        }
        __syncthreads();
    }
    if (end && tid == 0) {
        gpsiq_chain_est_t e;
        e.r_hi = (uint64_t) (Rc >> 64); e.r_lo = (uint64_t) Rc; e.drift = Dc; e.carr = 0.0; e.f_carr = last_f; e.prn = last_prn;
        e.flags = s_any_seed ? GPSIQ_CHAIN_RESEEDED : 0; e.first_prn = 0; e.reserved = 0;
        end[i] = e;
    }
}

// ---- lanes -----------------------------------------------------------------------------------------------------------
template <int kSeg>
__global__ __launch_bounds__(kLaneThreads) void chain_lanes(const Prep *__restrict__ prep, const double *__restrict__ c_before,
                                                            int nblocks, int nchan, int nsamp, int max_seg, Rec *__restrict__ rec)
{
    constexpr int kBlocks = kLaneThreads / kSeg;               // blocks of one slot per workgroup
    // walker[k]: the addend of block b0 - 1 + k (raw storage: a __shared__ object may not have initialisers; setup_head writes every scalar)
    __shared__ __attribute__((aligned(16))) unsigned char walker_raw[sizeof(Walker) * (kBlocks + 1)];
    Walker *walker = reinterpret_cast<Walker *>(walker_raw);
    constexpr int kEnt = kSeg >= kTabMinSeg ? lane::kEntriesHost : 0;  // entries of a block's table of cycles (0: every cycle walked)
    __shared__ lane::Cycle cycles[kEnt ? kBlocks * kEnt : 1];         // cycles[blk * kEnt + k]: entry k of block blk's table
    __shared__ int usable[kBlocks + 1];
    const int groups = (nblocks + kBlocks - 1) / kBlocks;
    const int i = blockIdx.x / groups, b0 = (blockIdx.x % groups) * kBlocks;
    const int tid = threadIdx.x;
    // the walkers' scalars: one thread per addend
    if (tid <= kBlocks) {
        const int b = b0 - 1 + tid;
        double c = 0.0;
        if (b < 0) c = c_before[i];
        else if (b < nblocks) { const Prep p = prep[(size_t) b * nchan + i]; c = (p.flags & lane::kSkip) ? 0.0 : p.c; }
        const bool ok = c != 0.0 && __builtin_fabs(c) < 0.5;
        walker[tid].setup_head(ok ? c : 0.25);              // an addend the walk does not take: general, never walked
        usable[tid] = ok && !walker[tid].general;
    }
    __syncthreads();
    // their tables: one (addend, binade) pair per thread and round -- a 64-bit division each
    for (int q = tid; q < (kBlocks + 1) * lane::kTab; q += kLaneThreads) {
        const int w = q / lane::kTab, s = q % lane::kTab;
        if (usable[w] && s > Walker::kLow && s <= (int) (1022 - walker[w].ec)) walker[w].setup_piece(s);
    }
    __syncthreads();
    const int blk = tid / kSeg, t = tid % kSeg, b = b0 + blk;
    // the block's table of whole cycles: every lane of the block walks one, from post-wrap states spread over their range
    const bool live = b < nblocks && usable[blk + 1];
    Prep p = {0.0, 0.0, 0.0, lane::kSkip, 0};
    if (live) p = prep[(size_t) b * nchan + i];
    // (kEnt entries whatever the number of lanes: a thinner sample misses a tenth of the look-ups, and a wave waits for its unluckiest
    // lane -- with fewer lanes per block each walks kEnt / kSeg cycles, and a wave holds more blocks)
    const bool use_tab = live && kEnt > 0 && __builtin_fabs(p.c) * (double) nsamp > 2.0 * kEnt;
    if (use_tab) {
        for (int k = t; k < kEnt; k += kSeg) {
            lane::Cycle e;
            lane::build_cycle(walker[blk + 1], k, kEnt, &e);
            cycles[kEnt ? blk * kEnt + k : 0] = e;
        }
    }
    __syncthreads();
    int nseg = 0;
    Stretch st;
    st.r_in = 0.0; st.x_out = 0.0; st.x_end = 0.0; st.lo = 0; st.hi = 0; st.n_in = -1; st.n_out = -1; st.ok = 0; st.why = lane::kWhyAddend; st.sigma = 0; st.pad = 0;
    if (live) {
        nseg = lane::stretches(p.c, nsamp, max_seg < kSeg ? max_seg : kSeg);
        if (t < nseg) {
            const Walker *prev = (!(p.flags & lane::kSeed) && usable[blk]) ? &walker[blk] : nullptr;
            lane::walk_stretch(walker[blk + 1], prev, p, nsamp, t, nseg, use_tab ? &cycles[kEnt ? blk * kEnt : 0] : nullptr, kEnt, &st);
        }
    }
    // The stretches of a block joined into its map: a scan over the block's kSeg lanes (gpsiq_lane.h, "the join as a scan") --
    // every lane takes part in the cross-lane moves, lanes without a stretch as identities.
    const bool mine = live && t < nseg;
    const bool neg = walker[blk + 1].neg;
    const int64_t grid = neg ? 1 : 2;
    lane::JoinLane jl;
    jl.el = lane::join_identity(); jl.D = 0; jl.fail = 0; jl.pad = 0;
    Stretch before = st;                                       // of the stretch before, the join reads where and in which state it ended
    before.x_out = __shfl_up(st.x_out, 1, kSeg); before.n_out = __shfl_up(st.n_out, 1, kSeg);
    if (mine) jl = lane::join_lane(before, st, t, grid);
    lane::JoinMap inc = jl.el;
    for (int off = 1; off < kSeg; off <<= 1) {
        lane::JoinMap o;
        o.t0 = __shfl_up(inc.t0, off, kSeg); o.t1 = __shfl_up(inc.t1, off, kSeg);
        if (t >= off) inc = lane::join_compose(o, inc);
    }
    lane::JoinMap exc;
    exc.t0 = __shfl_up(inc.t0, 1, kSeg); exc.t1 = __shfl_up(inc.t1, 1, kSeg);
    if (t == 0) exc = lane::join_identity();
    lane::JoinTerm q;
    q.l = INT64_MIN; q.h = INT64_MAX; q.mask = 3; q.pad = 0;
    if (mine) q = lane::join_term(exc, jl, st, t, grid);
    int closed = q.mask;                                       // both branches closed from this stretch on: the prefix AND of the masks
    for (int off = 1; off < kSeg; off <<= 1) {
        const int o = __shfl_up(closed, off, kSeg);
        if (t >= off) closed &= o;
    }
    int64_t lo = q.l, hi = q.h;
    int mask = q.mask, first_fail = mine && jl.fail ? (t << 8 | jl.fail) : INT32_MAX, first_closed = mine && !closed ? t : INT32_MAX;
    for (int off = kSeg / 2; off >= 1; off >>= 1) {
        const int64_t ol = __shfl_xor(lo, off, kSeg), oh = __shfl_xor(hi, off, kSeg);
        const int om = __shfl_xor(mask, off, kSeg), of = __shfl_xor(first_fail, off, kSeg), oc = __shfl_xor(first_closed, off, kSeg);
        lo = ol > lo ? ol : lo; hi = oh < hi ? oh : hi; mask &= om;
        first_fail = of < first_fail ? of : first_fail; first_closed = oc < first_closed ? oc : first_closed;
    }
    const int last = nseg > 0 ? nseg - 1 : 0;
    lane::JoinMap all;
    all.t0 = __shfl(inc.t0, last, kSeg); all.t1 = __shfl(inc.t1, last, kSeg);
    const double x_end = __shfl(st.x_end, last, kSeg);
    if (b < nblocks && t == 0) {
        Rec r;
        r.xs = 0.0; r.e = 0.0; r.cum[0] = 0; r.cum[1] = 0; r.lo = 0; r.hi = 0; r.ok = 0; r.info = 0;
        if (nseg > 0) lane::join_finish(st, x_end, neg, all, lo, hi, mask, first_fail, first_closed, &r);
        rec[(size_t) b * nchan + i] = r;
    }
}

// which = 1: chain_prepare only, 2: chain_lanes only, 3: both
hipError_t launch_chain(const void *d_in, int in_stride, int nblocks, int nchan, double delt, int nsamp, const gpsiq_chain_est_t *d_start,
                        int max_seg, void *d_prep_, double *d_c_before, gpsiq_chain_est_t *d_end, void *d_maps, hipStream_t stream, int which)
{
    if (nblocks <= 0) return hipSuccess;
    Prep *d_prep = static_cast<Prep *>(d_prep_);
    Rec *d_rec = static_cast<Rec *>(d_maps);
    if (which & 1) hipLaunchKernelGGL(chain_prepare, dim3((unsigned) nchan), dim3(kPrepThreads), 0, stream, static_cast<const char *>(d_in), in_stride, nblocks, nchan, delt, nsamp, d_start, d_prep, d_c_before, d_end);
#define GPSIQ_LANES(S)                                                                                                    \
    hipLaunchKernelGGL(chain_lanes<S>, dim3((unsigned) (nchan * ((nblocks + kLaneThreads / S - 1) / (kLaneThreads / S)))), dim3(kLaneThreads), 0, stream, \
                       d_prep, d_c_before, nblocks, nchan, nsamp, max_seg, d_rec)
    if (max_seg < 1) max_seg = 1;
    if (!(which & 2)) return hipGetLastError();
    if (max_seg > 16) GPSIQ_LANES(32);
    else if (max_seg > 8) GPSIQ_LANES(16);
    else if (max_seg > 4) GPSIQ_LANES(8);
    else GPSIQ_LANES(4);                                        // fewer stretches than lanes per block: the spare lanes idle
#undef GPSIQ_LANES
    return hipGetLastError();
}

}  // namespace gpsiq
