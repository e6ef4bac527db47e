// gpsiq_eval_kernels.hip -- gfx950 kernels that take a batch of channel descriptors (gpsiq_chan_t, what the reference's host
// code leaves at gps.c:2766) to device-resident quantised descriptors without a host stage in between:
//
//   pack_raw        gpsiq_chan_t rows the device can read (HBM, or page-locked host memory) -> ev::DChan, 64 bytes per block and
//                   channel, + the launch parameters of the synthesis kernel (largest code step, most active channels, largest sum
//                   of amplitudes) reduced on the way.  Descriptors in pageable host memory are packed by the host pool instead.
//   chain_link_scan GPSIQ_NCO_REFERENCE: level 2 of the time-parallel carrier chain (gps.c:2821-2826 through the certified maps of
//                   gpsiq_chain_kernels.hip) as a segmented scan down each slot: a map is an integer translation that depends on the
//                   offset mod 4 only (gpsiq_eval.h, Link), so the chain that the host walked block after block is three LDS scans
//                   per 256 blocks.  Writes every block's start state into its DChan; a slot with a block whose map does not
//                   apply is counted, and the host walker takes that slot (gpsiq_device.cpp).
//   quantize_est    one lane per (block, channel): the quantiser (gps.c:2033-2064 outputs as inputs) with the carrier seeded from
//                   chain_prepare's ESTIMATE of the block's start state; the descriptors compacted per block (active channels first)
//                   straight into the set the synthesis kernel reads -- which starts behind this kernel, not behind the chain.
//   eval_blocks     when the chain has been linked: from the TRUE start states the Euclid descent for the candidate samples (window
//                   widened by estimate - truth), the drift enclosure, the patches appended to the call's list; what the enclosure
//                   cannot decide appended to the host walker's list (eval_chan of gpsiq_eval.h).
//   quantize_fixed  GPSIQ_NCO_FIXED: the quantiser from each block's own carr_phase, then the exact carrier prefix
//   carry_prefix    p_{k+1} = p_k + nsamp*step_k (mod 2^59) down each slot as a segmented scan (what chain_carrier does on the host).
// No MFMA (no contraction), nothing HBM-bound: 64 bytes in and 48 out per block and channel; the work is 64/128-bit integer and
// FP64 scalar-style arithmetic, one lane per channel, sixteen lanes per block so that a block's channels compact with one ballot.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "gpsiq_internal.h"
#include "gpsiq_eval.h"
#include "gpsiq_evalctl.h"

namespace gpsiq {

using ev::DChan;
using ev::Link;
using lane::Rec;

constexpr int kPackThreads = 128;           // (x 296 bytes of rows in LDS: 37 KB)
constexpr int kLinkThreads = 256;
constexpr int kEvalThreads = 128;

// reductions over the 16 lanes of a block (a wave holds four blocks)
__device__ inline long long sum16(long long v)
{
    for (int off = 8; off >= 1; off >>= 1) v += __shfl_xor(v, off, 16);
    return v;
}
__device__ inline unsigned long long wave_max_u64(unsigned long long v)
{
    for (int off = 32; off >= 1; off >>= 1) { const unsigned long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}
__device__ inline long long wave_max_i64(long long v)
{
    for (int off = 32; off >= 1; off >>= 1) { const long long o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

// the launch parameters one (block, channel) contributes; every lane of a wave calls this (inactive lanes with zeros)
__device__ inline void reduce_params(bool counts, unsigned long long code_step, long long amp, EvalCtrl *ctrl)
{
    const long long blk_amp = sum16(counts ? amp : 0);
    const long long blk_active = sum16(counts ? 1 : 0);
    const unsigned long long ms = wave_max_u64(counts ? code_step : 0);
    const long long ma = wave_max_i64(blk_amp), mc = wave_max_i64(blk_active);
    // (a thousand waves' atomics on three words take their turns at the L2: 35 of pack_raw's 43 us for 66 000 rows.  A wave looks first:
    // a maximum that is already there needs no atomic, and a stale look only costs one that was not needed)
    if ((threadIdx.x & 63) == 0) {
        if (ms > __hip_atomic_load(&ctrl->max_code_step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctrl->max_code_step, ms);
        if (ma > __hip_atomic_load(&ctrl->max_amp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctrl->max_amp, ma);
        if ((int) mc > __hip_atomic_load(&ctrl->max_active, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(&ctrl->max_active, (int) mc);
    }
}

// ---- pack ----------------------------------------------------------------------------------------------------------------
// A gpsiq_chan_t is 296 bytes: lanes that each read their own row make every load touch 64 cache lines (43 us per 66 000 rows).  The
// workgroup's rows are contiguous in memory, so they come in as one coalesced stream of 8-byte words into LDS, and a lane takes what it
// needs of its row from there (the struct read through a pointer to LDS: the same ev::pack_chan as on the host).
constexpr int kPackRows = kPackThreads;                        // rows a workgroup stages at most (16 lanes per block, nchan <= 16 of them live)
constexpr int kRowWords = (int) (sizeof(gpsiq_chan_t) / 8);    // 37
static_assert(sizeof(gpsiq_chan_t) % 8 == 0, "rows are streamed as 8-byte words");
__global__ __launch_bounds__(kPackThreads) void pack_raw(const gpsiq_chan_t *__restrict__ ch, int nblocks, int nchan, double delt,
                                                         DChan *__restrict__ out, EvalCtrl *__restrict__ ctrl)
{
    __shared__ __attribute__((aligned(16))) uint64_t rows[kPackRows * kRowWords];
    constexpr int kBlocksPerWg = kPackThreads / 16;
    const int blk0 = (int) blockIdx.x * kBlocksPerWg;
    const int nblk = nblocks - blk0 < kBlocksPerWg ? nblocks - blk0 : kBlocksPerWg;
    const size_t first = (size_t) blk0 * nchan;
    const int words = nblk > 0 ? nblk * nchan * kRowWords : 0;
    const uint64_t *src = reinterpret_cast<const uint64_t *>(ch + first);
    // (sixteen loads in flight per lane before the first one is waited for: three round trips to memory per workgroup, not thirty-seven)
    constexpr int kInFlight = 16;
    for (int w0 = threadIdx.x; w0 < words; w0 += kPackThreads * kInFlight) {
        uint64_t t[kInFlight];
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) { const int w = w0 + j * kPackThreads; t[j] = w < words ? src[w] : 0; }
#pragma unroll
        for (int j = 0; j < kInFlight; ++j) { const int w = w0 + j * kPackThreads; if (w < words) rows[w] = t[j]; }
    }
    __syncthreads();
    const int lb = (int) (threadIdx.x >> 4), i = (int) (threadIdx.x & 15);
    const int b = blk0 + lb;
    const bool live = lb < nblk && i < nchan;
    DChan d;
    d.f_carr = 0.0; d.carr_phase = 0.0; d.prn = 0; d.pos = 0; d.f_code = 0.0; d.code_phase = 0.0; d.gain = 0.0; d.nav = 0; d.start = 0.0;
    if (live) {
        ev::pack_chan(*reinterpret_cast<const gpsiq_chan_t *>(&rows[(size_t) (lb * nchan + i) * kRowWords]), &d);
        out[(size_t) b * nchan + i] = d;
    }
    const double code_inc = d.f_code * delt;
    const bool counts = live && d.prn > 0 && d.prn <= 32 && code_inc > 0.0 && code_inc < 2.0 && d.gain > -ev::kEvMaxGain && d.gain < ev::kEvMaxGain;
    reduce_params(counts, counts ? (unsigned long long) (long long) __builtin_rint(code_inc * 0x1p56) : 0ull,
                  counts ? (long long) (250.0 * __builtin_fabs(d.gain)) : 0ll, ctrl);
}

// ---- level 2 of the carrier chain ---------------------------------------------------------------------------------------------
// One workgroup per slot, 256 blocks at a time, the carry (offset, accumulator, whether it is known) handed from chunk to chunk
// and from piece to piece of a batch (carry[]).  chan / rec: the whole timeline; this launch covers blocks [b0, b0 + nb).
__global__ __launch_bounds__(kLinkThreads) void chain_link_scan(DChan *__restrict__ chan, const Rec *__restrict__ rec, int b0, int nb, int nchan,
                                                                double delt, LinkCarry *__restrict__ carry, EvalCtrl *__restrict__ ctrl, int piece)
{
    __shared__ Link sl[kLinkThreads];
    __shared__ int s_seed[kLinkThreads], s_fail[kLinkThreads];
    __shared__ double s_next[kLinkThreads];
    __shared__ LinkCarry s_carry;
    __shared__ unsigned s_unknown, s_linked;
    const int i = blockIdx.x, tid = threadIdx.x;
    if (tid == 0) {
        LinkCarry c;
        c.d = 0; c.y = 0.0; c.prn = 0; c.known = 0;
        if (b0 > 0) c = carry[i];
        s_carry = c;
        s_unknown = 0; s_linked = 0;
    }
    __syncthreads();
    for (int base = 0; base < nb; base += kLinkThreads) {
        const int b = b0 + base + tid;
        const bool live = base + tid < nb;
        const LinkCarry cy = s_carry;
        const size_t at = (size_t) b * nchan + i;
        double carr_phase = 0.0, f_carr = 0.0;
        int prn = 0, prev_prn = 0;
        Rec r;
        r.xs = 0.0; r.e = 0.0; r.cum[0] = 0; r.cum[1] = 0; r.lo = 0; r.hi = 0; r.ok = 0; r.info = 0;
        if (live) {
            const DChan d = chan[at];
            carr_phase = d.carr_phase; f_carr = d.f_carr; prn = d.prn > 0 ? d.prn : 0;
            r = rec[at];
            if (b > 0) { const int p = chan[at - nchan].prn; prev_prn = p > 0 ? p : 0; }
        }
        const bool active = live && prn > 0;
        const bool seed = active && (b == 0 || prev_prn != prn);
        bool el_ok = true;
        Link el = ev::link_identity();
        if (live) {
            if (!active) el = ev::link_element(true, 0, r, 0.0, &el_ok);
            else if (seed) {
                int64_t d0 = 0;
                const bool ex = lane::exact_units(carr_phase, r.xs, &d0);
                el = ev::link_element(true, ex ? d0 : 0, r, 0.0, &el_ok);
                el_ok = ex;
            } else el = ev::link_element(false, 0, rec[at - nchan], r.xs, &el_ok);
        }
        // inclusive scan of the maps
        sl[tid] = el;
        __syncthreads();
        for (int off = 1; off < kLinkThreads; off <<= 1) {
            Link a = sl[tid], p;
            const bool has = tid >= off;
            if (has) p = sl[tid - off];
            __syncthreads();
            if (has) sl[tid] = ev::link_compose(p, a);
            __syncthreads();
        }
        const int64_t dd = ev::link_apply(sl[tid], cy.d);
        double next = 0.0;
        const bool linked = active && el_ok && __builtin_fabs(f_carr * delt) < 0.5 && ev::link_check(r, dd, &next);
        // which blocks' start states are known: no block failed to link between the slot's last seed and this one
        s_seed[tid] = (live && (seed || !active)) ? tid : -1;
        s_fail[tid] = (active && !linked) ? tid : -1;
        s_next[tid] = next;
        __syncthreads();
        for (int off = 1; off < kLinkThreads; off <<= 1) {
            int a = s_seed[tid], f = s_fail[tid], pa = -1, pf = -1;
            const bool has = tid >= off;
            if (has) { pa = s_seed[tid - off]; pf = s_fail[tid - off]; }
            __syncthreads();
            if (has) { s_seed[tid] = pa > a ? pa : a; s_fail[tid] = pf > f ? pf : f; }
            __syncthreads();
        }
        // the chunk's carry is a seed at -1 when it is known, a failed block at -1 when it is not
        int last_seed = s_seed[tid], last_fail = tid > 0 ? s_fail[tid - 1] : -2;
        if (last_seed < 0) last_seed = -1;
        if (last_fail < 0) last_fail = cy.known ? -2 : -1;
        const bool known = last_fail < last_seed;
        if (live) {
            double x = 0.0;
            if (active) x = seed ? carr_phase : (tid > 0 ? s_next[tid - 1] : cy.y);
            if (!active || known) chan[at].start = x;
            if (active && !(known && linked)) atomicAdd(&s_unknown, 1u);      // the host walker's: this block, or one before it, does not link
            if (active && known && linked) atomicAdd(&s_linked, 1u);
        }
        const int n_here = nb - base < kLinkThreads ? nb - base : kLinkThreads;
        __syncthreads();
        if (tid == n_here - 1) {
            LinkCarry c;
            c.d = dd; c.y = next; c.prn = prn; c.known = (active && known && linked) ? 1 : 0;
            s_carry = c;
        }
        __syncthreads();
    }
    if (tid == 0) {
        carry[i] = s_carry;
        ctrl->unknown[piece][i] = (int) s_unknown;
        atomicAdd(&ctrl->linked, s_linked);
    }
}

// ---- repair: one slot's column out of the row-major arrays, and its start states back in ---------------------------------------
// (a slot with a block whose map does not apply is relinked by the host walker: it needs that slot's maps and chain inputs only)
__global__ __launch_bounds__(256) void gather_slot(const DChan *__restrict__ chan, const Rec *__restrict__ rec, int b0, int nb, int nchan, int slot,
                                                   EvalSlotRow *__restrict__ out)
{
    const int k = (int) (blockIdx.x * 256u + threadIdx.x);
    if (k >= nb) return;
    const size_t at = (size_t) (b0 + k) * nchan + slot;
    const DChan d = chan[at];
    EvalSlotRow r;
    r.f_carr = d.f_carr; r.carr_phase = d.carr_phase; r.start = d.start; r.prn = d.prn; r.pad = 0;
    const Rec m = rec[at];
    r.map.xs = m.xs; r.map.e = m.e; r.map.cum[0] = m.cum[0]; r.map.cum[1] = m.cum[1]; r.map.lo = m.lo; r.map.hi = m.hi; r.map.ok = m.ok; r.map.info = m.info;
    out[k] = r;
}
__global__ __launch_bounds__(256) void scatter_starts(DChan *__restrict__ chan, int b0, int nb, int nchan, int slot, const double *__restrict__ starts)
{
    const int k = (int) (blockIdx.x * 256u + threadIdx.x);
    if (k < nb) chan[(size_t) (b0 + k) * nchan + slot].start = starts[k];
}

// ---- the evaluation -----------------------------------------------------------------------------------------------------
struct DevChips {
    const DeviceTables *tab;
    __device__ unsigned operator()(int prn, unsigned chip) const { return (tab->prn_ext[prn - 1][chip >> 5] >> (chip & 31)) & 1u; }
};
struct DevEmit {
    gpsiq_patch_t *patches; unsigned cap; EvalCtrl *ctrl; uint32_t block; uint8_t slot;
    __device__ void operator()(uint32_t sample, uint16_t lut, uint8_t neg)
    {
        const unsigned k = atomicAdd(&ctrl->npatch, 1u);
        if (k < cap) { gpsiq_patch_t p; p.block = block; p.sample = sample; p.slot = slot; p.neg = neg; p.lut = lut; patches[k] = p; }
    }
};

__device__ inline void report_error(EvalCtrl *ctrl, size_t flat, int status)
{
    atomicMin(&ctrl->err_key, ((unsigned long long) flat << 8) | (unsigned long long) (status & 0xff));
}

// compacted store of a block's descriptors: active channels first (their order kept), zeroed slots after them
__device__ inline int store_compacted(gpsiq_qchan_t *__restrict__ qout, size_t row, int i, int nchan, bool active, const gpsiq_qchan_t &q)
{
    const unsigned long long m = __ballot(active);
    const unsigned bits = (unsigned) (m >> (threadIdx.x & 48)) & 0xffffu;
    const int slot = __popc(bits & ((1u << i) - 1u)), na = __popc(bits);
    if (active) qout[row + slot] = q;
    if (i >= na && i < nchan) {
        gpsiq_qchan_t z;
        z.carr_phase = 0; z.carr_step = 0; z.code_frac = 0; z.code_step = 0; z.gain = 0.0; z.nav_bits = 0; z.chip0 = 0; z.icode = 0; z.prn = 0;
        qout[row + i] = z;
    }
    return slot;
}

// The phase a block's descriptor is seeded from: est_rows + k * est_stride bytes holds a double per (block, channel) -- the
// estimate of the accumulator at the block's first sample (lane::Prep::est, 32-byte rows, written by chain_prepare), or the
// caller's start states (gpsiq_generate_seeded: 8-byte rows).
__device__ inline double seed_phase(const char *__restrict__ est_rows, int est_stride, size_t k)
{
    return *reinterpret_cast<const double *>(est_rows + k * (size_t) est_stride);
}

// GPSIQ_NCO_REFERENCE, first half: the descriptors of blocks [b0, b0 + nb), quantised with the carrier seeded from the ESTIMATE
// of each block's start state, compacted per block into the set the synthesis kernel reads.  The synthesis starts behind this
// kernel; the chain that gives the TRUE start states (chain_lanes, chain_link_scan) and the evaluation run beside it.
__global__ __launch_bounds__(kEvalThreads) void quantize_est(const DChan *__restrict__ chan, int b0, int nb, int nchan, double delt, int nsamp,
                                                             const char *__restrict__ est_rows, int est_stride,
                                                             gpsiq_qchan_t *__restrict__ qout, EvalCtrl *__restrict__ ctrl)
{
    const long gid = (long) blockIdx.x * kEvalThreads + threadIdx.x;
    const int b = b0 + (int) (gid >> 4), i = (int) (gid & 15);
    const bool live = b < b0 + nb && i < nchan;
    DChan d;
    d.f_carr = 0.0; d.carr_phase = 0.0; d.prn = 0; d.pos = 0; d.f_code = 0.0; d.code_phase = 0.0; d.gain = 0.0; d.nav = 0; d.start = 0.0;
    double est = 0.0;
    if (live) { d = chan[(size_t) b * nchan + i]; est = seed_phase(est_rows, est_stride, (size_t) b * nchan + i); }
    gpsiq_qchan_t q;
    const int qs = ev::eval_quantize(d, est, delt, nsamp, &q);
    if (live && qs != ev::kQOk) report_error(ctrl, (size_t) b * nchan + i, qs);
    (void) store_compacted(qout, (size_t) b * nchan, i, b < b0 + nb ? nchan : 0, live && d.prn > 0 && qs == ev::kQOk, q);
}

// Second half, when the chain has been linked: the TRUE start state of every block is in its DChan (or in starts[], the caller's).
// One lane per (block, channel) recomputes the descriptor quantize_est wrote (same arithmetic: no dependence on the compacted
// set), finds the samples where the double path from the true start leaves that closed form, and appends the patches.
__global__ __launch_bounds__(kEvalThreads) void eval_blocks(const DChan *__restrict__ chan, int b0, int nb, int nchan, double delt, int nsamp,
                                                            const DeviceTables *__restrict__ tab, const char *__restrict__ est_rows, int est_stride,
                                                            gpsiq_patch_t *__restrict__ patches, unsigned patch_cap,
                                                            EvalHostItem *__restrict__ hostlist, unsigned host_cap, EvalCtrl *__restrict__ ctrl,
                                                            const double *__restrict__ starts)
{
    const long gid = (long) blockIdx.x * kEvalThreads + threadIdx.x;
    const int b = b0 + (int) (gid >> 4), i = (int) (gid & 15);
    const bool live = b < b0 + nb && i < nchan;
    DChan d;
    d.f_carr = 0.0; d.carr_phase = 0.0; d.prn = 0; d.pos = 0; d.f_code = 0.0; d.code_phase = 0.0; d.gain = 0.0; d.nav = 0; d.start = 0.0;
    double est = 0.0;
    if (live) { d = chan[(size_t) b * nchan + i]; est = seed_phase(est_rows, est_stride, (size_t) b * nchan + i); }
    const double start = live && starts ? starts[(size_t) b * nchan + i] : d.start;     // starts: the caller's (gpsiq_generate_seeded)
    gpsiq_qchan_t q;
    const int qs = ev::eval_quantize(d, est, delt, nsamp, &q);                          // (refusals were reported by quantize_est)
    const bool active = live && d.prn > 0 && qs == ev::kQOk;
    // the channel's place among the block's active ones, as quantize_est compacted them
    const unsigned long long m = __ballot(active);
    const unsigned bits = (unsigned) (m >> (threadIdx.x & 48)) & 0xffffu;
    const int slot = __popc(bits & ((1u << i) - 1u));
    if (!active) return;
    DevChips chips = {tab};
    DevEmit emit = {patches, patch_cap, ctrl, (uint32_t) b, (uint8_t) slot};
    const int st = ev::eval_candidates(d, q, start, delt, nsamp, chips, emit);
    if (st < 0) report_error(ctrl, (size_t) b * nchan + i, -st);
    if (st == ev::kEvalHost) {
        const unsigned k = atomicAdd(&ctrl->nhost, 1u);
        if (k < host_cap) { EvalHostItem h; h.block = (uint32_t) b; h.chan = (uint16_t) i; h.slot = (uint16_t) slot; h.start = start; h.seed = q.carr_phase; hostlist[k] = h; }
    }
}

// ---- GPSIQ_NCO_FIXED: quantise, then the exact carrier prefix --------------------------------------------------------------
// every block from its own carr_phase (gpsiq_quantize_batch's first half); descriptors NOT compacted yet (carry_prefix does that)
__global__ __launch_bounds__(kEvalThreads) void quantize_fixed(const DChan *__restrict__ chan, int b0, int nb, int nchan, double delt, int nsamp,
                                                               gpsiq_qchan_t *__restrict__ qraw, EvalCtrl *__restrict__ ctrl)
{
    const long gid = (long) blockIdx.x * kEvalThreads + threadIdx.x;
    const int b = b0 + (int) (gid >> 4), i = (int) (gid & 15);
    if (!(b < b0 + nb && i < nchan)) return;
    const DChan d = chan[(size_t) b * nchan + i];
    gpsiq_qchan_t q;
    const int qs = ev::quantize_dchan(d, delt, nsamp, nullptr, &q);
    if (qs != ev::kQOk) report_error(ctrl, (size_t) b * nchan + i, qs);
    qraw[(size_t) b * nchan + i] = q;
}

// p_{b+1} = p_b + nsamp*step_b (mod 2^59) down slot i while it keeps its satellite; block 0 continues carry0 where cont0 says so
// (chain_carrier of gpsiq_host.cpp).  One workgroup per slot; then the block's descriptors are compacted in place by a second
// kernel (compact_blocks), since a slot's scan must not move rows another slot's scan still reads.
// The scan: four consecutive blocks per thread in registers, the threads' totals across the wave with cross-lane moves, the four
// waves' totals through LDS -- 1 024 blocks per round and two barriers (a Hillis-Steele scan over LDS took sixteen barriers per 256
// blocks: 44 us per 4 130 blocks, on the critical path in front of the first synthesis).
struct FixScan { uint64_t v; int flag; };
__device__ inline FixScan fix_join(const FixScan &p, FixScan a)                  // first p, then a: a segment start in a hides p
{
    if (!a.flag) { a.v = (p.v + a.v) & ev::kEvCarrMask; a.flag = p.flag; }
    return a;
}
constexpr int kFixPer = 4;                                                       // blocks per thread and round
__global__ __launch_bounds__(kLinkThreads) void carry_prefix(gpsiq_qchan_t *__restrict__ q, int b0, int nb, int nchan, int nsamp,
                                                             FixedCarry *__restrict__ carry)
{
    __shared__ FixScan s_wave[kLinkThreads / 64];
    __shared__ FixedCarry s_carry;
    const int i = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = carry[i];
    __syncthreads();
    constexpr uint64_t mask = ev::kEvCarrMask;
    constexpr int kRound = kLinkThreads * kFixPer;
    for (int base = 0; base < nb; base += kRound) {
        const FixedCarry cy = s_carry;
        const int n_here = nb - base < kRound ? nb - base : kRound;
        const int k0 = tid * kFixPer;                                             // this thread's first block inside the round
        FixScan run[kFixPer];                                                     // inclusive scan of the thread's own blocks
        bool cont[kFixPer];
        int last_prn = 0;                                                         // satellite of the thread's last live block (0: unused slot)
        int prev_prn = cy.prn;                                                    // (the launch's first block: what the piece before left)
        if (base + k0 > 0 && k0 < n_here) prev_prn = (int) q[(size_t) (b0 + base + k0 - 1) * nchan + i].prn;
#pragma unroll
        for (int j = 0; j < kFixPer; ++j) {
            const int k = k0 + j, b = b0 + base + k;
            const bool live = k < n_here;
            uint64_t phase = 0, step = 0;
            int prn = 0;
            if (live) { const gpsiq_qchan_t &d = q[(size_t) b * nchan + i]; phase = d.carr_phase; step = d.carr_step; prn = (int) d.prn; }
            const bool active = live && prn != 0;
            // a block seeds the slot from its own phase unless it continues the block before (same satellite); the call's first block
            // continues what the caller says it does (cont0 of chain_carrier)
            cont[j] = active && (b == 0 ? cy.cont != 0 : prev_prn == prn);
            const uint64_t adv = active ? (step * (uint64_t) nsamp) & mask : 0;
            FixScan mine;
            mine.flag = live && !cont[j] ? 1 : 0;
            mine.v = (mine.flag ? phase + adv : adv) & mask;
            run[j] = j ? fix_join(run[j - 1], mine) : mine;
            if (live) { prev_prn = prn; last_prn = active ? prn : 0; }
        }
        // the threads' totals: inclusive scan across the wave, then the waves before this one
        FixScan inc = run[kFixPer - 1];
        for (int off = 1; off < 64; off <<= 1) {
            FixScan o;
            o.v = __shfl_up(inc.v, off, 64); o.flag = __shfl_up(inc.flag, off, 64);
            if (lane >= off) inc = fix_join(o, inc);
        }
        if (lane == 63) s_wave[wave] = inc;
        FixScan exc;                                                              // everything before this thread's blocks, inside the round
        exc.v = __shfl_up(inc.v, 1, 64); exc.flag = __shfl_up(inc.flag, 1, 64);
        if (lane == 0) { exc.v = 0; exc.flag = 0; }
        __syncthreads();
        FixScan before; before.v = 0; before.flag = 0;
        for (int w = 0; w < wave; ++w) before = fix_join(before, s_wave[w]);
        exc = fix_join(before, exc);
        // the phase a block starts from: what the blocks before it leave behind
#pragma unroll
        for (int j = 0; j < kFixPer; ++j) {
            if (!cont[j]) continue;
            const FixScan p = j ? fix_join(exc, run[j - 1]) : exc;
            const uint64_t p0 = p.flag ? p.v : (cy.phase + p.v) & mask;
            q[(size_t) (b0 + base + k0 + j) * nchan + i].carr_phase = p0 & mask;
        }
        // the round's carry: the thread that holds the round's last block knows the inclusive total and the slot's satellite
        const int t_last = (n_here - 1) / kFixPer;
        __syncthreads();
        if (tid == t_last) {
            const int jl = (n_here - 1) % kFixPer;
            FixScan rl = run[0];                                                  // (selected, not indexed: the array stays in registers)
#pragma unroll
            for (int j = 1; j < kFixPer; ++j) if (j == jl) rl = run[j];
            const FixScan e = fix_join(exc, rl);
            FixedCarry c;
            c.phase = e.flag ? e.v : (cy.phase + e.v) & mask;
            c.prn = last_prn;
            c.cont = 1;
            s_carry = c;
        }
        __syncthreads();
    }
    if (tid == 0) carry[i] = s_carry;
}

// active channels first, in place (a block's sixteen lanes read their row, then write it)
__global__ __launch_bounds__(kEvalThreads) void compact_blocks(gpsiq_qchan_t *__restrict__ q, int b0, int nb, int nchan, EvalCtrl *__restrict__ ctrl)
{
    const long gid = (long) blockIdx.x * kEvalThreads + threadIdx.x;
    const int b = b0 + (int) (gid >> 4), i = (int) (gid & 15);
    const bool live = b < b0 + nb && i < nchan;
    gpsiq_qchan_t d;
    d.carr_phase = 0; d.carr_step = 0; d.code_frac = 0; d.code_step = 0; d.gain = 0.0; d.nav_bits = 0; d.chip0 = 0; d.icode = 0; d.prn = 0;
    if (live) d = q[(size_t) b * nchan + i];
    const bool active = live && d.prn != 0;
    // (every lane of the block has read before any writes: the ballot inside is the wave's meeting point)
    (void) store_compacted(q, (size_t) b * nchan, i, b < b0 + nb ? nchan : 0, active, d);
    (void) ctrl;
}

// ---- launches -------------------------------------------------------------------------------------------------------------
hipError_t launch_pack_raw(const gpsiq_chan_t *d_ch, int nblocks, int nchan, double delt, void *d_chan, EvalCtrl *d_ctrl, hipStream_t s)
{
    if (nblocks <= 0) return hipSuccess;
    const long threads = (long) nblocks * 16;
    hipLaunchKernelGGL(pack_raw, dim3((unsigned) ((threads + kPackThreads - 1) / kPackThreads)), dim3(kPackThreads), 0, s, d_ch, nblocks, nchan, delt,
                       static_cast<DChan *>(d_chan), d_ctrl);
    return hipGetLastError();
}

hipError_t launch_link_scan(void *d_chan, const void *d_maps, int b0, int nb, int nchan, double delt, LinkCarry *d_carry, EvalCtrl *d_ctrl, int piece,
                            hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(chain_link_scan, dim3((unsigned) nchan), dim3(kLinkThreads), 0, s, static_cast<DChan *>(d_chan), static_cast<const Rec *>(d_maps),
                       b0, nb, nchan, delt, d_carry, d_ctrl, piece);
    return hipGetLastError();
}

hipError_t launch_gather_slot(const void *d_chan, const void *d_maps, int b0, int nb, int nchan, int slot, EvalSlotRow *d_out, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(gather_slot, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, s, static_cast<const DChan *>(d_chan), static_cast<const Rec *>(d_maps), b0, nb, nchan, slot, d_out);
    return hipGetLastError();
}
hipError_t launch_scatter_starts(void *d_chan, int b0, int nb, int nchan, int slot, const double *d_starts, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    hipLaunchKernelGGL(scatter_starts, dim3((unsigned) ((nb + 255) / 256)), dim3(256), 0, s, static_cast<DChan *>(d_chan), b0, nb, nchan, slot, d_starts);
    return hipGetLastError();
}

hipError_t launch_quantize_est(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, const void *est_rows, int est_stride,
                               gpsiq_qchan_t *d_q, EvalCtrl *d_ctrl, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const long threads = (long) nb * 16;
    hipLaunchKernelGGL(quantize_est, dim3((unsigned) ((threads + kEvalThreads - 1) / kEvalThreads)), dim3(kEvalThreads), 0, s, static_cast<const DChan *>(d_chan),
                       b0, nb, nchan, delt, nsamp, static_cast<const char *>(est_rows), est_stride, d_q, d_ctrl);
    return hipGetLastError();
}

hipError_t launch_eval(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, const DeviceTables *tab, const void *est_rows, int est_stride,
                       gpsiq_patch_t *d_patches, unsigned patch_cap, EvalHostItem *d_host, unsigned host_cap, EvalCtrl *d_ctrl, const double *d_starts,
                       hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const long threads = (long) nb * 16;
    hipLaunchKernelGGL(eval_blocks, dim3((unsigned) ((threads + kEvalThreads - 1) / kEvalThreads)), dim3(kEvalThreads), 0, s, static_cast<const DChan *>(d_chan),
                       b0, nb, nchan, delt, nsamp, tab, static_cast<const char *>(est_rows), est_stride, d_patches, patch_cap, d_host, host_cap, d_ctrl, d_starts);
    return hipGetLastError();
}

hipError_t launch_quantize_fixed(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, gpsiq_qchan_t *d_q, FixedCarry *d_carry,
                                 EvalCtrl *d_ctrl, hipStream_t s)
{
    if (nb <= 0) return hipSuccess;
    const long threads = (long) nb * 16;
    const dim3 grid((unsigned) ((threads + kEvalThreads - 1) / kEvalThreads)), block(kEvalThreads);
    hipLaunchKernelGGL(quantize_fixed, grid, block, 0, s, static_cast<const DChan *>(d_chan), b0, nb, nchan, delt, nsamp, d_q, d_ctrl);
    hipLaunchKernelGGL(carry_prefix, dim3((unsigned) nchan), dim3(kLinkThreads), 0, s, d_q, b0, nb, nchan, nsamp, d_carry);
    hipLaunchKernelGGL(compact_blocks, grid, block, 0, s, d_q, b0, nb, nchan, d_ctrl);
    return hipGetLastError();
}

#ifdef GPSIQ_TEST_HOOKS
__global__ void test_corrupt_map(Rec *rec, size_t at) { rec[at].cum[0] += 4; rec[at].cum[1] += 4; }
hipError_t launch_test_corrupt_map(void *d_maps, int b0, int nb, int nchan, hipStream_t s)
{
    const char *e = std::getenv("GPSIQ_TEST_CORRUPT_MAP");
    int b = -1, i = -1;
    if (!e || std::sscanf(e, "%d,%d", &b, &i) != 2 || b < b0 || b >= b0 + nb || i < 0 || i >= nchan) return hipSuccess;
    hipLaunchKernelGGL(test_corrupt_map, dim3(1), dim3(1), 0, s, static_cast<Rec *>(d_maps), (size_t) b * nchan + i);
    return hipGetLastError();
}
#endif

}  // namespace gpsiq
