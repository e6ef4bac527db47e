// gpsiq_evaldev.cpp -- gpsiq_generate_batch / gpsiq_generate_seeded with the descriptors quantised and, in GPSIQ_NCO_REFERENCE,
// chained and evaluated ON THE DEVICE (kernels: gpsiq_chain_kernels.hip, gpsiq_eval_kernels.hip; the lane code: gpsiq_eval.h).
//
// What the reference's host code leaves per block and channel at gps.c:2766 (gpsiq_chan_t) goes in; per piece of the batch
//     pack -> [chain_prepare -> chain_lanes -> chain_link_scan] -> eval_blocks | quantize_fixed -> synth_tile (-> apply_patches)
// with no host stage between them.  What the host still does:
//   * descriptors in PAGEABLE host memory are cut down to 64 bytes each on the pool (the device cannot read them; page-locked
//     and device-resident descriptors are read where they lie);
//   * it waits for each piece's link (an event, the device busy with the pieces behind) to learn the synthesis kernel's launch
//     parameters and whether a slot has a block whose certified map does not apply -- then that slot's chain is walked by the host
//     walker (lane::link_block + the true walk, as rounds 4-5 did for every slot) and its start states go back up;
//   * the (block, channel) pairs whose candidates the drift enclosure cannot decide (1 in 10^4) get the host evaluation
//     (eval_block) while the device renders, and the merged, sorted patch list goes up before apply_patches.
// Results are those of the host path (gpsiq_generate_reference_host / the host quantiser) bit for bit: tests/eval_twin.cpp on
// the CPU, tests/test_gpu_device_eval.py and every T2 = 0 test on the GPU.
// Reference lines: gps.c:2033-2064 (what the quantiser takes in), 2775-2782 (index + truncation), 2789-2826 (the accumulators),
// 2208-2214 (re-seeding a slot).
#include <algorithm>
#include <array>
#include <cmath>
#include <cstdlib>
#include <cstring>

#include "gpsiq_ctx.h"
#include "gpsiq_eval.h"

using namespace gpsiq;

namespace gpsiq {
// gpsiq_exact.cpp: one channel of one block on the host (descriptor + patches from its start state)
int eval_block_host(const gpsiq_chan_t &ch, double start, const uint64_t *seed, double delt, int nsamp, int block, int slot, gpsiq_qchan_t *q,
                    std::vector<gpsiq_patch_t> *out);
}

namespace {

enum SrcKind { kSrcPageable = 0, kSrcPinned = 1, kSrcDevice = 2 };

SrcKind classify(const void *p, int device, const void **dev_ptr)
{
    *dev_ptr = nullptr;
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void) hipGetLastError(); return kSrcPageable; }
    if (a.type == hipMemoryTypeDevice && a.device == device) { *dev_ptr = p; return kSrcDevice; }
    if (a.type == hipMemoryTypeHost) return kSrcPinned;
    return kSrcPageable;               // unregistered, managed, another device's: through the host
}

constexpr unsigned kPatchCap = 1u << 16, kHostCap = 1u << 13;

int evd_reserve(gpsiq_ctx *c, size_t n, SrcKind kind, bool seeds)
{
    gpsiq_ctx::EvalDev &e = c->evd;
    int rc = gpsiq_chain_reserve(c, n);
    if (rc) return rc;
    if (!e.eval_stream) {
        // Chain and evaluation run BESIDE the synthesis, which floods the device with workgroups: their streams get the highest
        // priority, so that the few workgroups of prepare / lanes / link / evaluation are dispatched as synthesis workgroups retire
        // instead of behind all of them (the patches have to be there when the synthesis ends, not some time after it).
        int least = 0, greatest = 0;
        (void) hipDeviceGetStreamPriorityRange(&least, &greatest);
        // (MI355X, 2.6 Msps, 2 000 blocks: 1.82 ms per call against 2.20 at equal priorities; 25 Msps: 1.54 against 2.61)
        HIP_TRY(hipStreamCreateWithPriority(&e.eval_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipStreamCreateWithPriority(&e.chain_stream, hipStreamNonBlocking, greatest));
        HIP_TRY(hipEventCreate(&e.t_synth0));
        HIP_TRY(hipEventCreate(&e.t_synth1));
        for (auto &ev_ : e.linked) HIP_TRY(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
        for (auto &ev_ : e.evaluated) HIP_TRY(hipEventCreateWithFlags(&ev_, hipEventDisableTiming));
        HIP_TRY(hipEventCreateWithFlags(&e.joined, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **) &e.d_ctrl, sizeof(EvalCtrl)));
        HIP_TRY(hipHostMalloc((void **) &e.h_ctrl, (kEvalMaxPieces + 2) * sizeof(EvalCtrl), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **) &e.d_link, GPSIQ_MAX_CHAN * sizeof(LinkCarry)));
        HIP_TRY(hipHostMalloc((void **) &e.h_link, GPSIQ_MAX_CHAN * sizeof(LinkCarry), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **) &e.d_fix, GPSIQ_MAX_CHAN * sizeof(FixedCarry)));
        HIP_TRY(hipHostMalloc((void **) &e.h_fix, GPSIQ_MAX_CHAN * sizeof(FixedCarry), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **) &e.d_patches, kPatchCap * sizeof(gpsiq_patch_t)));
        HIP_TRY(hipHostMalloc((void **) &e.h_patches, kPatchCap * sizeof(gpsiq_patch_t), hipHostMallocDefault));
        e.patch_cap = kPatchCap;
        HIP_TRY(hipMalloc((void **) &e.d_host, kHostCap * sizeof(EvalHostItem)));
        HIP_TRY(hipHostMalloc((void **) &e.h_host, kHostCap * sizeof(EvalHostItem), hipHostMallocDefault));
        e.host_cap = kHostCap;
    }
    if (n > e.cap) {
        if (e.d_chan) (void) hipFree(e.d_chan);
        if (e.h_chan) (void) hipHostFree(e.h_chan);
        e.d_chan = nullptr; e.h_chan = nullptr; e.cap = 0;
        const size_t cap = n + n / 4 + 256;
        HIP_TRY(hipMalloc(&e.d_chan, cap * sizeof(ev::DChan)));
        HIP_TRY(hipHostMalloc(&e.h_chan, cap * sizeof(ev::DChan), hipHostMallocDefault));
        e.cap = cap;
    }
    if (kind == kSrcPinned && n > e.raw_cap) {
        if (e.d_raw) (void) hipFree(e.d_raw);
        e.d_raw = nullptr; e.raw_cap = 0;
        HIP_TRY(hipMalloc((void **) &e.d_raw, (n + n / 4 + 16) * sizeof(gpsiq_chan_t)));
        e.raw_cap = n + n / 4 + 16;
    }
    if (seeds && n > e.seeds_cap) {
        if (e.d_seeds) (void) hipFree(e.d_seeds);
        e.d_seeds = nullptr; e.seeds_cap = 0;
        HIP_TRY(hipMalloc((void **) &e.d_seeds, (n + n / 4 + 16) * sizeof(double)));
        e.seeds_cap = n + n / 4 + 16;
    }
    return GPSIQ_OK;
}

// Piece boundaries.  A piece's synthesis waits for its own descriptors only (pack, estimate, quantise), so pieces exist to start
// the first synthesis early and to stage piece k+1 under the synthesis of piece k; every further piece costs a launch ramp
// (~0.05 ms).  Descriptors the device reads where they lie stage in microseconds: one piece in the fixed-point model, a short
// head in GPSIQ_NCO_REFERENCE (chain_prepare is 16 workgroups walking the timeline: 25 us per 1 000 blocks).  Pageable
// descriptors are packed by the pool at ~14 x the synthesis rate (16 threads; 2 x with two): a head worth 0.35 ms of synthesis, so
// that the pack of what follows hides under it, then pieces eight times the one before (measured, 2.6 Msps: 2 000 blocks fixed
// model 1.56 ms per call against 1.74 with a 0.1 ms head, reference model 1.85 against 2.06).
// GPSIQ_PIECE_BLOCKS (blocks of the first piece; <= 0: one piece) for A/B, read per call.
void device_piece_ends(int nblocks, int nsamp, int nchan, bool pageable /* or page-locked: host memory */, bool reference, std::vector<int> *ends)
{
    const double t_block = (double) nsamp * (double) nchan / gpsiq_rate_kernel();
    long head = (long) ((pageable ? 0.35e-3 : 0.15e-3) / (t_block > 0.0 ? t_block : 1e-6)) + 1;
    if (head < 16) head = 16;
    if (!pageable && !reference) head = 0;
    if (const char *e = std::getenv("GPSIQ_PIECE_BLOCKS")) head = std::atol(e);
    const long growth = 8;
    if (head <= 0 || 2 * head > nblocks) { ends->push_back(nblocks); return; }
    long b = head, size = growth * head;
    ends->push_back((int) b);
    while (nblocks - b > size + size / 2 && (int) ends->size() < kEvalMaxPieces - 1) {
        b += size;
        ends->push_back((int) b);
        size *= growth;
    }
    ends->push_back(nblocks);
}

// descriptors in pageable memory: cut down to ev::DChan on the pool, with the synthesis kernel's launch parameters on the way
struct PackJob { const gpsiq_chan_t *ch; ev::DChan *out; const double *seeds; int nchan; double delt; uint64_t mx; int max_active; long max_amp; };
void pack_host(PackJob *pj, int nblocks)
{
    parallel_for(nblocks, 0, 64, [](void *p, int b0, int b1) {
        PackJob &j = *static_cast<PackJob *>(p);
        uint64_t mx = 0;
        int max_active = 0;
        long max_amp = 0;
        for (int b = b0; b < b1; ++b) {
            int na = 0;
            long amp = 0;
            for (int i = 0; i < j.nchan; ++i) {
                const size_t k = (size_t) b * j.nchan + i;
                if (b + 1 < b1) __builtin_prefetch(reinterpret_cast<const char *>(&j.ch[k + j.nchan]));
                ev::DChan d;
                ev::pack_chan(j.ch[k], &d);
                if (j.seeds) d.start = j.seeds[k];
                j.out[k] = d;
                const double code_inc = d.f_code * j.delt;
                if (d.prn > 0 && d.prn <= 32 && code_inc > 0.0 && code_inc < 2.0 && d.gain > -ev::kEvMaxGain && d.gain < ev::kEvMaxGain) {
                    const uint64_t cs = (uint64_t) (int64_t) __builtin_rint(code_inc * 0x1p56);
                    if (cs > mx) mx = cs;
                    amp += (long) (250.0 * std::fabs(d.gain));
                    ++na;
                }
            }
            if (na > max_active) max_active = na;
            if (amp > max_amp) max_amp = amp;
        }
        for (uint64_t cur = j.mx; mx > cur && !__sync_bool_compare_and_swap(&j.mx, cur, mx); cur = j.mx) {}
        for (int cur = j.max_active; max_active > cur && !__sync_bool_compare_and_swap(&j.max_active, cur, max_active); cur = j.max_active) {}
        for (long cur = j.max_amp; max_amp > cur && !__sync_bool_compare_and_swap(&j.max_amp, cur, max_amp); cur = j.max_amp) {}
    }, pj);
}

// one slot's chain on the host through the device's maps (link_slot of gpsiq_chain.cpp on the slot's gathered column), blocks
// [b0, b1): rows[b - b0]; every start state -> start_col[b - b0]; *x / *pv: the accumulator and satellite before block b0
// (b0 > 0), and after block b1 - 1 on return.  false: a state or Doppler outside the NCO format (*bad_block)
bool link_slot_host(const EvalSlotRow *rows, int b0, int b1, double delt, int nsamp, double *x_io, int *pv_io, double *start_col,
                    long *linked, long *walked, int *bad_block)
{
    static_assert(sizeof(gpsiq_chain_map_t) == sizeof(lane::Rec), "the map as the lanes write it");
    double x = *x_io;
    int pv = *pv_io;
    for (int b = b0; b < b1; ++b) {
        const EvalSlotRow &d = rows[b - b0];
        if (d.prn <= 0) { pv = 0; x = 0.0; start_col[b - b0] = 0.0; continue; }
        if (b == 0 || pv != d.prn) x = d.carr_phase;
        pv = d.prn;
        start_col[b - b0] = x;
        const double inc = d.f_carr * delt;
        if (!(x >= 0.0 && x <= 1.0) || !(std::fabs(inc) < 0.5)) { *bad_block = b; return false; }
        double y;
        if (lane::link_block(*reinterpret_cast<const lane::Rec *>(&d.map), x, &y)) { x = y; ++*linked; }
        else { x = chain_block_true(d.f_carr, delt, nsamp, x); ++*walked; }
    }
    *x_io = x; *pv_io = pv;
    return true;
}

bool patch_before(const gpsiq_patch_t &a, const gpsiq_patch_t &b)
{
    if (a.block != b.block) return a.block < b.block;
    if (a.sample != b.sample) return a.sample < b.sample;
    return a.slot < b.slot;
}

}  // namespace

namespace gpsiq { void chain_count(long linked, long walked); }

static uint64_t g_evd_stats[6];       // calls, (block, channel) pairs evaluated on the device, handed to the host walker, slots repaired, patches, calls that fell back

extern "C" void gpsiq_device_eval_stats(uint64_t out[6])
{
    if (!out) return;
    for (int k = 0; k < 6; ++k) out[k] = __atomic_load_n(&g_evd_stats[k], __ATOMIC_RELAXED);
}

extern "C" double gpsiq_device_eval_host_ms(const gpsiq_ctx_t *c) { return c ? c->evd.host_ms : 0.0; }

void gpsiq_evaldev_destroy(gpsiq_ctx *c)
{
    gpsiq_ctx::EvalDev &e = c->evd;
    if (e.d_chan) (void) hipFree(e.d_chan);
    if (e.h_chan) (void) hipHostFree(e.h_chan);
    if (e.d_raw) (void) hipFree(e.d_raw);
    if (e.d_seeds) (void) hipFree(e.d_seeds);
    if (e.d_ctrl) (void) hipFree(e.d_ctrl);
    if (e.h_ctrl) (void) hipHostFree(e.h_ctrl);
    if (e.d_link) (void) hipFree(e.d_link);
    if (e.h_link) (void) hipHostFree(e.h_link);
    if (e.d_fix) (void) hipFree(e.d_fix);
    if (e.h_fix) (void) hipHostFree(e.h_fix);
    if (e.d_patches) (void) hipFree(e.d_patches);
    if (e.h_patches) (void) hipHostFree(e.h_patches);
    if (e.d_slot) (void) hipFree(e.d_slot);
    if (e.h_slot) (void) hipHostFree(e.h_slot);
    if (e.d_col) (void) hipFree(e.d_col);
    if (e.h_col) (void) hipHostFree(e.h_col);
    if (e.h_items) (void) hipHostFree(e.h_items);
    if (e.d_host) (void) hipFree(e.d_host);
    if (e.h_host) (void) hipHostFree(e.h_host);
    for (auto &ev_ : e.linked) if (ev_) (void) hipEventDestroy(ev_);
    for (auto &ev_ : e.evaluated) if (ev_) (void) hipEventDestroy(ev_);
    if (e.joined) (void) hipEventDestroy(e.joined);
    if (e.t_synth0) (void) hipEventDestroy(e.t_synth0);
    if (e.t_synth1) (void) hipEventDestroy(e.t_synth1);
    if (e.eval_stream) (void) hipStreamDestroy(e.eval_stream);
    if (e.chain_stream) (void) hipStreamDestroy(e.chain_stream);
    e = gpsiq_ctx::EvalDev();
}

// how many entries the patch / host lists of a call may hold.  (-DGPSIQ_TEST_HOOKS: GPSIQ_TEST_LIST_CAP=n makes the lists n entries
// short, so that a test can see what a call does when they overflow -- the whole call again on the host path.)
static inline unsigned list_cap(unsigned cap)
{
#ifdef GPSIQ_TEST_HOOKS
    if (const char *s = std::getenv("GPSIQ_TEST_LIST_CAP")) { const long v = std::atol(s); if (v >= 0 && (unsigned long) v < cap) return (unsigned) v; }
#endif
    return cap;
}

// Whether the device path takes a batch: long enough for its fixed costs (a dozen launches, two or three waits: ~0.06 ms) to pay.
// GPSIQ_NCO_REFERENCE: from 48 blocks (measured, 2.6 and 25 Msps, 16 host threads: the host path is ahead below that).  Fixed-point
// model: the host quantiser costs ~0.2 us per descriptor and thread, so the device takes over at ~300 descriptors per host thread
// (16 threads: 300 blocks of 16 channels; two threads: 38).  GPSIQ_EVAL=host / device decides by hand (read per call).
static bool device_path_wanted(const gpsiq_ctx *c, int nblocks, int nchan)
{
    const char *e = std::getenv("GPSIQ_EVAL");
    if (e && !std::strcmp(e, "host")) return false;
    if (e && !std::strcmp(e, "device")) return nblocks >= 1;
    if (std::getenv("GPSIQ_CHAIN")) return false;          // where level 1 of the chain runs WITHIN the host evaluation: asks for that path
    if (c->nco_mode == GPSIQ_NCO_REFERENCE) return nblocks >= 48;
    return (long) nblocks * nchan >= 300L * (host_threads() > 0 ? host_threads() : 1);
}

int gpsiq_generate_device(gpsiq_ctx *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs, int sample_size,
                          void *dst, int dst_is_device, double *carr_phase_out, const double *seeds, int *handled)
{
    *handled = 0;
    if (!device_path_wanted(c, nblocks, nchan) || nsamp <= 0) return GPSIQ_OK;
    const bool reference = c->nco_mode == GPSIQ_NCO_REFERENCE || seeds != nullptr;
    const char *trace_env = std::getenv("GPSIQ_TRACE");
    const bool trace = trace_env != nullptr;
    const double t0 = gpsiq_wall_ms();
    HIP_TRY(hipSetDevice(c->device));
    gpsiq_ctx::EvalDev &e = c->evd;
    const size_t n = (size_t) nblocks * (size_t) nchan;
    const double delt = 1.0 / fs;
    const void *dev_src = nullptr;
    const SrcKind kind = classify(ch, c->device, &dev_src);
    int rc = evd_reserve(c, n, kind, seeds != nullptr);
    if (rc) return rc;
    *handled = 1;
    __atomic_fetch_add(&g_evd_stats[0], 1, __ATOMIC_RELAXED);
    e.host_ms = 0.0; e.last_nhost = 0; e.last_npatch = 0; e.last_repaired = 0;
    double host_part[3] = {0.0, 0.0, 0.0};               // pack of pageable rows, repair of slots, patch lists + the host walker's share (the trace's business)

    // the first and the last block's descriptors on the host: the continuation test of the fixed model, the phase handed out for
    // a slot that ends unused
    std::vector<gpsiq_chan_t> edge;
    const gpsiq_chan_t *first = ch, *last = ch + (size_t) (nblocks - 1) * nchan;
    if (kind == kSrcDevice) {
        edge.resize((size_t) 2 * nchan);
        HIP_TRY(hipMemcpy(edge.data(), ch, (size_t) nchan * sizeof(gpsiq_chan_t), hipMemcpyDeviceToHost));
        HIP_TRY(hipMemcpy(edge.data() + nchan, last, (size_t) nchan * sizeof(gpsiq_chan_t), hipMemcpyDeviceToHost));
        first = edge.data(); last = edge.data() + nchan;
    }

    // destination: straight into the caller's device memory where its rows are ours, else through the context's staging
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) sample_size, stride = (blk_bytes + 15) & ~(size_t) 15;
    const bool direct = dst_is_device && stride == blk_bytes && !((uintptr_t) dst & 3);
    if (!direct) { rc = gpsiq_ensure_out(c, stride * (size_t) nblocks); if (rc) return rc; }

    // the descriptor set the evaluation writes and the synthesis reads: one for the whole call
    int nsets = gpsiq_ctx::kSets;
    const int next = (c->cur + 1) % nsets;
    gpsiq_ctx::DescBuf &nb_ = c->buf[next];
    rc = gpsiq_wait_idle(nb_);
    if (rc) return rc;
    if (n > nb_.cap) {
        if (nb_.d) HIP_TRY(hipFree(nb_.d));
        nb_.d = nullptr; nb_.cap = 0;
        HIP_TRY(hipMalloc((void **) &nb_.d, n * sizeof(gpsiq_qchan_t)));
        nb_.cap = n;
    }

    std::vector<int> ends;
    device_piece_ends(nblocks, nsamp, nchan, kind != kSrcDevice, reference, &ends);          // (page-locked rows cross PCIe too: staged like pageable ones)
    const int npieces = (int) ends.size();
    hipStream_t S = e.chain_stream, E = e.eval_stream;
    ev::DChan *d_chan = static_cast<ev::DChan *>(e.d_chan), *h_chan = static_cast<ev::DChan *>(e.h_chan);

    // ---- phase A: per piece stage the rows, estimate (chain_prepare), quantise from the estimate, render ------------------------
    // The synthesis of a piece waits for nothing but its own descriptors: in GPSIQ_NCO_REFERENCE they are seeded from
    // chain_prepare's ESTIMATE of every block's start state (good to ~1e-11 cycle), and what the double path from the TRUE start
    // state does differently goes into the patch list when the chain has been linked (phase B), beside the running synthesis.
    EvalCtrl zero;
    std::memset(&zero, 0, sizeof zero);
    zero.err_key = ~0ull;
    e.h_ctrl[kEvalMaxPieces + 1] = zero;
    HIP_TRY(hipMemcpyAsync(e.d_ctrl, &e.h_ctrl[kEvalMaxPieces + 1], sizeof(EvalCtrl), hipMemcpyHostToDevice, S));
    bool cont0[GPSIQ_MAX_CHAN] = {};
    if (!reference) {
        for (int i = 0; i < nchan; ++i) {
            cont0[i] = first[i].prn > 0 && c->carry_prn[i] == first[i].prn && c->handed[i] == first[i].carr_phase;
            e.h_fix[i].phase = c->carry[i]; e.h_fix[i].prn = c->carry_prn[i]; e.h_fix[i].cont = cont0[i] ? 1 : 0;
        }
        HIP_TRY(hipMemcpyAsync(e.d_fix, e.h_fix, (size_t) nchan * sizeof(FixedCarry), hipMemcpyHostToDevice, S));
    }
    if (seeds) HIP_TRY(hipMemcpyAsync(e.d_seeds, seeds, n * sizeof(double), hipMemcpyHostToDevice, S));
    const bool chained = reference && !seeds;
    // where a block's descriptor takes its carrier phase from: Prep::est (32-byte rows, the double at offset 8), or the caller's states
    const char *est_rows = seeds ? reinterpret_cast<const char *>(e.d_seeds) : static_cast<const char *>(c->chain.d_prep) + 8;
    const int est_stride = seeds ? 8 : 32;
    PackJob pj = {ch, h_chan, nullptr, nchan, delt, 0, 0, 0};
    hipError_t he = hipSuccess;
    char err[400] = "";
    int max_active = 0;
    long max_amp = 0;
    uint64_t max_step = 0;
    int copies = 0;
    double t_first_launch = 0.0;
    int timed_blocks = 0;
    hipEvent_t *staged = e.evaluated;                       // [k]: piece k's descriptors are in the set (and a snapshot of the control block behind them)
    for (int k = 0; k < npieces && rc == GPSIQ_OK; ++k) {
        const int b0 = k ? ends[k - 1] : 0, nb = ends[k] - b0;
        const size_t off = (size_t) b0 * nchan, cnt = (size_t) nb * nchan;
        if (kind == kSrcDevice) he = launch_pack_raw(static_cast<const gpsiq_chan_t *>(dev_src) + off, nb, nchan, delt, d_chan + off, e.d_ctrl, S);
        else if (kind == kSrcPinned) {
            he = hipMemcpyAsync(e.d_raw + off, ch + off, cnt * sizeof(gpsiq_chan_t), hipMemcpyHostToDevice, S);
            if (he == hipSuccess) he = launch_pack_raw(e.d_raw + off, nb, nchan, delt, d_chan + off, e.d_ctrl, S);
        } else {
            const double tp = gpsiq_wall_ms();
            PackJob part = pj;
            part.ch = ch + off; part.out = h_chan + off;
            pack_host(&part, nb);
            pj.mx = part.mx > pj.mx ? part.mx : pj.mx; pj.max_active = part.max_active > pj.max_active ? part.max_active : pj.max_active;
            pj.max_amp = part.max_amp > pj.max_amp ? part.max_amp : pj.max_amp;
            e.host_ms += gpsiq_wall_ms() - tp; host_part[0] += gpsiq_wall_ms() - tp;
            he = hipMemcpyAsync(d_chan + off, h_chan + off, cnt * sizeof(ev::DChan), hipMemcpyHostToDevice, S);
        }
        if (he == hipSuccess && chained)
            he = launch_chain(d_chan + off, (int) sizeof(ev::DChan), nb, nchan, delt, nsamp, k ? c->chain.d_est + (size_t) k * GPSIQ_MAX_CHAN : nullptr, 0,
                              static_cast<char *>(c->chain.d_prep) + off * 32, c->chain.d_c_before + (size_t) k * GPSIQ_MAX_CHAN,
                              c->chain.d_est + (size_t) (k + 1) * GPSIQ_MAX_CHAN, c->chain.d_maps + off, S, 1);
        if (he == hipSuccess) {
            if (reference) he = launch_quantize_est(d_chan, b0, nb, nchan, delt, nsamp, est_rows, est_stride, nb_.d, e.d_ctrl, S);
            else he = launch_quantize_fixed(d_chan, b0, nb, nchan, delt, nsamp, nb_.d, e.d_fix, e.d_ctrl, S);
        }
        if (he == hipSuccess && kind != kSrcPageable) he = hipMemcpyAsync(&e.h_ctrl[k], e.d_ctrl, sizeof(EvalCtrl), hipMemcpyDeviceToHost, S);
        if (he == hipSuccess) he = hipEventRecord(staged[k], S);
        // the launch parameters of the synthesis kernel: from the pool's pack at once, from the device's pack behind the event
        if (he == hipSuccess && kind != kSrcPageable) he = hipEventSynchronize(staged[k]);
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, piece %d: %s", k, hipGetErrorString(he)); break; }
        if (kind == kSrcPageable) { max_step = pj.mx; max_active = pj.max_active; max_amp = pj.max_amp; }
        else {
            const EvalCtrl &ck = e.h_ctrl[k];
            max_step = std::max<uint64_t>(max_step, ck.max_code_step); max_active = std::max(max_active, ck.max_active); max_amp = std::max<long>(max_amp, (long) ck.max_amp);
        }
        hipStream_t s = gpsiq_piece_stream(c, k);
        he = hipStreamWaitEvent(s, staged[k], 0);
        if (he == hipSuccess) {
            const int v = max_step <= kRowsMaxCodeStep ? kSeg : max_step <= kHalfRowsMaxCodeStep ? kSegHalf : kGeneric;
            uint8_t *dev = direct ? static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes : static_cast<uint8_t *>(c->d_out) + (size_t) b0 * stride;
            // the last piece's synthesis is timed: the rate the piece sizes are planned with is a measured one (gpsiq_note_kernel_rate)
            if (k == npieces - 1) (void) hipEventRecord(e.t_synth0, s);
            he = launch_variant(v, nb_.d, nchan, nsamp, sample_size, dev, stride, b0, nb, c->d_tab, s, max_active > 0 ? max_active : 1, max_amp, nullptr);
            if (k == npieces - 1) { (void) hipEventRecord(e.t_synth1, s); timed_blocks = nb; }
            if (trace && k == 0) t_first_launch = gpsiq_wall_ms() - t0;
            if (he == hipSuccess && !direct) {
                // a piece crosses to the destination as soon as it is rendered (the next piece's kernel covers the copy); in
                // GPSIQ_NCO_REFERENCE the few blocks that hold a patched sample are copied once more behind apply_patches
                hipStream_t cs = c->copy_stream[copies & 1];
                he = hipEventRecord(c->chunk_done[copies & 1], s);
                if (he == hipSuccess) he = hipStreamWaitEvent(cs, c->chunk_done[copies & 1], 0);
                const hipMemcpyKind kd = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
                if (he == hipSuccess) {
                    if (stride == blk_bytes) he = hipMemcpyAsync(static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes, dev, blk_bytes * (size_t) nb, kd, cs);
                    else he = hipMemcpy2DAsync(static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes, blk_bytes, dev, stride, blk_bytes, (size_t) nb, kd, cs);
                }
                ++copies;
            }
        }
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, piece %d launch: %s", k, hipGetErrorString(he)); }
    }
    const double t_queued = gpsiq_wall_ms();

    // ---- phase B (GPSIQ_NCO_REFERENCE): the chain behind the descriptors, repair, the evaluation ---------------------------------
    // One launch each over the WHOLE timeline, whatever the synthesis pieces were: every launch of the chain has a latency floor
    // (a wave's walk, ~0.1 ms), and this path has to be through before the synthesis is (three pieces of 2 000 blocks: 2.03 ms per
    // call against 1.84 with one).
    bool host_owned[GPSIQ_MAX_CHAN] = {};
    double host_end[GPSIQ_MAX_CHAN] = {};
    int host_last_prn[GPSIQ_MAX_CHAN] = {};
    const std::vector<int> chain_ends(1, nblocks);
    const int nchain = 1;
    if (rc == GPSIQ_OK && chained) {
        int max_seg = 32;
        if (const char *sv = std::getenv("GPSIQ_CHAIN_STRETCHES")) { const int v = std::atoi(sv); if (v >= 1 && v <= 32) max_seg = v; }
        for (int k = 0; k < nchain && he == hipSuccess; ++k) {
            const int b0 = k ? chain_ends[k - 1] : 0, nb = chain_ends[k] - b0;
            const size_t off = (size_t) b0 * nchan;
            he = launch_chain(d_chan + off, (int) sizeof(ev::DChan), nb, nchan, delt, nsamp, nullptr, max_seg, static_cast<char *>(c->chain.d_prep) + off * 32,
                              c->chain.d_c_before + (size_t) k * GPSIQ_MAX_CHAN, nullptr, c->chain.d_maps + off, S, 2);
#ifdef GPSIQ_TEST_HOOKS          // fault injection (tests/test_gpu_verify.py builds a library with it): one map made wrong without making it unusable
            if (he == hipSuccess) he = launch_test_corrupt_map(c->chain.d_maps, b0, nb, nchan, S);
#endif
            if (he == hipSuccess) he = launch_link_scan(d_chan, c->chain.d_maps, b0, nb, nchan, delt, e.d_link, e.d_ctrl, k, S);
            if (he == hipSuccess) he = hipMemcpyAsync(&e.h_ctrl[k], e.d_ctrl, sizeof(EvalCtrl), hipMemcpyDeviceToHost, S);
            if (he == hipSuccess) he = hipEventRecord(e.linked[k], S);
        }
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, chain: %s", hipGetErrorString(he)); }
    }
    for (int k = 0; k < nchain && rc == GPSIQ_OK && reference; ++k) {
        const int b0 = k ? chain_ends[k - 1] : 0, nb = chain_ends[k] - b0;
        if (chained) {
            he = hipEventSynchronize(e.linked[k]);
            if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, piece %d: %s", k, hipGetErrorString(he)); break; }
            const EvalCtrl &ck = e.h_ctrl[k];
            bool need = false;
            for (int i = 0; i < nchan; ++i) need = need || ck.unknown[k][i] > 0 || host_owned[i];
            if (need) {
                // A block whose certified map does not apply (a slow block, a sign change, a range the estimate missed): the host
                // walker takes that slot from this piece on -- the slot's column of maps and rows comes back, it is linked / walked from
                // the state it enters the piece with (lane::link_block + the true walk, as rounds 4-5 did for every slot), and its
                // start states go back up before the piece is evaluated.  The device's own scan goes on for the other slots, and
                // the synthesis is not held up: it runs from the estimates.
                const double tr = gpsiq_wall_ms();
                const int lead = b0 > 0 ? 1 : 0;                                   // the row before the piece too: which satellite the slot had
                if ((size_t) nb + 1 > e.slot_cap) {
                    if (e.d_slot) (void) hipFree(e.d_slot);
                    if (e.h_slot) (void) hipHostFree(e.h_slot);
                    if (e.d_col) (void) hipFree(e.d_col);
                    if (e.h_col) (void) hipHostFree(e.h_col);
                    e.d_slot = nullptr; e.h_slot = nullptr; e.d_col = nullptr; e.h_col = nullptr; e.slot_cap = 0;
                    const size_t cap = (size_t) nb + 1 + 256;
                    he = hipMalloc((void **) &e.d_slot, cap * sizeof(EvalSlotRow));
                    if (he == hipSuccess) he = hipHostMalloc((void **) &e.h_slot, cap * sizeof(EvalSlotRow), hipHostMallocDefault);
                    if (he == hipSuccess) he = hipMalloc((void **) &e.d_col, cap * sizeof(double));
                    if (he == hipSuccess) he = hipHostMalloc((void **) &e.h_col, cap * sizeof(double), hipHostMallocDefault);
                    if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, repair buffers: %s", hipGetErrorString(he)); break; }
                    e.slot_cap = cap;
                }
                for (int i = 0; i < nchan && rc == GPSIQ_OK; ++i) {
                    if (!(ck.unknown[k][i] > 0 || host_owned[i])) continue;
                    // the slot's column (map + chain inputs + the starts the scan wrote), one small copy instead of every slot's rows
                    he = launch_gather_slot(d_chan, c->chain.d_maps, b0 - lead, nb + lead, nchan, i, e.d_slot, E);
                    if (he == hipSuccess) he = hipMemcpyAsync(e.h_slot, e.d_slot, (size_t) (nb + lead) * sizeof(EvalSlotRow), hipMemcpyDeviceToHost, E);
                    if (he == hipSuccess) he = hipStreamSynchronize(E);
                    if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, repair: %s", hipGetErrorString(he)); break; }
                    if (!host_owned[i]) {
                        // the state the slot enters the piece with: the scan knew every start up to its first block that does not link
                        host_last_prn[i] = lead && e.h_slot[0].prn > 0 ? e.h_slot[0].prn : 0;
                        host_end[i] = e.h_slot[lead].start;
                    }
                    long linked = 0, walked = 0;
                    int bad = -1;
                    if (!link_slot_host(e.h_slot + lead, b0, b0 + nb, delt, nsamp, &host_end[i], &host_last_prn[i], e.h_col, &linked, &walked, &bad)) {
                        rc = GPSIQ_E_RANGE; std::snprintf(err, sizeof err, "block %d: carrier phase or Doppler outside the NCO format", bad);
                        break;
                    }
                    chain_count(linked, walked);
                    he = hipMemcpyAsync(e.d_col, e.h_col, (size_t) nb * sizeof(double), hipMemcpyHostToDevice, E);
                    if (he == hipSuccess) he = launch_scatter_starts(d_chan, b0, nb, nchan, i, e.d_col, E);
                    if (he == hipSuccess) he = hipStreamSynchronize(E);                   // (the staging is reused for the next slot)
                    if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, repair upload: %s", hipGetErrorString(he)); break; }
                    if (!host_owned[i]) { host_owned[i] = true; ++e.last_repaired; }
                }
                e.host_ms += gpsiq_wall_ms() - tr; host_part[1] += gpsiq_wall_ms() - tr;
                if (rc != GPSIQ_OK) break;
            }
        }
        he = hipStreamWaitEvent(E, chained ? e.linked[k] : staged[npieces - 1], 0);
        if (he == hipSuccess)
            he = launch_eval(d_chan, b0, nb, nchan, delt, nsamp, c->d_tab, est_rows, est_stride, e.d_patches, list_cap(e.patch_cap), e.d_host, list_cap(e.host_cap), e.d_ctrl,
                             seeds ? e.d_seeds : nullptr, E);
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, piece %d evaluation: %s", k, hipGetErrorString(he)); }
    }

    // ---- the end of the evaluation: errors, the host walker's share, the patches -----------------------------------------------
    EvalCtrl &fin = e.h_ctrl[kEvalMaxPieces];
    fin = zero;
    size_t npatch_total = 0;
    bool fall_back = false;
    if (rc == GPSIQ_OK) {
        // (the fixed model's kernels ran on the staging stream: the results' way back waits for the last piece's)
        he = hipStreamWaitEvent(E, staged[npieces - 1], 0);
        if (he == hipSuccess) he = hipMemcpyAsync(&fin, e.d_ctrl, sizeof(EvalCtrl), hipMemcpyDeviceToHost, E);
        if (he == hipSuccess && reference) {
            // the first entries of both lists ride along (nearly always all there are)
            he = hipMemcpyAsync(e.h_patches, e.d_patches, 1024 * sizeof(gpsiq_patch_t), hipMemcpyDeviceToHost, E);
            if (he == hipSuccess) he = hipMemcpyAsync(e.h_host, e.d_host, 256 * sizeof(EvalHostItem), hipMemcpyDeviceToHost, E);
            if (he == hipSuccess && !seeds) he = hipMemcpyAsync(e.h_link, e.d_link, (size_t) nchan * sizeof(LinkCarry), hipMemcpyDeviceToHost, E);
        }
        if (he == hipSuccess && !reference) he = hipMemcpyAsync(e.h_fix, e.d_fix, (size_t) nchan * sizeof(FixedCarry), hipMemcpyDeviceToHost, E);
        if (he == hipSuccess) he = hipStreamSynchronize(E);
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, results: %s", hipGetErrorString(he)); }
    }
    if (rc == GPSIQ_OK && fin.err_key != ~0ull) {
        // the descriptor the device quantiser refused, in the host quantiser's words
        const size_t flat = (size_t) (fin.err_key >> 8);
        const int status = (int) (fin.err_key & 0xff);
        gpsiq_chan_t bad;
        if (kind == kSrcDevice) (void) hipMemcpy(&bad, ch + flat, sizeof bad, hipMemcpyDeviceToHost);
        else bad = ch[flat];
        gpsiq_qchan_t tmp;
        if (status == ev::kQStart) {
            rc = GPSIQ_E_RANGE;
            std::snprintf(err, sizeof err, "block %d: prn %d: start phase outside [0, 1]", (int) (flat / nchan), bad.prn);
        } else {
            if (reference && !(bad.carr_phase >= 0.0 && bad.carr_phase < 1.0)) bad.carr_phase = 0.0;       // (not the evaluation's business: eval_block)
            const int qrc = quantize_one(bad, delt, nsamp, nullptr, &tmp, nullptr);
            rc = qrc != GPSIQ_OK ? qrc : ev::qstatus_code(status);
            if (qrc != GPSIQ_OK) std::snprintf(err, sizeof err, "block %d: %.280s", (int) (flat / nchan), gpsiq_last_error());
            else std::snprintf(err, sizeof err, "block %d: descriptor refused by the device quantiser (status %d)", (int) (flat / nchan), status);
        }
    }
    std::vector<gpsiq_patch_t> patches;
    if (rc == GPSIQ_OK && reference) {
        if (fin.npatch > list_cap(e.patch_cap) || fin.nhost > list_cap(e.host_cap)) fall_back = true;       // more than the lists hold (a rate far outside the design range)
        else {
            if (fin.npatch > 1024) he = hipMemcpy(e.h_patches + 1024, e.d_patches + 1024, (size_t) (fin.npatch - 1024) * sizeof(gpsiq_patch_t), hipMemcpyDeviceToHost);
            if (he == hipSuccess && fin.nhost > 256) he = hipMemcpy(e.h_host + 256, e.d_host + 256, (size_t) (fin.nhost - 256) * sizeof(EvalHostItem), hipMemcpyDeviceToHost);
            if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, lists: %s", hipGetErrorString(he)); }
        }
        if (rc == GPSIQ_OK && !fall_back) {
            const double th = gpsiq_wall_ms();
            patches.assign(e.h_patches, e.h_patches + fin.npatch);
            if (fin.nhost) {
                // what the device left to the host walker: those channels' patches from eval_block, in place of whatever the device
                // had emitted for them before it gave up
                std::vector<uint64_t> keys;
                for (unsigned k = 0; k < fin.nhost; ++k) keys.push_back((uint64_t) e.h_host[k].block << 8 | e.h_host[k].slot);
                std::sort(keys.begin(), keys.end());
                patches.erase(std::remove_if(patches.begin(), patches.end(), [&](const gpsiq_patch_t &p) {
                    return std::binary_search(keys.begin(), keys.end(), (uint64_t) p.block << 8 | p.slot); }), patches.end());
                // (descriptors that lie in device memory: the few the walker needs come back in one go)
                if (kind == kSrcDevice) {
                    if (fin.nhost > e.items_cap) {
                        if (e.h_items) (void) hipHostFree(e.h_items);
                        e.h_items = nullptr; e.items_cap = 0;
                        if (hipHostMalloc((void **) &e.h_items, ((size_t) fin.nhost + 64) * sizeof(gpsiq_chan_t), hipHostMallocDefault) == hipSuccess) e.items_cap = fin.nhost + 64;
                    }
                    he = e.items_cap >= fin.nhost ? hipSuccess : hipErrorOutOfMemory;
                    for (unsigned k = 0; k < fin.nhost && he == hipSuccess; ++k)
                        he = hipMemcpyAsync(&e.h_items[k], ch + (size_t) e.h_host[k].block * nchan + e.h_host[k].chan, sizeof(gpsiq_chan_t), hipMemcpyDeviceToHost, E);
                    if (he == hipSuccess) he = hipStreamSynchronize(E);
                    if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, descriptors for the host walker: %s", hipGetErrorString(he)); }
                }
                if (rc == GPSIQ_OK) {
                    // (one walk of a block's samples each, ~0.1 ms: side by side on the pool, the lists joined in the items' order)
                    struct WJob { const EvalHostItem *items; const gpsiq_chan_t *own, *all; int nchan, nsamp; double delt; std::vector<gpsiq_patch_t> *out; int *rcs; std::array<char, 300> *texts; };
                    std::vector<std::vector<gpsiq_patch_t>> found(fin.nhost);
                    std::vector<int> rcs(fin.nhost, GPSIQ_OK);
                    std::vector<std::array<char, 300>> texts(fin.nhost);
                    WJob wj = {e.h_host, kind == kSrcDevice ? e.h_items : nullptr, ch, nchan, nsamp, delt, found.data(), rcs.data(), texts.data()};
                    parallel_for((int) fin.nhost, 0, 1, [](void *p, int k0, int k1) {
                        WJob &j = *static_cast<WJob *>(p);
                        for (int k = k0; k < k1; ++k) {
                            const EvalHostItem &h = j.items[k];
                            const gpsiq_chan_t &one = j.own ? j.own[k] : j.all[(size_t) h.block * j.nchan + h.chan];
                            gpsiq_qchan_t q;
                            j.rcs[k] = eval_block_host(one, h.start, &h.seed, j.delt, j.nsamp, (int) h.block, (int) h.slot, &q, &j.out[k]);
                            if (j.rcs[k] != GPSIQ_OK) std::snprintf(j.texts[k].data(), j.texts[k].size(), "block %u: %.280s", h.block, gpsiq_last_error());
                        }
                    }, &wj);
                    for (unsigned k = 0; k < fin.nhost; ++k) {
                        if (rcs[k] != GPSIQ_OK && rc == GPSIQ_OK) { rc = rcs[k]; std::snprintf(err, sizeof err, "%s", texts[k].data()); }
                        patches.insert(patches.end(), found[k].begin(), found[k].end());
                    }
                }
            }
            std::sort(patches.begin(), patches.end(), patch_before);
            npatch_total = patches.size();
            e.host_ms += gpsiq_wall_ms() - th; host_part[2] += gpsiq_wall_ms() - th;
        }
    }
    // the patches go behind the synthesis of every piece: join the two piece streams on the first
    if (rc == GPSIQ_OK && reference && !fall_back) {
        he = hipSuccess;
        if (npieces > 1) { he = hipEventRecord(e.joined, c->stream2); if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, e.joined, 0); }
        if (he == hipSuccess && npatch_total) {
            if (npatch_total > nb_.patch_cap) {
                if (nb_.d_patch) (void) hipFree(nb_.d_patch);
                nb_.d_patch = nullptr; nb_.patch_cap = 0;
                const size_t cap = npatch_total < 256 ? 256 : npatch_total;
                he = hipMalloc((void **) &nb_.d_patch, cap * sizeof(gpsiq_patch_t));
                if (he == hipSuccess) nb_.patch_cap = cap;
            }
            if (he == hipSuccess) {
                std::memcpy(e.h_patches, patches.data(), npatch_total * sizeof(gpsiq_patch_t));
                he = hipMemcpyAsync(nb_.d_patch, e.h_patches, npatch_total * sizeof(gpsiq_patch_t), hipMemcpyHostToDevice, c->stream);
            }
            uint8_t *dev = direct ? static_cast<uint8_t *>(dst) : static_cast<uint8_t *>(c->d_out);
            if (he == hipSuccess) he = launch_patches(nb_.d, nchan, nsamp, sample_size, dev, stride, 0, nblocks, c->d_tab, nb_.d_patch, (int) npatch_total, c->stream);
        }
        if (he == hipSuccess && !direct && npatch_total) {
            // the blocks apply_patches touched go to the destination again -- behind the pieces' own copies, which may still be
            // on their way (join the copy streams) -- or the whole timeline when that is most of it
            const hipMemcpyKind kd = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
            for (int k = 0; k < 2 && he == hipSuccess; ++k) {
                he = hipEventRecord(c->chunk_done[k], c->copy_stream[k]);
                if (he == hipSuccess) he = hipStreamWaitEvent(c->stream, c->chunk_done[k], 0);
            }
            size_t touched = 0;
            for (size_t k = 0; k < patches.size(); ++k) touched += k == 0 || patches[k].block != patches[k - 1].block;
            if (he == hipSuccess && touched * 4 > (size_t) nblocks) {
                if (stride == blk_bytes) he = hipMemcpyAsync(dst, c->d_out, blk_bytes * (size_t) nblocks, kd, c->stream);
                else he = hipMemcpy2DAsync(dst, blk_bytes, c->d_out, stride, blk_bytes, (size_t) nblocks, kd, c->stream);
            } else
                for (size_t k = 0; k < patches.size() && he == hipSuccess; ++k)
                    if (k == 0 || patches[k].block != patches[k - 1].block)
                        he = hipMemcpyAsync(static_cast<uint8_t *>(dst) + (size_t) patches[k].block * blk_bytes,
                                            static_cast<uint8_t *>(c->d_out) + (size_t) patches[k].block * stride, blk_bytes, kd, c->stream);
        }
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, patches: %s", hipGetErrorString(he)); }
    }

    // ---- GPSIQ_CHAIN_VERIFY=N: every N-th block that went through its certified map is also walked serially ---------------------
    // (on a few host threads, from the start state the scan gave it, when everything of the call has been queued and only the
    // synthesis is still running: gps.c:2821-2826 is the judge)
    const int verify_n = reference && !seeds ? chain_verify_every() : 0;
    unsigned verified = 0;
    if (rc == GPSIQ_OK && verify_n > 0) {
        const double tv = gpsiq_wall_ms();
        he = hipMemcpyAsync(h_chan, d_chan, n * sizeof(ev::DChan), hipMemcpyDeviceToHost, E);
        if (he == hipSuccess) he = hipMemcpyAsync(e.h_link, e.d_link, (size_t) nchan * sizeof(LinkCarry), hipMemcpyDeviceToHost, E);
        if (he == hipSuccess) he = hipStreamSynchronize(E);
        if (he != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation, verify: %s", hipGetErrorString(he)); }
        else {
            struct VJob { const ev::DChan *ch; const LinkCarry *end; const bool *skip; int nblocks, nchan, nsamp, every; double delt; long bad; unsigned count; };
            VJob vj = {h_chan, e.h_link, host_owned, nblocks, nchan, nsamp, verify_n, delt, -1, 0};
            parallel_for(nblocks, 4, 64, [](void *p, int b0, int b1) {
                VJob &j = *static_cast<VJob *>(p);
                unsigned count = 0;
                for (int b = b0; b < b1; ++b)
                    for (int i = 0; i < j.nchan; ++i) {
                        if ((b + i) % j.every || j.skip[i]) continue;
                        const ev::DChan &d = j.ch[(size_t) b * j.nchan + i];
                        if (d.prn <= 0 || !(d.start >= 0.0 && d.start <= 1.0) || !(std::fabs(d.f_carr * j.delt) < 0.5)) continue;
                        double want;
                        if (b + 1 < j.nblocks) {
                            const ev::DChan &nx = j.ch[(size_t) (b + 1) * j.nchan + i];
                            if (nx.prn != d.prn) continue;                       // the slot ends or is re-seeded: nobody reads this block's end
                            want = nx.start;
                        } else if (j.end[i].known) want = j.end[i].y;
                        else continue;
                        const double got = chain_block_true(d.f_carr, j.delt, j.nsamp, d.start);
                        ++count;
                        if (bits_of(got) != bits_of(want)) __sync_val_compare_and_swap(&j.bad, -1L, (long) b * j.nchan + i);
                    }
                __sync_fetch_and_add(&j.count, count);
            }, &vj);
            verified = vj.count;
            if (vj.bad >= 0) {
                rc = GPSIQ_E_VERIFY;
                std::snprintf(err, sizeof err, "block %ld slot %ld: the state after the block through its certified map is not the serial walk's (GPSIQ_CHAIN_VERIFY=%d)",
                              vj.bad / nchan, vj.bad % nchan, verify_n);
            }
        }
        e.host_ms += gpsiq_wall_ms() - tv;
    }

    // ---- drain: on every path nothing may still be writing the caller's buffer or reading our staging -------------------------
    const double t_drain = gpsiq_wall_ms();
    const hipError_t d0 = hipStreamSynchronize(S), d1 = hipStreamSynchronize(E), d2 = hipStreamSynchronize(c->stream), d3 = hipStreamSynchronize(c->stream2);
    const hipError_t d4 = hipStreamSynchronize(c->copy_stream[0]), d5 = hipStreamSynchronize(c->copy_stream[1]);
    if (rc == GPSIQ_OK)
        for (hipError_t d : {d0, d1, d2, d3, d4, d5})
            if (d != hipSuccess) { rc = GPSIQ_E_DEVICE; std::snprintf(err, sizeof err, "device evaluation: %s", hipGetErrorString(d)); break; }
    if (rc != GPSIQ_OK) return fail(rc, "%s", err);
    if (timed_blocks > 0) {
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, e.t_synth0, e.t_synth1) == hipSuccess && ms > 0.02f)
            gpsiq_note_kernel_rate((double) timed_blocks * (double) nsamp * (double) nchan / (ms * 1e-3));
    }
    if (fall_back) {
        // the lists overflowed: the host path does the whole call again (dst is rewritten)
        __atomic_fetch_add(&g_evd_stats[5], 1, __ATOMIC_RELAXED);
        return gpsiq_generate_reference_host(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, carr_phase_out, seeds);
    }

    // ---- the resident set and the carried state ------------------------------------------------------------------------------
    c->cur = next;
    c->d_desc = nb_.d;
    c->nblocks = nblocks; c->nchan = nchan; c->max_code_step = max_step; c->max_active = max_active > 0 ? max_active : 1; c->max_amplitude = max_amp;
    nb_.npatch = (int) npatch_total;
    nb_.active_per_block.assign((size_t) nblocks, (uint8_t) nchan);
    nb_.upload_pending = false;
    if (reference) {
        if (carr_phase_out && !seeds)
            for (int i = 0; i < nchan; ++i) {
                if (host_owned[i]) carr_phase_out[i] = host_last_prn[i] ? host_end[i] : last[i].carr_phase;
                else carr_phase_out[i] = e.h_link[i].prn > 0 ? e.h_link[i].y : last[i].carr_phase;
            }
        chain_count((long) fin.linked, 0);
    } else {
        for (int i = 0; i < nchan; ++i) {
            c->carry_prn[i] = e.h_fix[i].prn;
            c->carry[i] = e.h_fix[i].prn ? e.h_fix[i].phase : 0;
            c->handed[i] = e.h_fix[i].prn ? carr_phase_to_double(c->carry[i]) : 0.0;
            if (carr_phase_out) carr_phase_out[i] = c->handed[i];
        }
    }
    e.last_nhost = fin.nhost; e.last_npatch = (unsigned) npatch_total;
    __atomic_fetch_add(&g_evd_stats[1], (uint64_t) n, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_evd_stats[2], (uint64_t) fin.nhost, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_evd_stats[3], (uint64_t) e.last_repaired, __ATOMIC_RELAXED);
    __atomic_fetch_add(&g_evd_stats[4], (uint64_t) npatch_total, __ATOMIC_RELAXED);
    if (trace)
        std::fprintf(stderr, "[gpsiq trace] device evaluation (%s, descriptors %s), %d blocks in %d pieces (head %d): queued by %.3f ms, first synthesis launched at %.3f ms, "
                             "draining from %.3f ms, whole call %.3f ms; host stages %.3f ms (pack %.3f, repair %.3f, lists + walker %.3f); %u patches, %u channels to the host walker, %u slots repaired, %u blocks verified\n",
                     reference ? "reference NCO" : "fixed-point NCO", kind == kSrcDevice ? "in device memory" : kind == kSrcPinned ? "page-locked" : "pageable (packed by the pool)",
                     nblocks, npieces, ends[0], t_queued - t0, t_first_launch, t_drain - t0, gpsiq_wall_ms() - t0, e.host_ms, host_part[0], host_part[1], host_part[2], (unsigned) npatch_total, fin.nhost, e.last_repaired, verified);
    return GPSIQ_OK;
}
