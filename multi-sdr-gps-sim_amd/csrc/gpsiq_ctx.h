// gpsiq_ctx.h -- the device context of libgpsiq (include/gpsiq.h: gpsiq_ctx_t), shared by the translation units of the device half:
// gpsiq_device.cpp (context, resident descriptors, launches, the drop-in calls) and gpsiq_evaldev.cpp (the batch calls whose
// descriptors are quantised / evaluated on the device).
#ifndef GPSIQ_CTX_H
#define GPSIQ_CTX_H

#include <hip/hip_runtime.h>

#include <vector>

#include "gpsiq_internal.h"
#include "gpsiq_evalctl.h"

struct gpsiq_ctx {
    int           device = -1;
    hipStream_t   stream = nullptr;
    hipStream_t   stream2 = nullptr;                     // every other piece of a batch in pieces (piece_stream below)
    gpsiq::DeviceTables *d_tab = nullptr;
    hipStream_t   copy_stream[2] = {nullptr, nullptr};   // device-to-host copies of the batch calls
    // resident descriptors, kSets buffers taken in turn: a new set is staged and uploaded into a buffer the latest launches
    // are NOT reading, so gpsiq_set_descriptors never waits for the device to go idle -- only, if it is still in flight, for the
    // launch from kSets sets ago that used the same buffer.  Four: the pieces of a batch alternate between two streams and a
    // piece's kernel shares the device with its neighbour's, so with two sets piece k+2 waited for a piece k that had been
    // slowed down by piece k+1 (GPSIQ_DESC_SETS=2 for the A/B, read per set)
    struct DescBuf {
        gpsiq_qchan_t *d = nullptr;  size_t cap = 0;      // device copy, in descriptors
        gpsiq_qchan_t *h = nullptr;  size_t hcap = 0;     // page-locked staging of the compacted descriptors
        // one event per stream that has launched on this buffer since it was last known idle: a launch on stream B must
        // not hide a longer one still running on stream A (when more than kUses streams are in play the extra ones
        // are chained behind the first, which then covers them)
        struct Use { hipStream_t s = nullptr; hipEvent_t ev = nullptr; bool active = false; };
        static constexpr int kUses = 4;
        Use            use[kUses];
        bool           in_use = false;
        std::vector<uint8_t> active_per_block;            // active channels of every resident block (patch validation)
        // patches that go with this set (GPSIQ_NCO_REFERENCE): per buffer, so that the next set's list can be uploaded
        // while launches of this one are still applying theirs
        gpsiq_patch_t *d_patch = nullptr;
        size_t         patch_cap = 0;
        int            npatch = 0;
        gpsiq_patch_t *h_patch = nullptr;  size_t h_patch_cap = 0;   // page-locked staging of the list (asynchronous sets)
        // a set staged without waiting (the pieces of a batch): the uploads are on up_stream, `uploaded` is recorded behind
        // them, and every launch on the set waits for it on its own stream
        hipEvent_t     uploaded = nullptr;
        bool           upload_pending = false;
    } buf[4];
    static constexpr int kSets = 4;
    hipStream_t    up_stream = nullptr;  // descriptor / patch uploads: never behind a running kernel
    int            cur = 0;             // buf[cur] holds the resident set
    gpsiq_qchan_t *d_desc = nullptr;    // == buf[cur].d
    int            nblocks = 0, nchan = 0;
    uint64_t       max_code_step = 0;
    int            max_active = 0;      // most active channels in any resident block
    long           max_amplitude = 0;   // largest sum over a block's channels of (int)(250*|gain|): bound on |I|, |Q|
    int            nco_mode = GPSIQ_NCO_FIXED;
    // scratch of the kernel variants that need some (segm: the sign masks of one launch)
    void          *d_scratch = nullptr;
    size_t         scratch_cap = 0;
    // staging for the synchronous entry points
    void          *d_out = nullptr;
    size_t         out_cap = 0;
    hipEvent_t     chunk_done[2] = {nullptr, nullptr};
    // gpsiq_generate_block_async: a small ring of per-block descriptor / output staging, each with the event that
    // says its block has landed
    struct AsyncSlot {
        gpsiq_qchan_t *d = nullptr, *h = nullptr;
        gpsiq_patch_t *d_patch = nullptr, *h_patch = nullptr;   // GPSIQ_NCO_REFERENCE: the block's patches, staged page-locked
        size_t         patch_cap = 0;
        void          *out = nullptr;
        size_t         out_cap = 0;
        hipEvent_t     done = nullptr;
        bool           busy = false;
    } aslot[4];
    int anext = 0;
    // carrier carry per slot (gpsiq_generate_block)
    uint64_t carry[GPSIQ_MAX_CHAN] = {};
    double   handed[GPSIQ_MAX_CHAN] = {};
    int      carry_prn[GPSIQ_MAX_CHAN] = {};
    // GPSIQ_NCO_REFERENCE batch calls: quantised descriptors and start states of the timeline being worked through, kept
    // between calls (a fresh 1.5 MB per call is four hundred page faults on the thread everything else waits for)
    std::vector<gpsiq_qchan_t> ref_q;
    std::vector<double>        ref_start;
    // the carrier chain of GPSIQ_NCO_REFERENCE on the device (gpsiq_chain_maps_device): inputs, estimates and maps of the
    // timeline being worked through, device side and page-locked staging, kept between calls
    struct Chain {
        size_t             cap = 0;                    // blocks x channels all of these hold
        gpsiq_chain_in_t  *d_in = nullptr, *h_in = nullptr;
        void              *d_prep = nullptr;           // lane::Prep, 32 bytes each
        gpsiq_chain_map_t *d_maps = nullptr, *h_maps = nullptr;
        gpsiq_chain_est_t *d_est = nullptr, *h_est = nullptr;       // [3][GPSIQ_MAX_CHAN]: start, end of the first launch (= start of a second), end
        double            *d_c_before = nullptr;
        hipStream_t        stream = nullptr, back = nullptr;   // uploads + kernels; the maps' way back + the callbacks (never in the kernels' way)
        hipEvent_t         t0 = nullptr, t1 = nullptr, landed = nullptr;   // landed: the maps of the last range queued are in h_maps
        hipEvent_t         walked[2] = {nullptr, nullptr};     // a launch's kernels are done
        float              last_ms = 0.0f;             // device time of the last call's two kernels
    } chain;
    // batch calls whose descriptors are quantised / evaluated on the device (gpsiq_evaldev.cpp; shares chain's buffers and streams)
    struct EvalDev {
        size_t          cap = 0;                       // block-channels d_chan / h_chan hold
        void           *d_chan = nullptr, *h_chan = nullptr;   // ev::DChan rows: device, page-locked staging (host-packed sources)
        gpsiq_chan_t   *d_raw = nullptr;  size_t raw_cap = 0;   // page-locked sources: the raw descriptors in HBM
        double         *d_seeds = nullptr; size_t seeds_cap = 0; // start states given by the caller (gpsiq_generate_seeded)
        gpsiq::EvalCtrl *d_ctrl = nullptr, *h_ctrl = nullptr;   // h_ctrl[kEvalMaxPieces + 1]: a snapshot behind every piece, one at the end
        gpsiq::LinkCarry *d_link = nullptr, *h_link = nullptr;  // [GPSIQ_MAX_CHAN]
        gpsiq::FixedCarry *d_fix = nullptr, *h_fix = nullptr;
        gpsiq_patch_t  *d_patches = nullptr, *h_patches = nullptr;  unsigned patch_cap = 0;
        gpsiq::EvalHostItem *d_host = nullptr, *h_host = nullptr;   unsigned host_cap = 0;
        gpsiq_chan_t   *h_items = nullptr;  unsigned items_cap = 0;                       // device-resident descriptors the host walker needs, page-locked
        gpsiq::EvalSlotRow *d_slot = nullptr, *h_slot = nullptr;  size_t slot_cap = 0;    // repair: one slot's column (+ one row before), device and page-locked
        double         *d_col = nullptr, *h_col = nullptr;                                 // ... and its start states on the way back
        hipStream_t     eval_stream = nullptr, chain_stream = nullptr;   // highest priority: beside the synthesis, ahead of its workgroups
        hipEvent_t      linked[gpsiq::kEvalMaxPieces] = {}, evaluated[gpsiq::kEvalMaxPieces] = {}, joined = nullptr, t_synth0 = nullptr, t_synth1 = nullptr;
        // statistics of the last call (gpsiq_evaldev_stats)
        double          host_ms = 0.0;                 // host thread-time spent on the call's descriptors (pack, repair, walker)
        unsigned        last_nhost = 0, last_npatch = 0, last_repaired = 0;
    } evd;
};

#define HIP_TRY(expr)                                                                        \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess)                                                                \
            return fail(GPSIQ_E_DEVICE, "%s: %s", #expr, hipGetErrorString(e_));             \
    } while (0)


// helpers of gpsiq_device.cpp the other translation unit uses
double gpsiq_wall_ms();
int gpsiq_wait_idle(gpsiq_ctx::DescBuf &b);
int gpsiq_mark_use(gpsiq_ctx::DescBuf &b, hipStream_t s);
int gpsiq_ensure_out(gpsiq_ctx *c, size_t bytes);
hipStream_t gpsiq_piece_stream(gpsiq_ctx *c, int k);
int gpsiq_chain_reserve(gpsiq_ctx *c, size_t n);
double gpsiq_rate_kernel();
void gpsiq_note_kernel_rate(double channel_samples_per_s);      // a measured rate of the synthesis kernel (running mean)
// GPSIQ_NCO_REFERENCE batch with chain link and evaluation on host threads (the path of rounds 4-5; fallback of the device path)
int gpsiq_generate_reference_host(gpsiq_ctx *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs,
                                  int sample_size, void *dst, int dst_is_device, double *carr_phase_out, const double *seeds);
// gpsiq_evaldev.cpp: the same call (either NCO model) with the descriptors quantised / evaluated on the device.  *handled = 0:
// not taken (too short a batch, switched off), the caller goes on with the host path
int gpsiq_generate_device(gpsiq_ctx *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs, int sample_size,
                          void *dst, int dst_is_device, double *carr_phase_out, const double *seeds, int *handled);
void gpsiq_evaldev_destroy(gpsiq_ctx *c);

namespace gpsiq {
hipError_t launch_variant(int variant, const gpsiq_qchan_t *desc, int nchan, int nsamp, int sample_size,
                          void *dst, size_t block_stride, int block0, int nblocks,
                          const DeviceTables *tab, hipStream_t stream, int max_active, long max_amplitude, void *scratch);
size_t variant_scratch_bytes(int variant, int nsamp, int nblocks);
hipError_t launch_patches(const gpsiq_qchan_t *desc, int nchan, int nsamp, int sample_size, void *dst, size_t block_stride,
                          int block0, int nblocks, const DeviceTables *tab, const gpsiq_patch_t *patches, int npatch,
                          hipStream_t stream);
hipError_t launch_chain(const void *d_in, int in_stride, int nblocks, int nchan, double delt, int nsamp, const gpsiq_chain_est_t *d_start,
                        int max_seg, void *d_prep, double *d_c_before, gpsiq_chain_est_t *d_end, void *d_maps, hipStream_t stream, int which = 3);
int chain_link(const gpsiq_chain_in_t *in, const void *maps, int nblocks, int nchan, double delt, int nsamp,
               const double *carr_in, const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn);
double chain_block_true(double f_carr, double delt, int nsamp, double start);
}
#endif
