// gpsiq_evalctl.h -- what gpsiq_device.cpp (host) and gpsiq_eval_kernels.hip (device) share about the device-side evaluation of
// a batch: the control block the kernels report into, the carries handed from piece to piece, and the launches.
#ifndef GPSIQ_EVALCTL_H
#define GPSIQ_EVALCTL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/gpsiq.h"
#include "gpsiq_tables.h"

namespace gpsiq {

constexpr int kEvalMaxPieces = 8;            // pieces of one batch call (a head, then what is left in growing pieces)

// One per call, in device memory; copied to page-locked host memory behind the kernels that write it.
struct EvalCtrl {
    unsigned long long max_code_step;        // launch parameters of the synthesis kernel, over everything packed so far
    long long          max_amp;
    int                max_active;
    unsigned           npatch;               // patches appended by eval_blocks (may exceed the list's capacity: then the host redoes the call)
    unsigned           nhost;                // (block, channel) pairs handed to the host walker
    unsigned           linked;               // blocks linked through their certified map (statistics)
    unsigned long long err_key;              // smallest (flat index << 8 | ev::QStatus) of a descriptor the quantiser refuses; ~0: none
    int                unknown[kEvalMaxPieces][GPSIQ_MAX_CHAN];   // per piece and slot: blocks that do not link through their map, or follow one that does not
};

struct LinkCarry { int64_t d; double y; int32_t prn; int32_t known; };     // GPSIQ_NCO_REFERENCE: offset, accumulator after the last block
struct FixedCarry { uint64_t phase; int32_t prn; int32_t cont; };          // GPSIQ_NCO_FIXED: exact carrier phase after the last block
struct EvalHostItem { uint32_t block; uint16_t chan, slot; double start; uint64_t seed; };   // slot: the channel's place among the block's active ones; start: its accumulator; seed: the carrier phase its descriptor was seeded with

// one block of one slot as the repair fetches it: the block's certified map and what the chain reads of its descriptor (88 bytes)
struct EvalSlotRow { gpsiq_chain_map_t map; double f_carr, carr_phase, start; int32_t prn, pad; };
hipError_t launch_gather_slot(const void *d_chan, const void *d_maps, int b0, int nb, int nchan, int slot, EvalSlotRow *d_out, hipStream_t s);
hipError_t launch_scatter_starts(void *d_chan, int b0, int nb, int nchan, int slot, const double *d_starts, hipStream_t s);
hipError_t launch_pack_raw(const gpsiq_chan_t *d_ch, int nblocks, int nchan, double delt, void *d_chan, EvalCtrl *d_ctrl, hipStream_t s);
hipError_t launch_link_scan(void *d_chan, const void *d_maps, int b0, int nb, int nchan, double delt, LinkCarry *d_carry, EvalCtrl *d_ctrl, int piece,
                            hipStream_t s);
hipError_t launch_quantize_est(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, const void *est_rows, int est_stride,
                               gpsiq_qchan_t *d_q, EvalCtrl *d_ctrl, hipStream_t s);
hipError_t launch_eval(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, const DeviceTables *tab, const void *est_rows, int est_stride,
                       gpsiq_patch_t *d_patches, unsigned patch_cap, EvalHostItem *d_host, unsigned host_cap, EvalCtrl *d_ctrl, const double *d_starts,
                       hipStream_t s);
hipError_t launch_quantize_fixed(const void *d_chan, int b0, int nb, int nchan, double delt, int nsamp, gpsiq_qchan_t *d_q, FixedCarry *d_carry,
                                 EvalCtrl *d_ctrl, hipStream_t s);

#ifdef GPSIQ_TEST_HOOKS
// GPSIQ_TEST_CORRUPT_MAP="block,slot" in the environment: that block's map gets another end offset (both parity branches)
hipError_t launch_test_corrupt_map(void *d_maps, int b0, int nb, int nchan, hipStream_t s);
#endif

}  // namespace gpsiq
#endif
