// gpsiq_internal.h — shared between the host half and the device half of libgpsiq.
#ifndef GPSIQ_INTERNAL_H
#define GPSIQ_INTERNAL_H

#define GPSIQ_BUILDING_LIBRARY 1
#include "../../include/gpsiq_extras.h"      // gpsiq.h (the boundary) + gpsiq_rows.h (section 8f rows) + the frozen convenience set
#include "gpsiq_plumbing.h"                  // the library's own plumbing (hidden symbols, reached through gpsiq_plumbing())
#include "gpsiq_tables.h"

#include <pthread.h>
#include <atomic>
#include <cstdio>
#include <memory>
#include <vector>

namespace gpsiq {

int  fail(int code, const char *fmt, ...) __attribute__((format(printf, 2, 3)));
int  set_error(int code, const char *text);
void ca_code(int prn, uint8_t chips[GPSIQ_CA_SEQ_LEN]);
void build_device_tables(DeviceTables *t);
uint64_t carr_phase_to_fixed(double cycles);
double   carr_phase_to_double(uint64_t fixed);
int  quantize_one(const gpsiq_chan_t &ch, double delt, int nsamp, const uint64_t *carry_in,
                  gpsiq_qchan_t *q, uint64_t *carry_out);

// Quantise nblocks x nchan descriptors (threaded, every block from its own carr_phase), then
// chain the carrier exactly: block 0 slot i starts from carry0[i] where cont0[i] is set, a later
// block continues the previous one while the slot keeps its PRN, and re-seeds from its own
// carr_phase otherwise.  carry_end / last_prn (may be null) receive the state after the last block.
// The serial half of quantize_timeline: p_{k+1} = p_k + nsamp*step_k (mod 2^59) down each slot of an already
// quantised (self-seeded) timeline; cont0/carry0 as in quantize_timeline.
void chain_carrier(gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, const bool *cont0, const uint64_t *carry0,
                   uint64_t *carry_end, int *last_prn);
int quantize_timeline(const gpsiq_chan_t *ch, int nblocks, int nchan, double delt, int nsamp,
                      const bool *cont0, const uint64_t *carry0, gpsiq_qchan_t *q,
                      uint64_t *carry_end, int *last_prn);

// GPSIQ_NCO_REFERENCE (gpsiq_exact.cpp): quantise a timeline with every block seeded from the carrier
// phase the reference's double accumulator holds at its start, and list the samples where the double
// path differs from the closed form.  carr_end / last_prn (may be null): state after the last block.
// carr_in / prn_in (both or neither; may be null): the accumulator and satellite of every slot after the block before
// block 0, when this timeline continues one that an earlier call walked (block 0 then only re-seeds a slot whose PRN changed).
int reference_timeline(const gpsiq_chan_t *ch, int nblocks, int nchan, double delt, int nsamp,
                       gpsiq_qchan_t *q, std::vector<gpsiq_patch_t> *patches, double *carr_end, int *last_prn,
                       const double *carr_in = nullptr, const int *prn_in = nullptr);

// The same as a walker that can be consumed piece by piece while it runs (gpsiq_exact.cpp): the timeline is cut into pieces,
// piece k = blocks [ends[k-1], ends[k]); per piece and channel one CHAIN task (serial down the channel: the start state of
// every block) and one EVAL task (descriptors + patches from the start states, any thread); piece k is complete when every
// channel's EVAL (chain_only: CHAIN) of it has finished.
struct CodeCache;
struct RefWalk {
    RefWalk(const gpsiq_chan_t *ch, int nblocks, int nchan, double delt, int nsamp, gpsiq_qchan_t *q,
            const double *carr_in, const int *prn_in, const std::vector<int> &piece_ends);
    ~RefWalk();
    RefWalk(const RefWalk &) = delete;
    RefWalk &operator=(const RefWalk &) = delete;
    void run();                                   // returns when every task has run; call from any ONE thread (it works too)
    int  wait_piece(size_t k);                    // blocks until piece k is complete; the first error so far (GPSIQ_OK if none)
    // the patches of piece k (complete: wait_piece(k) first, or after run()), sorted by (block, sample, slot); relative: block
    // indices counted from the piece's first block
    void take_patches(size_t k, std::vector<gpsiq_patch_t> *out, bool relative);
    size_t npieces() const { return ends.size(); }
    void abort() { abort_flag.store(true, std::memory_order_release); }     // the remaining tasks finish at once (pieces still complete)

    const gpsiq_chan_t *ch; int nblocks, nchan, nsamp; double delt; gpsiq_qchan_t *q;
    std::vector<int> ends;
    bool have_in;
    double carr_in[GPSIQ_MAX_CHAN]; int prn_in[GPSIQ_MAX_CHAN];
    double carr_end[GPSIQ_MAX_CHAN]; int last_prn[GPSIQ_MAX_CHAN];       // valid after run()
    int rc = GPSIQ_OK; char err[320] = "";
    // set before run():
    const gpsiq_chain_in_t *in = nullptr;         // chain inputs in compact form instead of ch (chain_only)
    const double *seeds = nullptr;                // [nblocks][nchan] start states known: no chain tasks
    bool    chain_only = false;                   // no EVAL tasks (q, ch may be null when `in` is given)
    double *start_out = nullptr;                  // where the chain publishes the start states (default: own storage)
    const void *maps = nullptr;                   // [nblocks][nchan] certified maps of the blocks (gpsiq_lane.h): the chain tasks link
    void release_maps(int upto);                  // ... through them; only blocks [0, maps_upto) may be touched yet (default: all)
    std::atomic<int> maps_upto{0x7fffffff};
private:
    void work();
    void chain_task(int i, size_t k);
    void eval_task(int i, size_t k, CodeCache *codes);
    void finish_piece(size_t k);
    void set_error(int code, const char *text, int block);
    std::vector<std::vector<gpsiq_patch_t>> patches;                    // [channel][piece]: written by that EVAL task alone
    pthread_mutex_t mu;
    pthread_cond_t cv, task_cv;
    std::unique_ptr<std::atomic<int>[]> done;
    std::atomic<bool> abort_flag{false};
    bool no_drift = false;                        // GPSIQ_NO_DRIFT: walk every candidate (A/B and tests)
    int  verify_every = 0;                        // GPSIQ_CHAIN_VERIFY
    // scheduler state, under mu
    size_t chain_next[GPSIQ_MAX_CHAN], eval_next[GPSIQ_MAX_CHAN];
    bool   chain_busy[GPSIQ_MAX_CHAN];
    double carr[GPSIQ_MAX_CHAN]; int prev[GPSIQ_MAX_CHAN];               // the chain's state per channel (owned by the running CHAIN task)
    const double *start = nullptr;
    std::vector<double> own_start;
};

// level 2 of the time-parallel chain for one block (gpsiq_chain.cpp): the accumulator after block `at` of maps from its true start
// x; false: the block's map does not apply (walk it).  chain_count: the process-wide statistics (gpsiq_chain_stats).
bool chain_step_mapped(const void *maps, size_t at, double x, double *next);
void chain_prefetch_map(const void *maps, size_t at);
void chain_count(long linked, long walked);
int  chain_verify_every();                        // GPSIQ_CHAIN_VERIFY=N: every N-th linked block is also walked; 0: off

// Host threads this process may use: online CPUs, capped by GPSIQ_THREADS (read once).
int host_threads();
// Run entry(ctx) on the calling thread and on up to nthreads - 1 workers of the shared pool at the same time (<= 0: as many
// as host_threads() allows; never more than that).  entry hands out its own work and returns when none is left; run_job
// returns when every thread that entered has returned.  Jobs of several callers share the pool (gpsiq_host.cpp).
void run_job(void (*entry)(void *ctx), void *ctx, int nthreads);
// Run fn(ctx, begin, end) over [0, n) on up to nthreads host threads (<= 0: one per online
// CPU, but at least `grain` items per thread; never more than host_threads()).  Returns after all parts are done.
void parallel_for(int n, int nthreads, int grain, void (*fn)(void *ctx, int begin, int end), void *ctx);

// |gain| bound of every entry point that takes a gain: (int)(250*|gain|) of 16 channels must fit the 32-bit sums with
// room to spare, and the LUT build converts table*gain to int.  (The reference's gains are below 2.)
constexpr double kMaxGain = 4.0e6;

// Kernel variants (gpsiq_launch's `variant`).
enum Variant {
    kAuto = 0,      // fast when every resident descriptor allows it, else generic
    kGeneric = 1,   // one sample per thread, full-width closed form per sample (any rate)
    kRows = 2,      // 64-sample rows per wave, incremental NCOs, LDS-staged windows
    kRowsX = 3,     // same rows, channel-inner loop order with all NCO state in registers
    kTile = 4,      // rowsx with trimmed per-tile overhead, 64 rows per wave (32768-sample tiles)
    kSeg = 5,       // tile kernel, each wave running several consecutive 64-row chunks
    kSegHalf = 6,   // seg with one window per 32 samples: for sample rates down to 1.023 Msps
    kSegMask = 7,   // high sample rates: per-(channel,row) 64-bit sign masks from a pre-pass, applied as EXEC masks
    kSegBoth = 8,   // seg's plain-add core with a both-polarity LUT (sign concatenated above the index), 16-wave workgroups
    kNumVariants
};

// The row kernel needs all 64 lanes of a row inside one 32-chip window:
// 63*code_step + (1 chip) <= 32 chips.
constexpr uint64_t kRowsMaxCodeStep = ((UINT64_C(31) << GPSIQ_CODE_FRAC_BITS) - 1) / 63;
// With a window per half row (32 lanes): 31*code_step + (1 chip) <= 32 chips, i.e. up to one
// chip per sample (fs >= 1.023 Msps).
constexpr uint64_t kHalfRowsMaxCodeStep = ((UINT64_C(31) << GPSIQ_CODE_FRAC_BITS) - 1) / 31;

}  // namespace gpsiq
#endif
