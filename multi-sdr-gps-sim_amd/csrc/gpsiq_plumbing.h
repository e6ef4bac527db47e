/*
 * gpsiq_plumbing.h -- the library's own plumbing, for time-sharded hosts and the tests: the two halves of the GPSIQ_NCO_REFERENCE
 * host evaluation, the carrier chain parallel in time, statistics, kernel timing.  NOT part of the boundary (include/gpsiq.h) and not
 * exported: libgpsiq.so exports ONE entry for all of it,
 *     void *gpsiq_plumbing(const char *name);      the address of the plumbing function of that name, NULL if there is none
 * (gpsiq/__init__.py resolves its ctypes signatures through it; C callers use the typed wrappers at the end of this header).
 * Reference lines: gps.c:2821-2826 (the carrier accumulator these functions walk), gps.c:2208-2214 (re-seeding a slot).
 */
#ifndef GPSIQ_PLUMBING_H
#define GPSIQ_PLUMBING_H

#include "../../include/gpsiq.h"

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
__attribute__((visibility("default")))
#endif
void *gpsiq_plumbing(const char *name);
/* Besides the functions declared below, gpsiq_plumbing() knows four names that are the library's internals as libgpsiq_rows.so (the
 * rows either side of the path: refresh, nav message, RINEX, motion -- include/gpsiq_rows.h, gpsiq_extras.h) needs them, so that the
 * two libraries share ONE worker pool, ONE quantiser and ONE error text per thread (csrc/gpsiq_rows_link.cpp is the other end):
 * "set_error", "parallel_for", "quantize_one", "chain_carrier" (csrc/gpsiq_internal.h has their C++ signatures). */

/* The tables the library builds in place of the reference's, read back (tests hold them against the oracle's).
 * C/A code of one PRN as 0/1 chips (codegen() gps.c:272-309); the carrier LUTs (cosTable512 / sinTable512 gps.c:145-213). */
int  gpsiq_prn_code(int prn, uint8_t chips[GPSIQ_CA_SEQ_LEN]);
void gpsiq_carrier_table(int16_t cos512[512], int16_t sin512[512]);

/* One shard of a time-sharded run in GPSIQ_NCO_REFERENCE, whatever the context's mode: render nblocks blocks whose start
 * states are known (carr_start[nblocks][nchan], this range's rows of gpsiq_reference_chain; ch[b][i].carr_phase is not
 * read) into dst, host or device as gpsiq_generate_batch -- evaluated and rendered in pieces like it, with no reference
 * to the blocks before the range.  Synchronous.  Does not touch the carrier continuation state. */
int gpsiq_generate_seeded(gpsiq_ctx_t *ctx, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs,
                          int sample_size, const double *carr_start, void *dst, int dst_is_device);

/* The two halves of gpsiq_reference_batch on their own, for hosts that spread GPSIQ_NCO_REFERENCE over devices or processes.
 * Only the carrier chain is serial in time (gps.c:2821-2826: block b of a channel starts where the double accumulator left
 * block b-1); everything else about a block follows from its start state alone.
 *   gpsiq_reference_chain   the chain: the double every channel's accumulator holds at the START of every block.  It reads
 *                           three fields per channel and block (gpsiq_chain_in_t; gpsiq_chain_inputs extracts them), so a
 *                           process that refreshed only its own blocks can be sent the rest (24 bytes each), and the channels
 *                           are independent: rank r may walk channels [c0, c1) of the WHOLE timeline (pass those columns,
 *                           nchan = c1 - c0) while the other ranks walk theirs -- nchan-fold parallel, one thread per channel.
 *                           carr_in / prn_in (both or neither): accumulator and satellite of every slot after the block before
 *                           block 0 when the timeline continues an earlier one.  carr_start is [nblocks][nchan] (0.0 for an unused
 *                           slot); carr_end / last_prn (may be NULL): the state after the last block.
 *   gpsiq_reference_seeded  the rest, for blocks whose start states are known (carr_start[nblocks][nchan] from the chain;
 *                           ch[b][i].carr_phase itself is not read): descriptors seeded from them and the patches, exactly the
 *                           rows gpsiq_reference_batch over the whole timeline gives for these blocks (patch block indices count
 *                           from this call's block 0).  Blocks are independent: any split over threads, devices, processes.
 * Both host only, threaded over the shared pool (GPSIQ_THREADS). */
typedef struct gpsiq_chain_in {
    double  f_carr;       /* gpsiq_chan_t.f_carr */
    double  carr_phase;   /* gpsiq_chan_t.carr_phase: read where the slot is (re-)allocated (gps.c:2208-2214) */
    int32_t prn;          /* <= 0: unused slot */
    int32_t reserved;
} gpsiq_chain_in_t;
/* Counts since the process started, over every GPSIQ_NCO_REFERENCE evaluation: out[0] accumulator states that had to be known
 * (candidate samples; carrier and code count separately), out[1] of them decided from the block's start state by the drift
 * enclosure, out[2] / out[3] carrier / code states that took a walk of the accumulator from the block's start. */
void gpsiq_reference_stats(uint64_t out[4]);
void gpsiq_chain_inputs(const gpsiq_chan_t *ch, int n /* nblocks*nchan */, gpsiq_chain_in_t *out);
int gpsiq_reference_chain(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                          const double *carr_in, const int32_t *prn_in,
                          double *carr_start, double *carr_end, int32_t *last_prn);
int gpsiq_reference_seeded(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp, const double *carr_start,
                           gpsiq_qchan_t *out, gpsiq_patch_t *patches, int max_patches, int *npatches);

/* The chain, parallel in time (csrc/gpsiq_lane.h has the method).  x += c in double is a TRANSLATION on a whole residue class
 * of start states, so every block can be walked on its own from a representative start state near an estimate of the true
 * one (exact real arithmetic + modelled rounding drift), which yields a certified map of the block
 *     start xs + d*2^-53, lo <= d <= hi   ->   end e + (d + cum[parity of d])*2^-53;
 * the chain proper is then one exact subtraction, range check and addition per block (gpsiq_chain_link), with a true walk
 * of the rare block whose map does not apply.  carr_start / carr_end / last_prn are gpsiq_reference_chain's, bit for bit.
 *   gpsiq_chain_maps     level 1 of blocks [0, nblocks): every block independent of every other -- host threads here,
 *                        gpsiq_chain_maps_device runs the same walk with one GPU lane per stretch of a block.  start[nchan]:
 *                        the estimator state before block 0 (NULL: the timeline begins here, block 0 seeds every slot from its
 *                        own carr_phase; a continued timeline passes the accumulator itself: GPSIQ_CHAIN_EXACT, carr, prn,
 *                        f_carr of the block before).  max_stretches: pieces a block is cut into at most (<= 0: default).
 *                        end[nchan] (may be NULL): the estimator state after the last block.
 *   gpsiq_chain_link     level 2: the chain over these blocks from carr_in / prn_in (as gpsiq_reference_chain).
 *   gpsiq_chain_summary  what a RANGE of blocks does to the estimator, for ranks that each hold their own blocks only:
 *   gpsiq_chain_fold     first every rank summarises its range with start = NULL (the phase pass: exact phase advance, no
 *                        drift) and all ranks gather the summaries; fold(summaries of the ranks before me) is where my range
 *                        starts; summarising again from there adds the modelled drift (it needs the absolute phase), one more
 *                        gather and fold give the start for gpsiq_chain_maps.  After the maps (the expensive part, fully
 *                        parallel over ranks) the true states are relayed: rank r links its range from rank r-1's carr_end /
 *                        last_prn (16 doubles per rank; gpsiq/shard.py::reference_chain_by_time). */
typedef struct gpsiq_chain_est {
    uint64_t r_hi, r_lo;  /* phase in 2^-128 cycle, exact real arithmetic (the wrap is the overflow) */
    double   drift;       /* modelled rounding drift accumulated since the slot was seeded */
    double   carr;        /* GPSIQ_CHAIN_EXACT: the accumulator itself.  In a summary with first_prn: block 0's carr_phase */
    double   f_carr;      /* f_carr of the block before (the next block walks its tail) */
    int32_t  prn;         /* satellite of the block before; 0: none (the next block seeds itself) */
    int32_t  flags;       /* GPSIQ_CHAIN_* */
    int32_t  first_prn;   /* summary of the phase pass: block 0's satellite where it was taken to continue the slot */
    int32_t  reserved;
} gpsiq_chain_est_t;
#define GPSIQ_CHAIN_EXACT    1
#define GPSIQ_CHAIN_RESEEDED 2   /* summary: the slot was (re-)seeded inside the range, the state is absolute */
#define GPSIQ_CHAIN_EMPTY    4   /* summary of no blocks */
typedef struct gpsiq_chain_map {
    double  xs, e;        /* representative state at the block's first sample; state after the block */
    int64_t cum[2], lo, hi;   /* units of 2^-53; cum[p]: d an even / odd number of steps of the wrap's grid (a tie on a wrap sends odd ones aside) */
    int32_t ok, info;     /* ok 0: no map, the block is walked; bit p: holds for parity p.  info: bits 0-7 units per grid step */
} gpsiq_chain_map_t;
int gpsiq_chain_maps(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                     const gpsiq_chain_est_t *start, int max_stretches, gpsiq_chain_map_t *maps, gpsiq_chain_est_t *end);
int gpsiq_chain_link(const gpsiq_chain_in_t *in, const gpsiq_chain_map_t *maps, int nblocks, int nchan, double fs, int nsamp,
                     const double *carr_in, const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn);
int gpsiq_chain_summary(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                        const gpsiq_chain_est_t *start, gpsiq_chain_est_t *sum);
int gpsiq_chain_fold(const gpsiq_chain_est_t *sums /* [nranges][nchan] */, int nranges, int nchan, gpsiq_chain_est_t *out);
/* The relay of a time-sharded chain in ONE exchange instead of one per rank.  What a RANGE of blocks does to a slot's accumulator,
 * as a function of the state it enters with: the range's maps composed (csrc/gpsiq_eval.h, Link: integer translations that depend
 * on the entry offset mod 4), with the ranges of entry offsets for which every block of it links.
 *   gpsiq_chain_range       rank r, after level 1 of its own blocks: one record per slot (all-gather them, 160 bytes each);
 *   gpsiq_chain_range_fold  every rank, the same arithmetic on the same records: the (accumulator, satellite) every range is entered
 *                           with -- carr / prn / known [nranges + 1][nchan], row r: before range r, the last row: after the whole
 *                           timeline.  A range that has a block whose map does not apply for the state it is entered with cannot
 *                           be composed: what follows it in that slot is unknown (known 0) until the slot is re-seeded -- or until
 *                           the rank that owns the range has linked it and published the state it ended on: true_end / true_prn
 *                           [nranges][nchan] with true_known[nranges] (1: that rank's rows are final; may all be NULL).  Returns 1
 *                           when every state is known.  A rank links its own blocks (gpsiq_chain_link) once its row is known in all
 *                           slots; gpsiq/shard.py::reference_chain_by_time iterates fold and exchange until every rank has.
 * `abs_end` / `last_prn`: the state after the range when its first block seeds the slot itself (gpsiq_chain_link with no state
 * handed in) -- also the state after it whatever came before when the slot is re-seeded or unused somewhere inside (restart). */
typedef struct gpsiq_chain_range {
    double  xs, e;                 /* representative start of the range's first block; end of its last block for entry offset 0 */
    int64_t t[4], lo[4], hi[4];    /* entry offset d = (x - xs) / 2^-53, r = d mod 4: the offset at the last block is d + t[r], if lo[r] <= d <= hi[r] */
    int64_t cum_last[2];           /* the last block's own map: the end is e + (d_last + cum_last[parity]) * 2^-53 */
    double  abs_end;
    int32_t first_prn, last_prn;   /* satellite of the range's first / last block (0: unused) */
    int32_t ok;                    /* bit r: residue r usable */
    int32_t restart;               /* the slot is re-seeded or unused inside the range: the end does not depend on the entry state */
    int32_t nblocks, grid_last;    /* 0 blocks: transparent */
} gpsiq_chain_range_t;
int gpsiq_chain_range(const gpsiq_chain_in_t *in, const gpsiq_chain_map_t *maps, int nblocks, int nchan, double fs, int nsamp,
                      gpsiq_chain_range_t *out /* [nchan] */);
int gpsiq_chain_range_fold(const gpsiq_chain_range_t *ranges /* [nranges][nchan] */, int nranges, int nchan,
                           const double *true_end, const int32_t *true_prn, const uint8_t *true_known,
                           double *carr, int32_t *prn, uint8_t *known /* [nranges + 1][nchan] each */);

/* out[0] blocks linked through their map, out[1] blocks walked from their true start, since the process started */
void gpsiq_chain_stats(uint64_t out[2]);
/* gpsiq_chain_maps on the context's device: one lane per stretch of a block (max_stretches <= 0: 32, or GPSIQ_CHAIN_STRETCHES), `in` and `maps` host
 * memory; kernel_ms (may be NULL): device time of the two kernels.  Synchronous.  `end` is the ESTIMATOR's state after the last
 * block (phase, drift, satellite, f_carr) as gpsiq_chain_maps gives it, with two differences: `carr` is 0 and GPSIQ_CHAIN_EXACT is
 * never set (level 1 has an estimate, not the accumulator: only gpsiq_chain_link knows that), and GPSIQ_CHAIN_RESEEDED is also set
 * when block 0 continued an exact state handed in through `start`.  In GPSIQ_NCO_REFERENCE gpsiq_generate_batch
 * walks the chain of a batch this way itself (48 blocks or more; GPSIQ_CHAIN=host keeps the serial walk on host threads). */
int gpsiq_chain_maps_device(gpsiq_ctx_t *ctx, const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                            const gpsiq_chain_est_t *start, int max_stretches, gpsiq_chain_map_t *maps, gpsiq_chain_est_t *end,
                            float *kernel_ms);


/* ---- measurement ----------------------------------------------------------------------------------------------------------- */
/* Time iters back-to-back launches with HIP events on hip_stream; returns the mean
 * kernel-launch duration in milliseconds in *ms_per_launch. */
int gpsiq_time_launches(gpsiq_ctx_t *ctx, int block0, int nblocks, int nsamp, int sample_size,
                        void *dst, size_t block_stride_bytes, void *hip_stream, int variant,
                        int iters, float *ms_per_launch);
int         gpsiq_num_variants(void);
const char *gpsiq_variant_name(int variant);


/* device evaluation of the batch calls (csrc/gpsiq_evaldev.cpp): since the process started -- calls taken, (block, channel) pairs
 * evaluated on the device, pairs handed to the host walker, slots whose chain the host repaired, patches, calls that fell back */
void gpsiq_device_eval_stats(uint64_t out[6]);
/* host time the context's last device-evaluated batch spent on descriptors (pack of pageable rows, repair, the host walker's share) */
double gpsiq_device_eval_host_ms(const gpsiq_ctx_t *ctx);

#ifdef __cplusplus
}
#endif

/* typed access for C callers outside the library (host/gpsiq_shard.c): same names, resolved through gpsiq_plumbing() */
#ifndef GPSIQ_BUILDING_LIBRARY
static inline void gpsiq_p_chain_inputs(const gpsiq_chan_t *ch, int n, gpsiq_chain_in_t *out)
{
    typedef void (*fn_t)(const gpsiq_chan_t *, int, gpsiq_chain_in_t *);
    ((fn_t) gpsiq_plumbing("gpsiq_chain_inputs"))(ch, n, out);
}
static inline int gpsiq_p_reference_chain(const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp, const double *carr_in,
                                          const int32_t *prn_in, double *carr_start, double *carr_end, int32_t *last_prn)
{
    typedef int (*fn_t)(const gpsiq_chain_in_t *, int, int, double, int, const double *, const int32_t *, double *, double *, int32_t *);
    return ((fn_t) gpsiq_plumbing("gpsiq_reference_chain"))(in, nblocks, nchan, fs, nsamp, carr_in, prn_in, carr_start, carr_end, last_prn);
}
static inline int gpsiq_p_generate_seeded(gpsiq_ctx_t *ctx, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs, int sample_size,
                                          const double *carr_start, void *dst, int dst_is_device)
{
    typedef int (*fn_t)(gpsiq_ctx_t *, const gpsiq_chan_t *, int, int, int, double, int, const double *, void *, int);
    return ((fn_t) gpsiq_plumbing("gpsiq_generate_seeded"))(ctx, ch, nblocks, nchan, nsamp, fs, sample_size, carr_start, dst, dst_is_device);
}
static inline int gpsiq_p_prn_code(int prn, uint8_t *chips)
{
    typedef int (*fn_t)(int, uint8_t *);
    return ((fn_t) gpsiq_plumbing("gpsiq_prn_code"))(prn, chips);
}
static inline void gpsiq_p_carrier_table(int16_t *cos512, int16_t *sin512)
{
    typedef void (*fn_t)(int16_t *, int16_t *);
    ((fn_t) gpsiq_plumbing("gpsiq_carrier_table"))(cos512, sin512);
}
#endif
#endif
