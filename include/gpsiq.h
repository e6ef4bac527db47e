/*
 * gpsiq.h — C-ABI of libgpsiq, the MI355X-native GPS L1 C/A IQ synthesiser.
 *
 * This library replaces ONE thing in Mictronics/multi-sdr-gps-sim: the per-sample
 * synthesis loop and its int8/int16 pack that are inlined in gps_thread_ep()
 * (reference gps.c:2767-2836 and gps.c:2839-2846).  Everything either side of it
 * (RINEX/ephemeris/pseudorange host model above, fifo.h / sdr_* sinks below) stays
 * as it is.  The reference has no function boundary at this point, so the boundary
 * is defined here: plain C, plain pointers and sizes, no C++/torch types.
 *
 * Call-site map (reference file:line -> entry point that replaces it)
 *   gps.c:2767-2846  per-block sample loop + pack      -> gpsiq_generate_block()
 *   gps.c:2703-2933  the 10 Hz block loop, run ahead   -> gpsiq_generate_batch()
 *   gps.c:2847-2865  HackRF 262144-element chunking /
 *                    iqfile+Pluto one-block hand-off   -> gpsiq_chunker_*()
 *   gps.c:272-309    codegen() C/A sequence             -> built in-library (read back: gpsiq_prn_code, csrc/gpsiq_plumbing.h)
 *   gps.c:145-213    sinTable512 / cosTable512          -> built in-library (read back: gpsiq_carrier_table, ibid.)
 *   gps.h:213-236    channel_t (fields the loop reads)  -> gpsiq_chan_t
 *   fifo.h:19-63     the block FIFO (API kept)           -> multi-sdr-gps-sim_amd/host/fifo.[ch]
 *   the rows either side of the path (host refresh, nav words, RINEX readers) -> include/gpsiq_rows.h
 *   reference host / CLI pieces outside section 8 (frozen convenience set)                    -> include/gpsiq_extras.h
 *
 * WHAT A MAINTAINER OF THE REFERENCE HAS TO READ.  The drop-in boundary is fifteen functions:
 *   gpsiq_create / gpsiq_destroy / gpsiq_set_nco_mode / gpsiq_last_error            the context
 *   gpsiq_generate_block (+ _async / gpsiq_wait)                                    gps.c:2767-2846, one call per 0.1 s block
 *   gpsiq_generate_batch / gpsiq_generate_batch_multi                               the same loop run ahead, one or several GPUs
 *   gpsiq_chunker_init / _push / _reserve / _commit                                 gps.c:2847-2865, the fifo hand-off
 *   gpsiq_host_alloc / gpsiq_host_free                                              page-locked fifo buffers
 * with gpsiq_chan_t as the only input type: the sections marked [boundary].  The sections marked [sharding] are the
 * resident-descriptor path and the time-axis sharding recipe: multi-GPU hosts, bench.py.
 *
 * NCO definition ("identical fixed-point NCO word widths", BASELINE.json north_star).
 * The reference advances both NCOs with sequential double additions (gps.h:17
 * FLOAT_CARR_PHASE; gps.c:2789, 2821).  libgpsiq and its CPU oracle evaluate the
 * same recurrences in closed form on integers:
 *   carrier: phase accumulator of GPSIQ_CARR_FRAC_BITS = 59 bits (cycles, wraps mod 1),
 *            LUT index = top 9 bits               (gps.c:2775 floor(carr_phase*512))
 *   code   : chip counter + GPSIQ_CODE_FRAC_BITS = 56 fractional bits
 *                                                  (gps.c:2789-2817)
 * for sample n of a block, with every quantity an exact integer:
 *   P(n) = (carr_phase + n*carr_step) mod 2^59          idx  = P(n) >> 50
 *   T(n) = code_frac + n*code_step                      A(n) = chip0 + (T(n) >> 56)
 *   chip = A(n) % 1023      period = A(n) / 1023        bit  = (icode + period) / 20
 *   neg  = prn_chip[chip] ^ ((nav_bits >> bit) & 1)   (dataBit*codeCA == -1  <=>  neg == 1)
 *   I   += neg ? -TC[idx] : TC[idx]   with TC[k] = (int)(cosTable512[k]*gain)  (C truncation)
 *   Q   += neg ? -TS[idx] : TS[idx]   with TS[k] = (int)(sinTable512[k]*gain)
 * and the outputs are (short)I,(short)Q (gps.c:2834-2835), or (signed char)((short)x >> 4)
 * for 8-bit sinks (gps.c:2845).  SURVEY.md section 0 fact 3 explains why >= 56 bits.
 */
#ifndef GPSIQ_H
#define GPSIQ_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* libgpsiq.so is built with -fvisibility=hidden: what these headers declare is what it exports */
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

#define GPSIQ_MAX_CHAN        16    /* reference MAX_CHAN is 12 (gps.h:36); BASELINE configs 3/5 use 16 */
#define GPSIQ_N_DWRD          60    /* gps.h:52 N_DWRD */
#define GPSIQ_CA_SEQ_LEN      1023  /* gps.h:58 */
#define GPSIQ_CARR_FRAC_BITS  59
#define GPSIQ_CODE_FRAC_BITS  56
#define GPSIQ_MAX_NAV_BITS    32    /* nav bits one block may touch (20 ms each) */
#define GPSIQ_HACKRF_CHUNK    262144 /* sdr.h:33 HACKRF_TRANSFER_BUFFER_SIZE, in IQ elements */

/* sample formats == simulator_t.sample_size in bytes (gps-sim.h:26-27 SC08 / SC16) */
#define GPSIQ_SC08 1
#define GPSIQ_SC16 2

/* sink kinds == sdr_type_t (gps-sim.h:30-32); only affects chunking, gps.c:2847-2865 */
#define GPSIQ_SINK_IQFILE   1
#define GPSIQ_SINK_HACKRF   2
#define GPSIQ_SINK_PLUTOSDR 3

/* error codes: 0 ok, negative on failure (the sdr_* convention, sdr.c:48-66, uses 0 / -1) */
#define GPSIQ_OK             0
#define GPSIQ_E_ARG         -1   /* bad argument (NULL, nchan > GPSIQ_MAX_CHAN, prn out of range ...) */
#define GPSIQ_E_RANGE       -2   /* descriptor outside what the NCO format represents */
#define GPSIQ_E_DEVICE      -3   /* HIP runtime error, see gpsiq_last_error() */
#define GPSIQ_E_NOMEM       -4
#define GPSIQ_E_STATE       -5   /* call order (e.g. launch before descriptors are resident) */
#define GPSIQ_E_VERIFY      -6   /* GPSIQ_CHAIN_VERIFY: a block linked through its certified map ended on another state than the serial walk
                                    (gps.c:2821-2826) from the same start: the output of this call is not to be trusted; block and slot in gpsiq_last_error() */

/* Per-channel block descriptor: exactly the channel_t fields (gps.h:213-236) plus
 * gain[i] (gps.c:2300, 2756-2763) that the sample loop reads, with the state the
 * host refresh leaves at gps.c:2766.  dataBit/codeCA are not carried: they are
 * functions of (dwrd, iword, ibit) and (ca, code_phase) (gps.c:2058-2059). */
typedef struct gpsiq_chan {
    int32_t  prn;                 /* 1..32; <= 0 marks an unused slot (gps.c:2772) */
    int32_t  iword;               /* 0..59  word index into dwrd  (gps.c:2049) */
    int32_t  ibit;                /* 0..29  bit inside the word   (gps.c:2052) */
    int32_t  icode;               /* 0..19  code period inside the bit (gps.c:2055) */
    double   f_carr;              /* Hz, Doppler (gps.c:2042) */
    double   f_code;              /* Hz, chipping rate (gps.c:2043) */
    double   carr_phase;          /* cycles in [0,1) (gps.c:2214); see gpsiq_generate_block */
    double   code_phase;          /* chips in [0,1023) (gps.c:2047) */
    double   gain;                /* gain[i] (gps.c:2756) */
    uint32_t dwrd[GPSIQ_N_DWRD];  /* nav words, bits 29..0 used (gps.c:2811) */
} gpsiq_chan_t;

/* Quantised descriptor: what the device kernel (and the oracle's closed form)
 * consumes.  48 bytes; one per (block, channel slot). */
typedef struct gpsiq_qchan {
    uint64_t carr_phase;  /* [0, 2^59) units of 2^-59 cycle */
    int64_t  carr_step;   /* per sample, units of 2^-59 cycle, |step| < 2^58 */
    uint64_t code_frac;   /* [0, 2^56) units of 2^-56 chip */
    uint64_t code_step;   /* per sample, units of 2^-56 chip, < 2^57 (2 chips/sample) */
    double   gain;
    uint32_t nav_bits;    /* bit b = data bit of the b-th nav-bit period this block touches */
    uint16_t chip0;       /* 0..1022 */
    uint8_t  icode;       /* 0..19 */
    uint8_t  prn;         /* 1..32, 0 = unused slot */
} gpsiq_qchan_t;

typedef struct gpsiq_ctx gpsiq_ctx_t;

/* NCO models.  GPSIQ_NCO_FIXED (default): the closed form on integers defined above; every block is
 * an independent function of its descriptor, so the time axis shards freely.  GPSIQ_NCO_REFERENCE: the
 * reference's own double accumulators (gps.c:2789-2792, 2821-2826), reproduced exactly: a whole run
 * equals the reference element for element and carr_phase is handed out as the reference's accumulator
 * leaves it.  The device runs the same synthesis kernels plus a fix-up of the few samples per 10^7 where the two models
 * differ.  A batch (gpsiq_generate_batch, 48 blocks or more) needs no host thread for it: quantiser, carrier chain (parallel
 * in time: a certified map per block, linked by a scan) and the search for those samples are kernels
 * (csrc/gpsiq_evaldev.cpp); descriptors that lie in page-locked (gpsiq_host_alloc) or device memory are read where they
 * are, pageable ones are first cut down to 64 bytes each on the host pool.  The block call and short batches walk the
 * carrier on host threads (~2 us per block and channel, csrc/gpsiq_exact.cpp); GPSIQ_EVAL=host takes every batch there. */
#define GPSIQ_NCO_FIXED      0
#define GPSIQ_NCO_REFERENCE  1

/* One sample of one channel where the reference's double path takes another LUT entry or sign than the
 * fixed-point closed form: the device recomputes that sample with (lut, neg) for this channel. */
typedef struct gpsiq_patch {
    uint32_t block;   /* index into the resident descriptor timeline */
    uint32_t sample;  /* 0 .. nsamp-1 */
    uint8_t  slot;    /* channel, in device order: the block's active channels counted from 0 */
    uint8_t  neg;     /* 1: dataBit*codeCA == -1 (gps.c:2781-2782) */
    uint16_t lut;     /* iTable, gps.c:2775 */
} gpsiq_patch_t;

/* ---- [boundary] library / tables (no device needed) ----------------------- */
const char *gpsiq_version(void);
/* [sharding / measurement] first 16 hex digits of the SHA-256 over the device-code sources (DEVSRC of csrc/Makefile) this library was built from: a
 * profile taken from one library (profiles/pmc_*.json record it) is only replayed next to measurements of the same one */
const char *gpsiq_kernels_id(void);
/* last error text of the calling thread ("" if none) */
const char *gpsiq_last_error(void);
/* (the tables the library builds in place of codegen() gps.c:272-309 and cosTable512 / sinTable512 gps.c:145-213 can be read back
 * for checking: gpsiq_prn_code / gpsiq_carrier_table, csrc/gpsiq_plumbing.h) */

/* ---- [sharding] the quantiser on its own -------------------------------------
 * Quantise nchan descriptors for a block of nsamp samples at fs Hz.
 * delt = 1.0/fs as in gps.c:2298; steps are rint(f*delt*2^F).
 * carry_in: if non-NULL, carry_in[i] replaces the phase derived from ch[i].carr_phase.
 * carry_out: if non-NULL, receives the exact carrier phase after nsamp samples
 *            (the only state the loop hands to the next block, gps.c:2821).
 * Unused slots (prn <= 0) produce prn = 0.  out has nchan entries.
 * Range checks (GPSIQ_E_RANGE), the same in every entry point that takes gpsiq_chan_t: |f_carr/fs| < 0.5,
 * 0 < f_code/fs < 2, carr_phase in [0,1), code_phase in [0,1023), iword/ibit/icode inside dwrd, and gain finite
 * with |gain| < 4e6 -- gpsiq_set_descriptors applies the same bound to descriptors quantised elsewhere, so a set
 * quantised on one node is not rejected late on another. */
int gpsiq_quantize(const gpsiq_chan_t *ch, int nchan, double fs, int nsamp,
                   gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out);

/* The same for a whole timeline of nblocks consecutive blocks, ch[nblocks][nchan] ->
 * out[nblocks][nchan], quantised on host threads and chained with the exact carrier prefix
 * p_{k+1} = p_k + nsamp*step_k (mod 2^59): block 0 starts from carry_in (if non-NULL) or its
 * own carr_phase; a later block continues the previous one while the slot keeps its PRN and
 * re-seeds from its own carr_phase when the slot is re-allocated (gps.c:2208-2210).
 * This is what gpsiq_generate_batch does internally; a multi-GPU host calls it once and gives
 * every device a contiguous slice of `out` (gpsiq_shard_range + gpsiq_set_descriptors), so
 * each shard starts bit-exactly where its predecessor ends and no device talks to another. */
int gpsiq_quantize_batch(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp,
                         gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out);

/* GPSIQ_NCO_REFERENCE form of gpsiq_quantize_batch.  Block 0 starts from ch[0][i].carr_phase (the
 * double the previous call handed out, or allocateChannel's value); later blocks continue the
 * reference's accumulator and re-seed from their own carr_phase when the slot's PRN changes.  out is
 * [nblocks][nchan]; patches receives up to max_patches entries sorted by (block, sample, slot),
 * *npatches their number (GPSIQ_E_RANGE if there is not enough room: call again with *npatches);
 * carr_phase_out[nchan] (may be NULL) the accumulator after the last block.  Host only. */
int gpsiq_reference_batch(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp,
                          gpsiq_qchan_t *out, gpsiq_patch_t *patches, int max_patches, int *npatches,
                          double *carr_phase_out);

/* The two halves of gpsiq_reference_batch on their own (gpsiq_reference_chain / _seeded), the carrier chain parallel in time
 * (gpsiq_chain_maps / _link / _summary / _fold / _maps_device) and the statistics counters are the library's own plumbing: the
 * batch calls below use them internally, time-sharded hosts (gpsiq/shard.py, host/gpsiq_shard.c) and the tests reach them through
 * multi-sdr-gps-sim_amd/csrc/gpsiq_plumbing.h.  They are not exported symbols. */

/* Contiguous balanced split of a block timeline over `world` devices/processes:
 * rank r owns [*begin, *end); the first nblocks % world ranks own one block more. */
int gpsiq_shard_range(int nblocks, int rank, int world, int *begin, int *end);

/* Sharding the HOST side too (GPSIQ_NCO_FIXED): a rank quantises only its own blocks,
 *   gpsiq_quantize_batch(ch_own, n_own, ..., carry_in = NULL)      block 0 seeded from its own carr_phase,
 *   gpsiq_shard_carry(q_own, ...)  -> 16 x 32 bytes                what its range does to each slot's carrier,
 *   (all ranks exchange those records: an all-gather of 512 bytes, setup only, not on the data path)
 *   gpsiq_shard_seed(q_own, ..., all_records, rank)                adds the exact prefix of the ranks before it,
 * after which q_own equals the same rows of gpsiq_quantize_batch over the whole timeline. */
typedef struct gpsiq_shard_carry {
    uint64_t end_phase;   /* carrier phase after the range's last block, with block 0 seeded from its own carr_phase */
    uint64_t advance;     /* sum of nsamp*carr_step over the range, mod 2^59 (meaningful while the slot keeps one PRN) */
    int32_t  first_prn, last_prn;   /* PRN in the first / last block of the range (0 = unused slot) */
    int32_t  reseeded;    /* the slot changed PRN inside the range: end_phase does not depend on the ranks before */
    int32_t  nblocks;     /* 0: an empty range, transparent to the chain */
} gpsiq_shard_carry_t;

int gpsiq_shard_carry(const gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, gpsiq_shard_carry_t *out /* [nchan] */);
int gpsiq_shard_seed(gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp,
                     const gpsiq_shard_carry_t *all /* [world][nchan], rank-major */, int rank);

/* ---- [boundary] device context and the drop-in calls ------------------------ */
/* device = HIP device ordinal.  GPSIQ_E_DEVICE when no GPU is present: the library has no CPU path. */
int  gpsiq_create(gpsiq_ctx_t **ctx, int device);
void gpsiq_destroy(gpsiq_ctx_t *ctx);

/* Select the NCO model of gpsiq_generate_block / gpsiq_generate_batch (default GPSIQ_NCO_FIXED). */
int  gpsiq_set_nco_mode(gpsiq_ctx_t *ctx, int mode);

/* Drop-in for one pass of gps.c:2767-2846: synthesise one block of nsamp complex
 * samples from the channel state at gps.c:2766 and write 2*nsamp IQ elements
 * (int8 for GPSIQ_SC08, int16 for GPSIQ_SC16) to the HOST buffer dst
 * (iq->data8 / iq->data16 of the acquired fifo buffer).  Synchronous.
 * carr_phase_out[i] (may be NULL) receives the carrier phase after the block, as
 * the loop leaves it in chan[i].carr_phase.  The context remembers the exact
 * 59-bit phase per slot: when the next call passes back the same prn and the same
 * carr_phase double it handed out, the exact value is continued; any other value
 * (allocateChannel re-initialising a slot, gps.c:2208-2214) re-seeds from the double.
 * (Staged through a slot of its own: the resident set of gpsiq_set_descriptors is not touched.) */
int gpsiq_generate_block(gpsiq_ctx_t *ctx, const gpsiq_chan_t *ch, int nchan,
                         int nsamp, double fs, int sample_size,
                         void *dst, double *carr_phase_out);

/* The same call without waiting for the device (both NCO models; in GPSIQ_NCO_REFERENCE the host walks the block's
 * carrier first, so the block's patches are queued with it): it returns once the descriptor upload, the
 * kernel and the copy into dst are queued; carr_phase_out is final on return (the host computes it), dst -- which
 * must be page-locked (gpsiq_host_alloc) -- is complete after gpsiq_wait().  Calls queue in order on the context's
 * stream, so a generator thread can prepare block k+1 (host model, fifo hand-off of block k-1) while block k is
 * synthesised and copied; at most four blocks are in flight, the fifth call waits for the first. */
int gpsiq_generate_block_async(gpsiq_ctx_t *ctx, const gpsiq_chan_t *ch, int nchan,
                               int nsamp, double fs, int sample_size,
                               void *dst_pinned, double *carr_phase_out);
/* Wait until everything queued by gpsiq_generate_block_async has landed. */
int gpsiq_wait(gpsiq_ctx_t *ctx);

/* Run-ahead form of the 10 Hz loop (gps.c:2703): ch is [nblocks][nchan], all blocks
 * prepared by the host model first (it never reads the loop's output except
 * carr_phase).  Block 0 seeds the carrier from ch[0][i].carr_phase under the same rule as
 * gpsiq_generate_block (the value handed out by the previous call continues exactly); later
 * blocks continue exactly, re-seeding a slot from its carr_phase only when its prn changes.
 * dst receives nblocks*2*nsamp elements; dst_is_device != 0 means dst is a device
 * pointer on the context's device (no D2H).  carr_phase_out[nchan] (may be NULL) receives
 * the carrier phase after the last block, to be put into the next batch's block 0.
 * ch may lie in pageable host memory, in page-locked host memory (gpsiq_host_alloc) or in device memory of the context's device:
 * a long batch (48 blocks or more in GPSIQ_NCO_REFERENCE; in the fixed-point model ~300 descriptors per host thread) is quantised -- and in
 * GPSIQ_NCO_REFERENCE chained and evaluated -- on the device, which reads page-locked and device-resident descriptors where
 * they lie (pageable ones are first cut down to 64 bytes each by the host pool); shorter batches, and every batch under
 * GPSIQ_EVAL=host, take the host quantiser / walker.  Either way a long batch is worked through in pieces, piece k+1 staged
 * under the kernel of piece k, and the call returns when everything has landed.  Descriptors are range-checked on the way: when a
 * descriptor is refused the call returns the error (the first refused block, in the host quantiser's words) after other
 * pieces have been rendered -- dst and the resident descriptor set are then undefined; the carried phases are untouched. */
int gpsiq_generate_batch(gpsiq_ctx_t *ctx, const gpsiq_chan_t *ch, int nblocks, int nchan,
                         int nsamp, double fs, int sample_size,
                         void *dst, int dst_is_device, double *carr_phase_out);

/* The batch call over several devices from ONE process (SURVEY.md 8b "batch extension"; the reference is
 * one process, gps-sim.c:314).  ctx[ndev]: one context per GPU (several contexts on one device work too).
 * The timeline is quantised once with the exact carrier prefix (or walked once in GPSIQ_NCO_REFERENCE,
 * taken from ctx[0]'s mode), cut into ndev contiguous block ranges (gpsiq_shard_range) and rendered by
 * one host thread per context; nothing passes between the devices.  Results land in timeline order:
 *   host_dst != NULL: one host buffer of nblocks*2*nsamp elements (page-locked: gpsiq_host_alloc);
 *   else dev_dst[i]:  device buffer on ctx[i]'s device for range i (its blocks packed, no padding).
 * Carrier continuation between calls follows gpsiq_generate_batch, with ctx[0] keeping the state. */
int gpsiq_generate_batch_multi(gpsiq_ctx_t *const *ctx, int ndev, const gpsiq_chan_t *ch, int nblocks, int nchan,
                               int nsamp, double fs, int sample_size, void *host_dst, void *const *dev_dst,
                               double *carr_phase_out);

/* One shard of a time-sharded run: synthesise nblocks already-quantised blocks
 * (a contiguous slice of gpsiq_quantize_batch's output, which carries the exact carrier
 * phase of every block) into dst, host or device as above.  Synchronous.  Does not touch
 * the carrier continuation state of gpsiq_generate_block/_batch. */
int gpsiq_generate_quantized(gpsiq_ctx_t *ctx, const gpsiq_qchan_t *q, int nblocks, int nchan,
                             int nsamp, int sample_size, void *dst, int dst_is_device);

/* Page-locked host memory for fifo buffers (hipHostMalloc): a device-to-host copy into it
 * is a single DMA.  NULL on failure.  Usable as the allocator of host/fifo.c. */
void *gpsiq_host_alloc(size_t bytes);
void  gpsiq_host_free(void *p);

/* ---- [sharding] resident-descriptor path (benchmarks, time-sharded multi-GPU) -------- */
/* (One shard of a time-sharded run in GPSIQ_NCO_REFERENCE is rendered from its blocks' start states by gpsiq_generate_seeded, which
 * goes with the carrier chain of csrc/gpsiq_plumbing.h and is declared there.) */
/* Copy nblocks*nchan quantised descriptors ([nblocks][nchan]) to the device. */
int gpsiq_set_descriptors(gpsiq_ctx_t *ctx, const gpsiq_qchan_t *q, int nblocks, int nchan);
/* Patches that go with the resident descriptors (gpsiq_reference_batch); every later gpsiq_launch
 * applies those of the blocks it synthesises, on the same stream, after the kernel.  n = 0 clears
 * them; gpsiq_set_descriptors clears them too.  A patch's slot counts its block's ACTIVE channels
 * (prn != 0) from 0, not the caller's channel index: slot >= that count is GPSIQ_E_RANGE. */
int gpsiq_set_patches(gpsiq_ctx_t *ctx, const gpsiq_patch_t *patches, int n);
/* Launch synthesis of blocks [block0, block0+nblocks) of the resident descriptors into
 * the DEVICE buffer dst; block b is written at dst + (b-block0)*block_stride_bytes
 * (block_stride_bytes >= 2*nsamp*sample_size, multiple of 4).  Asynchronous on
 * hip_stream (a hipStream_t passed as void*; NULL = the HIP null stream).
 * variant selects the kernel: 0 = default, see gpsiq_variant_name(). */
int gpsiq_launch(gpsiq_ctx_t *ctx, int block0, int nblocks, int nsamp, int sample_size,
                 void *dst, size_t block_stride_bytes, void *hip_stream, int variant);
/* (The "segm" variant and the patches of GPSIQ_NCO_REFERENCE use buffers owned by the context: launches of one
 * context that use them belong on one stream at a time.  The default kernels have no such state: launches of one
 * resident set may run on several streams at once, and gpsiq_set_descriptors waits for every one of them before it
 * reuses the descriptor buffer they read.) */
int gpsiq_synchronize(gpsiq_ctx_t *ctx, void *hip_stream);
/* ---- [boundary] hand-off to fifo.h buffers (gps.c:2847-2865) -------------------------- */
/* Element-exact restatement of the chunking rules, independent of the FIFO
 * implementation: the caller supplies acquire/enqueue callbacks with the fifo.h
 * semantics (fifo.h:45-55).  struct layout of the buffers is fifo.h:19-25. */
typedef struct gpsiq_iq_buf {       /* field-for-field struct iq_buf, fifo.h:19-25 */
    signed char  *data8;
    signed short *data16;
    unsigned int  totalLength;
    unsigned int  validLength;
    struct gpsiq_iq_buf *next;
} gpsiq_iq_buf_t;

typedef struct gpsiq_chunker {
    gpsiq_iq_buf_t *(*acquire)(void *user);           /* fifo_acquire  */
    void            (*enqueue)(void *user, gpsiq_iq_buf_t *buf); /* fifo_enqueue */
    void            *user;
    gpsiq_iq_buf_t  *cur;          /* buffer being filled (gps.c:2698) */
    int              sink_kind;    /* GPSIQ_SINK_* */
    int              sample_size;  /* GPSIQ_SC08 / GPSIQ_SC16 */
} gpsiq_chunker_t;

int gpsiq_chunker_init(gpsiq_chunker_t *ck, int sink_kind, int sample_size,
                       gpsiq_iq_buf_t *(*acquire)(void *), void (*enqueue)(void *, gpsiq_iq_buf_t *),
                       void *user);
/* Push one block of nelem = 2*nsamp already-packed elements (int8 or int16 per
 * sample_size) through the rules of gps.c:2839-2865.  Returns number of buffers
 * enqueued, or negative on error (acquire returned NULL = FIFO halted). */
int gpsiq_chunker_push(gpsiq_chunker_t *ck, const void *elems, size_t nelem);
/* In-place form for the sinks that take one block per buffer (iqfile, Pluto; gps.c:2860-2865):
 * gpsiq_chunker_reserve returns where the next nelem elements go inside the buffer being
 * filled (iq->data8 / iq->data16 at validLength), so that gpsiq_generate_block can write its
 * block there directly -- with page-locked fifo buffers the device-to-host copy lands in
 * place and the block is never copied on the host.  It returns NULL (no error recorded) when
 * the block cannot be contiguous in one buffer (HackRF's 262144-element chunks, or a buffer
 * that is too small): use gpsiq_chunker_push then.  gpsiq_chunker_commit hands over what was
 * written at the reserved position (validLength += nelem, enqueue, acquire the next buffer);
 * returns the number of buffers enqueued (1) or negative on error. */
void *gpsiq_chunker_reserve(gpsiq_chunker_t *ck, size_t nelem);
int   gpsiq_chunker_commit(gpsiq_chunker_t *ck, size_t nelem);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* GPSIQ_H */
