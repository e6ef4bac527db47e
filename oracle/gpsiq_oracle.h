/*
 * gpsiq_oracle.h — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 * The product (libgpsiq) never links, loads or calls anything in oracle/.
 *
 * Two restatements of reference gps.c:2767-2846 (Mictronics/multi-sdr-gps-sim):
 *   oracle_block_float  — the loop as the reference runs it: sequential double NCOs
 *                         (gps.h:17 FLOAT_CARR_PHASE), same operation order.
 *   oracle_block_fixed  — the closed-form integer NCO model of include/gpsiq.h that
 *                         the HIP kernels implement (the parity gate for the GPU).
 * Parity pin: both are checked against oracle/_ref (the reference's own source lines
 * compiled where they lie) and against tests/golden/ captured from it.
 */
#ifndef GPSIQ_ORACLE_H
#define GPSIQ_ORACLE_H

#include "../include/gpsiq.h"

#ifdef __cplusplus
extern "C" {
#endif

/* gps.c:145-213 */
int  oracle_sin512(int k);
int  oracle_cos512(int k);
/* gps.c:272-309 */
int  oracle_codegen(int prn, uint8_t ca[GPSIQ_CA_SEQ_LEN]);

/* gps.c:2767-2846, float form.  ch is copied (the caller's array is not mutated);
 * carr_phase_out[i] / code state after the block are returned through the optional
 * out arrays (carr_phase_out may be NULL).  dst: 2*nsamp int8 or int16 elements. */
int  oracle_block_float(const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                        int sample_size, void *dst, double *carr_phase_out);

/* The same loop with no per-sample recurrence: both double accumulators are piecewise linear in
 * exact integers (one piece per binade the phase passes through), built by real double additions
 * at the piece edges and integer jumps in between; every sample is then evaluated by piece
 * lookup.  Must equal oracle_block_float / the reference bit for bit, carr_phase_out included. */
/* the carrier accumulator alone, gps.c:2821-2826 nsamp times */
double oracle_carrier_chain(double carr_phase, double carr_inc, long nsamp);
int  oracle_block_float_closed(const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                               int sample_size, void *dst, double *carr_phase_out);

/* include/gpsiq.h quantisation rules */
int  oracle_quantize(const gpsiq_chan_t *ch, int nchan, double fs, int nsamp,
                     gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out);

/* include/gpsiq.h closed form, evaluated independently per sample with 128-bit ints */
int  oracle_block_fixed(const gpsiq_qchan_t *q, int nchan, int nsamp,
                        int sample_size, void *dst);
/* the same samples, but only the range [n0, n0+cnt) of the block (for spot checks at
 * BASELINE sizes): dst receives 2*cnt elements */
int  oracle_block_fixed_range(const gpsiq_qchan_t *q, int nchan, long n0, long cnt,
                              int sample_size, void *dst);
/* same result as oracle_block_fixed, NCOs advanced incrementally (timed CPU port) */
int  oracle_block_fixed_seq(const gpsiq_qchan_t *q, int nchan, int nsamp,
                            int sample_size, void *dst);

/* gps.c:2839-2865 chunking rules, restated on flat arrays: given nblocks blocks of
 * nelem elements pushed in order, returns in chunk_len[] the validLength of every
 * buffer enqueued (HackRF: 262144 each, partial kept; iqfile/Pluto: nelem each). */
int  oracle_chunk_plan(int sink_kind, size_t nelem, int nblocks,
                       size_t *chunk_len, int max_chunks);

#ifdef __cplusplus
}
#endif
#endif
