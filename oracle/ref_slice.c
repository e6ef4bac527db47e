/*
 * ref_slice.c — harness around the reference's OWN hot-path source lines.
 * TEST INFRASTRUCTURE ONLY; builds oracle/_ref/libgpsref.so (see oracle/Makefile).
 *
 * Why a slice: the hot path of Mictronics/multi-sdr-gps-sim is not a function, it is
 * inlined in gps_thread_ep() (gps.c:2282), and gps_thread_ep() cannot be driven
 * without the RINEX/almanac/curl/ncurses setup around it.  So the Makefile cuts the
 * line ranges of the path out of /root/reference/gps.c where it lies — tables
 * gps.c:145-213, codegen gps.c:272-309, subGpsTime gps.c:1096-1103, computeCodePhase
 * gps.c:2033-2064, the sample loop + pack + fifo hand-off gps.c:2767-2865 — into
 * temporary .inc files that are #included below, compiles, and deletes them.  No
 * reference text is kept in this repository; the reference's own headers (gps.h,
 * gps-sim.h, sdr.h, fifo.h) are included from /root/reference directly.  The only
 * things supplied here are the locals of gps_thread_ep() the loop reads and a
 * capturing implementation of the two fifo.h calls it makes.
 *
 * The two compile-time constants the BASELINE configs change (SURVEY.md section 0
 * fact 5) are made run-time: TX_SAMPLERATE (sdr.h:21) and MAX_CHAN (gps.h:36).
 */
#define _GNU_SOURCE
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdbool.h>

#include "sdr.h"     /* -> gps-sim.h -> gps.h ; NUM_IQ_SAMPLES, IQ_BUFFER_SIZE, HACKRF_TRANSFER_BUFFER_SIZE */
#include "fifo.h"    /* struct iq_buf, fifo_acquire, fifo_enqueue */

#include "../include/gpsiq.h"   /* gpsiq_chan_t: the descriptor layout the tests pass in */

static int ref_fs = 3000000;
static int ref_nchan = 12;
#undef TX_SAMPLERATE
#define TX_SAMPLERATE ref_fs
#undef MAX_CHAN
#define MAX_CHAN ref_nchan

#include "ref_tables.inc"            /* gps.c:145-213  sinTable512, cosTable512 */
#include "ref_codegen.inc"           /* gps.c:272-309  codegen() */
#include "ref_subgpstime.inc"        /* gps.c:1096-1103 subGpsTime() */
#include "ref_computecodephase.inc"  /* gps.c:2033-2064 computeCodePhase() */

/* ---- capturing fifo (the tap SURVEY.md section 0 fact 6 asks for) --------- */
static struct {
    struct iq_buf buf[2];
    int           cur;
    unsigned      buf_len;
    int           sample_size;
    unsigned char *out;         /* captured elements, in enqueue order */
    size_t        out_cap, out_len;      /* in elements */
    size_t       *chunk_len;
    int           max_chunks, nchunks;
    int           overflow;
} cap;

struct iq_buf *fifo_acquire(void)
{
    struct iq_buf *b = &cap.buf[cap.cur];
    cap.cur ^= 1;
    b->validLength = 0;       /* fifo.c:143 */
    b->next = NULL;
    return b;
}

void fifo_enqueue(struct iq_buf *b)
{
    size_t esz = (size_t) cap.sample_size;
    if (cap.out_len + b->validLength > cap.out_cap || cap.nchunks >= cap.max_chunks) {
        cap.overflow = 1;
        return;
    }
    const void *src = cap.sample_size == (int) SC16 ? (const void *) b->data16 : (const void *) b->data8;
    memcpy(cap.out + cap.out_len * esz, src, (size_t) b->validLength * esz);
    cap.out_len += b->validLength;
    cap.chunk_len[cap.nchunks++] = b->validLength;
}

/* ---- exports --------------------------------------------------------------- */
int ref_tables(int *sin512, int *cos512)
{
    if (sizeof sinTable512 != 512 * sizeof(int) || sizeof cosTable512 != 512 * sizeof(int))
        return -1;
    memcpy(sin512, sinTable512, sizeof sinTable512);
    memcpy(cos512, cosTable512, sizeof cosTable512);
    return 0;
}

int ref_codegen(int prn, int *ca)
{
    codegen(ca, prn);
    return 0;
}

/* computeCodePhase(): rho0/rho1 ranges and times in, the per-block state out. */
int ref_compute_code_phase(double rho0_range, int rho0_week, double rho0_sec,
                           int g0_week, double g0_sec, double rho1_range, double dt,
                           const uint32_t *dwrd, int prn, gpsiq_chan_t *out)
{
    static channel_t ch;
    range_t rho1;
    memset(&ch, 0, sizeof ch);
    memset(&rho1, 0, sizeof rho1);
    ch.prn = prn;
    codegen(ch.ca, prn);
    for (int k = 0; k < N_DWRD; k++) ch.dwrd[k] = dwrd[k];
    ch.rho0.range = rho0_range; ch.rho0.g.week = rho0_week; ch.rho0.g.sec = rho0_sec;
    ch.g0.week = g0_week; ch.g0.sec = g0_sec;
    rho1.range = rho1_range;
    computeCodePhase(&ch, rho1, dt);
    out->prn = prn; out->iword = ch.iword; out->ibit = ch.ibit; out->icode = ch.icode;
    out->f_carr = ch.f_carr; out->f_code = ch.f_code; out->code_phase = ch.code_phase;
    return (ch.dataBit == (int) ((dwrd[ch.iword] >> (29 - ch.ibit)) & 1u) * 2 - 1 &&
            ch.codeCA == ch.ca[(int) ch.code_phase] * 2 - 1) ? 0 : -2;
}

/*
 * Run nblocks consecutive passes of gps.c:2767-2865 on descriptors desc[nblocks][nchan].
 * Per block the code-side state and gain are re-seeded from the descriptor exactly as
 * the host refresh leaves them at gps.c:2766 (computeCodePhase gps.c:2047-2059);
 * carr_phase is taken from the descriptor in block 0 (and whenever a slot's prn
 * changes, as allocateChannel does, gps.c:2214) and otherwise carried by the loop
 * itself (gps.c:2821).  carr_out[nblocks][nchan] receives chan[i].carr_phase after
 * each block.  Captured elements go to out in fifo_enqueue order.
 */
int ref_run_blocks(const gpsiq_chan_t *desc, int nblocks, int nchan, int fs,
                   int sample_size, int sdr_type,
                   void *out, size_t out_cap_elems, size_t *out_elems,
                   size_t *chunk_len, int max_chunks, int *nchunks, double *carr_out)
{
    if (nchan < 1 || nchan > GPSIQ_MAX_CHAN || fs < 10 || nblocks < 0)
        return -1;
    ref_fs = fs;
    ref_nchan = nchan;

    /* the locals of gps_thread_ep() the included lines use (gps.c:2283-2320) */
    static simulator_t sim;
    simulator_t *simulator = &sim;
    static channel_t chan[GPSIQ_MAX_CHAN];
    double gain[GPSIQ_MAX_CHAN];
    const double delt = 1.0 / (double) TX_SAMPLERATE;   /* gps.c:2298 */
    int i, ip, qp, iTable, isamp;
    short *iq_buff = calloc(IQ_BUFFER_SIZE, 2);          /* gps.c:2695 */
    int rc = 0;

    memset(&sim, 0, sizeof sim);
    sim.sample_size = sample_size;
    sim.sdr_type = (sdr_type_t) sdr_type;
    memset(chan, 0, sizeof chan);

    memset(&cap, 0, sizeof cap);
    cap.sample_size = sample_size;
    cap.buf_len = sdr_type == SDR_HACKRF ? HACKRF_TRANSFER_BUFFER_SIZE : (unsigned) IQ_BUFFER_SIZE;
    for (int b = 0; b < 2; b++) {          /* fifo_create(), fifo.c:33-65 */
        if (sample_size == (int) SC16) cap.buf[b].data16 = calloc(cap.buf_len, 2);
        else                           cap.buf[b].data8 = calloc(cap.buf_len, 1);
        cap.buf[b].totalLength = cap.buf_len;
    }
    cap.out = out; cap.out_cap = out_cap_elems;
    cap.chunk_len = chunk_len; cap.max_chunks = max_chunks;

    struct iq_buf *iq = fifo_acquire();                   /* gps.c:2698 */

    for (int blk = 0; blk < nblocks && rc == 0; blk++) {
        const gpsiq_chan_t *d = desc + (size_t) blk * nchan;
        for (i = 0; i < nchan; i++) {
            if (d[i].prn <= 0) { chan[i].prn = 0; continue; }
            if (chan[i].prn != d[i].prn) {                 /* (re)allocation: gps.c:2193-2214 */
                chan[i].prn = d[i].prn;
                codegen(chan[i].ca, chan[i].prn);
                chan[i].carr_phase = d[i].carr_phase;
            }
            for (int k = 0; k < N_DWRD; k++) chan[i].dwrd[k] = d[i].dwrd[k];
            chan[i].f_carr = d[i].f_carr;                 /* gps.c:2042 */
            chan[i].f_code = d[i].f_code;                 /* gps.c:2043 */
            chan[i].code_phase = d[i].code_phase;         /* gps.c:2047 */
            chan[i].iword = d[i].iword; chan[i].ibit = d[i].ibit; chan[i].icode = d[i].icode;
            chan[i].codeCA = chan[i].ca[(int) chan[i].code_phase]*2 - 1;                                   /* gps.c:2058 */
            chan[i].dataBit = (int) ((chan[i].dwrd[chan[i].iword]>>(29 - chan[i].ibit)) & 0x1UL)*2 - 1;   /* gps.c:2059 */
            gain[i] = d[i].gain;                          /* gps.c:2756 */
        }

#include "ref_loop.inc"              /* gps.c:2767-2865 */

        if (carr_out)
            for (i = 0; i < nchan; i++)
                carr_out[(size_t) blk * nchan + i] = chan[i].prn > 0 ? chan[i].carr_phase : 0.0;
        if (cap.overflow) rc = -3;
    }
    (void) delt; (void) ip; (void) qp; (void) iTable; (void) isamp; (void) iq;

    *out_elems = cap.out_len;
    *nchunks = cap.nchunks;
    for (int b = 0; b < 2; b++) { free(cap.buf[b].data16); free(cap.buf[b].data8); }
    free(iq_buff);
    return rc;
}
