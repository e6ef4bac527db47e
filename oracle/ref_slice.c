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
#include "gui.h"     /* status_color_t, gui_status_wprintw() (called by validate_parityN) */
#include "almanac.h" /* almanac_gps_t (eph2sbf argument) */

#include "../include/gpsiq_extras.h"   /* gpsiq_chan_t and the structs of the rows either side (gpsiq_rows.h): the layouts the tests pass in */

static int ref_fs = 3000000;
static int ref_nchan = 12;
#undef TX_SAMPLERATE
#define TX_SAMPLERATE ref_fs
#undef MAX_CHAN
#define MAX_CHAN ref_nchan

#include <zlib.h>
#include "ref_rinexdate.inc"         /* gps.c:138      rinex_date[] */
#include "ref_tables.inc"            /* gps.c:145-213  sinTable512, cosTable512 */
#include "ref_antpat.inc"            /* gps.c:216-221  ant_pat_db[] */
#include "ref_vec.inc"               /* gps.c:243-266  subVect, normVect, dotProd */
#include "ref_codegen.inc"           /* gps.c:272-309  codegen() */
#include "ref_date2gps.inc"          /* gps.c:315-337  date2gps() */
#include "ref_gps2date.inc"          /* gps.c:339-355  gps2date() */
#include "ref_geo.inc"               /* gps.c:361-499  xyz2llh, llh2xyz, ltcmat, ecef2neu, neu2azel */
#include "ref_satpos.inc"            /* gps.c:508-611  satpos() */
#include "ref_subgpstime.inc"        /* gps.c:1096-1103 subGpsTime() */
#include "ref_expd.inc"              /* gps.c:1079-1094 replaceExpDesignator() */
#include "ref_incgpstime.inc"        /* gps.c:1105-1124 incGpsTime() */
#include "ref_rinex2.inc"            /* gps.c:1131-1505 readRinex2() */
#include "ref_rinex3.inc"            /* gps.c:1512-1891 readRinex3() */
#include "ref_iono.inc"              /* gps.c:1893-1964 ionosphericDelay() */
#include "ref_range.inc"             /* gps.c:1972-2026 computeRange() */
#include "ref_computecodephase.inc"  /* gps.c:2033-2064 computeCodePhase() */
#include "ref_sbfid.inc"             /* gps.c:224-234  sbf4_svId[], sbf5_svId[] */
#include "ref_eph2sbf.inc"           /* gps.c:617-884  eph2sbf() */
#include "ref_parity.inc"            /* gps.c:890-1072 countBits, decode_wordN, validate_parityN, computeChecksum */
#include "ref_navmsg.inc"            /* gps.c:2066-2140 generateNavMsg() */
#include "ref_allocsat.inc"          /* gps.c:236       allocatedSat[] */
#include "ref_allocate.inc"          /* gps.c:2142-2235 checkSatVisibility(), allocateChannel() */
#include "ref_usermotion.inc"        /* gps.c:2253-2277 readUserMotion() */
#include "ref_almanac.inc"           /* almanac.c:19, 29-53, 73-184  almanac_init(), almanac_read_file() */

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

/* ---- host refresh: the reference's own lines around the loop ---- */
static void load_eph(ephem_t *e, const gpsiq_ephem_t *in)
{
    memset(e, 0, sizeof *e);
    e->vflg = 1;
    e->toe.sec = in->toe_sec; e->toc.sec = in->toc_sec;
    e->m0 = in->m0; e->n = in->n; e->ecc = in->ecc; e->sqrta = in->sqrta; e->sq1e2 = in->sq1e2;
    e->A = in->A; e->aop = in->aop; e->omg0 = in->omg0; e->omgkdot = in->omgkdot;
    e->inc0 = in->inc0; e->idot = in->idot;
    e->cuc = in->cuc; e->cus = in->cus; e->cic = in->cic; e->cis = in->cis; e->crc = in->crc; e->crs = in->crs;
    e->af0 = in->af0; e->af1 = in->af1; e->af2 = in->af2; e->tgd = in->tgd;
}

static void load_iono(ionoutc_t *io, const gpsiq_iono_t *in)
{
    memset(io, 0, sizeof *io);
    io->enable = in->enable; io->vflg = in->vflg;
    io->alpha0 = in->alpha[0]; io->alpha1 = in->alpha[1]; io->alpha2 = in->alpha[2]; io->alpha3 = in->alpha[3];
    io->beta0 = in->beta[0]; io->beta1 = in->beta[1]; io->beta2 = in->beta[2]; io->beta3 = in->beta[3];
}

/* satpos() + computeRange() for one satellite: pos[3], vel[3], clk[2], then
 * range, rate, d, az, el, iono_delay. */
int ref_compute_range(const gpsiq_ephem_t *eph_in, const gpsiq_iono_t *iono_in, int week, double sec,
                      const double *xyz, double *out14)
{
    ephem_t e; ionoutc_t io; gpstime_t g = { week, sec }; range_t rho;
    double p[3] = { xyz[0], xyz[1], xyz[2] };
    load_eph(&e, eph_in); load_iono(&io, iono_in);
    satpos(e, g, out14, out14 + 3, out14 + 6);
    computeRange(&rho, e, &io, g, p);
    out14[8] = rho.range; out14[9] = rho.rate; out14[10] = rho.d;
    out14[11] = rho.azel[0]; out14[12] = rho.azel[1]; out14[13] = rho.iono_delay;
    return 0;
}

int ref_inc_gps_time(int *week, double *sec, double dt)
{
    gpstime_t g = { *week, *sec };
    g = incGpsTime(g, dt);
    *week = g.week; *sec = g.sec;
    return 0;
}

/*
 * The host side of gps_thread_ep() from channel allocation to the state at gps.c:2766, for
 * nblocks blocks: rho0 / carr_phase initialised as allocateChannel() does (gps.c:2199-2214,
 * restated below with the reference's computeRange), antenna pattern gps.c:2688-2689 and
 * the per-block refresh gps.c:2731-2765 included verbatim, grx advanced by incGpsTime as
 * gps.c:2692 / gps.c:2932 do.  xyz is [nblocks+1][3]: xyz[0] the allocation position,
 * xyz[k+1] the position of block k.  out is [nblocks][nchan].
 */
static int refresh_epochs(const gpsiq_ephem_t *eph_in, const gpsiq_iono_t *iono_in, int week, double sec,
                          const double *xyz_in, int nblocks, int nchan, int sdr_type,
                          const gpsiq_track_t *trk_in, const uint32_t *sbf_in, const int *ipage_in,
                          gpsiq_chan_t *out, double *carr_init)
{
    if (nchan < 1 || nchan > GPSIQ_MAX_CHAN) return -1;
    ref_nchan = nchan;
    static simulator_t sim;
    simulator_t *simulator = &sim;
    static channel_t chan[GPSIQ_MAX_CHAN];
    static ephem_t eph[1][MAX_SAT];
    ionoutc_t ionoutc;
    double gain[GPSIQ_MAX_CHAN], ant_pat[37], path_loss, ant_gain;
    int i, sv, ibs, ieph = 0, iumd, igrx;
    gpstime_t grx = { week, sec };
    double (*xyz)[3] = (double (*)[3]) xyz_in;

    memset(&sim, 0, sizeof sim);
    sim.sdr_type = (sdr_type_t) sdr_type;
    memset(chan, 0, sizeof chan);
    memset(eph, 0, sizeof eph);
    load_iono(&ionoutc, iono_in);

    for (i = 0; i < nchan; i++) {                       /* allocateChannel(), gps.c:2183-2214 */
        if (trk_in[i].prn <= 0) continue;
        range_t rho;
        double ref[3] = { 0.0 }, r_ref, r_xyz, phase_ini;
        chan[i].prn = trk_in[i].prn;
        sv = chan[i].prn - 1;
        load_eph(&eph[0][sv], &eph_in[i]);
        codegen(chan[i].ca, chan[i].prn);
        for (int k = 0; k < N_DWRD; k++) chan[i].dwrd[k] = trk_in[i].dwrd[k];
        chan[i].g0.week = trk_in[i].g0_week; chan[i].g0.sec = trk_in[i].g0_sec;
        if (sbf_in) {                                               /* chan.sbf / chan.ipage as eph2sbf + generateNavMsg(init) left them */
            for (int p = 0; p < N_SBF_PAGE; p++)
                for (int w = 0; w < N_DWRD_SBF; w++)
                    chan[i].sbf[p][w] = sbf_in[((size_t) i * N_SBF_PAGE + p) * N_DWRD_SBF + w];
            chan[i].ipage = ipage_in[i];
        }
        computeRange(&rho, eph[0][sv], &ionoutc, grx, xyz[0]);      /* gps.c:2199 */
        chan[i].rho0 = rho;
        r_xyz = rho.range;
        computeRange(&rho, eph[0][sv], &ionoutc, grx, ref);         /* gps.c:2206 */
        r_ref = rho.range;
        phase_ini = (2.0 * r_ref - r_xyz) / LAMBDA_L1;              /* gps.c:2209 */
        chan[i].carr_phase = phase_ini - floor(phase_ini);          /* gps.c:2211 */
        if (carr_init) carr_init[i] = chan[i].carr_phase;
    }

#include "ref_antinit.inc"           /* gps.c:2688-2689 ant_pat[] */

    grx = incGpsTime(grx, 0.1);                                     /* gps.c:2692 */
    for (iumd = 1; iumd <= nblocks; iumd++) {
#include "ref_refresh.inc"           /* gps.c:2731-2765 */
        for (i = 0; i < nchan; i++) {
            gpsiq_chan_t *o = &out[(size_t) (iumd - 1) * nchan + i];
            memset(o, 0, sizeof *o);
            o->prn = chan[i].prn;
            if (chan[i].prn <= 0) continue;
            o->iword = chan[i].iword; o->ibit = chan[i].ibit; o->icode = chan[i].icode;
            o->f_carr = chan[i].f_carr; o->f_code = chan[i].f_code;
            o->carr_phase = chan[i].carr_phase; o->code_phase = chan[i].code_phase;
            o->gain = gain[i];
            for (int k = 0; k < N_DWRD; k++) o->dwrd[k] = (uint32_t) chan[i].dwrd[k];
        }
        if (sbf_in) {                /* the 30 s navigation-message refresh, gps.c:2870 and 2878-2885 */
#include "ref_igrx.inc"
#include "ref_navroll.inc"
            }                        /* closes the block opened at gps.c:2879 */
        }
        grx = incGpsTime(grx, 0.1);                                 /* gps.c:2932 */
    }
    (void) path_loss; (void) ant_gain; (void) ibs; (void) ieph; (void) igrx;
    return 0;
}

int ref_refresh_blocks(const gpsiq_ephem_t *eph_in, const gpsiq_iono_t *iono_in, int week, double sec,
                       const double *xyz_in, int nblocks, int nchan, int sdr_type,
                       const gpsiq_track_t *trk_in, gpsiq_chan_t *out, double *carr_init)
{
    return refresh_epochs(eph_in, iono_in, week, sec, xyz_in, nblocks, nchan, sdr_type, trk_in, NULL, NULL, out, carr_init);
}

/* The same loop with the reference's 30 s navigation-message refresh in it (gps.c:2870,
 * 2878-2885: generateNavMsg(grx, &chan[i], 0) whenever grx is a multiple of 30 s).
 * sbf is [nchan][53][10] (eph2sbf output per channel), ipage the page counter after
 * generateNavMsg(init). */
int ref_refresh_epochs(const gpsiq_ephem_t *eph_in, const gpsiq_iono_t *iono_in, int week, double sec,
                       const double *xyz_in, int nblocks, int nchan, int sdr_type,
                       const gpsiq_track_t *trk_in, const uint32_t *sbf_in, const int *ipage_in,
                       gpsiq_chan_t *out, double *carr_init)
{
    if (!sbf_in || !ipage_in) return -1;
    return refresh_epochs(eph_in, iono_in, week, sec, xyz_in, nblocks, nchan, sdr_type, trk_in, sbf_in, ipage_in, out, carr_init);
}

/* ---- navigation message --------------------------------------- */
/* validate_parityN() reports a word whose parity fails its two independent checkers
 * (gps.c:926-1001, 907-924) through the TUI; here the report is counted. */
static int parity_complaints;
void gui_status_wprintw(status_color_t clr, const char *fmt, ...) { (void) clr; (void) fmt; parity_complaints++; }
int ref_parity_complaints(void) { return parity_complaints; }

unsigned ref_nav_parity(unsigned source, int nib) { return (unsigned) computeChecksum(source, nib); }

static void load_nav_eph(ephem_t *e, const gpsiq_nav_eph_t *in)
{
    memset(e, 0, sizeof *e);
    e->vflg = 1;
    e->toe.week = in->toe_week; e->toe.sec = in->toe_sec; e->toc.week = in->toe_week; e->toc.sec = in->toc_sec;
    e->iode = in->iode; e->iodc = in->iodc;
    e->deltan = in->deltan; e->cuc = in->cuc; e->cus = in->cus; e->cic = in->cic; e->cis = in->cis;
    e->crc = in->crc; e->crs = in->crs; e->ecc = in->ecc; e->sqrta = in->sqrta; e->m0 = in->m0;
    e->omg0 = in->omg0; e->inc0 = in->inc0; e->aop = in->aop; e->omgdot = in->omgdot; e->idot = in->idot;
    e->af0 = in->af0; e->af1 = in->af1; e->af2 = in->af2; e->tgd = in->tgd;
}

int ref_nav_subframes(const gpsiq_nav_eph_t *eph_in, const gpsiq_nav_utc_t *utc, const gpsiq_nav_alm_sv_t *alm_in,
                      uint32_t *out /* [53][10] */)
{
    static ephem_t e; static ionoutc_t io; static almanac_gps_t alm;
    static unsigned long sbf[N_SBF_PAGE][N_DWRD_SBF];
    load_nav_eph(&e, eph_in);
    memset(&io, 0, sizeof io);
    io.enable = 1; io.vflg = utc->vflg;
    io.alpha0 = utc->alpha[0]; io.alpha1 = utc->alpha[1]; io.alpha2 = utc->alpha[2]; io.alpha3 = utc->alpha[3];
    io.beta0 = utc->beta[0]; io.beta1 = utc->beta[1]; io.beta2 = utc->beta[2]; io.beta3 = utc->beta[3];
    io.A0 = utc->A0; io.A1 = utc->A1; io.dtls = utc->dtls; io.tot = utc->tot; io.wnt = utc->wnt;
    memset(&alm, 0, sizeof alm);
    for (int sv = 0; alm_in && sv < MAX_SAT; sv++) {
        almanac_prn_t *a = &alm.sv[sv];
        a->svid = (unsigned short) alm_in[sv].svid; a->valid = alm_in[sv].valid;
        a->toa.week = alm_in[sv].toa_week; a->toa.sec = alm_in[sv].toa_sec;
        a->e = alm_in[sv].e; a->delta_i = alm_in[sv].delta_i; a->omegadot = alm_in[sv].omegadot;
        a->sqrta = alm_in[sv].sqrta; a->omega0 = alm_in[sv].omega0; a->aop = alm_in[sv].aop;
        a->m0 = alm_in[sv].m0; a->af0 = alm_in[sv].af0; a->af1 = alm_in[sv].af1;
    }
    memset(sbf, 0, sizeof sbf);
    eph2sbf(e, io, &alm, sbf);
    for (int p = 0; p < N_SBF_PAGE; p++)
        for (int w = 0; w < N_DWRD_SBF; w++) {
            if (sbf[p][w] >> 32) return -2;          /* every word must fit 32 bits */
            out[p * N_DWRD_SBF + w] = (uint32_t) sbf[p][w];
        }
    return 0;
}

int ref_nav_message(const uint32_t *sbf_in /* [53][10] */, int week, double sec, int init, gpsiq_nav_state_t *st)
{
    static channel_t ch;
    gpstime_t g = { week, sec };
    memset(&ch, 0, sizeof ch);
    for (int p = 0; p < N_SBF_PAGE; p++)
        for (int w = 0; w < N_DWRD_SBF; w++) ch.sbf[p][w] = sbf_in[p * N_DWRD_SBF + w];
    for (int k = 0; k < N_DWRD; k++) ch.dwrd[k] = st->dwrd[k];
    ch.ipage = st->ipage;
    generateNavMsg(g, &ch, init);
    for (int k = 0; k < N_DWRD; k++) {
        if (ch.dwrd[k] >> 32) return -2;
        st->dwrd[k] = (uint32_t) ch.dwrd[k];
    }
    st->ipage = ch.ipage; st->g0_week = ch.g0.week; st->g0_sec = ch.g0.sec;
    return 0;
}

/* ---- the whole host side of the block loop, allocation included ----------------------- */
static void load_full_eph(ephem_t *e, const gpsiq_rinex_eph_t *in)
{
    load_eph(e, &in->orbit);                    /* orbit + working variables */
    e->vflg = in->vflg; e->sva = in->sva; e->svh = in->svh; e->code = in->code; e->flag = in->flag; e->fit = in->fit;
    e->t.y = in->t_y; e->t.m = in->t_m; e->t.d = in->t_d; e->t.hh = in->t_hh; e->t.mm = in->t_mm; e->t.sec = in->t_sec;
    e->toc.week = in->toc_week; e->toc.sec = in->nav.toc_sec;
    e->toe.week = in->nav.toe_week; e->toe.sec = in->nav.toe_sec;
    e->iodc = in->nav.iodc; e->iode = in->nav.iode; e->deltan = in->nav.deltan; e->omgdot = in->nav.omgdot;
}

/* Channel allocation at the start (gps.c:2663-2675), then per block the refresh lines
 * (gps.c:2731-2765) and, at every multiple of 30 s, the navigation-message refresh and the
 * ephemeris-set switch and the re-allocation (gps.c:2870, 2878-2885, 2889-2906, 2909).
 * eph_sets is [nsets][32], ieph0 the set to start with; xyz is [nblocks+1][3] with xyz[0] the position
 * allocateChannel() always uses (gps.c:2675, 2909).  out is [nblocks][nchan]; nsat (may be
 * NULL) receives allocateChannel()'s return value of every call, at most max_nsat entries. */
int ref_run_host(const gpsiq_rinex_eph_t *eph_sets, int nsets, int ieph0, const gpsiq_nav_utc_t *utc, int week, double sec,
                 const double *xyz_in, int nblocks, int nchan, int sdr_type, gpsiq_chan_t *out,
                 int *nsat, int max_nsat, int *ieph_end)
{
    if (nchan < 1 || nchan > GPSIQ_MAX_CHAN || nsets < 1 || nsets > EPHEM_ARRAY_SIZE || ieph0 < 0 || ieph0 >= nsets) return -1;
    ref_nchan = nchan;
    static simulator_t sim;
    simulator_t *simulator = &sim;
    static channel_t chan[GPSIQ_MAX_CHAN];
    static ephem_t eph[EPHEM_ARRAY_SIZE + 1][MAX_SAT];     /* one spare all-invalid set: gps.c:2890 looks at eph[ieph + 1] */
    static almanac_gps_t alm_store;
    almanac_gps_t *alm = &alm_store;
    ionoutc_t ionoutc;
    double gain[GPSIQ_MAX_CHAN], ant_pat[37], path_loss, ant_gain, elvmask = 0.0, dt;
    int i, sv, ibs, ieph = ieph0, iumd, igrx, ncalls = 0;
    gpstime_t grx = { week, sec };
    double (*xyz)[3] = (double (*)[3]) xyz_in;

    memset(&sim, 0, sizeof sim);
    sim.sdr_type = (sdr_type_t) sdr_type;
    memset(chan, 0, sizeof chan);
    memset(&alm_store, 0, sizeof alm_store);
    memset(&ionoutc, 0, sizeof ionoutc);
    ionoutc.enable = 1; ionoutc.vflg = utc->vflg;
    ionoutc.alpha0 = utc->alpha[0]; ionoutc.alpha1 = utc->alpha[1]; ionoutc.alpha2 = utc->alpha[2]; ionoutc.alpha3 = utc->alpha[3];
    ionoutc.beta0 = utc->beta[0]; ionoutc.beta1 = utc->beta[1]; ionoutc.beta2 = utc->beta[2]; ionoutc.beta3 = utc->beta[3];
    ionoutc.A0 = utc->A0; ionoutc.A1 = utc->A1; ionoutc.dtls = utc->dtls; ionoutc.tot = utc->tot; ionoutc.wnt = utc->wnt;
    memset(eph, 0, sizeof eph);
    for (int k = 0; k < nsets; k++)
        for (sv = 0; sv < MAX_SAT; sv++)
            if (eph_sets[k * MAX_SAT + sv].vflg) load_full_eph(&eph[k][sv], &eph_sets[k * MAX_SAT + sv]);

    for (i = 0; i < MAX_CHAN; i++) chan[i].prn = 0;                 /* gps.c:2664-2665 */
    for (sv = 0; sv < MAX_SAT; sv++) allocatedSat[sv] = -1;         /* gps.c:2668-2669 */
    i = allocateChannel(chan, alm, eph[ieph], ionoutc, grx, xyz[0], elvmask);   /* gps.c:2675 */
    if (nsat && ncalls < max_nsat) nsat[ncalls] = i;
    ncalls++;

#include "ref_antinit.inc"           /* gps.c:2688-2689 ant_pat[] */

    grx = incGpsTime(grx, 0.1);                                     /* gps.c:2692 */
    for (iumd = 1; iumd <= nblocks; iumd++) {
#include "ref_refresh.inc"           /* gps.c:2731-2765 */
        for (i = 0; i < nchan; i++) {
            gpsiq_chan_t *o = &out[(size_t) (iumd - 1) * nchan + i];
            memset(o, 0, sizeof *o);
            o->prn = chan[i].prn;
            if (chan[i].prn <= 0) continue;
            o->iword = chan[i].iword; o->ibit = chan[i].ibit; o->icode = chan[i].icode;
            o->f_carr = chan[i].f_carr; o->f_code = chan[i].f_code;
            o->carr_phase = chan[i].carr_phase; o->code_phase = chan[i].code_phase;
            o->gain = gain[i];
            for (int k = 0; k < N_DWRD; k++) o->dwrd[k] = (uint32_t) chan[i].dwrd[k];
        }
#include "ref_igrx.inc"              /* gps.c:2870 */
#include "ref_navroll.inc"           /* gps.c:2878-2885 */
#include "ref_ephrefresh.inc"        /* gps.c:2889-2906 switch to the next ephemeris set, new subframes */
            i = allocateChannel(chan, alm, eph[ieph], ionoutc, grx, xyz[0], elvmask);   /* gps.c:2909 */
            if (nsat && ncalls < max_nsat) nsat[ncalls] = i;
            ncalls++;
        }                            /* closes the block opened at gps.c:2879 */
        grx = incGpsTime(grx, 0.1);                                 /* gps.c:2932 */
    }
    (void) path_loss; (void) ant_gain; (void) ibs;
    if (ieph_end) *ieph_end = ieph;
    return ncalls;
}

/* ---- RINEX readers --------------------------------------------- */
int ref_read_rinex(int version, const char *path, gpsiq_rinex_eph_t *out /* [13][32] */, gpsiq_nav_utc_t *utc,
                   char *date21)
{
    static ephem_t eph[EPHEM_ARRAY_SIZE][MAX_SAT];
    ionoutc_t io;
    memset(eph, 0, sizeof eph);
    memset(&io, 0, sizeof io);
    ref_nchan = MAX_SAT;   /* MAX_CHAN is not used by the readers; keep the macro harmless */
    int n = version == 3 ? readRinex3(eph, &io, path) : readRinex2(eph, &io, path);
    memset(out, 0, sizeof *out * EPHEM_ARRAY_SIZE * 32);
    for (int i = 0; i < EPHEM_ARRAY_SIZE; i++)
        for (int sv = 0; sv < 32; sv++) {
            const ephem_t *e = &eph[i][sv];
            gpsiq_rinex_eph_t *o = &out[i * 32 + sv];
            if (!e->vflg) continue;
            o->vflg = 1; o->sva = e->sva; o->svh = e->svh; o->code = e->code; o->flag = e->flag;
            o->t_y = e->t.y; o->t_m = e->t.m; o->t_d = e->t.d; o->t_hh = e->t.hh; o->t_mm = e->t.mm; o->t_sec = e->t.sec;
            o->fit = e->fit; o->toc_week = e->toc.week;
            gpsiq_ephem_t *b = &o->orbit; gpsiq_nav_eph_t *v = &o->nav;
            b->toe_sec = v->toe_sec = e->toe.sec; b->toc_sec = v->toc_sec = e->toc.sec;
            v->toe_week = e->toe.week; v->iode = e->iode; v->iodc = e->iodc;
            b->m0 = v->m0 = e->m0; b->n = e->n; b->ecc = v->ecc = e->ecc; b->sqrta = v->sqrta = e->sqrta;
            b->sq1e2 = e->sq1e2; b->A = e->A; b->aop = v->aop = e->aop; b->omg0 = v->omg0 = e->omg0;
            b->omgkdot = e->omgkdot; v->omgdot = e->omgdot; b->inc0 = v->inc0 = e->inc0; b->idot = v->idot = e->idot;
            b->cuc = v->cuc = e->cuc; b->cus = v->cus = e->cus; b->cic = v->cic = e->cic; b->cis = v->cis = e->cis;
            b->crc = v->crc = e->crc; b->crs = v->crs = e->crs; v->deltan = e->deltan;
            b->af0 = v->af0 = e->af0; b->af1 = v->af1 = e->af1; b->af2 = v->af2 = e->af2; b->tgd = v->tgd = e->tgd;
        }
    memset(utc, 0, sizeof *utc);
    utc->vflg = io.vflg; utc->dtls = io.dtls; utc->tot = io.tot; utc->wnt = io.wnt;
    utc->alpha[0] = io.alpha0; utc->alpha[1] = io.alpha1; utc->alpha[2] = io.alpha2; utc->alpha[3] = io.alpha3;
    utc->beta[0] = io.beta0; utc->beta[1] = io.beta1; utc->beta[2] = io.beta2; utc->beta[3] = io.beta3;
    utc->A0 = io.A0; utc->A1 = io.A1;
    if (date21) memcpy(date21, rinex_date, 21);
    return n;
}

/* ---- the -T start-time overwrite (gps.c:2507-2513 gmin, gps.c:2534-2561 the shift of toc / toe / t and of the UTC reference) ---- */
int ref_time_overwrite(gpsiq_rinex_eph_t *sets /* [13][32] */, int neph, gpsiq_nav_utc_t *utc, int week, double sec)
{
    static ephem_t eph[EPHEM_ARRAY_SIZE][MAX_SAT];
    ionoutc_t ionoutc;
    gpstime_t g0, gmin, gtmp;
    datetime_t tmin, ttmp;
    int sv, i;
    memset(eph, 0, sizeof eph);
    memset(&ionoutc, 0, sizeof ionoutc);
    memset(&gmin, 0, sizeof gmin); memset(&tmin, 0, sizeof tmin);
    for (i = 0; i < neph; i++)
        for (sv = 0; sv < MAX_SAT; sv++)
            if (sets[i * 32 + sv].vflg) load_full_eph(&eph[i][sv], &sets[i * 32 + sv]);
    ionoutc.wnt = utc->wnt; ionoutc.tot = utc->tot;
    g0.week = week; g0.sec = sec;
#include "ref_gmin.inc"              /* gps.c:2507-2513 */
    {
#include "ref_overwrite.inc"         /* gps.c:2534-2561 */
    }
    for (i = 0; i < neph; i++)
        for (sv = 0; sv < MAX_SAT; sv++) {
            gpsiq_rinex_eph_t *o = &sets[i * 32 + sv];
            const ephem_t *e = &eph[i][sv];
            if (!o->vflg) continue;
            o->t_y = e->t.y; o->t_m = e->t.m; o->t_d = e->t.d; o->t_hh = e->t.hh; o->t_mm = e->t.mm; o->t_sec = e->t.sec;
            o->toc_week = e->toc.week; o->orbit.toc_sec = o->nav.toc_sec = e->toc.sec;
            o->nav.toe_week = e->toe.week; o->orbit.toe_sec = o->nav.toe_sec = e->toe.sec;
        }
    utc->wnt = ionoutc.wnt; utc->tot = ionoutc.tot;
    (void) tmin;
    return 0;
}

/* ---- where the receiver is: the reference's geodetic conversions and its user-motion reader ---- */
/* almanac_read_file() reads "almanac.sem" in the current directory; returns its CURLcode, the records in out[32] */
int ref_almanac_read(gpsiq_nav_alm_sv_t *out)
{
    int rc = (int) almanac_read_file();
    for (int sv = 0; sv < MAX_SAT; sv++) {
        const almanac_prn_t *a = &almanac_gps.sv[sv];
        memset(&out[sv], 0, sizeof out[sv]);
        out[sv].svid = a->svid; out[sv].valid = a->valid;
        out[sv].toa_week = a->toa.week; out[sv].toa_sec = a->toa.sec;
        out[sv].e = a->e; out[sv].delta_i = a->delta_i; out[sv].omegadot = a->omegadot;
        out[sv].sqrta = a->sqrta; out[sv].omega0 = a->omega0; out[sv].aop = a->aop;
        out[sv].m0 = a->m0; out[sv].af0 = a->af0; out[sv].af1 = a->af1;
    }
    return rc;
}
void ref_date2gps(int y, int m, int d, int hh, int mm, double sec, int *week, double *sow)
{
    datetime_t t = {y, m, d, hh, mm, sec};
    gpstime_t g;
    date2gps(&t, &g);
    *week = g.week; *sow = g.sec;
}
void ref_gps2date(int week, double sow, int *y, int *m, int *d, int *hh, int *mm, double *sec)
{
    gpstime_t g = {week, sow};
    datetime_t t;
    gps2date(&g, &t);
    *y = t.y; *m = t.m; *d = t.d; *hh = t.hh; *mm = t.mm; *sec = t.sec;
}
void ref_llh2xyz(const double *llh, double *xyz) { llh2xyz(llh, xyz); }
/* the target offset / interactive step: ltcmat of the start location, then the three lines gps.c:2354-2356 (== 2726-2728) */
void ref_add_neu(const double *llh, const double *neu_in, double *xyz_io)
{
    double tmat[3][3], neu[3] = {neu_in[0], neu_in[1], neu_in[2]};
    double xyz[1][3] = {{xyz_io[0], xyz_io[1], xyz_io[2]}};
    ltcmat(llh, tmat);
#include "ref_addneu.inc"            /* gps.c:2354-2356 */
    xyz_io[0] = xyz[0][0]; xyz_io[1] = xyz[0][1]; xyz_io[2] = xyz[0][2];
}
void ref_xyz2llh(const double *xyz, double *llh) { xyz2llh(xyz, llh); }
int ref_read_user_motion(const char *path, double *xyz_out, int max_points)
{
    static double xyz[USER_MOTION_SIZE][3];
    int n = readUserMotion(xyz, path);
    for (int k = 0; k < n && k < max_points; k++) { xyz_out[3 * k] = xyz[k][0]; xyz_out[3 * k + 1] = xyz[k][1]; xyz_out[3 * k + 2] = xyz[k][2]; }
    return n;
}
