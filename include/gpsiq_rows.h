/*
 * gpsiq_rows.h -- the rows either side of the hot path, each bit-identical to the reference lines it
 * restates: the per-block host refresh batched (gps.c:2731-2765), the navigation message words (gps.c:617-884, 1008-1072,
 * 2066-2140) and the RINEX navigation readers (gps.c:1131-1891).  For run-ahead hosts that do not keep the reference's C host
 * model; a port of the reference that keeps it needs include/gpsiq.h only.  Same library (libgpsiq.so), host only.
 */
#ifndef GPSIQ_ROWS_H
#define GPSIQ_ROWS_H

#include "gpsiq.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* ---- [next rows] per-block host refresh, batched ----------- */
/* What the reference does on the host just before every pass of the sample loop
 * (gps.c:2731-2765: computeRange -> computeCodePhase -> gain), for many 0.1 s blocks at
 * once.  Plain double-precision C on the host, same operation order as the reference, so
 * with the same libm the descriptors are identical; blocks are independent (each range
 * depends only on time and position; the Doppler of block k is the range difference to
 * block k-1), so the batch is spread over host threads.  Nav words are inputs
 * (the 30 s nav-message refresh, gps.c:2878-2885, stays with the caller). */
typedef struct gpsiq_ephem {       /* the ephem_t fields satpos()/computeRange() read (gps.h:155-196) */
    double toe_sec, toc_sec;       /* toe.sec, toc.sec */
    double m0, n, ecc, sqrta, sq1e2, A, aop, omg0, omgkdot, inc0, idot;
    double cuc, cus, cic, cis, crc, crs;
    double af0, af1, af2, tgd;
} gpsiq_ephem_t;

typedef struct gpsiq_iono {        /* ionoutc_t fields ionosphericDelay() reads (gps.h:198-206) */
    int32_t enable, vflg;
    double  alpha[4], beta[4];
} gpsiq_iono_t;

typedef struct gpsiq_track {       /* per-channel host state that persists between blocks */
    int32_t  prn;                  /* 1..32, <= 0 unused */
    int32_t  g0_week;  double g0_sec;      /* chan.g0: start of the nav-word buffer (gps.c:2045) */
    int32_t  rho0_week; double rho0_sec;   /* chan.rho0.g  */
    double   rho0_range;                   /* chan.rho0.range: pseudorange of the previous block (gps.c:2039) */
    double   carr_phase;                   /* initial carrier phase (gps.c:2208-2214) */
    uint32_t dwrd[GPSIQ_N_DWRD];
} gpsiq_track_t;

/* Initialise trk[i].rho0 and carr_phase at receiver time (week, sec) and position xyz the
 * way allocateChannel() does (gps.c:2199-2214).  prn, g0 and dwrd must be filled by the caller. */
int gpsiq_track_init(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                     const double xyz[3], gpsiq_track_t *trk, int nchan);

/* checkSatVisibility() (gps.c:2142-2162): geometric azimuth / elevation (radians, no light-time
 * correction) of one satellite from ECEF position xyz at receiver time (week, sec), and the test
 * elevation > elv_mask_deg.  Returns 1 visible, 0 not visible, negative on error; azel may be NULL.
 * (allocateChannel() itself always passes a mask of 0 degrees, gps.c:2175.) */
int gpsiq_sat_visibility(const gpsiq_ephem_t *eph, int week, double sec, const double xyz[3],
                         double elv_mask_deg, double azel[2]);

/* Blocks k = 0..nblocks-1 at receiver times t_k = incGpsTime^(k+1)(week, sec) (the reference
 * advances grx by 0.1 s before the first block, gps.c:2692, and after every block, gps.c:2932)
 * and positions xyz[k] (ECEF metres).  out is [nblocks][nchan]; trk is updated to the state
 * after the last block.  gain_x2 != 0 applies the Pluto factor (gps.c:2759-2763).
 * nthreads <= 0: one per online CPU. */
int gpsiq_refresh_batch(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                        const double *xyz, int nblocks, int nchan, int gain_x2,
                        gpsiq_track_t *trk, gpsiq_chan_t *out, int nthreads);

/* The same over several navigation-message epochs in ONE threaded pass (a run-ahead host refreshes the word
 * buffers every 30 s, gps.c:2878-2885, but the ranges -- the expensive part -- do not depend on them): epoch e
 * covers blocks [first_block[e], first_block[e+1]) (first_block[0] = 0, the last epoch ends at nblocks) and takes
 * dwrd / g0 from trk_epochs[e][c]; prn, rho0 and carr_phase come from trk_epochs[0], whose rho0 is updated to the
 * state after the last block.  Every epoch must hold the same satellites (one allocation per call). */
int gpsiq_refresh_epochs(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                         const double *xyz, int nblocks, int nchan, int gain_x2,
                         gpsiq_track_t *trk_epochs /* [nepochs][nchan] */, const int *first_block /* [nepochs] */,
                         int nepochs, gpsiq_chan_t *out, int nthreads);

/* gpsiq_refresh_epochs followed by gpsiq_quantize_batch(carry_in = NULL) in one pass over the blocks: the same
 * out[nblocks][nchan] gpsiq_qchan_t those two calls give (block 0 of a slot seeded from trk_epochs[0][c].carr_phase,
 * later blocks chained with the exact carrier prefix), without the double-precision descriptors -- 296 bytes per
 * channel and block, mostly the nav-word buffer -- ever being written to memory.  For a run-ahead host that feeds
 * gpsiq_set_descriptors / gpsiq_generate_quantized. */
int gpsiq_refresh_epochs_quantized(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                                   const double *xyz, int nblocks, int nchan, int gain_x2,
                                   gpsiq_track_t *trk_epochs /* [nepochs][nchan] */, const int *first_block /* [nepochs] */,
                                   int nepochs, double fs, int nsamp, gpsiq_qchan_t *out, int nthreads);

/* ---- [next rows] navigation message words ------------------- */
/* The 60-word rolling buffer dwrd[] the sample loop reads its data bits from
 * (gps.c:2811) is built by the reference from the broadcast ephemeris: eph2sbf()
 * (gps.c:617-884) packs 3 + 2*25 subframe pages, generateNavMsg() (gps.c:2066-2140) inserts
 * week number and TOW count, chains the (32,26) parity of computeChecksum() (gps.c:1008-1072)
 * from word to word and rolls the buffer by one 30 s frame.  Bit-exact restatements: */
#define GPSIQ_N_SBF_PAGE 53   /* gps.h:55: subframes 1-3 + 25 pages of subframes 4 and 5 */
#define GPSIQ_N_DWRD_SBF 10

typedef struct gpsiq_nav_eph {    /* the ephem_t fields eph2sbf() packs (gps.h:155-196) */
    int32_t toe_week, iode, iodc, reserved;
    double  toe_sec, toc_sec;
    double  deltan, cuc, cus, cic, cis, crc, crs, ecc, sqrta, m0, omg0, inc0, aop, omgdot, idot;
    double  af0, af1, af2, tgd;
} gpsiq_nav_eph_t;

typedef struct gpsiq_nav_utc {    /* ionoutc_t (gps.h:198-206) */
    int32_t vflg, dtls, tot, wnt;
    double  alpha[4], beta[4], A0, A1;
} gpsiq_nav_utc_t;

typedef struct gpsiq_nav_alm_sv { /* almanac_prn_t fields eph2sbf() reads (almanac.h:21-41) */
    uint32_t svid, valid;
    int32_t  toa_week, reserved;
    double   toa_sec, e, delta_i, omegadot, sqrta, omega0, aop, m0, af0, af1;
} gpsiq_nav_alm_sv_t;

typedef struct gpsiq_nav_state {  /* per channel: chan.dwrd, chan.ipage, chan.g0 */
    uint32_t dwrd[GPSIQ_N_DWRD];
    int32_t  ipage, g0_week;
    double   g0_sec;
} gpsiq_nav_state_t;

/* computeChecksum(): source bits 31..30 = D29*,D30* of the previous word, bits 29..6 = d1..d24.
 * nib != 0 solves d23,d24 so that D29 = D30 = 0 (words 2 and 10). */
uint32_t gpsiq_nav_parity(uint32_t source, int nib);
/* eph2sbf().  alm = 32 entries or NULL (--disable-almanac: every page-25/almanac slot empty). */
int gpsiq_nav_subframes(const gpsiq_nav_eph_t *eph, const gpsiq_nav_utc_t *utc,
                        const gpsiq_nav_alm_sv_t *alm,
                        uint32_t sbf[GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF]);
/* generateNavMsg(g = (week, sec), chan, init).  init != 0 at channel allocation (gps.c:2196),
 * 0 at every 30 s refresh (gps.c:2880-2885).  st->ipage selects the subframe 4/5 page and is advanced. */
int gpsiq_nav_message(const uint32_t sbf[GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF], int week, double sec,
                      int init, gpsiq_nav_state_t *st);

/* The 30 s refresh of every channel in one call (gps.c:2880-2885: generateNavMsg(grx, &chan[i], 0) for all allocated
 * channels): sbf is [nchan][GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF], st[nchan]. */
int gpsiq_nav_roll(const uint32_t *sbf, int nchan, int week, double sec, gpsiq_nav_state_t *st);

/* ---- [next rows] RINEX navigation files ---------------------- */
/* readRinex2() (gps.c:1131-1505) / readRinex3() (gps.c:1512-1891): fixed-column parse of a
 * GPS broadcast-ephemeris file (plain or gzip), records grouped into sets whenever the time
 * of clock advances by more than an hour, at most GPSIQ_EPHEM_SETS sets of 32 satellites. */
#define GPSIQ_EPHEM_SETS 13   /* gps.h:108 EPHEM_ARRAY_SIZE */
#define GPSIQ_MAX_SAT    32   /* gps.h:33 */

typedef struct gpsiq_rinex_eph {  /* one ephem_t (gps.h:155-196), in the groupings the other entry points take */
    int32_t vflg, sva, svh, code, flag;       /* validity, URA index, health (MSB set as the reference does), L2 code, L2P flag */
    int32_t t_y, t_m, t_d, t_hh, t_mm;        /* calendar time of clock */
    double  t_sec, fit;
    int32_t toc_week, reserved;
    gpsiq_ephem_t   orbit;                    /* what gpsiq_refresh_batch() takes (incl. working variables A, n, sq1e2, omgkdot) */
    gpsiq_nav_eph_t nav;                      /* what gpsiq_nav_subframes() takes */
} gpsiq_rinex_eph_t;

/* version: 2 or 3.  eph is [GPSIQ_EPHEM_SETS][GPSIQ_MAX_SAT]; utc receives the header's
 * ionosphere/UTC parameters (vflg set when all four header records were present, gps.c:1257-1259).
 * Returns the number of ephemeris sets (0 .. GPSIQ_EPHEM_SETS; the reference reports 14 for a file with more
 * than 13 hourly groups although it stores 13 -- the library does not), or the reference's error codes: -1 cannot open,
 * -2 wrong RINEX version for this reader, -3 not a GPS navigation file. */
int gpsiq_rinex_read(const char *path, int version, gpsiq_rinex_eph_t *eph, gpsiq_nav_utc_t *utc);
/* The set gps_thread_ep() would use for a start time (gps.c:2588-2608): first set with a
 * satellite whose toc is within one hour of (week, sec); -1 if none. */
int gpsiq_rinex_select(const gpsiq_rinex_eph_t *eph, int nsets, int week, double sec);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
