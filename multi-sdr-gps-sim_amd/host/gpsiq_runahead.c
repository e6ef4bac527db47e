/*
 * gpsiq_runahead.c — a whole scenario through the C-ABI, host code in C.
 *
 * What gps_thread_ep() does between reading the RINEX file and handing IQ blocks to the sink
 * (reference gps.c:2472-2933), with every step a libgpsiq entry point and the 10 Hz loop run
 * ahead of time, 30 s epoch by epoch:
 *
 *   gpsiq_rinex_read / gpsiq_rinex_select        readRinex2/3, ephemeris set for the start time
 *   allocate()  = the policy of allocateChannel() (gps.c:2164-2235) on gpsiq_sat_visibility,
 *                 gpsiq_nav_subframes, gpsiq_nav_message(init), gpsiq_track_init
 *   gpsiq_refresh_batch                          computeRange/computeCodePhase/gain, gps.c:2731-2765
 *   gpsiq_generate_batch                         the sample loop + pack, gps.c:2767-2846
 *   gpsiq_nav_message(roll), next ephemeris set,
 *   allocate()                                   the 30 s refresh, gps.c:2870-2909
 *
 *   gpsiq_runahead <rinex> <2|3> <week> <sec> <position> <nblocks> <nchan> <fs> <1|2> <out.bin>
 *
 * start time: GPS <week> <sec>, or the reference's -t form "YYYY/MM/DD,hh:mm:ss" as <week> with "-" as <sec>
 * (gps-sim.c:104; date2gps, gps.c:315-337, 2295: no leap seconds applied, as there).
 * position, one of
 *   xyz.bin        double[nblocks+1][3] ECEF metres, row 0 = start position (the one allocateChannel()
 *                  always uses, gps.c:2675, 2909), row k+1 = position of block k
 *   motion.csv     the reference's user-motion file (-m, gps.c:2253-2277, 2495-2505): "t,x,y,z" per 0.1 s,
 *                  same row meaning; a file shorter than nblocks+1 points shortens the run as the
 *                  reference's numd does
 *   lat,lon,h      a static receiver, degrees and metres (-l, gps.c:2337-2340, 2359-2363)
 * out.bin: the iqfile stream.  An optional eleventh argument names a SEM almanac file (the reference's almanac.sem,
 * almanac.c:73-184): its entries fill the almanac pages of subframes 4 and 5; without it those pages are empty, as with
 * the reference's --disable-almanac.
 */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsiq_extras.h"

#define BLOCKS_PER_CALL 100    /* bounds the page-locked staging buffer */

static int die(const char *what)
{
    fprintf(stderr, "gpsiq_runahead: %s: %s\n", what, gpsiq_last_error());
    return 1;
}

/* incGpsTime(.., 0.1) applied `steps` times to a whole-millisecond time (gps.c:1105-1124) */
static double time_after(double sec, long steps) { return round(round(sec * 1000.0) + 100.0 * (double) steps) / 1000.0; }

struct host_state {
    const gpsiq_nav_alm_sv_t *alm;                   /* 32 almanac entries, or NULL (--disable-almanac) */
    int nchan, week;
    double xyz0[3];
    const gpsiq_rinex_eph_t *eph;         /* the set in use: 32 satellites */
    const gpsiq_rinex_eph_t *sets;        /* all sets [nsets][32] */
    int nsets, ieph;
    gpsiq_nav_utc_t utc;
    gpsiq_iono_t iono;
    int allocated_sat[GPSIQ_MAX_SAT];
    gpsiq_ephem_t orbit[GPSIQ_MAX_CHAN];
    gpsiq_track_t trk[GPSIQ_MAX_CHAN];
    gpsiq_nav_state_t nav[GPSIQ_MAX_CHAN];
    uint32_t sbf[GPSIQ_MAX_CHAN][GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF];
};

/* allocateChannel(), gps.c:2164-2235: visible satellites, lowest PRN first, into the first free
 * channel; satellites that have set release theirs.  Returns the number of visible satellites. */
static int allocate(struct host_state *h, double t)
{
    int nsat = 0;
    for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) {
        const gpsiq_rinex_eph_t *e = &h->eph[sv];
        int vis = e->vflg ? gpsiq_sat_visibility(&e->orbit, h->week, t, h->xyz0, 0.0, NULL) : -1;
        if (vis == 1) {
            ++nsat;
            if (h->allocated_sat[sv] != -1) continue;
            for (int i = 0; i < h->nchan; ++i) {
                if (h->trk[i].prn != 0) continue;
                memset(&h->trk[i], 0, sizeof h->trk[i]);
                {   /* chan[i].ipage survives release and re-allocation of a slot: only generateNavMsg touches it (gps.c:2137-2139) */
                    const int32_t ipage = h->nav[i].ipage;
                    memset(&h->nav[i], 0, sizeof h->nav[i]);
                    h->nav[i].ipage = ipage;
                }
                h->trk[i].prn = sv + 1;
                h->orbit[i] = e->orbit;
                if (gpsiq_nav_subframes(&e->nav, &h->utc, h->alm, h->sbf[i]) != GPSIQ_OK) return -1;          /* gps.c:2190 */
                if (gpsiq_nav_message(h->sbf[i], h->week, t, 1, &h->nav[i]) != GPSIQ_OK) return -1;         /* gps.c:2193 */
                h->trk[i].g0_week = h->nav[i].g0_week; h->trk[i].g0_sec = h->nav[i].g0_sec;
                memcpy(h->trk[i].dwrd, h->nav[i].dwrd, sizeof h->trk[i].dwrd);
                if (gpsiq_track_init(&h->orbit[i], &h->iono, h->week, t, h->xyz0, &h->trk[i], 1) != GPSIQ_OK) return -1;   /* gps.c:2196-2214 */
                h->allocated_sat[sv] = i;
                break;
            }
        } else if (h->allocated_sat[sv] >= 0) {                                                      /* gps.c:2224-2231 */
            h->trk[h->allocated_sat[sv]].prn = 0;
            h->allocated_sat[sv] = -1;
        }
    }
    return nsat;
}

/* gps.c:2889-2906: when the first valid satellite of the next set has its time of clock less than an
 * hour ahead, that set takes over: orbits at once, subframes rebuilt (the word buffer picks them up at
 * the next refresh). */
static int refresh_ephemeris(struct host_state *h, double t)
{
    if (h->ieph + 1 >= h->nsets) return 0;
    const gpsiq_rinex_eph_t *nxt = h->sets + (size_t) (h->ieph + 1) * GPSIQ_MAX_SAT;
    for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) {
        if (!nxt[sv].vflg) continue;
        const double dt = (double) (nxt[sv].toc_week - h->week) * 604800.0 + (nxt[sv].nav.toc_sec - t);   /* subGpsTime, gps.c:1096-1103 */
        if (dt < 3600.0) {
            h->ieph++;
            h->eph = nxt;
            for (int i = 0; i < h->nchan; ++i) {
                if (h->trk[i].prn == 0) continue;
                const gpsiq_rinex_eph_t *e = &h->eph[h->trk[i].prn - 1];
                if (gpsiq_nav_subframes(&e->nav, &h->utc, h->alm, h->sbf[i]) != GPSIQ_OK) return -1;
                h->orbit[i] = e->orbit;
            }
        }
        break;
    }
    return 0;
}

int main(int argc, char **argv)
{
    if (argc != 11 && argc != 12) {
        fprintf(stderr, "usage: %s rinex 2|3 week sec xyz.bin|motion.csv|lat,lon,h nblocks nchan fs 1|2 out.bin [almanac.sem]\n", argv[0]);
        return 2;
    }
    const int version = atoi(argv[2]), nchan = atoi(argv[7]), ss = atoi(argv[9]);
    int nblocks = atoi(argv[6]), week = atoi(argv[3]);
    double sec0 = atof(argv[4]);
    const double fs = atof(argv[8]);
    if (strchr(argv[3], '/')) {                                                /* the -t form */
        int y, mo, d, hh, mi;
        double s;
        if (sscanf(argv[3], "%d/%d/%d,%d:%d:%lf", &y, &mo, &d, &hh, &mi, &s) != 6) { fprintf(stderr, "bad start time %s\n", argv[3]); return 2; }
        gpsiq_date_to_gps(y, mo, d, hh, mi, s, &week, &sec0);                 /* gps-sim.c:104, gps.c:2295 */
    }
    const int nsamp = (int) floor(fs / 10.0 + 0.5);                           /* NUM_IQ_SAMPLES, sdr.h:26 */
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN || (ss != 1 && ss != 2)) { fprintf(stderr, "bad arguments\n"); return 2; }

    static gpsiq_rinex_eph_t eph[GPSIQ_EPHEM_SETS][GPSIQ_MAX_SAT];
    static struct host_state h;
    const int nsets = gpsiq_rinex_read(argv[1], version, &eph[0][0], &h.utc);
    if (nsets <= 0) { fprintf(stderr, "gpsiq_runahead: cannot read %s (%d)\n", argv[1], nsets); return 1; }
    const int ieph = gpsiq_rinex_select(&eph[0][0], nsets, week, sec0);        /* gps.c:2588-2608 */
    if (ieph < 0) { fprintf(stderr, "gpsiq_runahead: no ephemeris for the start time\n"); return 1; }

    double *xyz = malloc(sizeof(double) * 3 * ((size_t) nblocks + 1));
    if (!xyz) { fprintf(stderr, "cannot allocate positions\n"); return 1; }
    const size_t plen = strlen(argv[5]);
    double llh[3];
    if (plen > 4 && strcmp(argv[5] + plen - 4, ".csv") == 0) {
        const int npoints = gpsiq_motion_read_csv(argv[5], xyz, nblocks + 1);
        if (npoints <= 0) { fprintf(stderr, "gpsiq_runahead: cannot read motion file %s\n", argv[5]); return 1; }   /* gps.c:2497-2500 */
        if (npoints - 1 < nblocks) nblocks = npoints - 1;                     /* numd points give numd - 1 blocks, gps.c:2703 */
    } else if (sscanf(argv[5], "%lf,%lf,%lf", &llh[0], &llh[1], &llh[2]) == 3) {
        llh[0] /= 57.2957795131; llh[1] /= 57.2957795131;                       /* R2D, gps.h:98 */
        gpsiq_llh_to_ecef(llh, xyz);
        for (int k = 1; k <= nblocks; ++k) memcpy(xyz + 3 * (size_t) k, xyz, sizeof(double) * 3);
    } else {
        FILE *fx = fopen(argv[5], "rb");
        if (!fx || fread(xyz, sizeof(double) * 3, (size_t) nblocks + 1, fx) != (size_t) nblocks + 1) { fprintf(stderr, "bad xyz file\n"); return 2; }
        fclose(fx);
    }

    static gpsiq_nav_alm_sv_t alm[GPSIQ_MAX_SAT];
    if (argc == 12) {
        const int nalm = gpsiq_almanac_read_sem(argv[11], alm);
        if (nalm < 0) { fprintf(stderr, "gpsiq_runahead: %s\n", gpsiq_last_error()); return 1; }
        if (nalm > 0) h.alm = alm;                                             /* no valid record: as without a file */
        for (int sv = 0; sv < GPSIQ_MAX_SAT && h.alm; ++sv) {                  /* gps.c:2640-2650: within four weeks of the start */
            if (!alm[sv].valid) continue;
            const double dt = (alm[sv].toa_sec - sec0) + (double) (alm[sv].toa_week - week) * 604800.0;
            if (dt < -4.0 * 604800.0 || dt > 4.0 * 604800.0) { fprintf(stderr, "gpsiq_runahead: invalid time of almanac\n"); return 1; }
        }
    }
    h.nchan = nchan; h.week = week; h.eph = eph[ieph]; h.sets = &eph[0][0]; h.nsets = nsets; h.ieph = ieph;
    memcpy(h.xyz0, xyz, sizeof h.xyz0);
    h.iono.enable = 1; h.iono.vflg = h.utc.vflg;
    memcpy(h.iono.alpha, h.utc.alpha, sizeof h.iono.alpha); memcpy(h.iono.beta, h.utc.beta, sizeof h.iono.beta);
    for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) h.allocated_sat[sv] = -1;      /* gps.c:2668-2669 */
    int nsat = allocate(&h, sec0);                                             /* gps.c:2675 */
    if (nsat < 0) return die("allocate");

    gpsiq_ctx_t *gq = NULL;
    if (gpsiq_create(&gq, 0) != GPSIQ_OK) return die("create");
    {   /* GPSIQ_NCO=reference: the reference's double accumulators, as in the binding of INTEGRATION.md section 2 */
        const char *m = getenv("GPSIQ_NCO");
        if (m && strcmp(m, "reference") == 0 && gpsiq_set_nco_mode(gq, GPSIQ_NCO_REFERENCE) != GPSIQ_OK) return die("nco mode");
    }
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) ss;
    void *buf = gpsiq_host_alloc(blk_bytes * BLOCKS_PER_CALL);
    gpsiq_chan_t *desc = malloc(sizeof *desc * BLOCKS_PER_CALL * (size_t) nchan);
    FILE *fo = fopen(argv[10], "wb");
    if (!buf || !desc || !fo) { fprintf(stderr, "cannot allocate / open output\n"); return 1; }

    double carr[GPSIQ_MAX_CHAN];
    int have_carr = 0, nalloc = 1;
    long done = 0;
    while (done < nblocks) {
        const double t = time_after(sec0, done);
        /* blocks up to and including the one generated at the next multiple of 30 s (igrx % 300 == 0, gps.c:2870-2878) */
        const long tenths = (long) llround(t * 10.0);
        long n = 300 - tenths % 300;
        const int roll = done + n <= nblocks;
        if (n > nblocks - done) n = nblocks - done;
        for (long b = 0; b < n; b += BLOCKS_PER_CALL) {
            const int nb = (int) (n - b < BLOCKS_PER_CALL ? n - b : BLOCKS_PER_CALL);
            if (gpsiq_refresh_batch(h.orbit, &h.iono, week, time_after(sec0, done + b), xyz + 3 * (done + b + 1), nb, nchan, 0,
                                    h.trk, desc, 0) != GPSIQ_OK) return die("refresh");
            /* carr_phase is the loop's own state (gps.c:2821): hand back what the library handed out; a
             * slot allocateChannel() just (re)initialised carries its phase_ini instead (gps.c:2209-2211) */
            for (int k = 0; k < nb; ++k)
                for (int i = 0; i < nchan; ++i)
                    desc[(size_t) k * nchan + i].carr_phase = (have_carr && k == 0 && carr[i] >= 0.0) ? carr[i] : h.trk[i].carr_phase;
            if (gpsiq_generate_batch(gq, desc, nb, nchan, nsamp, fs, ss, buf, 0, carr) != GPSIQ_OK) return die("generate");
            have_carr = 1;
            if (fwrite(buf, blk_bytes, (size_t) nb, fo) != (size_t) nb) { fprintf(stderr, "short write\n"); return 1; }
        }
        done += n;
        if (roll) {                                                            /* gps.c:2878-2885, 2909 */
            const double tr = time_after(sec0, done);
            int prn_before[GPSIQ_MAX_CHAN];
            for (int i = 0; i < nchan; ++i) {
                prn_before[i] = h.trk[i].prn;
                if (h.trk[i].prn <= 0) continue;
                if (gpsiq_nav_message(h.sbf[i], week, tr, 0, &h.nav[i]) != GPSIQ_OK) return die("nav refresh");
                memcpy(h.trk[i].dwrd, h.nav[i].dwrd, sizeof h.trk[i].dwrd);
                h.trk[i].g0_week = h.nav[i].g0_week; h.trk[i].g0_sec = h.nav[i].g0_sec;
            }
            if (refresh_ephemeris(&h, tr) != 0) return die("ephemeris refresh");
            nsat = allocate(&h, tr);
            if (nsat < 0) return die("allocate");
            ++nalloc;
            for (int i = 0; i < nchan; ++i)          /* a re-allocated slot starts from its own phase_ini */
                if (h.trk[i].prn != prn_before[i]) carr[i] = -1.0;
        }
    }
    fclose(fo);
    printf("%d blocks, %d channels, %d allocation passes, last nsat %d, ephemeris set %d of %d\n", nblocks, nchan, nalloc, nsat, h.ieph, nsets);
    gpsiq_host_free(buf);
    gpsiq_destroy(gq);
    free(desc); free(xyz);
    return 0;
}
