/* Exercises the host-only half of libgpsiq (quantiser incl. the worker pool, refresh, nav words,
 * RINEX readers, fifo hand-off) for AddressSanitizer / UndefinedBehaviorSanitizer.  Built by
 * tests/test_host_c.py from the library's host sources; no device code, no GPU.
 *   sanitize_host [rinex [scratch-file [2|3]]] */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsiq_extras.h"
#include "gpsiq_plumbing.h"     /* the tables read back: gpsiq_prn_code, gpsiq_carrier_table (linked in directly here) */

static uint64_t rng_state = 0x243f6a8885a308d3ull;
static uint64_t rnd(void) { rng_state += 0x9e3779b97f4a7c15ull; uint64_t z = rng_state; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull; return z ^ (z >> 31); }
static double uni(void) { return (double) (rnd() >> 11) / 9007199254740992.0; }

static gpsiq_iq_buf_t pool[4];
static int pool_next, enq_count;
static gpsiq_iq_buf_t *acq(void *u) { (void) u; gpsiq_iq_buf_t *b = &pool[pool_next++ & 3]; b->validLength = 0; return b; }
static void enq(void *u, gpsiq_iq_buf_t *b) { (void) u; (void) b; ++enq_count; }

#define CHECK(c) do { if (!(c)) { fprintf(stderr, "sanitize_host: check failed at line %d: %s (%s)\n", __LINE__, #c, gpsiq_last_error()); return 1; } } while (0)

int main(int argc, char **argv)
{
    /* ---- tables, quantiser, batch quantiser on the worker pool, error paths ---- */
    uint8_t chips[GPSIQ_CA_SEQ_LEN]; int16_t c512[512], s512[512];
    for (int prn = 1; prn <= 32; ++prn) CHECK(gpsiq_prn_code(prn, chips) == GPSIQ_OK);
    CHECK(gpsiq_prn_code(33, chips) < 0);
    gpsiq_carrier_table(c512, s512);
    const int nb = 3000, nc = 16;
    gpsiq_chan_t *ch = calloc((size_t) nb * nc, sizeof *ch);
    gpsiq_qchan_t *q = calloc((size_t) nb * nc, sizeof *q);
    for (int i = 0; i < nb * nc; ++i) {
        ch[i].prn = 1 + (int) (rnd() % 32); ch[i].iword = (int) (rnd() % 58); ch[i].ibit = (int) (rnd() % 30); ch[i].icode = (int) (rnd() % 20);
        ch[i].f_carr = (uni() - 0.5) * 1e4; ch[i].f_code = 1.023e6 + ch[i].f_carr / 1540.0;
        ch[i].carr_phase = uni(); ch[i].code_phase = uni() * 1023.0; ch[i].gain = uni();
        for (int k = 0; k < GPSIQ_N_DWRD; ++k) ch[i].dwrd[k] = (uint32_t) rnd() & 0x3fffffffu;
    }
    uint64_t carry[GPSIQ_MAX_CHAN];
    for (int rep = 0; rep < 3; ++rep) CHECK(gpsiq_quantize_batch(ch, nb, nc, 2.6e6, 260000, q, NULL, carry) == GPSIQ_OK);
    ch[1234 * nc + 5].icode = 20;
    CHECK(gpsiq_quantize_batch(ch, nb, nc, 2.6e6, 260000, q, NULL, carry) == GPSIQ_E_RANGE);
    CHECK(strstr(gpsiq_last_error(), "block 1234") != NULL);
    CHECK(gpsiq_quantize_batch(ch, 0, nc, 2.6e6, 260000, q, NULL, carry) == GPSIQ_OK);
    int b0, b1;
    CHECK(gpsiq_shard_range(35999, 7, 8, &b0, &b1) == GPSIQ_OK && b1 == 35999);
    ch[1234 * nc + 5].icode = 3;

    /* ---- GPSIQ_NCO_REFERENCE host half: carrier walk on the channel threads, candidate search, patches;
     *      slots that change satellite, unused slots, a phase on a LUT boundary with a tiny step (every sample
     *      is then a candidate), too small a patch buffer ---- */
    {
        const int nbr = 40;
        gpsiq_chan_t *cr = calloc((size_t) nbr * nc, sizeof *cr);
        memcpy(cr, ch, sizeof *cr * (size_t) nbr * nc);
        for (int b = 0; b < nbr; ++b)
            for (int c = 0; c < nc; ++c) {
                gpsiq_chan_t *e = &cr[b * nc + c];
                e->prn = c == 3 ? 0 : (c == 5 && b >= 17 ? 29 : c + 1);
                e->f_carr = cr[c].f_carr + 0.4 * b; e->f_code = 1.023e6 + e->f_carr / 1540.0;
                if (c == 7) { e->f_carr = -1e-9; e->f_code = 1.023e6; e->carr_phase = 0.5; }
                if (c == 8) { e->f_carr = 0.0; e->f_code = 1.023e6; e->carr_phase = 0.0; }
            }
        static gpsiq_patch_t patches[1 << 16];
        int np = -1; double carr_end[GPSIQ_MAX_CHAN];
        for (int fsk = 0; fsk < 2; ++fsk) {
            const double fs = fsk ? 25e6 : 2.6e6; const int ns = fsk ? 250000 : 26000;     /* short blocks: the walk does not care */
            CHECK(gpsiq_reference_batch(cr, nbr, nc, fs, ns, q, patches, 1 << 16, &np, carr_end) == GPSIQ_OK && np >= 0);
            for (int i = 1; i < np; ++i)
                CHECK(patches[i - 1].block < patches[i].block || (patches[i - 1].block == patches[i].block && patches[i - 1].sample <= patches[i].sample));
            for (int i = 0; i < np; ++i) CHECK(patches[i].block < (uint32_t) nbr && patches[i].sample < (uint32_t) ns && patches[i].lut < 512 && patches[i].slot < nc);
            if (np > 1) { int np2 = 0; CHECK(gpsiq_reference_batch(cr, nbr, nc, fs, ns, q, patches, 1, &np2, carr_end) == GPSIQ_E_RANGE && np2 == np); }
        }
        gpsiq_shard_carry_t sc[2][GPSIQ_MAX_CHAN];
        CHECK(gpsiq_quantize_batch(cr, 20, nc, 2.6e6, 26000, q, NULL, NULL) == GPSIQ_OK);
        CHECK(gpsiq_quantize_batch(cr + 20 * nc, 20, nc, 2.6e6, 26000, q + 20 * nc, NULL, NULL) == GPSIQ_OK);
        CHECK(gpsiq_shard_carry(q, 20, nc, 26000, sc[0]) == GPSIQ_OK && gpsiq_shard_carry(q + 20 * nc, 20, nc, 26000, sc[1]) == GPSIQ_OK);
        CHECK(gpsiq_shard_seed(q + 20 * nc, 20, nc, 26000, &sc[0][0], 1) == GPSIQ_OK);
        gpsiq_qchan_t *whole = calloc((size_t) nbr * nc, sizeof *whole);
        CHECK(gpsiq_quantize_batch(cr, nbr, nc, 2.6e6, 26000, whole, NULL, NULL) == GPSIQ_OK);
        CHECK(memcmp(whole + 20 * nc, q + 20 * nc, sizeof *whole * 20 * nc) == 0);
        free(whole); free(cr);
    }

    /* ---- fifo hand-off rules ---- */
    static int16_t store[4][520000];
    for (int i = 0; i < 4; ++i) { pool[i].data16 = store[i]; pool[i].data8 = NULL; pool[i].totalLength = 520000; }
    gpsiq_chunker_t ck;
    static int16_t blk[520000];
    for (int sink = GPSIQ_SINK_IQFILE; sink <= GPSIQ_SINK_PLUTOSDR; ++sink) {
        for (int i = 0; i < 4; ++i) pool[i].totalLength = sink == GPSIQ_SINK_HACKRF ? GPSIQ_HACKRF_CHUNK : 520000;
        CHECK(gpsiq_chunker_init(&ck, sink, GPSIQ_SC16, acq, enq, NULL) == GPSIQ_OK);
        for (int k = 0; k < 5; ++k) {
            void *w = gpsiq_chunker_reserve(&ck, 520000);
            if (w) { memcpy(w, blk, sizeof blk); CHECK(gpsiq_chunker_commit(&ck, 520000) == 1); }
            else CHECK(gpsiq_chunker_push(&ck, blk, 520000) >= 0);
        }
    }

    /* ---- RINEX reader on the file given (and on a truncated copy), nav words, refresh ---- */
    if (argc > 1) {
        static gpsiq_rinex_eph_t eph[GPSIQ_EPHEM_SETS][GPSIQ_MAX_SAT];
        gpsiq_nav_utc_t utc;
        const int version = argc > 3 ? atoi(argv[3]) : 2;
        const int nsets = gpsiq_rinex_read(argv[1], version, &eph[0][0], &utc);
        CHECK(nsets >= 1);
        CHECK(gpsiq_rinex_read("/nonexistent/file.21n", 2, &eph[0][0], &utc) == -1);
        if (argc > 2) {               /* every prefix length of the file in coarse steps: no over-read on short lines */
            FILE *f = fopen(argv[1], "rb"); CHECK(f);
            static char text[1 << 20]; const size_t len = fread(text, 1, sizeof text, f); fclose(f);
            for (size_t cut = 0; cut < len; cut += 97) {
                FILE *o = fopen(argv[2], "wb"); CHECK(o); fwrite(text, 1, cut, o); fclose(o);
                static gpsiq_rinex_eph_t e2[GPSIQ_EPHEM_SETS][GPSIQ_MAX_SAT]; gpsiq_nav_utc_t u2;
                (void) gpsiq_rinex_read(argv[2], 2, &e2[0][0], &u2);
                (void) gpsiq_rinex_read(argv[2], 3, &e2[0][0], &u2);
            }
            for (int it = 0; it < 300; ++it) {            /* random damage: bytes overwritten, lines cut or doubled */
                static char mut[1 << 20];
                memcpy(mut, text, len);
                size_t mlen = len;
                for (int k = 0; k < 1 + (int) (rnd() % 12); ++k) {
                    const size_t at = rnd() % len;
                    switch (rnd() % 4) {
                    case 0: mut[at] = (char) (rnd() & 0xff); break;
                    case 1: mut[at] = '\n'; break;
                    case 2: mut[at] = "0123456789.-+DEed "[rnd() % 18]; break;
                    default: if (at + 200 < mlen) { memmove(mut + at, mut + at + 1 + rnd() % 150, mlen - at - 160); mlen -= 160; } break;
                    }
                }
                FILE *o = fopen(argv[2], "wb"); CHECK(o); fwrite(mut, 1, mlen, o); fclose(o);
                static gpsiq_rinex_eph_t e2[GPSIQ_EPHEM_SETS][GPSIQ_MAX_SAT]; gpsiq_nav_utc_t u2;
                (void) gpsiq_rinex_read(argv[2], version, &e2[0][0], &u2);
            }
            CHECK(gpsiq_rinex_read(argv[1], version, &eph[0][0], &utc) == nsets);
        }
        const int week = eph[0][0].vflg ? eph[0][0].nav.toe_week : 2190;
        const int ieph = gpsiq_rinex_select(&eph[0][0], nsets, week, 270000.0);
        const gpsiq_rinex_eph_t *set = eph[ieph >= 0 ? ieph : 0];
        static uint32_t sbf[GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF];
        gpsiq_ephem_t orbit[GPSIQ_MAX_CHAN]; gpsiq_track_t trk[GPSIQ_MAX_CHAN]; gpsiq_iono_t iono;
        static gpsiq_track_t trk_first[GPSIQ_MAX_CHAN];
        memset(trk_first, 0, sizeof trk_first);
        memset(trk, 0, sizeof trk); memset(&iono, 0, sizeof iono);
        iono.enable = 1; iono.vflg = utc.vflg; memcpy(iono.alpha, utc.alpha, sizeof iono.alpha); memcpy(iono.beta, utc.beta, sizeof iono.beta);
        const double xyz0[3] = {-3959000.0, 3350000.0, 3699000.0};
        int n = 0;
        for (int sv = 0; sv < GPSIQ_MAX_SAT && n < GPSIQ_MAX_CHAN; ++sv) {
            if (!set[sv].vflg) continue;
            double azel[2];
            (void) gpsiq_sat_visibility(&set[sv].orbit, week, 270000.0, xyz0, 0.0, azel);
            gpsiq_nav_state_t st; memset(&st, 0, sizeof st);
            CHECK(gpsiq_nav_subframes(&set[sv].nav, &utc, NULL, sbf) == GPSIQ_OK);
            CHECK(gpsiq_nav_message(sbf, week, 270000.0, 1, &st) == GPSIQ_OK);
            trk_first[n].prn = sv + 1; trk_first[n].g0_week = st.g0_week; trk_first[n].g0_sec = st.g0_sec;   /* the frame of the start time */
            memcpy(trk_first[n].dwrd, st.dwrd, sizeof st.dwrd);
            for (int k = 1; k < 26; ++k) CHECK(gpsiq_nav_message(sbf, week, 270000.0 + 30.0 * k, 0, &st) == GPSIQ_OK);
            trk[n].prn = sv + 1; trk[n].g0_week = st.g0_week; trk[n].g0_sec = st.g0_sec; memcpy(trk[n].dwrd, st.dwrd, sizeof st.dwrd);
            orbit[n++] = set[sv].orbit;
        }
        CHECK(n > 0);
        CHECK(gpsiq_track_init(orbit, &iono, week, 270000.0, xyz0, trk, n) == GPSIQ_OK);
        const int nblk = 5000;
        double *xyz = malloc(sizeof(double) * 3 * nblk);
        for (int k = 0; k < nblk; ++k) { xyz[3 * k] = xyz0[0] + 0.3 * k; xyz[3 * k + 1] = xyz0[1]; xyz[3 * k + 2] = xyz0[2] - 0.1 * k; }
        gpsiq_chan_t *out = malloc(sizeof *out * (size_t) nblk * n);
        gpsiq_track_t trk0[GPSIQ_MAX_CHAN];
        memcpy(trk0, trk, sizeof trk0);                           /* the state before the first block ... */
        for (int c = 0; c < n; ++c) {                             /* ... with the word buffer of the start time */
            trk0[c].g0_week = trk_first[c].g0_week; trk0[c].g0_sec = trk_first[c].g0_sec;
            memcpy(trk0[c].dwrd, trk_first[c].dwrd, sizeof trk0[c].dwrd);
        }
        for (int threads = 0; threads <= 3; threads += 3) CHECK(gpsiq_refresh_batch(orbit, &iono, week, 270000.0, xyz, nblk, n, 0, trk, out, threads) == GPSIQ_OK);
        /* several epochs in one pass, plain and fused with the quantiser */
        gpsiq_track_t trk_ep[3][GPSIQ_MAX_CHAN];
        const int nq = 290, first[3] = {0, nq / 3, nq / 3 + 1};      /* one word buffer lasts 30 s = 300 blocks */
        for (int e = 0; e < 3; ++e) memcpy(trk_ep[e], trk0, sizeof(gpsiq_track_t) * (size_t) n);
        gpsiq_track_t *te = malloc(sizeof(gpsiq_track_t) * 3 * (size_t) n);
        for (int e = 0; e < 3; ++e) memcpy(te + (size_t) e * n, trk_ep[e], sizeof(gpsiq_track_t) * (size_t) n);
        CHECK(gpsiq_refresh_epochs(orbit, &iono, week, 270000.0, xyz, nq, n, 0, te, first, 3, out, 0) == GPSIQ_OK);
        gpsiq_qchan_t *qa = malloc(sizeof *qa * (size_t) nblk * n), *qb = malloc(sizeof *qb * (size_t) nblk * n);
        CHECK(gpsiq_quantize_batch(out, nq, n, 2.6e6, 260000, qa, NULL, NULL) == GPSIQ_OK);
        for (int e = 0; e < 3; ++e) memcpy(te + (size_t) e * n, trk_ep[e], sizeof(gpsiq_track_t) * (size_t) n);
        CHECK(gpsiq_refresh_epochs_quantized(orbit, &iono, week, 270000.0, xyz, nq, n, 0, te, first, 3, 2.6e6, 260000, qb, 3) == GPSIQ_OK);
        CHECK(memcmp(qa, qb, sizeof *qa * (size_t) nq * n) == 0);
        CHECK(gpsiq_refresh_epochs_quantized(orbit, &iono, week, 270000.0, xyz, nq, n, 0, te, first, 3, 0.0, 260000, qb, 0) == GPSIQ_E_ARG);
        free(qa); free(qb); free(te);
        free(out); free(xyz);
    }
    free(q); free(ch);
    printf("ok\n");
    return 0;
}
