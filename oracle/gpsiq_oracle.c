/*
 * gpsiq_oracle.c — CPU restatement of the reference hot path.  TEST INFRASTRUCTURE ONLY
 * (see gpsiq_oracle.h).  Written from the behaviour of Mictronics/multi-sdr-gps-sim
 * gps.c; every function cites the reference lines it follows.
 *
 * Parity pin: tests/test_oracle_vs_ref.py compares every function here with
 * oracle/_ref/libgpsref.so (the reference's own lines compiled in place) and
 * tests/test_golden.py with the committed captures under tests/golden/.
 *
 * Build: see oracle/Makefile (gcc -O2 -std=c11 -ffp-contract=off: no FMA contraction,
 * so the double arithmetic is the same IEEE sequence the reference executes).
 */
#include "gpsiq_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------------- */
/* Carrier LUT.  gps.c:145-178 sinTable512, gps.c:180-213 cosTable512.
 * The reference tables obey sin[k] = sin[255-k] (k < 256), sin[k+256] = -sin[k],
 * cos[k] = sin[(k+128) % 512]; the first quarter wave is the data below
 * (= round(250 sin(2 pi (k+.5)/512)) except entry 35, which is 105, not 106). */
static const short quarter_wave[128] = {
      2,   5,   8,  11,  14,  17,  20,  23,  26,  29,  32,  35,  38,  41,  44,  47,
     50,  53,  56,  59,  62,  65,  68,  71,  74,  77,  80,  83,  86,  89,  91,  94,
     97, 100, 103, 105, 108, 111, 114, 116, 119, 122, 125, 127, 130, 132, 135, 138,
    140, 143, 145, 148, 150, 153, 155, 157, 160, 162, 164, 167, 169, 171, 173, 176,
    178, 180, 182, 184, 186, 188, 190, 192, 194, 196, 198, 200, 202, 204, 205, 207,
    209, 210, 212, 214, 215, 217, 218, 220, 221, 223, 224, 225, 227, 228, 229, 230,
    232, 233, 234, 235, 236, 237, 238, 239, 240, 241, 241, 242, 243, 244, 244, 245,
    245, 246, 247, 247, 248, 248, 248, 249, 249, 249, 249, 250, 250, 250, 250, 250,
};

int oracle_sin512(int k)
{
    k &= 511;
    int h = k & 255;
    int v = quarter_wave[h < 128 ? h : 255 - h];
    return k < 256 ? v : -v;
}

int oracle_cos512(int k)
{
    return oracle_sin512(k + 128);
}

/* ------------------------------------------------------------------------- */
/* C/A code.  gps.c:272-309 codegen(): two 10-stage LFSRs started all-ones,
 * G1 feedback stages 3,10 (gps.c:297), G2 feedback stages 2,3,6,8,9,10 (gps.c:298),
 * output G1[i] xor G2[i - delay[prn]] (gps.c:306-307) with the ICD-GPS-200 G2 delays
 * (gps.c:273-278).  The reference works on +-1 and maps (1 - g1*g2)/2; here on bits. */
static const short g2_delay[32] = {
      5,   6,   7,   8,  17,  18, 139, 140, 141, 251, 252, 254, 255, 256, 257, 258,
    469, 470, 471, 472, 473, 474, 509, 512, 513, 514, 515, 516, 859, 860, 861, 862,
};

int oracle_codegen(int prn, uint8_t ca[GPSIQ_CA_SEQ_LEN])
{
    uint8_t g1[GPSIQ_CA_SEQ_LEN], g2[GPSIQ_CA_SEQ_LEN];
    unsigned r1 = 0x3ff, r2 = 0x3ff; /* bit s-1 = stage s */

    if (prn < 1 || prn > 32)
        return GPSIQ_E_ARG;
    for (int i = 0; i < GPSIQ_CA_SEQ_LEN; i++) {
        g1[i] = (r1 >> 9) & 1;
        g2[i] = (r2 >> 9) & 1;
        unsigned f1 = ((r1 >> 2) ^ (r1 >> 9)) & 1;
        unsigned f2 = ((r2 >> 1) ^ (r2 >> 2) ^ (r2 >> 5) ^ (r2 >> 7) ^ (r2 >> 8) ^ (r2 >> 9)) & 1;
        r1 = ((r1 << 1) | f1) & 0x3ff;
        r2 = ((r2 << 1) | f2) & 0x3ff;
    }
    int d = g2_delay[prn - 1];
    for (int i = 0; i < GPSIQ_CA_SEQ_LEN; i++)
        ca[i] = g1[i] ^ g2[(i + GPSIQ_CA_SEQ_LEN - d) % GPSIQ_CA_SEQ_LEN];
    return GPSIQ_OK;
}

/* ------------------------------------------------------------------------- */
/* gps.c:2839-2846: int16 passthrough, or arithmetic >>4 then wrap to signed char. */
static void pack_elems(const short *iq, size_t nelem, int sample_size, void *dst)
{
    if (sample_size == GPSIQ_SC16) {
        memcpy(dst, iq, nelem * sizeof(short));
    } else {
        signed char *d8 = (signed char *) dst;
        for (size_t k = 0; k < nelem; k++)
            d8[k] = (signed char) (iq[k] >> 4);
    }
}

/* ------------------------------------------------------------------------- */
/* The loop as the reference runs it.  gps.c:2767-2836 (+ gps.c:2058-2059 for the
 * initial dataBit / codeCA, gps.c:2298 for delt). */
int oracle_block_float(const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                       int sample_size, void *dst, double *carr_phase_out)
{
    if (!ch || !dst || nchan < 0 || nchan > GPSIQ_MAX_CHAN || nsamp < 0)
        return GPSIQ_E_ARG;
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16)
        return GPSIQ_E_ARG;

    struct st {
        int on, iword, ibit, icode, data, code;
        double carr, codeph, carr_inc, code_inc, gain;
        const uint32_t *dwrd;
        uint8_t ca[GPSIQ_CA_SEQ_LEN];
    } *s = calloc((size_t) (nchan ? nchan : 1), sizeof *s);
    short *iq = malloc(sizeof(short) * 2 * (size_t) (nsamp ? nsamp : 1));
    if (!s || !iq) { free(s); free(iq); return GPSIQ_E_NOMEM; }

    const double delt = 1.0 / fs;                                   /* gps.c:2298 */
    int rc = GPSIQ_OK;
    for (int c = 0; c < nchan; c++) {
        s[c].on = ch[c].prn > 0;                                    /* gps.c:2772 */
        if (!s[c].on) continue;
        if (oracle_codegen(ch[c].prn, s[c].ca) != GPSIQ_OK) { rc = GPSIQ_E_ARG; goto out; }
        if (ch[c].iword < 0 || ch[c].iword >= GPSIQ_N_DWRD || ch[c].ibit < 0 || ch[c].ibit >= 30 ||
            !(ch[c].code_phase >= 0.0 && ch[c].code_phase < GPSIQ_CA_SEQ_LEN)) { rc = GPSIQ_E_RANGE; goto out; }
        s[c].iword = ch[c].iword; s[c].ibit = ch[c].ibit; s[c].icode = ch[c].icode;
        s[c].carr = ch[c].carr_phase; s[c].codeph = ch[c].code_phase;
        s[c].carr_inc = ch[c].f_carr * delt;                         /* gps.c:2821 operand */
        s[c].code_inc = ch[c].f_code * delt;                         /* gps.c:2789 operand */
        s[c].gain = ch[c].gain; s[c].dwrd = ch[c].dwrd;
        s[c].code = s[c].ca[(int) s[c].codeph] * 2 - 1;              /* gps.c:2058 */
        s[c].data = (int) ((s[c].dwrd[s[c].iword] >> (29 - s[c].ibit)) & 1u) * 2 - 1; /* gps.c:2059 */
    }

    for (int n = 0; n < nsamp; n++) {
        int i_acc = 0, q_acc = 0;                                    /* gps.c:2768-2769 */
        for (int c = 0; c < nchan; c++) {
            struct st *p = &s[c];
            if (!p->on) continue;
            int k = (int) floor(p->carr * 512.0);                    /* gps.c:2775 */
            int ip = (int) (p->data * p->code * oracle_cos512(k) * p->gain); /* gps.c:2781 */
            int qp = (int) (p->data * p->code * oracle_sin512(k) * p->gain); /* gps.c:2782 */
            i_acc += ip;                                             /* gps.c:2785 */
            q_acc += qp;
            p->codeph += p->code_inc;                                /* gps.c:2789 */
            if (p->codeph >= GPSIQ_CA_SEQ_LEN) {                     /* gps.c:2791-2814 */
                p->codeph -= GPSIQ_CA_SEQ_LEN;
                if (++p->icode >= 20) {
                    p->icode = 0;
                    if (++p->ibit >= 30) {
                        p->ibit = 0;
                        if (++p->iword >= GPSIQ_N_DWRD) { rc = GPSIQ_E_RANGE; goto out; }
                    }
                    p->data = (int) ((p->dwrd[p->iword] >> (29 - p->ibit)) & 1u) * 2 - 1;
                }
            }
            p->code = p->ca[(int) p->codeph] * 2 - 1;                /* gps.c:2817 */
            p->carr += p->carr_inc;                                  /* gps.c:2821 */
            if (p->carr >= 1.0) p->carr -= 1.0;                      /* gps.c:2823-2826 */
            else if (p->carr < 0.0) p->carr += 1.0;
        }
        iq[2 * n] = (short) i_acc;                                   /* gps.c:2834 */
        iq[2 * n + 1] = (short) q_acc;                               /* gps.c:2835 */
    }
    pack_elems(iq, 2 * (size_t) nsamp, sample_size, dst);
    if (carr_phase_out)
        for (int c = 0; c < nchan; c++)
            carr_phase_out[c] = s[c].on ? s[c].carr : ch[c].carr_phase;
out:
    free(s); free(iq);
    return rc;
}

/* The carrier accumulator alone: gps.c:2821-2826, nsamp times.  (For the tests of the product's wrap-to-wrap table.) */
double oracle_carrier_chain(double carr_phase, double carr_inc, long nsamp)
{
    for (long n = 0; n < nsamp; n++) {
        carr_phase += carr_inc;                                       /* gps.c:2821 */
        if (carr_phase >= 1.0) carr_phase -= 1.0;                     /* gps.c:2823-2826 */
        else if (carr_phase < 0.0) carr_phase += 1.0;
    }
    return carr_phase;
}

/* ------------------------------------------------------------------------- */
/* include/gpsiq.h quantisation rules. */
#define CARR_F GPSIQ_CARR_FRAC_BITS
#define CODE_F GPSIQ_CODE_FRAC_BITS
#define CARR_MASK ((UINT64_C(1) << CARR_F) - 1)
#define CODE_MASK ((UINT64_C(1) << CODE_F) - 1)

int oracle_quantize(const gpsiq_chan_t *ch, int nchan, double fs, int nsamp,
                    gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out)
{
    if (!ch || !out || nchan < 0 || nchan > GPSIQ_MAX_CHAN || nsamp < 0 || !(fs > 0.0))
        return GPSIQ_E_ARG;
    const double delt = 1.0 / fs;                                    /* gps.c:2298 */
    for (int c = 0; c < nchan; c++) {
        gpsiq_qchan_t *q = &out[c];
        memset(q, 0, sizeof *q);
        if (ch[c].prn <= 0) { if (carry_out) carry_out[c] = 0; continue; }
        if (ch[c].prn > 32) return GPSIQ_E_ARG;
        double cinc = ch[c].f_carr * delt, kinc = ch[c].f_code * delt;
        if (!(fabs(cinc) < 0.5) || !(kinc > 0.0 && kinc < 2.0)) return GPSIQ_E_RANGE;
        if (!(ch[c].carr_phase >= 0.0 && ch[c].carr_phase < 1.0)) return GPSIQ_E_RANGE;
        if (!(ch[c].code_phase >= 0.0 && ch[c].code_phase < GPSIQ_CA_SEQ_LEN)) return GPSIQ_E_RANGE;
        if (ch[c].iword < 0 || ch[c].iword >= GPSIQ_N_DWRD || ch[c].ibit < 0 || ch[c].ibit >= 30 ||
            ch[c].icode < 0 || ch[c].icode >= 20) return GPSIQ_E_RANGE;
        q->prn = (uint8_t) ch[c].prn;
        q->icode = (uint8_t) ch[c].icode;
        q->gain = ch[c].gain;
        q->carr_step = llrint(ldexp(cinc, CARR_F));
        q->carr_phase = carry_in ? (carry_in[c] & CARR_MASK)
                                 : ((uint64_t) floor(ldexp(ch[c].carr_phase, CARR_F)) & CARR_MASK);
        int chip0 = (int) ch[c].code_phase;
        q->chip0 = (uint16_t) chip0;
        q->code_frac = (uint64_t) floor(ldexp(ch[c].code_phase - (double) chip0, CODE_F)) & CODE_MASK;
        q->code_step = (uint64_t) llrint(ldexp(kinc, CODE_F));
        /* nav bits the block touches: walk (iword, ibit) as gps.c:2799-2811 does */
        u128 tmax = (u128) q->code_frac + (u128) q->code_step * (u128) (nsamp > 0 ? nsamp - 1 : 0);
        uint64_t amax = (uint64_t) chip0 + (uint64_t) (tmax >> CODE_F);
        uint64_t nbits = ((uint64_t) q->icode + amax / GPSIQ_CA_SEQ_LEN) / 20 + 1;
        if (nbits > GPSIQ_MAX_NAV_BITS) return GPSIQ_E_RANGE;
        for (unsigned b = 0; b < nbits; b++) {
            unsigned pos = (unsigned) ch[c].ibit + b;
            unsigned w = (unsigned) ch[c].iword + pos / 30, bp = pos % 30;
            if (w >= GPSIQ_N_DWRD) return GPSIQ_E_RANGE;
            q->nav_bits |= ((ch[c].dwrd[w] >> (29 - bp)) & 1u) << b;  /* gps.c:2811 */
        }
        if (carry_out)
            carry_out[c] = (q->carr_phase + (uint64_t) q->carr_step * (uint64_t) nsamp) & CARR_MASK;
    }
    return GPSIQ_OK;
}

/* ------------------------------------------------------------------------- */
/* Per-channel gain LUT: TC[k] = (int)(cosTable512[k]*gain), TS likewise
 * (gps.c:2781-2782 with dataBit*codeCA factored out: truncation is odd-symmetric). */
static void gain_lut(double gain, int tc[512], int ts[512])
{
    for (int k = 0; k < 512; k++) {
        tc[k] = (int) (oracle_cos512(k) * gain);
        ts[k] = (int) (oracle_sin512(k) * gain);
    }
}

int oracle_block_fixed_range(const gpsiq_qchan_t *q, int nchan, long n0, long cnt,
                             int sample_size, void *dst)
{
    if (!q || !dst || nchan < 0 || nchan > GPSIQ_MAX_CHAN || n0 < 0 || cnt < 0)
        return GPSIQ_E_ARG;
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16)
        return GPSIQ_E_ARG;
    static _Thread_local int tc[GPSIQ_MAX_CHAN][512], ts[GPSIQ_MAX_CHAN][512];
    static _Thread_local uint8_t ca[GPSIQ_MAX_CHAN][GPSIQ_CA_SEQ_LEN];
    for (int c = 0; c < nchan; c++) {
        if (!q[c].prn) continue;
        if (oracle_codegen(q[c].prn, ca[c]) != GPSIQ_OK) return GPSIQ_E_ARG;
        gain_lut(q[c].gain, tc[c], ts[c]);
    }
    short *iq = malloc(sizeof(short) * 2 * (size_t) (cnt ? cnt : 1));
    if (!iq) return GPSIQ_E_NOMEM;
    for (long j = 0; j < cnt; j++) {
        uint64_t n = (uint64_t) (n0 + j);
        int i_acc = 0, q_acc = 0;
        for (int c = 0; c < nchan; c++) {
            if (!q[c].prn) continue;
            uint64_t P = (q[c].carr_phase + (uint64_t) q[c].carr_step * n) & CARR_MASK;
            int idx = (int) (P >> (CARR_F - 9));
            u128 T = (u128) q[c].code_frac + (u128) q[c].code_step * (u128) n;
            uint64_t A = (uint64_t) q[c].chip0 + (uint64_t) (T >> CODE_F);
            unsigned chip = (unsigned) (A % GPSIQ_CA_SEQ_LEN);
            uint64_t period = A / GPSIQ_CA_SEQ_LEN;
            unsigned bit = (unsigned) ((q[c].icode + period) / 20);
            unsigned neg = ca[c][chip] ^ ((q[c].nav_bits >> (bit & 31)) & 1u);
            i_acc += neg ? -tc[c][idx] : tc[c][idx];
            q_acc += neg ? -ts[c][idx] : ts[c][idx];
        }
        iq[2 * j] = (short) i_acc;
        iq[2 * j + 1] = (short) q_acc;
    }
    pack_elems(iq, 2 * (size_t) cnt, sample_size, dst);
    free(iq);
    return GPSIQ_OK;
}

int oracle_block_fixed(const gpsiq_qchan_t *q, int nchan, int nsamp, int sample_size, void *dst)
{
    return oracle_block_fixed_range(q, nchan, 0, nsamp, sample_size, dst);
}

/* Same samples with both NCOs advanced by addition: the straightforward single-core
 * CPU implementation of the fixed-point model (timed as the "port" CPU baseline). */
int oracle_block_fixed_seq(const gpsiq_qchan_t *q, int nchan, int nsamp, int sample_size, void *dst)
{
    if (!q || !dst || nchan < 0 || nchan > GPSIQ_MAX_CHAN || nsamp < 0)
        return GPSIQ_E_ARG;
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16)
        return GPSIQ_E_ARG;
    struct st {
        uint64_t P, dP, T, dT;
        unsigned chip, icode, bit, nav, neg_data;
        int tc[512], ts[512];
        uint8_t ca[GPSIQ_CA_SEQ_LEN];
    } *s = calloc(GPSIQ_MAX_CHAN, sizeof *s);
    short *iq = malloc(sizeof(short) * 2 * (size_t) (nsamp ? nsamp : 1));
    if (!s || !iq) { free(s); free(iq); return GPSIQ_E_NOMEM; }
    int act[GPSIQ_MAX_CHAN], na = 0;
    for (int c = 0; c < nchan; c++) {
        if (!q[c].prn) continue;
        struct st *p = &s[na];
        if (oracle_codegen(q[c].prn, p->ca) != GPSIQ_OK) { free(s); free(iq); return GPSIQ_E_ARG; }
        gain_lut(q[c].gain, p->tc, p->ts);
        p->P = q[c].carr_phase; p->dP = (uint64_t) q[c].carr_step;
        p->T = q[c].code_frac; p->dT = q[c].code_step;
        p->chip = q[c].chip0; p->icode = q[c].icode; p->bit = 0; p->nav = q[c].nav_bits;
        p->neg_data = p->nav & 1u;
        act[na++] = c;
    }
    (void) act;
    for (int n = 0; n < nsamp; n++) {
        int i_acc = 0, q_acc = 0;
        for (int a = 0; a < na; a++) {
            struct st *p = &s[a];
            int idx = (int) ((p->P >> (CARR_F - 9)) & 511);
            unsigned neg = p->ca[p->chip] ^ p->neg_data;
            i_acc += neg ? -p->tc[idx] : p->tc[idx];
            q_acc += neg ? -p->ts[idx] : p->ts[idx];
            p->P += p->dP;
            p->T += p->dT;
            p->chip += (unsigned) (p->T >> CODE_F);
            p->T &= CODE_MASK;
            while (p->chip >= GPSIQ_CA_SEQ_LEN) {
                p->chip -= GPSIQ_CA_SEQ_LEN;
                if (++p->icode >= 20) {
                    p->icode = 0;
                    p->bit++;
                    p->neg_data = (p->nav >> (p->bit & 31)) & 1u;
                }
            }
        }
        iq[2 * n] = (short) i_acc;
        iq[2 * n + 1] = (short) q_acc;
    }
    pack_elems(iq, 2 * (size_t) nsamp, sample_size, dst);
    free(s); free(iq);
    return GPSIQ_OK;
}

/* ------------------------------------------------------------------------- */
/* gps.c:2839-2865.  HackRF: every element goes into the current buffer, the buffer is
 * enqueued exactly when validLength reaches 262144 and a partial buffer is kept for
 * the next block (gps.c:2847-2856).  iqfile / Pluto: the whole block in one buffer,
 * enqueued once per block (gps.c:2860-2865). */
int oracle_chunk_plan(int sink_kind, size_t nelem, int nblocks, size_t *chunk_len, int max_chunks)
{
    int n = 0;
    size_t fill = 0;
    for (int b = 0; b < nblocks; b++) {
        if (sink_kind == GPSIQ_SINK_HACKRF) {
            for (size_t k = 0; k < nelem; k++) {
                if (++fill == GPSIQ_HACKRF_CHUNK) {
                    if (n >= max_chunks) return GPSIQ_E_RANGE;
                    chunk_len[n++] = fill;
                    fill = 0;
                }
            }
        } else {
            if (n >= max_chunks) return GPSIQ_E_RANGE;
            chunk_len[n++] = nelem;
        }
    }
    return n;
}

/* ------------------------------------------------------------------------- */
/* The float loop in closed form.  gps.c:2789-2792 and gps.c:2821-2826 advance their phases
 * with  x += c  in double precision.  While x stays inside one binade [2^(E-1), 2^E) it is
 * a multiple of that binade's ulp u, and x + c rounds to x + S with one constant
 * S = rnd(c/u)*u (ties go to even, and from the second addition inside the binade on the
 * parity is settled, so the tie case is constant too).  The trajectory is therefore
 * piecewise linear: a handful of pieces per carrier cycle / per code period, and the
 * state of sample n inside a piece is  (m0 + (n-n0)*dm) * 2^q  in exact integers.
 * walk() builds the pieces: real double additions at the piece edges (binade crossings,
 * the  -= 1023.0  and  -/+= 1.0  wraps: these ARE the reference's operations), an integer
 * jump over the steady run in between.  oracle_block_float_closed() then evaluates every
 * sample by piece lookup, with no per-sample recurrence; it must equal oracle_block_float
 * (and the reference) bit for bit, carried carr_phase included. */
typedef struct { long n0, len; int64_t m0, dm; int q; int wrapped; } fseg_t;
typedef struct { fseg_t *s; size_t n, cap; } fsegs_t;

static int seg_push(fsegs_t *v, long n0, long len, int64_t m0, int64_t dm, int q, int wrapped)
{
    if (v->n == v->cap) {
        size_t nc = v->cap ? 2 * v->cap : 1024;
        fseg_t *p = realloc(v->s, nc * sizeof *p);
        if (!p) return -1;
        v->s = p; v->cap = nc;
    }
    v->s[v->n++] = (fseg_t) {n0, len, m0, dm, q, wrapped};
    return 0;
}

/* x >= 0  ->  x = m * 2^q, m < 2^53 (m >= 2^52 unless x == 0) */
static void split(double x, int64_t *m, int *q)
{
    int e;
    double f = frexp(x, &e);
    *m = (int64_t) ldexp(f, 53);
    *q = x == 0.0 ? 0 : e - 53;
}

/* kind 0: code phase  (y = x + c; if (y >= 1023) y -= 1023, gps.c:2789-2792)
 * kind 1: carrier     (y = x + c; if (y >= 1) y -= 1; else if (y < 0) y += 1, gps.c:2821-2826) */
static double fstep(double x, double c, int kind, int *wrapped)
{
    double y = x + c;
    *wrapped = 0;
    if (kind == 0) {
        if (y >= (double) GPSIQ_CA_SEQ_LEN) { y -= (double) GPSIQ_CA_SEQ_LEN; *wrapped = 1; }
    } else {
        if (y >= 1.0) { y -= 1.0; *wrapped = 1; }
        else if (y < 0.0) { y += 1.0; *wrapped = 1; }
    }
    return y;
}

static int same_binade(double a, double b)
{
    int ea, eb;
    if (a <= 0.0 || b <= 0.0) return 0;
    frexp(a, &ea); frexp(b, &eb);
    return ea == eb;
}

/* states s_0 .. s_nsamp (s_n is what sample n uses; s_nsamp is what the loop leaves behind) */
static int walk(double x, double c, int kind, long nsamp, fsegs_t *v, double *x_end)
{
    long n = 0;
    int w0 = 0;                                   /* was s_n produced by a wrap */
    const double top = kind == 0 ? (double) GPSIQ_CA_SEQ_LEN : 1.0;
    while (n <= nsamp) {
        int wy, wz, ww;
        double y = fstep(x, c, kind, &wy), z = fstep(y, c, kind, &wz), w = fstep(z, c, kind, &ww);
        if (!wy && !wz && !ww && same_binade(y, z) && same_binade(z, w) && n + 2 <= nsamp) {
            /* steady from z on: s_{n+2+j} = z + j*S while the value stays inside z's binade and short of the wrap */
            int64_t mz, mw; int qz, qw;
            split(z, &mz, &qz); split(w, &mw, &qw);
            const int64_t dm = mw - mz;            /* same exponent: same binade */
            long J;
            if (dm == 0) J = nsamp;                /* the addend is below half an ulp: the phase stands still */
            else if (dm > 0) {
                int64_t lim = (int64_t) 1 << 53;    /* next binade */
                if (ldexp((double) lim, qz) > top) lim = (int64_t) ldexp(top, -qz);   /* the wrap comes first */
                J = (long) ((lim - 1 - mz) / dm);
            } else {
                /* downwards the run stays strictly above the binade's first value: a sum that falls below 2^52 ulps is
                 * rounded on the finer grid of the binade underneath, so that step is left to a real addition */
                J = mz > ((int64_t) 1 << 52) ? (long) ((mz - ((int64_t) 1 << 52) - 1) / -dm) : 0;
            }
            if (n + 2 + J > nsamp) J = nsamp - (n + 2);
            int64_t mx, my; int qx, qy;
            split(x, &mx, &qx); split(y, &my, &qy);
            if (seg_push(v, n, 1, mx, 0, qx, w0) || seg_push(v, n + 1, 1, my, 0, qy, 0) ||
                seg_push(v, n + 2, J + 1, mz, dm, qz, 0)) return -1;
            x = ldexp((double) (mz + (int64_t) J * dm), qz);
            n += 2 + J;
            w0 = 0;
            if (n == nsamp) { *x_end = x; return 0; }
            /* s_n is the last value of the run and is already covered: continue from its successor */
            x = fstep(x, c, kind, &w0);
            n++;
            continue;
        }
        int64_t mx; int qx;
        split(x, &mx, &qx);
        if (seg_push(v, n, 1, mx, 0, qx, w0)) return -1;
        if (n == nsamp) { *x_end = x; return 0; }
        x = y; w0 = wy; n++;
    }
    return 0;
}

/* floor(m * 2^(q + up)) for m >= 0 */
static int64_t seg_floor(int64_t m, int q, int up)
{
    int sh = -(q + up);
    if (sh <= 0) return m << -sh;
    return sh >= 63 ? 0 : m >> sh;
}

int oracle_block_float_closed(const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                              int sample_size, void *dst, double *carr_phase_out)
{
    if (!ch || !dst || nchan < 0 || nchan > GPSIQ_MAX_CHAN || nsamp < 0)
        return GPSIQ_E_ARG;
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16)
        return GPSIQ_E_ARG;
    const double delt = 1.0 / fs;                                    /* gps.c:2298 */
    int *acc = calloc(2 * (size_t) (nsamp ? nsamp : 1), sizeof *acc);
    short *iq = malloc(sizeof(short) * 2 * (size_t) (nsamp ? nsamp : 1));
    fsegs_t code = {0, 0, 0}, carr = {0, 0, 0};
    int rc = GPSIQ_OK;
    if (!acc || !iq) { rc = GPSIQ_E_NOMEM; goto out; }
    for (int c = 0; c < nchan; c++) {
        if (carr_phase_out) carr_phase_out[c] = ch[c].carr_phase;
        if (ch[c].prn <= 0) continue;                                 /* gps.c:2772 */
        uint8_t ca[GPSIQ_CA_SEQ_LEN];
        int tc[512], ts[512];
        if (oracle_codegen(ch[c].prn, ca) != GPSIQ_OK) { rc = GPSIQ_E_ARG; goto out; }
        gain_lut(ch[c].gain, tc, ts);        /* (int)(+-table*gain) = +-(int)(table*gain): truncation is odd */
        code.n = carr.n = 0;
        double code_end, carr_end;
        if (walk(ch[c].code_phase, ch[c].f_code * delt, 0, nsamp, &code, &code_end) ||
            walk(ch[c].carr_phase, ch[c].f_carr * delt, 1, nsamp, &carr, &carr_end)) { rc = GPSIQ_E_NOMEM; goto out; }
        if (carr_phase_out) carr_phase_out[c] = carr_end;
        int iword = ch[c].iword, ibit = ch[c].ibit, icode = ch[c].icode;
        int data = (int) ((ch[c].dwrd[iword] >> (29 - ibit)) & 1u);   /* gps.c:2059 */
        size_t ik = 0, ic = 0;
        for (long n = 0; n < nsamp; n++) {
            while (n >= code.s[ik].n0 + code.s[ik].len) ik++;
            while (n >= carr.s[ic].n0 + carr.s[ic].len) ic++;
            const fseg_t *sk = &code.s[ik], *sc = &carr.s[ic];
            if (sk->wrapped && n == sk->n0) {                         /* gps.c:2791-2812, done by the step that produced s_n */
                if (++icode >= 20) {
                    icode = 0;
                    if (++ibit >= 30) {
                        ibit = 0;
                        if (++iword >= GPSIQ_N_DWRD) { rc = GPSIQ_E_RANGE; goto out; }
                    }
                    data = (int) ((ch[c].dwrd[iword] >> (29 - ibit)) & 1u);
                }
            }
            const int chip = (int) seg_floor(sk->m0 + (int64_t) (n - sk->n0) * sk->dm, sk->q, 0);   /* gps.c:2817 */
            const int k = (int) seg_floor(sc->m0 + (int64_t) (n - sc->n0) * sc->dm, sc->q, 9);      /* gps.c:2775 */
            const int pos = ca[chip] == (unsigned) data;              /* dataBit*codeCA == +1 */
            acc[2 * n] += pos ? tc[k] : -tc[k];
            acc[2 * n + 1] += pos ? ts[k] : -ts[k];
        }
    }
    for (long k = 0; k < 2 * (long) nsamp; k++) iq[k] = (short) acc[k];   /* gps.c:2834-2835 */
    pack_elems(iq, 2 * (size_t) nsamp, sample_size, dst);
out:
    free(acc); free(iq); free(code.s); free(carr.s);
    return rc;
}
