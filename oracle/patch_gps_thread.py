#!/usr/bin/env python3
"""Apply INTEGRATION.md section 2 to a scratch copy of the reference's gps.c.

    python oracle/patch_gps_thread.py /root/reference/gps.c out.c

TEST INFRASTRUCTURE (used by oracle/Makefile for _ref/gps-sim-gpsiq).  Nothing of the reference is
stored here: the script holds only the binding code a maintainer would add, and the anchors (line
number + the text expected there) that say where it goes; a different revision of gps.c fails loudly.
The sample loop and its pack / fifo hand-off (gps.c:2767-2865) are cut out and replaced by one
gpsiq_generate_block() call plus the chunker; everything else of gps_thread_ep() is left as it is.
GPSIQ_NCO=fixed in the environment selects the fixed-point model, the default here is
GPSIQ_NCO_REFERENCE (the output then equals the unpatched program's byte for byte)."""
import sys

src, dst = sys.argv[1], sys.argv[2]
L = open(src).read().split("\n")


def expect(lineno, text):
    got = L[lineno - 1]
    if text not in got:
        sys.exit(f"patch_gps_thread: line {lineno} of {src} is {got!r}, expected to contain {text!r}")


expect(26, '#include "gps-sim.h"')
expect(2320, "short *iq_buff = NULL;")
expect(2695, "iq_buff = calloc(IQ_BUFFER_SIZE, 2);")
expect(2698, "struct iq_buf *iq = fifo_acquire();")
expect(2766, "")
expect(2767, "for (isamp = 0; isamp < NUM_IQ_SAMPLES; isamp++) {")
expect(2865, "        }")
expect(2868, "// Update navigation message and channel allocation every 30 seconds")
expect(2941, "free(iq_buff);")

INCLUDE = '''#include "gpsiq.h"                                   /* libgpsiq: the C-ABI that replaces gps.c:2767-2865 */
/* callbacks with the fifo.h signatures; gpsiq_iq_buf_t is field-for-field struct iq_buf (fifo.h:19-25) */
static gpsiq_iq_buf_t *gq_acquire(void *u) { (void) u; return (gpsiq_iq_buf_t *) fifo_acquire(); }
static void gq_enqueue(void *u, gpsiq_iq_buf_t *b) { (void) u; fifo_enqueue((struct iq_buf *) b); }'''

DECLS = '''    gpsiq_ctx_t *gq = NULL;
    gpsiq_chunker_t gq_ck;
    gpsiq_chan_t gq_ch[MAX_CHAN];
    double gq_carr[MAX_CHAN];
    void *gq_blk = NULL;'''

ALLOC = '''    gq_blk = gpsiq_host_alloc((size_t) IQ_BUFFER_SIZE * (size_t) simulator->sample_size);   /* page-locked staging for the HackRF chunking only: the block's copy off the device lands in it */'''

INIT = '''    {
        const char *gq_nco = getenv("GPSIQ_NCO");
        if (gq_blk == NULL || gpsiq_create(&gq, 0) != GPSIQ_OK ||
            gpsiq_set_nco_mode(gq, (gq_nco && !strcmp(gq_nco, "fixed")) ? GPSIQ_NCO_FIXED : GPSIQ_NCO_REFERENCE) != GPSIQ_OK ||
            gpsiq_chunker_init(&gq_ck, simulator->sdr_type, simulator->sample_size, gq_acquire, gq_enqueue, NULL) != GPSIQ_OK) {
            gui_status_wprintw(RED, "gpsiq: %s\\n", gpsiq_last_error());
            goto end_gps_thread;
        }
    }'''

CALL = '''        /* libgpsiq: one call synthesises the block (gps.c:2767-2846) ... */
        for (i = 0; i < MAX_CHAN; i++) {
            gq_ch[i].prn = chan[i].prn;
            if (chan[i].prn <= 0) continue;
            gq_ch[i].iword = chan[i].iword;
            gq_ch[i].ibit = chan[i].ibit;
            gq_ch[i].icode = chan[i].icode;
            gq_ch[i].f_carr = chan[i].f_carr;
            gq_ch[i].f_code = chan[i].f_code;
            gq_ch[i].carr_phase = chan[i].carr_phase;
            gq_ch[i].code_phase = chan[i].code_phase;
            gq_ch[i].gain = gain[i];
            for (int k = 0; k < N_DWRD; k++) gq_ch[i].dwrd[k] = (uint32_t) chan[i].dwrd[k];
        }
        {
            /* iqfile / Pluto: straight into the fifo buffer; HackRF: through the staging block */
            void *gq_where = gpsiq_chunker_reserve(&gq_ck, IQ_BUFFER_SIZE);
            if (gpsiq_generate_block(gq, gq_ch, MAX_CHAN, NUM_IQ_SAMPLES, (double) TX_SAMPLERATE,
                                     simulator->sample_size, gq_where ? gq_where : gq_blk, gq_carr) != GPSIQ_OK) {
                gui_status_wprintw(RED, "gpsiq: %s\\n", gpsiq_last_error());
                goto end_gps_thread;
            }
            for (i = 0; i < MAX_CHAN; i++)              /* what the loop leaves behind (gps.c:2821) */
                if (chan[i].prn > 0) chan[i].carr_phase = gq_carr[i];
            /* ... and the chunker hands it to the fifo (gps.c:2839-2865) */
            if ((gq_where ? gpsiq_chunker_commit(&gq_ck, IQ_BUFFER_SIZE)
                          : gpsiq_chunker_push(&gq_ck, gq_blk, IQ_BUFFER_SIZE)) < 0)
                break;                                  /* fifo halted */
        }'''

FREE = '''    gpsiq_destroy(gq);
    gpsiq_host_free(gq_blk);'''

out = []
for n, line in enumerate(L, 1):
    if n == 26:
        out += [line, INCLUDE]
    elif n == 2320:
        out += [line, DECLS]
    elif n == 2695:
        out.append(ALLOC)
    elif n == 2698:
        out.append(INIT)
    elif n == 2767:
        out.append(CALL)
    elif 2767 < n <= 2865:
        continue
    elif n == 2941:
        out += [FREE, line]
    else:
        out.append(line)
open(dst, "w").write("\n".join(out))
