/*
 * gpsiq_play.c — the reference's thread structure around libgpsiq, in C.
 *
 * A generator thread runs the 10 Hz block loop of gps_thread_ep() (reference gps.c:2703)
 * with the sample loop replaced by gpsiq_generate_block() and the hand-off of
 * gps.c:2839-2865 by gpsiq_chunker_push(); a sink thread drains the fifo.h FIFO into a
 * file the way the iqfile sink does (reference sdr_iqfile.c:22-47).  The per-block channel
 * state that the reference's host model would compute (gps.c:2731-2765) is read from a
 * descriptor file instead, so the program exercises exactly the path this repository
 * replaces, end to end, with host code in C.
 *
 *   gpsiq_play <descriptors.bin> <out.bin> [iqfile|hackrf|pluto]
 *
 * descriptors.bin: struct play_header, then gpsiq_chan_t[nblocks][nchan].
 */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "fifo.h"
#include "gpsiq.h"

struct play_header {
    char     magic[8];         /* "GPSIQD1" */
    uint32_t nblocks, nchan, sample_size, nsamp;
    double   fs;
};

#define NUM_FIFO_BUFFERS 8     /* reference sdr.h:24 */

static gpsiq_iq_buf_t *acq(void *u) { (void) u; return (gpsiq_iq_buf_t *) fifo_acquire(); }
static void enq(void *u, gpsiq_iq_buf_t *b) { (void) u; fifo_enqueue((struct iq_buf *) b); }

struct sink_arg { FILE *fp; unsigned sample_size; unsigned long long elems; int failed; };

static void *sink_thread(void *p)
{
    struct sink_arg *a = p;
    for (;;) {
        struct iq_buf *b = fifo_dequeue();               /* sdr_iqfile.c:39 */
        if (!b) break;
        const void *src = a->sample_size == GPSIQ_SC16 ? (const void *) b->data16 : (const void *) b->data8;
        if (fwrite(src, a->sample_size, b->validLength, a->fp) != b->validLength) a->failed = 1;
        a->elems += b->validLength;
        fifo_release(b);                                 /* sdr_iqfile.c:47 */
    }
    return NULL;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s descriptors.bin out.bin [iqfile|hackrf|pluto]\n", argv[0]); return 2; }
    int sink = GPSIQ_SINK_IQFILE;
    if (argc > 3) {
        if (!strcmp(argv[3], "hackrf")) sink = GPSIQ_SINK_HACKRF;
        else if (!strcmp(argv[3], "pluto")) sink = GPSIQ_SINK_PLUTOSDR;
        else if (strcmp(argv[3], "iqfile")) { fprintf(stderr, "unknown sink %s\n", argv[3]); return 2; }
    }
    FILE *fd = fopen(argv[1], "rb");
    struct play_header h;
    if (!fd || fread(&h, sizeof h, 1, fd) != 1 || memcmp(h.magic, "GPSIQD1", 8)) { fprintf(stderr, "bad descriptor file\n"); return 2; }
    if (h.nchan < 1 || h.nchan > GPSIQ_MAX_CHAN) { fprintf(stderr, "bad nchan\n"); return 2; }
    gpsiq_chan_t *desc = malloc(sizeof *desc * (size_t) h.nblocks * h.nchan);
    if (!desc || fread(desc, sizeof *desc, (size_t) h.nblocks * h.nchan, fd) != (size_t) h.nblocks * h.nchan) { fprintf(stderr, "short descriptor file\n"); return 2; }
    fclose(fd);

    gpsiq_ctx_t *gq = NULL;
    if (gpsiq_create(&gq, 0) != GPSIQ_OK) { fprintf(stderr, "gpsiq: %s\n", gpsiq_last_error()); return 1; }

    const unsigned block_elems = 2 * h.nsamp;                                   /* IQ_BUFFER_SIZE, sdr.h:29 */
    const unsigned buf_elems = sink == GPSIQ_SINK_HACKRF ? GPSIQ_HACKRF_CHUNK : block_elems;  /* sdr_hackrf.c:215 / sdr_iqfile.c:59 */
    fifo_set_allocator(gpsiq_host_alloc, gpsiq_host_free);                      /* page-locked fifo buffers */
    if (!fifo_create(NUM_FIFO_BUFFERS, buf_elems, h.sample_size)) { fprintf(stderr, "fifo_create failed\n"); return 1; }

    struct sink_arg sa = { fopen(argv[2], "wb"), h.sample_size, 0, 0 };
    if (!sa.fp) { perror(argv[2]); return 1; }
    pthread_t st;
    pthread_create(&st, NULL, sink_thread, &sa);

    /* ---- the generator loop (what gps_thread_ep does between gps.c:2698 and gps.c:2933) ---- */
    gpsiq_chunker_t ck;
    void *blk = gpsiq_host_alloc((size_t) block_elems * h.sample_size);
    double carr[GPSIQ_MAX_CHAN];
    int rc = gpsiq_chunker_init(&ck, sink, (int) h.sample_size, acq, enq, NULL);
    gpsiq_chan_t cur[GPSIQ_MAX_CHAN];
    for (uint32_t b = 0; rc == GPSIQ_OK && b < h.nblocks; ++b) {
        memcpy(cur, desc + (size_t) b * h.nchan, sizeof *cur * h.nchan);
        for (uint32_t i = 0; b > 0 && i < h.nchan; ++i)          /* chan[i].carr_phase persists (gps.c:2821) */
            if (cur[i].prn > 0 && cur[i].prn == desc[(size_t) (b - 1) * h.nchan + i].prn)
                cur[i].carr_phase = carr[i];
        /* iqfile / Pluto take one block per buffer: synthesise straight into the page-locked fifo
           buffer; HackRF's 262144-element chunks (gps.c:2847-2856) go through the staging block */
        void *inplace = gpsiq_chunker_reserve(&ck, block_elems);
        rc = gpsiq_generate_block(gq, cur, (int) h.nchan, (int) h.nsamp, h.fs, (int) h.sample_size, inplace ? inplace : blk, carr);
        if (rc != GPSIQ_OK) break;
        int n = inplace ? gpsiq_chunker_commit(&ck, block_elems) : gpsiq_chunker_push(&ck, blk, block_elems);
        if (n < 0) rc = n;
    }
    if (rc != GPSIQ_OK) fprintf(stderr, "gpsiq: %s\n", gpsiq_last_error());

    fifo_wait_next();            /* let the sink drain what was enqueued */
    fifo_halt();
    pthread_join(st, NULL);
    fclose(sa.fp);
    printf("%u blocks, %llu elements written, sink %d\n", h.nblocks, sa.elems, sink);
    gpsiq_host_free(blk);
    fifo_destroy();
    gpsiq_destroy(gq);
    free(desc);
    return (rc != GPSIQ_OK || sa.failed) ? 1 : 0;
}
