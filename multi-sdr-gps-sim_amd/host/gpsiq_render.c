/*
 * gpsiq_render.c — a whole scenario on every GPU of the node from ONE process, host code in C.
 *
 * The reference is a single process (gps-sim.c:314 starts its one gps thread); the offline equivalent
 * of its block loop (gps.c:2703-2933) on a multi-GPU node is one call:
 *
 *   gpsiq_generate_batch_multi(ctx[ngpu], ...)     the timeline is quantised once (exact carrier prefix),
 *                                                  cut into ngpu contiguous block ranges, each rendered by
 *                                                  its own context from its own host thread; no traffic
 *                                                  between the devices; blocks land in timeline order.
 *
 *   gpsiq_render <descriptors.bin> <out.bin> [contexts [reference]]
 *
 * contexts: how many contexts to use (default: one per visible GPU); more contexts than GPUs are spread
 * round-robin (that is how the tests exercise the multi-context path on a one-GPU box).  "reference" selects
 * GPSIQ_NCO_REFERENCE (the reference's double NCOs, exactly).  descriptors.bin as for gpsiq_play; the output
 * is the iqfile stream (one block per buffer, sdr_iqfile.c:59).  The timeline is rendered in slices of
 * SLICE blocks so that the page-locked staging stays bounded; the carrier phase handed out by one call goes
 * into block 0 of the next (gps.c:2821).
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsiq.h"

struct play_header {
    char     magic[8];         /* "GPSIQD1" */
    uint32_t nblocks, nchan, sample_size, nsamp;
    double   fs;
};

#define SLICE 256
#define MAX_CTX 64

static int die(const char *what)
{
    fprintf(stderr, "gpsiq_render: %s: %s\n", what, gpsiq_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 3) { fprintf(stderr, "usage: %s descriptors.bin out.bin [contexts [reference]]\n", argv[0]); return 2; }
    FILE *fd = fopen(argv[1], "rb");
    struct play_header h;
    if (!fd || fread(&h, sizeof h, 1, fd) != 1 || memcmp(h.magic, "GPSIQD1", 8)) { fprintf(stderr, "bad descriptor file\n"); return 2; }
    if (h.nchan < 1 || h.nchan > GPSIQ_MAX_CHAN) { fprintf(stderr, "bad nchan\n"); return 2; }
    const size_t n = (size_t) h.nblocks * h.nchan;
    gpsiq_chan_t *desc = malloc(sizeof *desc * (n ? n : 1));
    if (!desc || fread(desc, sizeof *desc, n, fd) != n) { fprintf(stderr, "short descriptor file\n"); return 2; }
    fclose(fd);

    /* one context per GPU: gpsiq_create fails past the last device */
    gpsiq_ctx_t *gq[MAX_CTX];
    int ngpu = 0, nctx = argc > 3 ? atoi(argv[3]) : 0;
    while (ngpu < MAX_CTX && gpsiq_create(&gq[ngpu], ngpu) == GPSIQ_OK) ++ngpu;
    if (ngpu == 0) return die("no GPU");
    if (nctx <= 0) nctx = ngpu;
    if (nctx > MAX_CTX) nctx = MAX_CTX;
    for (int i = ngpu; i < nctx; ++i)
        if (gpsiq_create(&gq[i], i % ngpu) != GPSIQ_OK) return die("create");
    if (argc > 4 && !strcmp(argv[4], "reference") && gpsiq_set_nco_mode(gq[0], GPSIQ_NCO_REFERENCE) != GPSIQ_OK) return die("mode");

    const size_t blk_bytes = (size_t) 2 * h.nsamp * h.sample_size;
    void *buf = gpsiq_host_alloc(blk_bytes * SLICE);
    FILE *fo = fopen(argv[2], "wb");
    if (!buf || !fo) { fprintf(stderr, "cannot allocate / open output\n"); return 1; }
    double carr[GPSIQ_MAX_CHAN];
    int failed = 0;
    for (uint32_t b = 0; b < h.nblocks && !failed; b += SLICE) {
        const int nb = h.nblocks - b < SLICE ? (int) (h.nblocks - b) : SLICE;
        gpsiq_chan_t *d = desc + (size_t) b * h.nchan;
        const gpsiq_chan_t *before = b > 0 ? desc + (size_t) (b - 1) * h.nchan : NULL;
        for (uint32_t i = 0; before && i < h.nchan; ++i)           /* what the loop left behind, gps.c:2821 */
            if (d[i].prn > 0 && d[i].prn == before[i].prn) d[i].carr_phase = carr[i];
        if (gpsiq_generate_batch_multi(gq, nctx, d, nb, (int) h.nchan, (int) h.nsamp, h.fs, (int) h.sample_size,
                                       buf, NULL, carr) != GPSIQ_OK) { failed = die("generate"); break; }
        if (fwrite(buf, blk_bytes, (size_t) nb, fo) != (size_t) nb) failed = 1;
    }
    fclose(fo);
    printf("%u blocks on %d context(s) over %d GPU(s)\n", h.nblocks, nctx, ngpu);
    gpsiq_host_free(buf);
    for (int i = 0; i < (nctx > ngpu ? nctx : ngpu); ++i) gpsiq_destroy(gq[i]);
    free(desc);
    return failed;
}
