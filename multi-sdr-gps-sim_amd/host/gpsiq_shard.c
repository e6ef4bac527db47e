/*
 * gpsiq_shard.c — one rank of a time-sharded offline run, host code in C.
 *
 * The reference renders a scenario with one thread walking the 10 Hz block loop
 * (gps.c:2703-2933).  Blocks only depend on their own channel state plus the carrier phase
 * the loop carries (gps.c:2821), and that phase has an exact prefix in the fixed-point
 * model, so the timeline splits into contiguous block ranges, one per GPU, with no traffic
 * between them.  Every rank runs this program:
 *
 *   gpsiq_shard <descriptors.bin> <out.part> <rank> <world> [device]
 *
 * It quantises the WHOLE timeline on the host (cheap: threaded, ~1 us per block), takes its
 * own range, renders it on its GPU and writes that part of the iqfile stream (the iqfile
 * sink is one block per buffer, reference sdr_iqfile.c:59, so parts concatenate to the
 * file the single-thread run would have written).  descriptors.bin as for gpsiq_play.
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

#define BLOCKS_PER_CALL 64     /* bounds the page-locked staging buffer */

static int die(const char *what)
{
    fprintf(stderr, "gpsiq_shard: %s: %s\n", what, gpsiq_last_error());
    return 1;
}

int main(int argc, char **argv)
{
    if (argc < 5) { fprintf(stderr, "usage: %s descriptors.bin out.part rank world [device]\n", argv[0]); return 2; }
    const int rank = atoi(argv[3]), world = atoi(argv[4]);
    const int device = argc > 5 ? atoi(argv[5]) : rank;
    FILE *fd = fopen(argv[1], "rb");
    struct play_header h;
    if (!fd || fread(&h, sizeof h, 1, fd) != 1 || memcmp(h.magic, "GPSIQD1", 8)) { fprintf(stderr, "bad descriptor file\n"); return 2; }
    if (h.nchan < 1 || h.nchan > GPSIQ_MAX_CHAN) { fprintf(stderr, "bad nchan\n"); return 2; }
    const size_t n = (size_t) h.nblocks * h.nchan;
    gpsiq_chan_t *desc = malloc(sizeof *desc * (n ? n : 1));
    gpsiq_qchan_t *q = malloc(sizeof *q * (n ? n : 1));
    if (!desc || !q || fread(desc, sizeof *desc, n, fd) != n) { fprintf(stderr, "short descriptor file\n"); return 2; }
    fclose(fd);

    /* the whole timeline, so that this shard starts from the exact carried carrier phase */
    if (gpsiq_quantize_batch(desc, (int) h.nblocks, (int) h.nchan, h.fs, (int) h.nsamp, q, NULL, NULL) != GPSIQ_OK)
        return die("quantise");
    int b0, b1;
    if (gpsiq_shard_range((int) h.nblocks, rank, world, &b0, &b1) != GPSIQ_OK) return die("shard");

    gpsiq_ctx_t *gq = NULL;
    if (gpsiq_create(&gq, device) != GPSIQ_OK) return die("create");
    const size_t blk_bytes = (size_t) 2 * h.nsamp * h.sample_size;
    void *buf = gpsiq_host_alloc(blk_bytes * BLOCKS_PER_CALL);
    FILE *fo = fopen(argv[2], "wb");
    if (!buf || !fo) { fprintf(stderr, "cannot allocate / open output\n"); return 1; }
    int failed = 0;
    for (int b = b0; b < b1 && !failed; b += BLOCKS_PER_CALL) {
        const int nb = b1 - b < BLOCKS_PER_CALL ? b1 - b : BLOCKS_PER_CALL;
        if (gpsiq_generate_quantized(gq, q + (size_t) b * h.nchan, nb, (int) h.nchan, (int) h.nsamp,
                                     (int) h.sample_size, buf, 0) != GPSIQ_OK) { failed = die("generate"); break; }
        if (fwrite(buf, blk_bytes, (size_t) nb, fo) != (size_t) nb) failed = 1;
    }
    fclose(fo);
    printf("rank %d/%d: blocks [%d, %d) of %u on device %d\n", rank, world, b0, b1, h.nblocks, device);
    gpsiq_host_free(buf);
    gpsiq_destroy(gq);
    free(q);
    free(desc);
    return failed;
}
