/*
 * gpsiq_shard.c — one rank of a time-sharded offline run, host code in C.
 *
 * The reference renders a scenario with one thread walking the 10 Hz block loop
 * (gps.c:2703-2933).  Blocks only depend on their own channel state plus the carrier phase
 * the loop carries (gps.c:2821), and that phase has an exact prefix in the fixed-point
 * model, so the timeline splits into contiguous block ranges, one per GPU, with no traffic
 * between them.  Every rank runs this program:
 *
 *   gpsiq_shard <descriptors.bin> <out.part> <rank> <world> [device [reference]]
 *
 * It quantises the WHOLE timeline on the host (cheap: threaded, ~1 us per block), takes its
 * own range, renders it on its GPU and writes that part of the iqfile stream (the iqfile
 * sink is one block per buffer, reference sdr_iqfile.c:59, so parts concatenate to the
 * file the single-thread run would have written).  descriptors.bin as for gpsiq_play.
 *
 * With "reference" the rank renders its range in GPSIQ_NCO_REFERENCE (the reference's own double accumulators: the bytes the
 * reference program writes).  The only thing serial in time there is the carrier chain (gps.c:2821-2826): the rank walks it over
 * the whole timeline on the host (gpsiq_reference_chain, ~2.5 us per block and channel, a thread per channel; ranks that can
 * talk walk a few channels each and exchange 8 bytes per channel and block instead, gpsiq/shard.py) and hands its own
 * blocks' start states to gpsiq_generate_seeded, which evaluates and renders them with no reference to the blocks before.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "gpsiq.h"
#include "gpsiq_plumbing.h"      /* the chain on its own: not part of the boundary, reached through gpsiq_plumbing() */

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
    const int reference = argc > 6 && !strcmp(argv[6], "reference");
    FILE *fd = fopen(argv[1], "rb");
    struct play_header h;
    if (!fd || fread(&h, sizeof h, 1, fd) != 1 || memcmp(h.magic, "GPSIQD1", 8)) { fprintf(stderr, "bad descriptor file\n"); return 2; }
    if (h.nchan < 1 || h.nchan > GPSIQ_MAX_CHAN) { fprintf(stderr, "bad nchan\n"); return 2; }
    const size_t n = (size_t) h.nblocks * h.nchan;
    gpsiq_chan_t *desc = malloc(sizeof *desc * (n ? n : 1));
    gpsiq_qchan_t *q = malloc(sizeof *q * (n ? n : 1));
    if (!desc || !q || fread(desc, sizeof *desc, n, fd) != n) { fprintf(stderr, "short descriptor file\n"); return 2; }
    fclose(fd);

    double *start = NULL;
    if (reference) {
        /* the chain over the whole timeline: the double every channel's accumulator holds at the start of every block */
        gpsiq_chain_in_t *cin = malloc(sizeof *cin * (n ? n : 1));
        start = malloc(sizeof *start * (n ? n : 1));
        if (!cin || !start) { fprintf(stderr, "out of memory\n"); return 1; }
        gpsiq_p_chain_inputs(desc, (int) n, cin);
        if (gpsiq_p_reference_chain(cin, (int) h.nblocks, (int) h.nchan, h.fs, (int) h.nsamp, NULL, NULL, start, NULL, NULL) != GPSIQ_OK)
            return die("chain");
        free(cin);
    } else if (gpsiq_quantize_batch(desc, (int) h.nblocks, (int) h.nchan, h.fs, (int) h.nsamp, q, NULL, NULL) != GPSIQ_OK) {
        /* the whole timeline, so that this shard starts from the exact carried carrier phase */
        return die("quantise");
    }
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
        const int rc = reference
            ? gpsiq_p_generate_seeded(gq, desc + (size_t) b * h.nchan, nb, (int) h.nchan, (int) h.nsamp, h.fs, (int) h.sample_size,
                                    start + (size_t) b * h.nchan, buf, 0)
            : gpsiq_generate_quantized(gq, q + (size_t) b * h.nchan, nb, (int) h.nchan, (int) h.nsamp, (int) h.sample_size, buf, 0);
        if (rc != GPSIQ_OK) { failed = die("generate"); break; }
        if (fwrite(buf, blk_bytes, (size_t) nb, fo) != (size_t) nb) failed = 1;
    }
    fclose(fo);
    printf("rank %d/%d: blocks [%d, %d) of %u on device %d%s\n", rank, world, b0, b1, h.nblocks, device, reference ? ", GPSIQ_NCO_REFERENCE" : "");
    free(start);
    gpsiq_host_free(buf);
    gpsiq_destroy(gq);
    free(q);
    free(desc);
    return failed;
}
