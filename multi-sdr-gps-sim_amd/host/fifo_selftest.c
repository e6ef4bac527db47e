/*
 * fifo_selftest.c — stress test of host/fifo.c (no GPU): a fast producer and a jittery
 * consumer push numbered blocks through an 8-buffer FIFO; every block must arrive exactly
 * once and in order (the reference's FIFO loses blocks here),
 * fifo_wait_full must not miss its wake-up, and fifo_halt must release both sides.
 */
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "fifo.h"

enum { NBUF = 8, BUFLEN = 4096 };
static int nblocks = 20000;
static int bad;

static void nap(unsigned *seed, int max_us)
{
    *seed = *seed * 1103515245u + 12345u;
    struct timespec ts = {0, (long) ((*seed >> 8) % (unsigned) (max_us + 1)) * 1000L};
    if (ts.tv_nsec) nanosleep(&ts, NULL);
}

static void *producer(void *arg)
{
    (void) arg;
    for (int i = 0; i < nblocks; ++i) {
        struct iq_buf *b = fifo_acquire();
        if (!b) { ++bad; return NULL; }
        if (b->validLength != 0) ++bad;                 /* acquire resets the fill level */
        for (int k = 0; k < BUFLEN; ++k) b->data16[k] = (short) (i + k);
        b->validLength = BUFLEN;
        fifo_enqueue(b);
    }
    return NULL;
}

static void *consumer(void *arg)
{
    unsigned seed = 7;
    int expect = 0;
    (void) arg;
    for (;;) {
        struct iq_buf *b = fifo_dequeue();
        if (!b) break;
        if (b->validLength != BUFLEN) ++bad;
        for (int k = 0; k < BUFLEN; k += 97)
            if (b->data16[k] != (short) (expect + k)) { ++bad; break; }
        ++expect;
        fifo_release(b);
        if ((expect & 63) == 0) nap(&seed, 200);       /* slower than the producer now and then */
    }
    if (expect != nblocks) { fprintf(stderr, "consumer saw %d of %d blocks\n", expect, nblocks); ++bad; }
    return NULL;
}

static void *blocked_acquire(void *arg) { (void) arg; return fifo_acquire(); }
static void *blocked_dequeue(void *arg) { (void) arg; return fifo_dequeue(); }

int main(int argc, char **argv)
{
    if (argc > 1) nblocks = atoi(argv[1]);
    if (!fifo_create(NBUF, BUFLEN, sizeof(short))) return 2;
    pthread_t p, c;
    pthread_create(&p, NULL, producer, NULL);
    fifo_wait_full();                                   /* sdr_*_run pattern: prefill, then start the sink */
    pthread_create(&c, NULL, consumer, NULL);
    pthread_join(p, NULL);
    fifo_wait_next();
    fifo_halt();
    pthread_join(c, NULL);
    if (fifo_acquire() != NULL) ++bad;                  /* halted: producers get NULL */
    if (fifo_dequeue() != NULL) ++bad;
    fifo_destroy();

    /* halt while both sides are blocked */
    if (!fifo_create(2, 16, 1)) return 2;
    struct iq_buf *a = fifo_acquire(), *b2 = fifo_acquire();
    if (!a || !b2 || !a->data8 || a->data16) ++bad;
    pthread_create(&p, NULL, blocked_acquire, NULL);   /* blocks: no free buffer */
    pthread_create(&c, NULL, blocked_dequeue, NULL);   /* blocks: nothing queued */
    struct timespec ts = {0, 50 * 1000 * 1000};
    nanosleep(&ts, NULL);
    fifo_halt();
    void *r1, *r2;
    pthread_join(p, &r1);
    pthread_join(c, &r2);
    if (r1 || r2) ++bad;
    fifo_enqueue(a);                                    /* after halt: straight back to the free list */
    fifo_release(b2);
    fifo_destroy();
    printf(bad ? "FAIL (%d)\n" : "ok\n", bad);
    return bad ? 1 : 0;
}
