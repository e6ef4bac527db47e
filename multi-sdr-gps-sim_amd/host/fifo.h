/*
 * fifo.h — the IQ block FIFO between the generator thread and an SDR sink thread.
 *
 * API-compatible with the reference's fifo.h (Mictronics/multi-sdr-gps-sim fifo.h:19-63):
 * same struct layout, same nine functions, same meaning of totalLength / validLength
 * (counted in IQ ELEMENTS, not complex samples), so the reference's sdr_iqfile.c /
 * sdr_hackrf.c / sdr_pluto.c compile against it unchanged.  The implementation (fifo.c) is
 * new: every wait is predicate-guarded, the queue is a real FIFO (the reference never
 * advances its tail pointer, fifo.c:166-168, and drops blocks whenever the producer is
 * ahead — SURVEY.md section 0 fact 6), and buffers can be page-locked so the GPU's
 * device-to-host copy lands in them directly.
 */
#ifndef GPSIQ_FIFO_H
#define GPSIQ_FIFO_H

#include <stdbool.h>
#include <stddef.h>
#include <stdint.h>

struct iq_buf {
    signed char  *data8;        /* 8-bit IQ elements, or NULL */
    signed short *data16;       /* 16-bit IQ elements, or NULL */
    unsigned int  totalLength;  /* capacity in elements */
    unsigned int  validLength;  /* filled elements */
    struct iq_buf *next;
};

/* Optional: allocate buffer storage with these instead of calloc/free (e.g. page-locked
 * memory from gpsiq_host_alloc).  Call before fifo_create. */
void fifo_set_allocator(void *(*alloc_fn)(size_t bytes), void (*free_fn)(void *p));

bool fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned sample_size);
void fifo_destroy(void);
void fifo_wait_next(void);      /* until the queue is empty (or halted) */
void fifo_wait_full(void);      /* until every buffer is queued (or halted) */
void fifo_halt(void);           /* wake everyone, queued buffers go back to the free list */
struct iq_buf *fifo_acquire(void);              /* producer: a free buffer, NULL once halted */
void fifo_enqueue(struct iq_buf *buf);          /* producer: hand a filled buffer over */
struct iq_buf *fifo_dequeue(void);              /* consumer: oldest filled buffer, NULL once halted */
void fifo_release(struct iq_buf *buf);          /* consumer: give it back */

#endif
