/*
 * fifo.c — drop-free bounded block FIFO with the reference's fifo.h API (see fifo.h).
 * One mutex, four conditions, all waits in `while (predicate)` loops.
 */
#include "fifo.h"

#include <pthread.h>
#include <stdlib.h>
#include <string.h>

static pthread_mutex_t lock = PTHREAD_MUTEX_INITIALIZER;
static pthread_cond_t  cv_free = PTHREAD_COND_INITIALIZER;     /* a buffer became free */
static pthread_cond_t  cv_data = PTHREAD_COND_INITIALIZER;     /* a buffer was queued */
static pthread_cond_t  cv_full = PTHREAD_COND_INITIALIZER;     /* all buffers are queued */
static pthread_cond_t  cv_empty = PTHREAD_COND_INITIALIZER;    /* the queue drained */

static struct iq_buf *q_head, *q_tail;   /* filled buffers, oldest first */
static struct iq_buf *free_list;
static struct iq_buf *all_bufs;          /* array of every buffer, for destroy */
static unsigned n_bufs, n_queued;
static bool halted;

static void *(*alloc_fn)(size_t) = NULL;
static void (*free_fn)(void *) = NULL;

void fifo_set_allocator(void *(*a)(size_t), void (*f)(void *))
{
    alloc_fn = a;
    free_fn = f;
}

static void *buf_alloc(size_t bytes)
{
    void *p = alloc_fn ? alloc_fn(bytes) : malloc(bytes);
    if (p) memset(p, 0, bytes);
    return p;
}

static void buf_free(void *p)
{
    if (!p) return;
    if (free_fn) free_fn(p); else free(p);
}

bool fifo_create(unsigned buffer_count, unsigned buffer_size, unsigned sample_size)
{
    pthread_mutex_lock(&lock);
    bool ok = false;
    if (all_bufs || buffer_count == 0) goto out;
    all_bufs = calloc(buffer_count, sizeof *all_bufs);
    if (!all_bufs) goto out;
    n_bufs = buffer_count;
    n_queued = 0;
    q_head = q_tail = free_list = NULL;
    halted = false;
    ok = true;
    for (unsigned i = 0; i < buffer_count; ++i) {
        struct iq_buf *b = &all_bufs[i];
        if (sample_size == sizeof(signed short))
            b->data16 = buf_alloc((size_t) buffer_size * sizeof(signed short));
        else
            b->data8 = buf_alloc((size_t) buffer_size);
        if (!b->data16 && !b->data8) ok = false;
        b->totalLength = buffer_size;
        b->next = free_list;
        free_list = b;
    }
    if (!ok) {
        for (unsigned i = 0; i < buffer_count; ++i) { buf_free(all_bufs[i].data8); buf_free(all_bufs[i].data16); }
        free(all_bufs);
        all_bufs = NULL; free_list = NULL; n_bufs = 0;
    }
out:
    pthread_mutex_unlock(&lock);
    return ok;
}

void fifo_destroy(void)
{
    pthread_mutex_lock(&lock);
    for (unsigned i = 0; i < n_bufs; ++i) { buf_free(all_bufs[i].data8); buf_free(all_bufs[i].data16); }
    free(all_bufs);
    all_bufs = NULL; q_head = q_tail = free_list = NULL;
    n_bufs = n_queued = 0;
    pthread_mutex_unlock(&lock);
}

void fifo_halt(void)
{
    pthread_mutex_lock(&lock);
    halted = true;
    while (q_head) {                       /* nothing more will be consumed */
        struct iq_buf *b = q_head;
        q_head = b->next;
        b->next = free_list;
        free_list = b;
    }
    q_tail = NULL;
    n_queued = 0;
    pthread_cond_broadcast(&cv_free);
    pthread_cond_broadcast(&cv_data);
    pthread_cond_broadcast(&cv_full);
    pthread_cond_broadcast(&cv_empty);
    pthread_mutex_unlock(&lock);
}

void fifo_wait_full(void)
{
    pthread_mutex_lock(&lock);
    while (!halted && n_queued < n_bufs)
        pthread_cond_wait(&cv_full, &lock);
    pthread_mutex_unlock(&lock);
}

void fifo_wait_next(void)
{
    pthread_mutex_lock(&lock);
    while (!halted && n_queued > 0)
        pthread_cond_wait(&cv_empty, &lock);
    pthread_mutex_unlock(&lock);
}

struct iq_buf *fifo_acquire(void)
{
    struct iq_buf *b = NULL;
    pthread_mutex_lock(&lock);
    while (!halted && !free_list)
        pthread_cond_wait(&cv_free, &lock);
    if (!halted) {
        b = free_list;
        free_list = b->next;
        b->next = NULL;
        b->validLength = 0;
    }
    pthread_mutex_unlock(&lock);
    return b;
}

void fifo_enqueue(struct iq_buf *b)
{
    if (!b) return;
    pthread_mutex_lock(&lock);
    if (halted) {
        b->next = free_list;
        free_list = b;
        pthread_cond_signal(&cv_free);
    } else {
        b->next = NULL;
        if (q_tail) q_tail->next = b; else q_head = b;
        q_tail = b;                          /* the step the reference's fifo.c:166-168 lacks */
        if (++n_queued == n_bufs) pthread_cond_broadcast(&cv_full);
        pthread_cond_signal(&cv_data);
    }
    pthread_mutex_unlock(&lock);
}

struct iq_buf *fifo_dequeue(void)
{
    struct iq_buf *b = NULL;
    pthread_mutex_lock(&lock);
    while (!halted && !q_head)
        pthread_cond_wait(&cv_data, &lock);
    if (!halted) {
        b = q_head;
        q_head = b->next;
        if (!q_head) q_tail = NULL;
        b->next = NULL;
        if (--n_queued == 0) pthread_cond_broadcast(&cv_empty);
    }
    pthread_mutex_unlock(&lock);
    return b;
}

void fifo_release(struct iq_buf *b)
{
    if (!b) return;
    pthread_mutex_lock(&lock);
    b->next = free_list;
    free_list = b;
    pthread_cond_signal(&cv_free);
    pthread_mutex_unlock(&lock);
}
