// gpsiq_host.cpp — host-side half of the C-ABI in include/gpsiq.h: tables, the
// double -> fixed-point descriptor quantiser, and the fifo hand-off rules.  No device
// code here; no dependency on anything under oracle/.
#include "gpsiq_internal.h"

#include <cmath>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <pthread.h>
#include <sched.h>
#include <time.h>
#include <unistd.h>
#include <vector>

namespace gpsiq {

static thread_local char g_err[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

int set_error(int code, const char *text)          // fail() for a text formatted elsewhere (libgpsiq_rows.so, through gpsiq_plumbing)
{
    snprintf(g_err, sizeof g_err, "%s", text ? text : "");
    return code;
}

// ---- the host worker pool ---------------------------------------------------------------------------------
// Spawning threads per call costs more than the work of a 4000-block batch on a 256-core host (measured: 65
// pthread_create/join ~ 2.4 ms vs 0.6 ms of quantising), so the workers are created once, on first use, and sleep on a
// condition variable between jobs.  A JOB is a function that any number of threads may run at the same time until it
// returns (it hands out its own work: an atomic chunk counter in parallel_for, the task scheduler of RefWalk); submitting
// one puts `helpers` tickets on a list, every idle worker takes a ticket of the oldest job that still has one, and the
// submitting thread runs the function too -- so a job always makes progress, however busy the pool is, and SEVERAL JOBS
// SHARE THE POOL (two contexts walking a reference-NCO timeline on two host threads, a batched refresh running ahead
// while a walk is in flight): none of them falls back to one thread, and together they never run more threads than the
// pool has.  Its size is the number of online CPUs, capped by GPSIQ_THREADS (several ranks of a time-sharded run share one
// host): the cap holds for every caller, also for those that ask for a thread count of their own.  The workers are
// detached and never exit; the library is not meant to be dlclose()d.
namespace {
constexpr int kMaxWorkers = 63;
struct Job {
    void (*entry)(void *);
    void  *ctx;
    int    tickets;        // helpers still wanted
    int    running;        // helpers inside entry()
    Job   *next;
};
struct Pool {
    pthread_mutex_t m;
    pthread_cond_t  go, done;
    int             nworkers;
    Job            *head, *tail;         // jobs with tickets left, oldest first
};
Pool g_pool = {PTHREAD_MUTEX_INITIALIZER, PTHREAD_COND_INITIALIZER, PTHREAD_COND_INITIALIZER, 0, nullptr, nullptr};

void *pool_worker(void *)
{
    Pool &p = g_pool;
    pthread_mutex_lock(&p.m);
    for (;;) {
        while (!p.head) pthread_cond_wait(&p.go, &p.m);
        Job *j = p.head;
        __atomic_add_fetch(&j->running, 1, __ATOMIC_RELAXED);
        if (--j->tickets == 0) { p.head = j->next; if (!p.head) p.tail = nullptr; }
        pthread_mutex_unlock(&p.m);
        j->entry(j->ctx);
        pthread_mutex_lock(&p.m);
        if (__atomic_sub_fetch(&j->running, 1, __ATOMIC_RELEASE) == 0 && j->tickets == 0) pthread_cond_broadcast(&p.done);
    }
    return nullptr;
}

void pool_after_fork_in_child()          // the child has none of the parent's threads
{
    Pool &p = g_pool;
    pthread_mutex_init(&p.m, nullptr);
    pthread_cond_init(&p.go, nullptr);
    pthread_cond_init(&p.done, nullptr);
    p.nworkers = 0;
    p.head = p.tail = nullptr;
}
}  // namespace

// CPUs this process may really use: the online count, cut to the scheduler affinity mask and to the container's CPU
// bandwidth limit (cgroup v2 cpu.max, cgroup v1 cpu.cfs_quota_us / cpu.cfs_period_us) -- a box with 256 hardware threads and
// a 16-CPU quota runs 16 threads' worth of work, and 64 runnable threads there only take turns.
static int granted_cpus()
{
    long c = sysconf(_SC_NPROCESSORS_ONLN);
    int n = c > 0 ? (int) c : 1;
    cpu_set_t set;
    if (sched_getaffinity(0, sizeof set, &set) == 0) { const int a = CPU_COUNT(&set); if (a > 0 && a < n) n = a; }
    long long quota = -1, period = -1;
    if (FILE *f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[32] = "";
        if (std::fscanf(f, "%31s %lld", q, &period) == 2 && std::strcmp(q, "max") != 0) quota = std::atoll(q);
        std::fclose(f);
    } else {
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) { if (std::fscanf(g, "%lld", &quota) != 1) quota = -1; std::fclose(g); }
        if (FILE *g = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (std::fscanf(g, "%lld", &period) != 1) period = -1; std::fclose(g); }
    }
    if (quota > 0 && period > 0) { const int lim = (int) ((quota + period - 1) / period); if (lim >= 1 && lim < n) n = lim; }
    return n;
}

int host_threads()
{
    static const int n = [] {
        const char *e = std::getenv("GPSIQ_THREADS");
        const int cap = e ? std::atoi(e) : 0;
        int g = granted_cpus();
        if (cap > 0 && g > cap) g = cap;
        if (g > kMaxWorkers + 1) g = kMaxWorkers + 1;
        return g < 1 ? 1 : g;
    }();
    return n;
}

void run_job(void (*entry)(void *), void *ctx, int nthreads)
{
    const int lim = host_threads();
    if (nthreads <= 0 || nthreads > lim) nthreads = lim;
    if (nthreads <= 1) { entry(ctx); return; }
    Pool &p = g_pool;
    Job job = {entry, ctx, 0, 0, nullptr};
    pthread_mutex_lock(&p.m);
    static bool atfork_set = false;
    if (!atfork_set) { pthread_atfork(nullptr, nullptr, pool_after_fork_in_child); atfork_set = true; }
    while (p.nworkers < nthreads - 1) {
        pthread_t th;
        pthread_attr_t at;
        pthread_attr_init(&at);
        pthread_attr_setdetachstate(&at, PTHREAD_CREATE_DETACHED);
        const int rc = pthread_create(&th, &at, pool_worker, nullptr);
        pthread_attr_destroy(&at);
        if (rc != 0) break;
        ++p.nworkers;
    }
    job.tickets = nthreads - 1 < p.nworkers ? nthreads - 1 : p.nworkers;
    if (job.tickets > 0) {
        if (p.tail) p.tail->next = &job; else p.head = &job;
        p.tail = &job;
        // one wake-up per helper wanted: a broadcast to fifteen sleepers for a job that wants two is ~60 us of futex calls on
        // the submitting thread (GPSIQ_TRACE=2 inside gpsiq_generate_batch: a 141-block piece quantised in 0.12 ms instead of 0.03)
        if (job.tickets >= p.nworkers) pthread_cond_broadcast(&p.go);
        else for (int t = 0; t < job.tickets; ++t) pthread_cond_signal(&p.go);
    }
    pthread_mutex_unlock(&p.m);
    entry(ctx);
    pthread_mutex_lock(&p.m);
    if (job.tickets > 0) {                                 // helpers that never came: take the job off the list
        job.tickets = 0;
        Job **pp = &p.head, *prev = nullptr;
        while (*pp && *pp != &job) { prev = *pp; pp = &(*pp)->next; }
        if (*pp) { *pp = job.next; if (p.tail == &job) p.tail = prev; }
    }
    // helpers still inside entry() are microseconds from leaving it (the work is handed out in chunks and none is left): poll
    // for them before sleeping -- a sleep here costs this thread a wake-up latency of its own
    if (__atomic_load_n(&job.running, __ATOMIC_ACQUIRE)) {
        pthread_mutex_unlock(&p.m);
        for (int i = 0; i < 4000 && __atomic_load_n(&job.running, __ATOMIC_ACQUIRE); ++i) {
#if defined(__x86_64__)
            __builtin_ia32_pause();
#endif
        }
        pthread_mutex_lock(&p.m);
    }
    while (__atomic_load_n(&job.running, __ATOMIC_ACQUIRE)) pthread_cond_wait(&p.done, &p.m);
    pthread_mutex_unlock(&p.m);
}

namespace {
struct Chunks { void (*fn)(void *, int, int); void *ctx; long n, chunk, next; };
void run_chunks(void *arg)
{
    Chunks &c = *static_cast<Chunks *>(arg);
    for (;;) {
        const long b = __atomic_fetch_add(&c.next, c.chunk, __ATOMIC_RELAXED);
        if (b >= c.n) break;
        const long e = b + c.chunk < c.n ? b + c.chunk : c.n;
        c.fn(c.ctx, (int) b, (int) e);
    }
}
}  // namespace

void parallel_for(int n, int nthreads, int grain, void (*fn)(void *, int, int), void *ctx)
{
    if (n <= 0) return;
    if (grain < 1) grain = 1;
    const int lim = host_threads();
    if (nthreads <= 0) {
        nthreads = lim;
        const int useful = n / grain + 1;
        if (nthreads > useful) nthreads = useful;
    }
    if (nthreads > lim) nthreads = lim;                     // GPSIQ_THREADS holds for explicit counts too
    if (nthreads > n) nthreads = n;
    if (nthreads <= 1) { fn(ctx, 0, n); return; }
    const long per = ((long) n + (long) nthreads * 4 - 1) / ((long) nthreads * 4);
    Chunks c = {fn, ctx, n, per > grain ? per : grain, 0};
    run_job(run_chunks, &c, nthreads);
}

// C/A code of one PRN (replaces codegen(), reference gps.c:272-309): G1 = x^10+x^3+1,
// G2 = x^10+x^9+x^8+x^6+x^3+x^2+1, both all-ones at chip 0, chip = G1[i] ^ G2[i-delay].
void ca_code(int prn, uint8_t chips[GPSIQ_CA_SEQ_LEN])
{
    uint8_t g1[GPSIQ_CA_SEQ_LEN], g2[GPSIQ_CA_SEQ_LEN];
    uint32_t a = 0x3ff, b = 0x3ff;
    for (int i = 0; i < GPSIQ_CA_SEQ_LEN; ++i) {
        g1[i] = (a >> 9) & 1u;
        g2[i] = (b >> 9) & 1u;
        uint32_t fa = __builtin_parity(a & 0x204u);   // stages 3, 10
        uint32_t fb = __builtin_parity(b & 0x3a6u);   // stages 2, 3, 6, 8, 9, 10
        a = ((a << 1) | fa) & 0x3ffu;
        b = ((b << 1) | fb) & 0x3ffu;
    }
    int lag = kG2Delay[prn - 1];
    for (int i = 0; i < GPSIQ_CA_SEQ_LEN; ++i) {
        int j = i - lag;
        if (j < 0) j += GPSIQ_CA_SEQ_LEN;
        chips[i] = g1[i] ^ g2[j];
    }
}

void build_device_tables(DeviceTables *t)
{
    std::memset(t, 0, sizeof *t);
    for (int prn = 1; prn <= 32; ++prn) {
        uint8_t ca[GPSIQ_CA_SEQ_LEN];
        ca_code(prn, ca);
        for (int j = 0; j < kPrnExtWords * 32; ++j)
            if (ca[j % GPSIQ_CA_SEQ_LEN])
                t->prn_ext[prn - 1][j >> 5] |= 1u << (j & 31);
    }
    std::memcpy(t->quarter_wave, kQuarterWave, sizeof kQuarterWave);
}

static const uint64_t kCarrMask = (UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1;
static const uint64_t kCodeMask = (UINT64_C(1) << GPSIQ_CODE_FRAC_BITS) - 1;

uint64_t carr_phase_to_fixed(double cycles)
{
    return (uint64_t) std::floor(std::ldexp(cycles, GPSIQ_CARR_FRAC_BITS)) & kCarrMask;
}

double carr_phase_to_double(uint64_t fixed)
{
    return std::ldexp((double) (fixed & kCarrMask), -GPSIQ_CARR_FRAC_BITS);
}

// One channel, one block.  Returns GPSIQ_OK or an error with text set.
int quantize_one(const gpsiq_chan_t &ch, double delt, int nsamp, const uint64_t *carry_in,
                 gpsiq_qchan_t *q, uint64_t *carry_out)
{
    std::memset(q, 0, sizeof *q);
    if (ch.prn <= 0) {
        if (carry_out) *carry_out = 0;
        return GPSIQ_OK;
    }
    if (ch.prn > 32) return fail(GPSIQ_E_ARG, "prn %d out of range 1..32", ch.prn);
    const double carr_inc = ch.f_carr * delt;   // the operand of gps.c:2821
    const double code_inc = ch.f_code * delt;   // the operand of gps.c:2789
    if (!(std::fabs(carr_inc) < 0.5))
        return fail(GPSIQ_E_RANGE, "prn %d: |f_carr/fs| = %g not < 0.5 cycle/sample", ch.prn, carr_inc);
    if (!(code_inc > 0.0 && code_inc < 2.0))
        return fail(GPSIQ_E_RANGE, "prn %d: f_code/fs = %g outside (0, 2) chips/sample", ch.prn, code_inc);
    if (!(ch.carr_phase >= 0.0 && ch.carr_phase < 1.0))
        return fail(GPSIQ_E_RANGE, "prn %d: carr_phase %g outside [0,1)", ch.prn, ch.carr_phase);
    if (!(ch.code_phase >= 0.0 && ch.code_phase < (double) GPSIQ_CA_SEQ_LEN))
        return fail(GPSIQ_E_RANGE, "prn %d: code_phase %g outside [0,1023)", ch.prn, ch.code_phase);
    if (ch.iword < 0 || ch.iword >= GPSIQ_N_DWRD || ch.ibit < 0 || ch.ibit > 29 || ch.icode < 0 || ch.icode > 19)
        return fail(GPSIQ_E_RANGE, "prn %d: iword/ibit/icode = %d/%d/%d out of range", ch.prn, ch.iword, ch.ibit, ch.icode);
    if (!(ch.gain > -kMaxGain && ch.gain < kMaxGain))       // also rejects NaN; the same bound gpsiq_set_descriptors applies
        return fail(GPSIQ_E_RANGE, "prn %d: gain %g not finite or |gain| >= %g", ch.prn, ch.gain, kMaxGain);

    q->prn = (uint8_t) ch.prn;
    q->icode = (uint8_t) ch.icode;
    q->gain = ch.gain;
    q->carr_step = std::llrint(std::ldexp(carr_inc, GPSIQ_CARR_FRAC_BITS));
    q->carr_phase = carry_in ? (*carry_in & kCarrMask) : carr_phase_to_fixed(ch.carr_phase);
    const int whole = (int) ch.code_phase;
    q->chip0 = (uint16_t) whole;
    q->code_frac = (uint64_t) std::floor(std::ldexp(ch.code_phase - (double) whole, GPSIQ_CODE_FRAC_BITS)) & kCodeMask;
    q->code_step = (uint64_t) std::llrint(std::ldexp(code_inc, GPSIQ_CODE_FRAC_BITS));

    // Nav data bits this block can reach, in the order the loop walks them
    // (gps.c:2795-2811: 20 code periods per bit, 30 bits per word).
    const unsigned __int128 last = (unsigned __int128) q->code_frac +
        (unsigned __int128) q->code_step * (unsigned __int128) (nsamp > 0 ? nsamp - 1 : 0);
    const uint64_t chips_end = (uint64_t) whole + (uint64_t) (last >> GPSIQ_CODE_FRAC_BITS);
    const uint64_t nbits = ((uint64_t) ch.icode + chips_end / GPSIQ_CA_SEQ_LEN) / 20 + 1;
    if (nbits > GPSIQ_MAX_NAV_BITS)
        return fail(GPSIQ_E_RANGE, "prn %d: block spans %llu nav bits (max %d)", ch.prn,
                    (unsigned long long) nbits, GPSIQ_MAX_NAV_BITS);
    int word = ch.iword, bit = ch.ibit;
    for (unsigned b = 0; b < nbits; ++b) {
        if (word >= GPSIQ_N_DWRD)
            return fail(GPSIQ_E_RANGE, "prn %d: block runs past dwrd[%d]", ch.prn, GPSIQ_N_DWRD - 1);
        q->nav_bits |= ((ch.dwrd[word] >> (29 - bit)) & 1u) << b;
        if (++bit == 30) { bit = 0; ++word; }
    }
    if (carry_out)
        *carry_out = (q->carr_phase + (uint64_t) q->carr_step * (uint64_t) nsamp) & kCarrMask;
    return GPSIQ_OK;
}

int quantize_timeline(const gpsiq_chan_t *ch, int nblocks, int nchan, double delt, int nsamp,
                      const bool *cont0, const uint64_t *carry0, gpsiq_qchan_t *q,
                      uint64_t *carry_end, int *last_prn)
{
    // pass 1, on host threads: everything but the carrier carry is independent per block
    struct QJob { const gpsiq_chan_t *ch; gpsiq_qchan_t *q; int nchan, nsamp; double delt; int rc; char err[320]; };
    QJob qj = {ch, q, nchan, nsamp, delt, GPSIQ_OK, ""};
    parallel_for(nblocks, 0, 64, [](void *p, int b0, int b1) {
        QJob &j = *static_cast<QJob *>(p);
        for (int b = b0; b < b1; ++b)
            for (int i = 0; i < j.nchan; ++i) {
                int rc = quantize_one(j.ch[(size_t) b * j.nchan + i], j.delt, j.nsamp, nullptr, &j.q[(size_t) b * j.nchan + i], nullptr);
                if (rc != GPSIQ_OK && __sync_bool_compare_and_swap(&j.rc, GPSIQ_OK, rc))
                    std::snprintf(j.err, sizeof j.err, "block %d: %.280s", b, gpsiq_last_error());
            }
    }, &qj);
    if (qj.rc != GPSIQ_OK) return fail(qj.rc, "%s", qj.err);
    chain_carrier(q, nblocks, nchan, nsamp, cont0, carry0, carry_end, last_prn);
    return GPSIQ_OK;
}

// serial and touching only q (cache-resident): p_{k+1} = p_k + nsamp*step_k (mod 2^59)
void chain_carrier(gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, const bool *cont0, const uint64_t *carry0,
                   uint64_t *carry_end, int *last_prn)
{
    uint64_t carry[GPSIQ_MAX_CHAN] = {};
    int prev_prn[GPSIQ_MAX_CHAN] = {};
    const uint64_t mask = (UINT64_C(1) << GPSIQ_CARR_FRAC_BITS) - 1;
    for (int b = 0; b < nblocks; ++b)
        for (int i = 0; i < nchan; ++i) {
            gpsiq_qchan_t &qq = q[(size_t) b * nchan + i];
            const int prn = qq.prn;
            const bool cont = prn && (b == 0 ? (cont0 && cont0[i]) : prev_prn[i] == prn);
            if (cont) qq.carr_phase = (b == 0 ? carry0[i] : carry[i]) & mask;
            carry[i] = prn ? (qq.carr_phase + (uint64_t) qq.carr_step * (uint64_t) nsamp) & mask : 0;
            prev_prn[i] = prn;
        }
    for (int i = 0; i < nchan; ++i) {
        if (carry_end) carry_end[i] = carry[i];
        if (last_prn) last_prn[i] = prev_prn[i];
    }
}

}  // namespace gpsiq

using namespace gpsiq;

extern "C" {

const char *gpsiq_version(void) { return "gpsiq 0.3 (gfx950)"; }

#ifndef GPSIQ_KERNELS_SHA16
#define GPSIQ_KERNELS_SHA16 "unknown"       // host-only test builds (no device code linked)
#endif
const char *gpsiq_kernels_id(void) { return GPSIQ_KERNELS_SHA16; }

const char *gpsiq_last_error(void) { return g_err; }

int gpsiq_prn_code(int prn, uint8_t chips[GPSIQ_CA_SEQ_LEN])
{
    if (prn < 1 || prn > 32 || !chips) return fail(GPSIQ_E_ARG, "prn %d out of range 1..32", prn);
    ca_code(prn, chips);
    return GPSIQ_OK;
}

void gpsiq_carrier_table(int16_t cos512[512], int16_t sin512_out[512])
{
    for (int k = 0; k < 512; ++k) {
        if (sin512_out) sin512_out[k] = (int16_t) sin512(k);
        if (cos512) cos512[k] = (int16_t) sin512(k + 128);
    }
}

int gpsiq_quantize(const gpsiq_chan_t *ch, int nchan, double fs, int nsamp,
                   gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out)
{
    if (!ch || !out) return fail(GPSIQ_E_ARG, "null descriptor pointer");
    if (nchan < 0 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "nchan %d outside 0..%d", nchan, GPSIQ_MAX_CHAN);
    if (nsamp < 0 || !(fs > 0.0)) return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g", nsamp, fs);
    const double delt = 1.0 / fs;   // gps.c:2298
    for (int c = 0; c < nchan; ++c) {
        int rc = quantize_one(ch[c], delt, nsamp, carry_in ? &carry_in[c] : nullptr, &out[c],
                              carry_out ? &carry_out[c] : nullptr);
        if (rc != GPSIQ_OK) return rc;
    }
    return GPSIQ_OK;
}

int gpsiq_quantize_batch(const gpsiq_chan_t *ch, int nblocks, int nchan, double fs, int nsamp,
                         gpsiq_qchan_t *out, const uint64_t *carry_in, uint64_t *carry_out)
{
    if ((!ch || !out) && nblocks) return fail(GPSIQ_E_ARG, "null descriptor pointer");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0)) return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g", nsamp, fs);
    bool cont0[GPSIQ_MAX_CHAN];
    for (int i = 0; i < nchan; ++i) cont0[i] = carry_in != nullptr;
    return quantize_timeline(ch, nblocks, nchan, 1.0 / fs, nsamp, cont0, carry_in, out, carry_out, nullptr);
}

int gpsiq_shard_range(int nblocks, int rank, int world, int *begin, int *end)
{
    if (nblocks < 0 || world < 1 || rank < 0 || rank >= world || !begin || !end)
        return fail(GPSIQ_E_ARG, "bad shard request: %d blocks, rank %d of %d", nblocks, rank, world);
    const int base = nblocks / world, extra = nblocks % world;       // the first `extra` ranks take one more
    *begin = rank * base + (rank < extra ? rank : extra);
    *end = *begin + base + (rank < extra ? 1 : 0);
    return GPSIQ_OK;
}

int gpsiq_shard_carry(const gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, gpsiq_shard_carry_t *out)
{
    if ((!q && nblocks) || !out) return fail(GPSIQ_E_ARG, "null argument");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN || nsamp < 0) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d / nsamp %d", nblocks, nchan, nsamp);
    for (int i = 0; i < nchan; ++i) {
        gpsiq_shard_carry_t &o = out[i];
        std::memset(&o, 0, sizeof o);
        o.nblocks = nblocks;
        int prev = 0;
        for (int b = 0; b < nblocks; ++b) {
            const gpsiq_qchan_t &d = q[(size_t) b * nchan + i];
            if (b == 0) o.first_prn = d.prn;
            else if (d.prn != prev) o.reseeded = 1;
            o.advance = (o.advance + (uint64_t) d.carr_step * (uint64_t) nsamp) & kCarrMask;
            o.end_phase = d.prn ? (d.carr_phase + (uint64_t) d.carr_step * (uint64_t) nsamp) & kCarrMask : 0;
            prev = d.prn;
        }
        o.last_prn = prev;
    }
    return GPSIQ_OK;
}

int gpsiq_shard_seed(gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, const gpsiq_shard_carry_t *all, int rank)
{
    if ((!q && nblocks) || !all) return fail(GPSIQ_E_ARG, "null argument");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN || nsamp < 0 || rank < 0) return fail(GPSIQ_E_ARG, "bad shard arguments");
    for (int i = 0; i < nchan && nblocks; ++i) {
        const int prn = q[i].prn;
        if (!prn) continue;
        // the phase the ranks before this one leave behind for the slot: walk back while the slot keeps this
        // PRN through whole ranges, to the range that seeded it (a re-allocation inside it, another PRN in
        // front of it, or rank 0), whose end_phase is absolute; then add the advances on the way forward
        uint64_t adv = 0;
        int r = rank - 1;
        bool found = false;
        for (; r >= 0; --r) {
            const gpsiq_shard_carry_t &s = all[(size_t) r * nchan + i];
            if (s.nblocks == 0) continue;
            if (s.last_prn != prn) break;                         // the slot held something else: this rank seeds itself
            if (s.reseeded || s.first_prn != prn || r == 0) { adv = (adv + s.end_phase) & kCarrMask; found = true; break; }
            // the whole of range r kept this PRN and continues its predecessor -- unless that one ended on another PRN
            int rp = r - 1;
            while (rp >= 0 && all[(size_t) rp * nchan + i].nblocks == 0) --rp;
            if (rp < 0 || all[(size_t) rp * nchan + i].last_prn != prn) { adv = (adv + s.end_phase) & kCarrMask; found = true; break; }
            adv = (adv + s.advance) & kCarrMask;
        }
        if (!found) continue;
        const uint64_t delta = (adv - q[i].carr_phase) & kCarrMask;   // true start minus the self-seeded start
        for (int b = 0; b < nblocks && q[(size_t) b * nchan + i].prn == prn; ++b) {
            gpsiq_qchan_t &d = q[(size_t) b * nchan + i];
            d.carr_phase = (d.carr_phase + delta) & kCarrMask;
        }
    }
    return GPSIQ_OK;
}

int gpsiq_chunker_init(gpsiq_chunker_t *ck, int sink_kind, int sample_size,
                       gpsiq_iq_buf_t *(*acquire)(void *), void (*enqueue)(void *, gpsiq_iq_buf_t *),
                       void *user)
{
    if (!ck || !acquire || !enqueue) return fail(GPSIQ_E_ARG, "null chunker argument");
    if (sink_kind < GPSIQ_SINK_IQFILE || sink_kind > GPSIQ_SINK_PLUTOSDR) return fail(GPSIQ_E_ARG, "bad sink kind %d", sink_kind);
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16) return fail(GPSIQ_E_ARG, "bad sample size %d", sample_size);
    ck->acquire = acquire; ck->enqueue = enqueue; ck->user = user;
    ck->sink_kind = sink_kind; ck->sample_size = sample_size;
    ck->cur = acquire(user);                       // gps.c:2698
    return ck->cur ? GPSIQ_OK : fail(GPSIQ_E_STATE, "fifo halted: acquire returned NULL");
}

// gps.c:2839-2865 on whole runs of elements instead of one element at a time.
int gpsiq_chunker_push(gpsiq_chunker_t *ck, const void *elems, size_t nelem)
{
    if (!ck || !elems) return fail(GPSIQ_E_ARG, "null chunker argument");
    if (!ck->cur) return fail(GPSIQ_E_STATE, "fifo halted");
    const size_t esz = (size_t) ck->sample_size;
    const unsigned char *src = (const unsigned char *) elems;
    int enq = 0;
    while (nelem) {
        gpsiq_iq_buf_t *b = ck->cur;
        size_t room = ck->sink_kind == GPSIQ_SINK_HACKRF
                          ? (size_t) GPSIQ_HACKRF_CHUNK - b->validLength     // gps.c:2849
                          : (size_t) b->totalLength - b->validLength;
        if (b->validLength + (room < nelem ? room : nelem) > b->totalLength)
            return fail(GPSIQ_E_RANGE, "fifo buffer of %u elements too small", b->totalLength);
        size_t n = room < nelem ? room : nelem;
        unsigned char *dstp = esz == 2 ? (unsigned char *) (b->data16 + b->validLength)
                                       : (unsigned char *) (b->data8 + b->validLength);
        std::memcpy(dstp, src, n * esz);
        b->validLength += (unsigned) n;
        src += n * esz;
        nelem -= n;
        if (ck->sink_kind == GPSIQ_SINK_HACKRF && b->validLength == GPSIQ_HACKRF_CHUNK) {
            ck->enqueue(ck->user, b);              // gps.c:2851-2853
            ++enq;
            if (!(ck->cur = ck->acquire(ck->user))) return fail(GPSIQ_E_STATE, "fifo halted");
        } else if (nelem) {
            return fail(GPSIQ_E_RANGE, "block does not fit the fifo buffer");
        }
    }
    if (ck->sink_kind != GPSIQ_SINK_HACKRF) {      // gps.c:2860-2865
        ck->enqueue(ck->user, ck->cur);
        ++enq;
        if (!(ck->cur = ck->acquire(ck->user))) return fail(GPSIQ_E_STATE, "fifo halted");
    }
    return enq;
}

void *gpsiq_chunker_reserve(gpsiq_chunker_t *ck, size_t nelem)
{
    if (!ck || !ck->cur || ck->sink_kind == GPSIQ_SINK_HACKRF) return nullptr;
    gpsiq_iq_buf_t *b = ck->cur;
    if ((size_t) b->validLength + nelem > (size_t) b->totalLength) return nullptr;
    return ck->sample_size == GPSIQ_SC16 ? (void *) (b->data16 + b->validLength) : (void *) (b->data8 + b->validLength);
}

int gpsiq_chunker_commit(gpsiq_chunker_t *ck, size_t nelem)
{
    if (!ck) return fail(GPSIQ_E_ARG, "null chunker argument");
    if (!ck->cur) return fail(GPSIQ_E_STATE, "fifo halted");
    if (ck->sink_kind == GPSIQ_SINK_HACKRF) return fail(GPSIQ_E_STATE, "HackRF chunks are not handed over in place");
    gpsiq_iq_buf_t *b = ck->cur;
    if ((size_t) b->validLength + nelem > (size_t) b->totalLength)
        return fail(GPSIQ_E_RANGE, "commit of %zu elements overruns the fifo buffer of %u", nelem, b->totalLength);
    b->validLength += (unsigned) nelem;
    ck->enqueue(ck->user, b);                      // gps.c:2860-2864
    if (!(ck->cur = ck->acquire(ck->user))) return fail(GPSIQ_E_STATE, "fifo halted");
    return 1;
}

}  // extern "C"
