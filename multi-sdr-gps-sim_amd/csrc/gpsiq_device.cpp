// gpsiq_device.cpp — device half of the C-ABI in include/gpsiq.h: context, resident
// descriptors, launches, the synchronous drop-in entry points.  HIP runtime only.
// There is deliberately no CPU path here: without a GPU gpsiq_create() fails.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <cstring>
#include <deque>
#include <new>
#include <vector>

#include "gpsiq_ctx.h"

using namespace gpsiq;

// GPSIQ_TRACE=1 in the environment prints the host-side phase times of the batch call to stderr
static double wall_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec * 1e3 + (double) ts.tv_nsec * 1e-6;
}

// Wait until no launch reads the buffer any more (every stream that used it), then forget the uses.
static int wait_idle(gpsiq_ctx::DescBuf &b)
{
    if (b.upload_pending) {              // also when nothing was ever launched on the set: its staging is about to be rewritten
        HIP_TRY(hipEventSynchronize(b.uploaded));
        b.upload_pending = false;
    }
    if (!b.in_use) return GPSIQ_OK;
    for (auto &u : b.use)
        if (u.active) { HIP_TRY(hipEventSynchronize(u.ev)); u.active = false; }
    b.in_use = false;
    return GPSIQ_OK;
}

// Record that stream s has just launched work reading the buffer.
static int mark_use(gpsiq_ctx::DescBuf &b, hipStream_t s)
{
    gpsiq_ctx::DescBuf::Use *slot = nullptr;
    for (auto &u : b.use)
        if (u.active && u.s == s) { slot = &u; break; }
    if (!slot)
        for (auto &u : b.use)
            if (!u.active) { slot = &u; break; }
    if (!slot) {
        // more streams than events: put this stream behind the first tracked one, whose event it then re-records
        slot = &b.use[0];
        HIP_TRY(hipStreamWaitEvent(s, slot->ev, 0));
    }
    if (!slot->ev) HIP_TRY(hipEventCreateWithFlags(&slot->ev, hipEventDisableTiming));
    HIP_TRY(hipEventRecord(slot->ev, s));
    slot->s = s; slot->active = true;
    b.in_use = true;
    return GPSIQ_OK;
}

static int ensure_out(gpsiq_ctx *c, size_t bytes)
{
    if (bytes > c->out_cap) {
        if (c->d_out) HIP_TRY(hipFree(c->d_out));
        c->d_out = nullptr; c->out_cap = 0;
        HIP_TRY(hipMalloc(&c->d_out, bytes));
        c->out_cap = bytes;
    }
    return GPSIQ_OK;
}

static int pick_variant(const gpsiq_ctx *c, int variant)
{
    if (variant == kAuto)
        return c->max_code_step <= kRowsMaxCodeStep ? kSeg : c->max_code_step <= kHalfRowsMaxCodeStep ? kSegHalf : kGeneric;
    return variant;
}

static int check_launch(const gpsiq_ctx *c, int block0, int nblocks, int nsamp, int sample_size,
                        const void *dst, size_t stride, int variant)
{
    if (!c) return fail(GPSIQ_E_ARG, "null context");
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16) return fail(GPSIQ_E_ARG, "bad sample size %d", sample_size);
    if (nsamp < 0 || nblocks < 0 || block0 < 0) return fail(GPSIQ_E_ARG, "negative size");
    if (!c->d_desc || nblocks > c->nblocks || block0 > c->nblocks - nblocks)        // no int overflow in the sum
        return fail(GPSIQ_E_STATE, "blocks [%d,+%d) not resident (have %d)", block0, nblocks, c->nblocks);
    if (!dst && nblocks && nsamp) return fail(GPSIQ_E_ARG, "null destination");
    if ((uintptr_t) dst & 3) return fail(GPSIQ_E_ARG, "destination %p not 4-byte aligned", dst);
    if (stride < (size_t) 2 * (size_t) nsamp * (size_t) sample_size || (stride & 3))
        return fail(GPSIQ_E_ARG, "block stride %zu too small or not a multiple of 4", stride);
    if (variant < 0 || variant >= kNumVariants) return fail(GPSIQ_E_ARG, "unknown variant %d", variant);
    if (variant == kSegHalf && c->max_code_step > kHalfRowsMaxCodeStep)
        return fail(GPSIQ_E_RANGE, "half-row kernel needs f_code/fs <= 1 chip per sample");
    if (variant >= kRows && variant != kSegHalf && c->max_code_step > kRowsMaxCodeStep)
        return fail(GPSIQ_E_RANGE, "row kernel needs f_code/fs <= 31/63 chip per sample");
    return GPSIQ_OK;
}

extern "C" {

int gpsiq_create(gpsiq_ctx_t **out, int device)
{
    if (!out) return fail(GPSIQ_E_ARG, "null context pointer");
    *out = nullptr;
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
        return fail(GPSIQ_E_DEVICE, "no HIP device (%s); libgpsiq has no CPU path", hipGetErrorString(e));
    if (device < 0 || device >= ndev) return fail(GPSIQ_E_ARG, "device %d outside 0..%d", device, ndev - 1);
    HIP_TRY(hipSetDevice(device));
    gpsiq_ctx *c = new (std::nothrow) gpsiq_ctx;
    if (!c) return fail(GPSIQ_E_NOMEM, "out of memory");
    c->device = device;
    DeviceTables *h = new (std::nothrow) DeviceTables;
    if (!h) { delete c; return fail(GPSIQ_E_NOMEM, "out of memory"); }
    build_device_tables(h);
    e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
        e = hipStreamCreateWithFlags(&c->copy_stream[i], hipStreamNonBlocking);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->chunk_done[i], hipEventDisableTiming);
    }
    if (e == hipSuccess) e = hipMalloc((void **) &c->d_tab, sizeof(DeviceTables));
    if (e == hipSuccess) e = hipMemcpy(c->d_tab, h, sizeof(DeviceTables), hipMemcpyHostToDevice);
    delete h;
    if (e != hipSuccess) {
        gpsiq_destroy(c);
        return fail(GPSIQ_E_DEVICE, "context setup: %s", hipGetErrorString(e));
    }
    *out = c;
    return GPSIQ_OK;
}

void gpsiq_destroy(gpsiq_ctx_t *c)
{
    if (!c) return;
    if (c->device >= 0) (void) hipSetDevice(c->device);
    (void) hipDeviceSynchronize();
    if (c->d_tab) (void) hipFree(c->d_tab);
    if (c->d_out) (void) hipFree(c->d_out);
    if (c->d_scratch) (void) hipFree(c->d_scratch);
    for (auto &a : c->aslot) {
        if (a.d) (void) hipFree(a.d);
        if (a.h) (void) hipHostFree(a.h);
        if (a.d_patch) (void) hipFree(a.d_patch);
        if (a.h_patch) (void) hipHostFree(a.h_patch);
        if (a.out) (void) hipFree(a.out);
        if (a.done) (void) hipEventDestroy(a.done);
    }
    for (int i = 0; i < gpsiq_ctx::kSets; ++i) {
        if (c->buf[i].d) (void) hipFree(c->buf[i].d);
        if (c->buf[i].h) (void) hipHostFree(c->buf[i].h);
        if (c->buf[i].d_patch) (void) hipFree(c->buf[i].d_patch);
        if (c->buf[i].h_patch) (void) hipHostFree(c->buf[i].h_patch);
        if (c->buf[i].uploaded) (void) hipEventDestroy(c->buf[i].uploaded);
        for (auto &u : c->buf[i].use)
            if (u.ev) (void) hipEventDestroy(u.ev);
    }
    for (int i = 0; i < 2; ++i) {
        if (c->chunk_done[i]) (void) hipEventDestroy(c->chunk_done[i]);
        if (c->copy_stream[i]) (void) hipStreamDestroy(c->copy_stream[i]);
    }
    gpsiq_evaldev_destroy(c);
    if (c->chain.d_in) (void) hipFree(c->chain.d_in);
    if (c->chain.h_in) (void) hipHostFree(c->chain.h_in);
    if (c->chain.d_prep) (void) hipFree(c->chain.d_prep);
    if (c->chain.d_maps) (void) hipFree(c->chain.d_maps);
    if (c->chain.h_maps) (void) hipHostFree(c->chain.h_maps);
    if (c->chain.d_est) (void) hipFree(c->chain.d_est);
    if (c->chain.h_est) (void) hipHostFree(c->chain.h_est);
    if (c->chain.d_c_before) (void) hipFree(c->chain.d_c_before);
    if (c->chain.t0) (void) hipEventDestroy(c->chain.t0);
    if (c->chain.t1) (void) hipEventDestroy(c->chain.t1);
    if (c->chain.landed) (void) hipEventDestroy(c->chain.landed);
    for (auto &e : c->chain.walked) if (e) (void) hipEventDestroy(e);
    if (c->chain.back) (void) hipStreamDestroy(c->chain.back);
    if (c->chain.stream) (void) hipStreamDestroy(c->chain.stream);
    if (c->stream) (void) hipStreamDestroy(c->stream);
    if (c->stream2) (void) hipStreamDestroy(c->stream2);
    if (c->up_stream) (void) hipStreamDestroy(c->up_stream);
    delete c;
}

void *gpsiq_host_alloc(size_t bytes)
{
    void *p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) {
        (void) fail(GPSIQ_E_NOMEM, "hipHostMalloc(%zu) failed", bytes);
        return nullptr;
    }
    return p;
}

void gpsiq_host_free(void *p)
{
    if (p) (void) hipHostFree(p);
}

}  // extern "C"

// patch list against the set it belongs to: slot counts the block's ACTIVE channels (device order), sorted by (block, sample)
static int check_patches(const gpsiq_ctx::DescBuf &b, int nblocks, const gpsiq_patch_t *patches, int n)
{
    for (int i = 0; i < n; ++i) {
        const gpsiq_patch_t &p = patches[i];
        if ((int64_t) p.block >= nblocks || p.slot >= b.active_per_block[p.block] || p.lut > 511 || p.neg > 1)
            return fail(GPSIQ_E_RANGE, "patch %d (block %u, slot %u, lut %u) outside the resident descriptors", i, p.block, p.slot, p.lut);
        if (i && (patches[i - 1].block > p.block || (patches[i - 1].block == p.block && patches[i - 1].sample > p.sample)))
            return fail(GPSIQ_E_ARG, "patches not sorted by (block, sample) at %d", i);
    }
    return GPSIQ_OK;
}

// gpsiq_set_descriptors (+ the set's patches in the same step).  no_wait: the uploads are queued on the context's upload
// stream and the call returns; launches on the set wait for them on the device (the pieces of a batch: the render thread
// queues piece after piece without a round trip to the device in between).
static int set_descriptors_impl(gpsiq_ctx_t *c, const gpsiq_qchan_t *q, int nblocks, int nchan, const gpsiq_patch_t *patches, int npatch,
                                bool no_wait)
{
    if (!c || !q) return fail(GPSIQ_E_ARG, "null argument");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t) nblocks * (size_t) nchan;
    const int next = (c->cur + 1) % gpsiq_ctx::kSets;
    gpsiq_ctx::DescBuf &nb = c->buf[next];              // a set not being read by the latest launches
    // the launch from kSets sets ago may still be reading this buffer (and its staging may still be
    // the source of an upload): wait for exactly that, not for the whole device
    { int wrc = wait_idle(nb); if (wrc) return wrc; }
    if (n > nb.hcap) {
        if (nb.h) HIP_TRY(hipHostFree(nb.h));
        nb.h = nullptr; nb.hcap = 0;
        HIP_TRY(hipHostMalloc((void **) &nb.h, n * sizeof(gpsiq_qchan_t), hipHostMallocDefault));
        nb.hcap = n;
    }
    // Device copy is compacted per block: active channels first, unused slots (zeroed)
    // after them.  The sum over channels is commutative modulo 2^16, so slot order is free.
    // Validation + compaction run on host threads straight into the page-locked staging buffer,
    // BEFORE anything resident is touched: a rejected set leaves the previous one in place.
    struct PJob { const gpsiq_qchan_t *q; gpsiq_qchan_t *out; uint8_t *active; int nchan; uint64_t mx; int max_active; long max_amp; int rc; size_t bad; };
    nb.active_per_block.resize((size_t) nblocks);
    PJob pj = {q, nb.h, nb.active_per_block.data(), nchan, 0, 0, 0, GPSIQ_OK, 0};
    const bool trace = std::getenv("GPSIQ_TRACE") != nullptr;
    const double t0 = trace ? wall_ms() : 0.0;
    parallel_for(nblocks, 0, 128, [](void *p, int b0, int b1) {
        PJob &j = *static_cast<PJob *>(p);
        uint64_t mx = 0;
        int max_active = 0;
        long max_amp = 0;
        for (int b = b0; b < b1; ++b) {
            int na = 0;
            long amp = 0;
            for (int s = 0; s < j.nchan; ++s) {
                const size_t i = (size_t) b * j.nchan + s;
                const gpsiq_qchan_t &d = j.q[i];
                if (!d.prn) continue;
                if (d.prn > 32 || d.chip0 >= GPSIQ_CA_SEQ_LEN || d.icode >= 20 || (d.code_frac >> GPSIQ_CODE_FRAC_BITS) ||
                    (d.code_step >> (GPSIQ_CODE_FRAC_BITS + 1)) || !(d.gain > -kMaxGain && d.gain < kMaxGain)) {
                    if (__sync_bool_compare_and_swap(&j.rc, GPSIQ_OK, d.prn > 32 ? GPSIQ_E_ARG : GPSIQ_E_RANGE)) j.bad = i;
                    continue;
                }
                if (d.code_step > mx) mx = d.code_step;
                amp += (long) (250.0 * std::fabs(d.gain));           // |(int)(table*gain)| <= (int)(250*|gain|), gps.c:2781-2782
                j.out[(size_t) b * j.nchan + na++] = d;
            }
            for (int s = na; s < j.nchan; ++s) std::memset(&j.out[(size_t) b * j.nchan + s], 0, sizeof(gpsiq_qchan_t));
            j.active[b] = (uint8_t) na;
            if (na > max_active) max_active = na;
            if (amp > max_amp) max_amp = amp;
        }
        for (uint64_t cur = j.mx; mx > cur && !__sync_bool_compare_and_swap(&j.mx, cur, mx); cur = j.mx) {}
        for (int cur = j.max_active; max_active > cur && !__sync_bool_compare_and_swap(&j.max_active, cur, max_active); cur = j.max_active) {}
        for (long cur = j.max_amp; max_amp > cur && !__sync_bool_compare_and_swap(&j.max_amp, cur, max_amp); cur = j.max_amp) {}
    }, &pj);
    if (pj.rc != GPSIQ_OK) return fail(pj.rc, "descriptor %zu outside the NCO format (prn %u)", pj.bad, q[pj.bad].prn);
    const size_t need = n ? n : 1;
    if (need > nb.cap) {
        if (nb.d) HIP_TRY(hipFree(nb.d));
        nb.d = nullptr; nb.cap = 0;
        HIP_TRY(hipMalloc((void **) &nb.d, need * sizeof(gpsiq_qchan_t)));
        nb.cap = need;
    }
    if (npatch > 0) {                                   // before anything resident is touched, like the descriptors
        const int prc = check_patches(nb, nblocks, patches, npatch);
        if (prc) return prc;
        if ((size_t) npatch > nb.patch_cap) {
            if (nb.d_patch) HIP_TRY(hipFree(nb.d_patch));
            nb.d_patch = nullptr; nb.patch_cap = 0;
            const size_t cap = (size_t) npatch < 256 ? 256 : (size_t) npatch;
            HIP_TRY(hipMalloc((void **) &nb.d_patch, cap * sizeof(gpsiq_patch_t)));
            nb.patch_cap = cap;
        }
        if ((size_t) npatch > nb.h_patch_cap) {
            if (nb.h_patch) HIP_TRY(hipHostFree(nb.h_patch));
            nb.h_patch = nullptr; nb.h_patch_cap = 0;
            const size_t cap = (size_t) npatch < 256 ? 256 : (size_t) npatch;
            HIP_TRY(hipHostMalloc((void **) &nb.h_patch, cap * sizeof(gpsiq_patch_t), hipHostMallocDefault));
            nb.h_patch_cap = cap;
        }
        std::memcpy(nb.h_patch, patches, (size_t) npatch * sizeof(gpsiq_patch_t));
    }
    if (n) {
        const double t1 = trace ? wall_ms() : 0.0;
        // on the context's upload stream (non-blocking, nothing else ever queued on it): overlaps whatever the caller's
        // streams and the context's own kernels are doing
        hipError_t e = hipMemcpyAsync(nb.d, nb.h, n * sizeof(gpsiq_qchan_t), hipMemcpyHostToDevice, c->up_stream);
        if (e == hipSuccess && npatch > 0)
            e = hipMemcpyAsync(nb.d_patch, nb.h_patch, (size_t) npatch * sizeof(gpsiq_patch_t), hipMemcpyHostToDevice, c->up_stream);
        if (e == hipSuccess && no_wait) {
            if (!nb.uploaded) e = hipEventCreateWithFlags(&nb.uploaded, hipEventDisableTiming);
            if (e == hipSuccess) e = hipEventRecord(nb.uploaded, c->up_stream);
            if (e == hipSuccess) nb.upload_pending = true;
        } else if (e == hipSuccess) {
            e = hipStreamSynchronize(c->up_stream);
        }
        if (e != hipSuccess) return fail(GPSIQ_E_DEVICE, "descriptor upload: %s", hipGetErrorString(e));
        if (trace)
            std::fprintf(stderr, "[gpsiq trace] descriptors %d blocks: validate+compact %.2f ms, upload %s %.2f ms\n",
                         nblocks, t1 - t0, no_wait ? "queued" : "done", wall_ms() - t1);
    }
    c->cur = next;
    c->d_desc = nb.d;
    c->nblocks = nblocks; c->nchan = nchan; c->max_code_step = pj.mx; c->max_active = pj.max_active; c->max_amplitude = pj.max_amp;
    nb.npatch = n ? npatch : 0;
    return GPSIQ_OK;
}

extern "C" {

int gpsiq_set_descriptors(gpsiq_ctx_t *c, const gpsiq_qchan_t *q, int nblocks, int nchan)
{
    return set_descriptors_impl(c, q, nblocks, nchan, nullptr, 0, false);
}

int gpsiq_set_patches(gpsiq_ctx_t *c, const gpsiq_patch_t *patches, int n)
{
    if (!c || (n > 0 && !patches) || n < 0) return fail(GPSIQ_E_ARG, "bad patch list");
    HIP_TRY(hipSetDevice(c->device));
    gpsiq_ctx::DescBuf &cb = c->buf[c->cur];
    { const int prc = check_patches(cb, c->nblocks, patches, n); if (prc) return prc; }
    cb.npatch = 0;
    if (n == 0) return GPSIQ_OK;                      // launches in flight took their count with them
    // launches of THIS set may still be applying the list that is replaced (the other buffer's launches have their own)
    { int wrc = wait_idle(cb); if (wrc) return wrc; }
    if ((size_t) n > cb.patch_cap) {
        if (cb.d_patch) HIP_TRY(hipFree(cb.d_patch));
        cb.d_patch = nullptr; cb.patch_cap = 0;
        const size_t cap = (size_t) n < 256 ? 256 : (size_t) n;
        HIP_TRY(hipMalloc((void **) &cb.d_patch, cap * sizeof(gpsiq_patch_t)));
        cb.patch_cap = cap;
    }
    hipError_t e = hipMemcpyAsync(cb.d_patch, patches, (size_t) n * sizeof(gpsiq_patch_t), hipMemcpyHostToDevice, c->up_stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->up_stream);
    if (e != hipSuccess) return fail(GPSIQ_E_DEVICE, "patch upload: %s", hipGetErrorString(e));
    cb.npatch = n;
    return GPSIQ_OK;
}

int gpsiq_set_nco_mode(gpsiq_ctx_t *c, int mode)
{
    if (!c) return fail(GPSIQ_E_ARG, "null context");
    if (mode != GPSIQ_NCO_FIXED && mode != GPSIQ_NCO_REFERENCE) return fail(GPSIQ_E_ARG, "unknown NCO mode %d", mode);
    if (mode != c->nco_mode)
        for (int i = 0; i < GPSIQ_MAX_CHAN; ++i) { c->carry_prn[i] = 0; c->carry[i] = 0; c->handed[i] = 0.0; }
    c->nco_mode = mode;
    return GPSIQ_OK;
}

// kernel + patches of blocks [block0, block0+nblocks) on stream s; marks the descriptor buffer as in use
static int launch_on(gpsiq_ctx *c, int v, int block0, int nblocks, int nsamp, int sample_size, void *dst, size_t stride, hipStream_t s)
{
    const size_t need = variant_scratch_bytes(v, nsamp, nblocks);
    if (need > c->scratch_cap) {
        // growing is rare (the first launch of a shape); hipFree waits for whatever still uses the old buffer
        if (c->d_scratch) HIP_TRY(hipFree(c->d_scratch));
        c->d_scratch = nullptr; c->scratch_cap = 0;
        HIP_TRY(hipMalloc(&c->d_scratch, need));
        c->scratch_cap = need;
    }
    if (c->buf[c->cur].upload_pending) HIP_TRY(hipStreamWaitEvent(s, c->buf[c->cur].uploaded, 0));    // a set staged without waiting
    hipError_t e = launch_variant(v, c->d_desc, c->nchan, nsamp, sample_size, dst, stride, block0, nblocks, c->d_tab, s,
                                  c->max_active, c->max_amplitude, need ? c->d_scratch : nullptr);
    if (e == hipSuccess && c->buf[c->cur].npatch)
        e = launch_patches(c->d_desc, c->nchan, nsamp, sample_size, dst, stride, block0, nblocks, c->d_tab, c->buf[c->cur].d_patch,
                           c->buf[c->cur].npatch, s);
    if (e != hipSuccess) return fail(GPSIQ_E_DEVICE, "launch: %s", hipGetErrorString(e));
    if (nblocks > 0 && nsamp > 0) return mark_use(c->buf[c->cur], s);
    return GPSIQ_OK;
}

int gpsiq_launch(gpsiq_ctx_t *c, int block0, int nblocks, int nsamp, int sample_size,
                 void *dst, size_t block_stride_bytes, void *hip_stream, int variant)
{
    int rc = check_launch(c, block0, nblocks, nsamp, sample_size, dst, block_stride_bytes, variant);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    return launch_on(c, pick_variant(c, variant), block0, nblocks, nsamp, sample_size, dst, block_stride_bytes, (hipStream_t) hip_stream);
}

int gpsiq_synchronize(gpsiq_ctx_t *c, void *hip_stream)
{
    if (!c) return fail(GPSIQ_E_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize((hipStream_t) hip_stream));
    return GPSIQ_OK;
}

int gpsiq_time_launches(gpsiq_ctx_t *c, int block0, int nblocks, int nsamp, int sample_size,
                        void *dst, size_t block_stride_bytes, void *hip_stream, int variant,
                        int iters, float *ms_per_launch)
{
    if (!ms_per_launch || iters < 1) return fail(GPSIQ_E_ARG, "bad timing arguments");
    int rc = check_launch(c, block0, nblocks, nsamp, sample_size, dst, block_stride_bytes, variant);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    hipStream_t s = (hipStream_t) hip_stream;
    const int v = pick_variant(c, variant);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    float ms = 0.f;
    hipError_t e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    if (e == hipSuccess) e = hipEventRecord(e0, s);
    for (int i = 0; i < iters && e == hipSuccess && rc == GPSIQ_OK; ++i)
        rc = launch_on(c, v, block0, nblocks, nsamp, sample_size, dst, block_stride_bytes, s);
    if (e == hipSuccess && rc == GPSIQ_OK) e = hipEventRecord(e1, s);
    if (e == hipSuccess && rc == GPSIQ_OK) e = hipEventSynchronize(e1);
    if (e == hipSuccess && rc == GPSIQ_OK) e = hipEventElapsedTime(&ms, e0, e1);
    if (e0) (void) hipEventDestroy(e0);                  // on every path
    if (e1) (void) hipEventDestroy(e1);
    if (rc != GPSIQ_OK) return rc;
    if (e != hipSuccess) return fail(GPSIQ_E_DEVICE, "timing: %s", hipGetErrorString(e));
    *ms_per_launch = ms / (float) iters;
    return GPSIQ_OK;
}

int gpsiq_num_variants(void) { return kNumVariants; }

const char *gpsiq_variant_name(int v)
{
    switch (v) {
    case kAuto: return "auto";
    case kGeneric: return "generic";
    case kRows: return "rows";
    case kRowsX: return "rowsx";
    case kTile: return "tile";
    case kSeg: return "seg";
    case kSegHalf: return "segh";
    case kSegMask: return "segm";
    case kSegBoth: return "segb";
    default: return "?";
    }
}

// ---- synchronous drop-in entry points ---------------------------------------

// Blocks per piece of a host-destination batch: the kernel of piece k+1 runs while piece k crosses
// PCIe (two copy streams, so consecutive copies queue back to back).  GPSIQ_PIECE_BLOCKS overrides;
// 0 = one kernel, then one copy (the round-1 behaviour, kept for A/B measurements).
static int d2h_chunk_blocks(size_t stride)
{
    if (const char *e = std::getenv("GPSIQ_PIECE_BLOCKS")) return std::atoi(e) > 0 ? std::atoi(e) : 0;   // read per call: A/B in one process
    const size_t target = (size_t) 32 << 20;                 // ~32 MiB per piece: >= 0.5 ms on the link, a few hundred workgroups
    const size_t n = (target + stride - 1) / stride;
    return (int) (n < 8 ? 8 : n);
}

static int run_to_host_or_device(gpsiq_ctx *c, const gpsiq_qchan_t *q, int nblocks, int nchan,
                                 int nsamp, int sample_size, void *dst, int dst_is_device)
{
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) sample_size;
    const size_t stride = (blk_bytes + 15) & ~(size_t) 15;
    int rc = gpsiq_set_descriptors(c, q, nblocks, nchan);
    if (rc) return rc;
    if (!nblocks || !nsamp) return GPSIQ_OK;
    if (dst_is_device && stride == blk_bytes && !((uintptr_t) dst & 3)) {
        rc = gpsiq_launch(c, 0, nblocks, nsamp, sample_size, dst, stride, c->stream, kAuto);
        if (rc) return rc;
        return gpsiq_synchronize(c, c->stream);
    }
    rc = ensure_out(c, stride * (size_t) nblocks);
    if (rc) return rc;
    const hipMemcpyKind kind = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
    const int chunk = d2h_chunk_blocks(stride);
    if (dst_is_device || chunk <= 0 || nblocks <= chunk) {
        rc = gpsiq_launch(c, 0, nblocks, nsamp, sample_size, c->d_out, stride, c->stream, kAuto);
        if (rc) return rc;
        if (stride == blk_bytes)
            HIP_TRY(hipMemcpyAsync(dst, c->d_out, blk_bytes * (size_t) nblocks, kind, c->stream));
        else
            HIP_TRY(hipMemcpy2DAsync(dst, blk_bytes, c->d_out, stride, blk_bytes, (size_t) nblocks, kind, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
        return GPSIQ_OK;
    }
    // pieces: kernel on c->stream, copy of piece k on copy_stream[k & 1] once its kernel has finished.
    // On an error in the middle the earlier pieces may still be copying into the caller's buffer: whatever happens, the
    // three streams are drained before this returns, so the caller may free or reuse dst.
    int k = 0;
    hipError_t e = hipSuccess;
    const char *what = "";
    for (int b0 = 0; b0 < nblocks && rc == GPSIQ_OK && e == hipSuccess; b0 += chunk, ++k) {
        const int nb = nblocks - b0 < chunk ? nblocks - b0 : chunk;
        uint8_t *piece = static_cast<uint8_t *>(c->d_out) + (size_t) b0 * stride;
        rc = gpsiq_launch(c, b0, nb, nsamp, sample_size, piece, stride, c->stream, kAuto);
        if (rc) break;
        hipStream_t cs = c->copy_stream[k & 1];
        what = "piece hand-over";
        e = hipEventRecord(c->chunk_done[k & 1], c->stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(cs, c->chunk_done[k & 1], 0);
        if (e != hipSuccess) break;
        what = "piece copy";
        if (stride == blk_bytes)                                  // rows are contiguous: one linear DMA
            e = hipMemcpyAsync(static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes, piece, blk_bytes * (size_t) nb, kind, cs);
        else
            e = hipMemcpy2DAsync(static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes, blk_bytes, piece, stride, blk_bytes,
                                 (size_t) nb, kind, cs);
    }
    const hipError_t s0 = hipStreamSynchronize(c->copy_stream[0]);
    const hipError_t s1 = hipStreamSynchronize(c->copy_stream[1]);
    const hipError_t s2 = hipStreamSynchronize(c->stream);
    if (rc) return rc;                                            // gpsiq_launch has set the text
    if (e != hipSuccess) return fail(GPSIQ_E_DEVICE, "%s: %s", what, hipGetErrorString(e));
    if (s0 != hipSuccess || s1 != hipSuccess || s2 != hipSuccess)
        return fail(GPSIQ_E_DEVICE, "batch pieces: %s", hipGetErrorString(s0 != hipSuccess ? s0 : s1 != hipSuccess ? s1 : s2));
    return GPSIQ_OK;
}

// The stream of piece k of a batch worked through in pieces.  On ONE stream a piece's kernel starts when the last workgroup of
// the piece before it has retired: every piece pays its own ramp-down (six pieces of a 2 000-block call: 1.75 ms of kernels
// against 1.50 ms in one launch, profiles/r05_chain_ab.txt).  Pieces write disjoint blocks and read their own descriptor set
// (four sets taken in turn; piece k+2 follows piece k on the same stream), so consecutive pieces alternate
// between two streams and the next piece's first workgroups fill the compute units the last ones of this piece leave.
// (One stream against two: profiles/r05_chain_ab.txt.)
static hipStream_t piece_stream(gpsiq_ctx *c, int k)
{
    return (k & 1) ? c->stream2 : c->stream;
}

// Blocks per piece of a long device-destination batch in the fixed-point model: ~1 ms of kernel, a few hundred microseconds of
// host work per piece.  GPSIQ_PIECE_BLOCKS overrides (read per call); <= 0: one piece.
static int batch_piece_blocks(int nblocks, int nsamp)
{
    long n = nsamp > 0 ? ((long) 1024 * 260000) / nsamp : 1024;
    if (n > 1024) n = 1024;
    if (n < 32) n = 32;
    if (const char *e = std::getenv("GPSIQ_PIECE_BLOCKS")) n = std::atoi(e);
    return n > 0 && 2 * n <= nblocks ? (int) n : nblocks;            // fewer than two pieces' worth: one piece
}

static int check_gen_args(const gpsiq_ctx *c, const void *ch, const void *dst, int nblocks, int nchan, int nsamp, double fs, int sample_size)
{
    if (!c || (!ch && nblocks) || (!dst && nblocks && nsamp)) return fail(GPSIQ_E_ARG, "null argument");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nsamp < 0 || !(fs > 0.0)) return fail(GPSIQ_E_ARG, "bad nsamp %d / fs %g", nsamp, fs);
    if (sample_size != GPSIQ_SC08 && sample_size != GPSIQ_SC16) return fail(GPSIQ_E_ARG, "bad sample size %d", sample_size);
    return GPSIQ_OK;
}

// ---- GPSIQ_NCO_REFERENCE: walk and render in pieces ------------------------------------------
// The carrier chain is serial in time on the host, the render is not: the timeline is cut into pieces, and while the device
// renders (and copies out) piece k the host threads (RefWalk, gpsiq_exact.cpp: a chain task and an evaluation task per channel
// and piece, taken piece-major from the shared pool) are in the pieces behind it.  RefRender is the device side of one context: begin() sizes the staging once, piece() queues
// descriptors + patches + kernel (+ the copy to the destination) without waiting, finish() drains.  generate_reference
// drives one from the calling thread; gpsiq_generate_batch_multi gives every device one, fed through a queue.
static int ref_chunk_blocks(int nblocks, int nsamp)
{
    // A piece costs the renderer ~0.1-0.2 ms (validate + compact + upload + launch) whatever its size and nothing on the
    // walkers' side, and should be enough samples for a launch that fills the chip: 256 blocks at 2.6 Msps (1.6 ms of
    // walking, 0.2 ms of kernel), fewer at higher rates where a block is more device work (25 Msps: 26 blocks = 66 M samples).
    long n = nsamp > 0 ? ((long) 256 * 260000) / nsamp : 256;
    if (n > 256) n = 256;
    if (n < 16) n = 16;
    if (const char *e = std::getenv("GPSIQ_PIECE_BLOCKS")) n = std::atoi(e);       // read per call: A/B in one process; <= 0: one piece
    return n > 0 && n < nblocks ? (int) n : nblocks;
}

struct RefRender {
    gpsiq_ctx *c = nullptr;
    int nchan = 0, nsamp = 0, ss = 0;
    uint8_t *dst = nullptr;          // destination of the range's first block
    bool dst_is_device = false, direct = false;
    size_t blk_bytes = 0, stride = 0;
    int k = 0, npiece = 0;

    int begin(gpsiq_ctx *ctx, int range_blocks, int nchan_, int nsamp_, int ss_, void *dst_, int dst_is_device_)
    {
        c = ctx; nchan = nchan_; nsamp = nsamp_; ss = ss_; dst = static_cast<uint8_t *>(dst_); dst_is_device = dst_is_device_ != 0; k = 0; npiece = 0;
        blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) ss;
        stride = (blk_bytes + 15) & ~(size_t) 15;
        direct = dst_is_device && stride == blk_bytes && !((uintptr_t) dst & 3);
        HIP_TRY(hipSetDevice(c->device));
        if (!direct && range_blocks && nsamp) return ensure_out(c, stride * (size_t) range_blocks);      // before anything is queued
        return GPSIQ_OK;
    }

    // blocks [b0, b0 + nb) of the range: q and patches (block indices relative to b0) belong to this piece
    int piece(const gpsiq_qchan_t *q, int b0, int nb, const std::vector<gpsiq_patch_t> &patches)
    {
        int rc = set_descriptors_impl(c, q, nb, nchan, patches.data(), (int) patches.size(), true);   // queued, not waited for
        if (rc) return rc;
        if (!nb || !nsamp) return GPSIQ_OK;
        uint8_t *dev = direct ? dst + (size_t) b0 * blk_bytes : static_cast<uint8_t *>(c->d_out) + (size_t) b0 * stride;
        hipStream_t s = piece_stream(c, npiece++);
        rc = gpsiq_launch(c, 0, nb, nsamp, ss, dev, stride, s, kAuto);
        if (rc || direct) return rc;
        hipStream_t cs = c->copy_stream[k & 1];
        HIP_TRY(hipEventRecord(c->chunk_done[k & 1], s));
        HIP_TRY(hipStreamWaitEvent(cs, c->chunk_done[k & 1], 0));
        const hipMemcpyKind kind = dst_is_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost;
        if (stride == blk_bytes)
            HIP_TRY(hipMemcpyAsync(dst + (size_t) b0 * blk_bytes, dev, blk_bytes * (size_t) nb, kind, cs));
        else
            HIP_TRY(hipMemcpy2DAsync(dst + (size_t) b0 * blk_bytes, blk_bytes, dev, stride, blk_bytes, (size_t) nb, kind, cs));
        ++k;
        return GPSIQ_OK;
    }

    // on every path, also after an error: nothing may still be writing the caller's buffer when the call returns
    int finish()
    {
        if (!c) return GPSIQ_OK;
        (void) hipSetDevice(c->device);
        const hipError_t s0 = hipStreamSynchronize(c->copy_stream[0]);
        const hipError_t s1 = hipStreamSynchronize(c->copy_stream[1]);
        hipError_t s2 = hipStreamSynchronize(c->stream);
        const hipError_t s3 = hipStreamSynchronize(c->stream2);
        if (s2 == hipSuccess) s2 = s3;
        if (s0 != hipSuccess || s1 != hipSuccess || s2 != hipSuccess)
            return fail(GPSIQ_E_DEVICE, "reference NCO pieces: %s", hipGetErrorString(s0 != hipSuccess ? s0 : s1 != hipSuccess ? s1 : s2));
        // every launch waited for its set's upload on the device and has finished; a set that was staged but never launched
        // (an error in between) may still be uploading: the upload stream is drained too before the flags are dropped
        if (hipStreamSynchronize(c->up_stream) != hipSuccess) return fail(GPSIQ_E_DEVICE, "reference NCO pieces: descriptor upload");
        for (auto &b : c->buf) b.upload_pending = false;
        return GPSIQ_OK;
    }
};

// piece boundaries of a range of `n` blocks starting at block `first` of the walk, `chunk` blocks each
// Piece sizes.  Nothing renders before the first piece is through all channels: half a chunk.  After that it depends on which
// side is the slower one.  HOST-bound (the rule at 2.6 - 10 Msps): a chunk, then two chunks each, and a chunk and half a chunk
// again at the end -- the last piece's kernel is all that is left after the host has finished.  KERNEL-bound (25 Msps: a block
// is 6.7 us of device work against ~3 us of host work per thread): every launch costs ~30 us beyond its share of one big
// launch (a 26-block piece is a single round of workgroups: 215 us measured against 183 us), so as few pieces as the host can
// keep ahead of -- each 2.2 x the one before (the host has piece k+1 ready before the kernel of piece k ends), no small tail.
static void piece_ends(int first, int n, int chunk, std::vector<int> *ends, bool kernel_bound = false)
{
    const int half = chunk > 1 ? chunk / 2 : 1;
    if (n <= 4 * chunk) {
        for (int b = chunk; b < n; b += chunk) ends->push_back(first + b);
        ends->push_back(first + n);
        return;
    }
    if (kernel_bound) {
        const int growth = 220;                           // per cent
        int b = 0, size = half;
        while (n - b > size + half) {                // what is left after this piece is worth a piece of its own
            b += size;
            ends->push_back(first + b);
            size = (int) (((long) size * growth + 50) / 100);
            if (size > 16 * chunk) size = 16 * chunk;
        }
        ends->push_back(first + n);
        return;
    }
    const int tail0 = n - chunk - half;               // the last two pieces: a chunk, half a chunk
    int b = half;
    ends->push_back(first + b);
    b += chunk;
    ends->push_back(first + b);
    while (tail0 - b >= 3 * chunk) { b += 2 * chunk; ends->push_back(first + b); }     // what is left (chunk .. 3 chunks) is one piece
    if (tail0 > b) ends->push_back(first + tail0);
    ends->push_back(first + n - half);
    ends->push_back(first + n);
}

// Whether a GPSIQ_NCO_REFERENCE batch is clearly kernel-bound, from the rates measured on MI355X + EPYC 9575F (DESIGN.md section
// 2): the kernel at 6.0e12 channel-samples/s, the host at 2.5 us + 0.8 us per 10^6 samples per block and channel on each of its
// threads (2.7 us at 2.6 Msps, 4.5 us at 25 Msps, in the call).  At 25 Msps on sixteen threads the two sides are within 1.5 x of
// each other and the symmetric ramp measured better (1.77 against 1.91 ms per 200 blocks): only a clear case takes the few
// growing pieces.
// The kernel's rate is MEASURED: every batch call that renders through the device evaluation times its last piece's synthesis
// with events and keeps a running mean in the context (gpsiq_evaldev.cpp); until the first such call, and for the host rate, the
// figures of MI355X + EPYC 9575F stand in.  GPSIQ_RATE_KERNEL (channel-samples per second, read per call) overrides both.
static std::atomic<double> g_rate_kernel_measured{0.0};        // a running mean over every context's calls (the placement rules below have no context at hand)
static double rate_kernel()
{
    if (const char *e = std::getenv("GPSIQ_RATE_KERNEL")) { const double v = std::atof(e); if (v > 1e9) return v; }
    const double m = g_rate_kernel_measured.load(std::memory_order_relaxed);
    return m > 0.0 ? m : 6.0e12;
}
static double rate_chain_us() { return 2.2; }      // microseconds per block and channel of the serial walk on one host thread

static bool ref_kernel_bound(int nsamp, int nchan)
{
    const int threads = host_threads() < nchan ? host_threads() : nchan;
    const double t_kernel = (double) nsamp * (double) nchan / rate_kernel();
    const double t_host = (double) nchan * (2.5e-6 + 0.8e-12 * (double) nsamp) / (double) (threads > 0 ? threads : 1);
    return t_kernel > 2.0 * t_host;
}

// ---- the carrier chain on the device -------------------------------------------------------------------------------
static int chain_reserve(gpsiq_ctx *c, size_t n)
{
    gpsiq_ctx::Chain &k = c->chain;
    if (!k.stream) {
        // (equal priorities: with the synthesis stream above the chain's, the second launch's maps came back late -- 0.98 ms
        // instead of 0.78 -- and the pieces behind the head waited for them: 2.6 ms per call instead of 2.4, profiles/r05_chain_ab.txt)
        HIP_TRY(hipStreamCreateWithFlags(&k.stream, hipStreamNonBlocking));
        HIP_TRY(hipStreamCreateWithFlags(&k.back, hipStreamNonBlocking));
        for (auto &e : k.walked) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
        HIP_TRY(hipEventCreate(&k.t0));
        HIP_TRY(hipEventCreate(&k.t1));
        HIP_TRY(hipEventCreateWithFlags(&k.landed, hipEventDisableTiming));
        HIP_TRY(hipMalloc((void **) &k.d_est, (kEvalMaxPieces + 1) * GPSIQ_MAX_CHAN * sizeof(gpsiq_chain_est_t)));
        HIP_TRY(hipHostMalloc((void **) &k.h_est, (kEvalMaxPieces + 1) * GPSIQ_MAX_CHAN * sizeof(gpsiq_chain_est_t), hipHostMallocDefault));
        HIP_TRY(hipMalloc((void **) &k.d_c_before, kEvalMaxPieces * GPSIQ_MAX_CHAN * sizeof(double)));
    }
    if (n <= k.cap) return GPSIQ_OK;
    if (k.d_in) (void) hipFree(k.d_in);
    if (k.h_in) (void) hipHostFree(k.h_in);
    if (k.d_prep) (void) hipFree(k.d_prep);
    if (k.d_maps) (void) hipFree(k.d_maps);
    if (k.h_maps) (void) hipHostFree(k.h_maps);
    k.d_in = nullptr; k.h_in = nullptr; k.d_prep = nullptr; k.d_maps = nullptr; k.h_maps = nullptr; k.cap = 0;
    const size_t cap = n + n / 4 + 256;
    HIP_TRY(hipMalloc((void **) &k.d_in, cap * sizeof(gpsiq_chain_in_t)));
    HIP_TRY(hipHostMalloc((void **) &k.h_in, cap * sizeof(gpsiq_chain_in_t), hipHostMallocDefault));
    HIP_TRY(hipMalloc(&k.d_prep, cap * 32));
    HIP_TRY(hipMalloc((void **) &k.d_maps, cap * sizeof(gpsiq_chain_map_t)));
    HIP_TRY(hipHostMalloc((void **) &k.h_maps, cap * sizeof(gpsiq_chain_map_t), hipHostMallocDefault));
    k.cap = cap;
    return GPSIQ_OK;
}

// Level 1 of the chain for blocks [b0, b0 + nb) of the inputs staged in c->chain.h_in, queued on the chain stream without
// waiting: upload, two kernels, the maps back into c->chain.h_maps (same rows), `landed` recorded behind them.  part 0 starts
// from `start` (host, may be null: the timeline begins here); part 1 continues where part 0's scan ended, on the device.
static int chain_queue(gpsiq_ctx *c, int part, int b0, int nb, int nchan, double fs, int nsamp, const gpsiq_chain_est_t *start, int max_stretches)
{
    gpsiq_ctx::Chain &k = c->chain;
    const size_t off = (size_t) b0 * (size_t) nchan, n = (size_t) nb * (size_t) nchan;
    if (max_stretches <= 0) {
        max_stretches = 32;
        if (const char *e = std::getenv("GPSIQ_CHAIN_STRETCHES")) { const int v = std::atoi(e); if (v >= 1 && v <= 32) max_stretches = v; }
    }
    HIP_TRY(hipMemcpyAsync(k.d_in + off, k.h_in + off, n * sizeof(gpsiq_chain_in_t), hipMemcpyHostToDevice, k.stream));
    const gpsiq_chain_est_t *d_start = nullptr;
    if (part == 1) d_start = k.d_est + GPSIQ_MAX_CHAN;
    else if (start) {
        std::memcpy(k.h_est, start, (size_t) nchan * sizeof(gpsiq_chain_est_t));
        HIP_TRY(hipMemcpyAsync(k.d_est, k.h_est, (size_t) nchan * sizeof(gpsiq_chain_est_t), hipMemcpyHostToDevice, k.stream));
        d_start = k.d_est;
    }
    if (part == 0) HIP_TRY(hipEventRecord(k.t0, k.stream));
    HIP_TRY(launch_chain(k.d_in + off, (int) sizeof(gpsiq_chain_in_t), nb, nchan, 1.0 / fs, nsamp, d_start, max_stretches, static_cast<char *>(k.d_prep) + off * 32,
                         k.d_c_before + part * GPSIQ_MAX_CHAN, k.d_est + (part + 1) * GPSIQ_MAX_CHAN, k.d_maps + off, k.stream));
    HIP_TRY(hipEventRecord(k.t1, k.stream));
    // the maps' way back on a stream of its own: the next launch's kernels follow these at once (a callback queued between
    // them held the second launch up by ~0.1 ms)
    HIP_TRY(hipEventRecord(k.walked[part], k.stream));
    HIP_TRY(hipStreamWaitEvent(k.back, k.walked[part], 0));
    HIP_TRY(hipMemcpyAsync(k.h_maps + off, k.d_maps + off, n * sizeof(gpsiq_chain_map_t), hipMemcpyDeviceToHost, k.back));
    HIP_TRY(hipMemcpyAsync(k.h_est + (part + 1) * GPSIQ_MAX_CHAN, k.d_est + (part + 1) * GPSIQ_MAX_CHAN, (size_t) nchan * sizeof(gpsiq_chain_est_t),
                           hipMemcpyDeviceToHost, k.back));
    HIP_TRY(hipEventRecord(k.landed, k.back));
    return GPSIQ_OK;
}

static int chain_drain(gpsiq_ctx *c)
{
    const hipError_t a = hipStreamSynchronize(c->chain.stream), b = hipStreamSynchronize(c->chain.back);
    if (a != hipSuccess || b != hipSuccess) return fail(GPSIQ_E_DEVICE, "carrier chain, level 1: %s", hipGetErrorString(a != hipSuccess ? a : b));
    return GPSIQ_OK;
}

// ... and waited for
static int chain_maps_staged(gpsiq_ctx *c, int nblocks, int nchan, double fs, int nsamp, const gpsiq_chain_est_t *start, int max_stretches,
                             gpsiq_chain_est_t *end)
{
    gpsiq_ctx::Chain &k = c->chain;
    int rc = chain_queue(c, 0, 0, nblocks, nchan, fs, nsamp, start, max_stretches);
    if (rc) { (void) chain_drain(c); return rc; }
    rc = chain_drain(c);
    if (rc) return rc;
    (void) hipEventElapsedTime(&k.last_ms, k.t0, k.t1);
    if (end) std::memcpy(end, k.h_est + GPSIQ_MAX_CHAN, (size_t) nchan * sizeof(gpsiq_chain_est_t));
    return GPSIQ_OK;
}

extern "C" int gpsiq_chain_maps_device(gpsiq_ctx_t *c, const gpsiq_chain_in_t *in, int nblocks, int nchan, double fs, int nsamp,
                                       const gpsiq_chain_est_t *start, int max_stretches, gpsiq_chain_map_t *maps, gpsiq_chain_est_t *end,
                                       float *kernel_ms)
{
    if (!c || ((!in || !maps) && nblocks)) return fail(GPSIQ_E_ARG, "null argument");
    if (nblocks < 0 || nchan < 1 || nchan > GPSIQ_MAX_CHAN || nsamp < 0 || !(fs > 0.0)) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d / nsamp %d / fs %g", nblocks, nchan, nsamp, fs);
    if (kernel_ms) *kernel_ms = 0.0f;
    if (nblocks == 0) return GPSIQ_OK;
    HIP_TRY(hipSetDevice(c->device));
    const size_t n = (size_t) nblocks * (size_t) nchan;
    int rc = chain_reserve(c, n);
    if (rc) return rc;
    std::memcpy(c->chain.h_in, in, n * sizeof(gpsiq_chain_in_t));
    rc = chain_maps_staged(c, nblocks, nchan, fs, nsamp, start, max_stretches, end);
    if (rc) return rc;
    std::memcpy(maps, c->chain.h_maps, n * sizeof(gpsiq_chain_map_t));
    if (kernel_ms) *kernel_ms = c->chain.last_ms;
    return GPSIQ_OK;
}

// Whether a GPSIQ_NCO_REFERENCE batch walks level 1 of its carrier chain on the device.  The serial walk on host threads costs
// ~2.2 us per block and channel on each of min(threads, channels) threads and hides under the kernel of the piece before when
// the kernel is the slower side (25 Msps on sixteen threads); level 1 on the device costs a launch latency of ~0.25 ms before
// anything renders and nearly nothing after that.  GPSIQ_CHAIN=host / device decides by hand (read per call: A/B in one process).
static bool chain_on_device(int nblocks, int nsamp, int nchan)
{
    const char *e = std::getenv("GPSIQ_CHAIN");
    if (e && !std::strcmp(e, "host")) return false;
    if (e && !std::strcmp(e, "device")) return nblocks > 1;
    if (nblocks < 48) return false;
    const int threads = host_threads() < nchan ? host_threads() : nchan;
    const double t_kernel = (double) nsamp * (double) nchan / rate_kernel();
    const double t_chain = rate_chain_us() * 1e-6 * (double) nchan / (double) (threads > 0 ? threads : 1);
    return t_chain > 0.5 * t_kernel;
}

// chain inputs of blocks [b0, b1) cut out of the descriptors into the page-locked staging: 296-byte descriptors, 24 bytes wanted
// of each -- memory-bound, so spread over the pool
static void chain_stage_inputs(gpsiq_ctx *c, const gpsiq_chan_t *ch, int b0, int b1, int nchan)
{
    struct Job { const gpsiq_chan_t *ch; gpsiq_chain_in_t *out; } job = {ch + (size_t) b0 * nchan, c->chain.h_in + (size_t) b0 * nchan};
    parallel_for((b1 - b0) * nchan, (b1 - b0) * nchan >= 8192 ? 0 : 1, 2048, [](void *p, int k0, int k1) {
        const Job &j = *static_cast<Job *>(p);
        gpsiq_chain_inputs(j.ch + k0, k1 - k0, j.out + k0);
    }, &job);
}

static void *run_walk(void *w) { static_cast<RefWalk *>(w)->run(); return nullptr; }

// ref_q / ref_start are kept between calls (a fresh 1.5 MB per call is four hundred page faults on the critical thread) -- but not
// at the size of the largest batch the context has ever seen: a call that needed less than an eighth of what is held (and what is
// held is more than 16 MB) gives the rest back
static void trim_retained(gpsiq_ctx *c, size_t need)
{
    if (c->ref_q.capacity() > (size_t) 8 * need && c->ref_q.capacity() * sizeof(gpsiq_qchan_t) > ((size_t) 16 << 20)) {
        std::vector<gpsiq_qchan_t>(need).swap(c->ref_q);
        std::vector<double>(c->ref_start.size() < need ? c->ref_start.size() : need).swap(c->ref_start);
    }
}

// GPSIQ_NCO_REFERENCE form of both drop-in calls: the carrier is the caller's double, walked exactly
static int generate_reference(gpsiq_ctx *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs,
                              int sample_size, void *dst, int dst_is_device, double *carr_phase_out, const double *seeds = nullptr)
{
    const char *trace_env = std::getenv("GPSIQ_TRACE");
    const bool trace = trace_env != nullptr, trace_pieces = trace && std::atoi(trace_env) >= 2;      // GPSIQ_TRACE=2: every piece
    const double t0 = trace ? wall_ms() : 0.0;
    double t_wait = 0.0, t_queue = 0.0;
    size_t npatch = 0;
    std::vector<gpsiq_qchan_t> &q = c->ref_q;                // every element is written by its evaluation task before it is read
    if (q.size() < (size_t) nblocks * (size_t) nchan) q.resize((size_t) nblocks * (size_t) nchan);
    if (!seeds && c->ref_start.size() < (size_t) nblocks * (size_t) nchan) c->ref_start.resize((size_t) nblocks * (size_t) nchan);
    std::vector<gpsiq_patch_t> patches;
    RefRender r;
    int rc = r.begin(c, nblocks, nchan, nsamp, sample_size, dst, dst_is_device);
    if (rc) return rc;
    const int chunk = ref_chunk_blocks(nblocks, nsamp);
    std::vector<int> ends;
    const bool dev_chain = !seeds && chain_on_device(nblocks, nsamp, nchan);
    // (few growing pieces also with the chain on the device were tried: a 600-block piece is 0.3 ms of evaluation before it can
    // render, and the device waits for it: 2.6 ms per call against 2.4 at 2.6 Msps, profiles/r05_chain_ab.txt)
    piece_ends(0, nblocks, chunk, &ends, ref_kernel_bound(nsamp, nchan));
    // The carrier chain: level 1 (every block's certified map) on the device, parallel in time (gpsiq_chain_kernels.hip); the chain
    // tasks then link block to block through the maps.  Nothing renders before the first maps are back, and a launch is ~0.25 ms
    // however small: the timeline goes in two launches -- a head whose kernels cover the second launch, then the rest --, and the
    // walkers are let into the rest when its maps have landed (RefWalk::release_maps).
    double t_chain[3] = {};
    int head = nblocks;
    if (dev_chain) {
        rc = chain_reserve(c, (size_t) nblocks * (size_t) nchan);
        if (rc) return rc;
        // the head: pieces worth ~0.4 ms of synthesis (the second launch's latency + its first piece's evaluation)
        const double t_block = (double) nsamp * (double) nchan / rate_kernel();
        int want = (int) (0.4e-3 / t_block) + 1;
        if (want > 0 && 2 * want < nblocks)                    // the first piece end at or beyond that (a bigger head measured better than a
            for (size_t k = 0; k < ends.size(); ++k)           // smaller one: 2.35 ms per call at 900 blocks, 2.49 at 256, 2.55 at 128)
                if (ends[k] >= want) { head = ends[k]; break; }
        if (2 * head > nblocks) head = nblocks;                // what is left would not be worth a launch of its own
    }
    RefWalk w(ch, nblocks, nchan, 1.0 / fs, nsamp, q.data(), nullptr, nullptr, ends);
    w.seeds = seeds;                                         // start states known (gpsiq_generate_seeded): evaluation tasks only
    if (!seeds) w.start_out = c->ref_start.data();
    struct Landed { RefWalk *w; int upto; double *at, t0; };
    Landed landed[2] = {{&w, head, &t_chain[1], t0}, {&w, nblocks, &t_chain[2], t0}};
    // runs on a thread of the HIP runtime when the maps of a launch are in host memory: the walkers may link through them
    auto on_landed = [](void *p) {
        Landed *l = static_cast<Landed *>(p);
        if (l->t0 != 0.0) *l->at = wall_ms() - l->t0;
        l->w->release_maps(l->upto);
    };
    if (dev_chain) {
        w.in = c->chain.h_in; w.maps = c->chain.h_maps; w.maps_upto.store(0);
        chain_stage_inputs(c, ch, 0, head, nchan);
        rc = chain_queue(c, 0, 0, head, nchan, fs, nsamp, nullptr, 0);
        if (rc == GPSIQ_OK && hipLaunchHostFunc(c->chain.back, on_landed, &landed[0]) != hipSuccess) rc = fail(GPSIQ_E_DEVICE, "hipLaunchHostFunc");
        if (rc == GPSIQ_OK && head < nblocks) {
            chain_stage_inputs(c, ch, head, nblocks, nchan);                               // under the head's kernels
            rc = chain_queue(c, 1, head, nblocks - head, nchan, fs, nsamp, nullptr, 0);
            if (rc == GPSIQ_OK && hipLaunchHostFunc(c->chain.back, on_landed, &landed[1]) != hipSuccess) rc = fail(GPSIQ_E_DEVICE, "hipLaunchHostFunc");
        }
        if (rc) { (void) chain_drain(c); return rc; }
        if (trace) t_chain[0] = wall_ms() - t0;
    }
    // one piece (a block call, a short batch): walk here, then render; else the walk runs on the pool, driven by a helper
    // thread, and this thread renders every piece as soon as all channels are through it
    pthread_t th;
    const bool threaded = w.npieces() > 1 && pthread_create(&th, nullptr, run_walk, &w) == 0;
    if (!threaded) w.run();
    for (size_t k = 0; k < w.npieces() && rc == GPSIQ_OK; ++k) {
        const double tw = trace ? wall_ms() : 0.0;
        rc = w.wait_piece(k);
        if (rc != GPSIQ_OK) { (void) fail(rc, "%s", w.err); break; }
        const double tp = trace ? wall_ms() : 0.0;
        w.take_patches(k, &patches, true);
        npatch += patches.size();
        const int b0 = k ? w.ends[k - 1] : 0;
        rc = r.piece(q.data() + (size_t) b0 * nchan, b0, w.ends[k] - b0, patches);
        if (trace) {
            const double tq = wall_ms();
            t_wait += tp - tw; t_queue += tq - tp;
            if (trace_pieces)
                std::fprintf(stderr, "[gpsiq trace]   piece %zu, blocks [%d, %d): ready at %.3f ms, queued at %.3f ms, %zu patches\n",
                             k, b0, w.ends[k], tp - t0, tq - t0, patches.size());
        }
    }
    char err[400] = "";
    if (rc != GPSIQ_OK) { std::snprintf(err, sizeof err, "%s", gpsiq_last_error()); w.abort(); }     // nothing further is walked for a call that has failed
    if (dev_chain) {
        // the callbacks have run when the chain's streams have drained -- unless the device reported an error, in which case the
        // walkers must not be left waiting for maps that never land (and must not link through whatever h_maps held before)
        const int crc = chain_drain(c);
        if (crc != GPSIQ_OK) {
            if (rc == GPSIQ_OK) { rc = crc; std::snprintf(err, sizeof err, "%s", gpsiq_last_error()); }
            w.abort();
        }
        w.release_maps(nblocks);
    }
    if (threaded) pthread_join(th, nullptr);                 // the walkers read ch and write q: never leave them running
    const double tf = trace ? wall_ms() : 0.0;
    const int frc = r.finish();
    if (rc != GPSIQ_OK) return fail(rc, "%s", err);
    if (w.rc != GPSIQ_OK) return fail(w.rc, "%s", w.err);
    if (frc != GPSIQ_OK) return frc;
    if (trace && dev_chain)
        std::fprintf(stderr, "[gpsiq trace] carrier chain, level 1 on the device: two launches queued by %.3f ms; the maps of the head (%d blocks) back at "
                             "%.3f ms, of the rest at %.3f ms\n", t_chain[0], head, t_chain[1], t_chain[2]);
    if (trace)
        std::fprintf(stderr, "[gpsiq trace] reference NCO, %d blocks in %zu pieces of %d: waited for the walkers %.2f ms (%zu patches), "
                             "validate + upload + launch %.2f ms, final wait %.2f ms, whole call %.2f ms\n",
                     nblocks, w.npieces(), chunk, t_wait, npatch, t_queue, wall_ms() - tf, wall_ms() - t0);
    if (carr_phase_out)
        for (int i = 0; i < nchan; ++i)
            carr_phase_out[i] = w.last_prn[i] ? w.carr_end[i] : ch[(size_t) (nblocks - 1) * nchan + i].carr_phase;
    trim_retained(c, (size_t) nblocks * (size_t) nchan);
    return GPSIQ_OK;
}

int gpsiq_generate_block(gpsiq_ctx_t *c, const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                         int sample_size, void *dst, double *carr_phase_out)
{
    if (!dst) return fail(GPSIQ_E_ARG, "null argument");
    int rc = check_gen_args(c, ch, dst, 1, nchan, nsamp, fs, sample_size);
    if (rc) return rc;
    // The asynchronous form -- descriptor upload, kernel (+ patches), copy into dst, all on the context's one stream -- and a
    // wait for THIS block: one host round trip where set_descriptors + launch + copy made two (84 -> ~60 us per block on MI355X).
    // Anything still queued by earlier _async calls completes first (same stream).  The continuation state goes back to what
    // it was if the device reports an error.
    uint64_t carry0[GPSIQ_MAX_CHAN];
    int      prn0[GPSIQ_MAX_CHAN];
    double   handed0[GPSIQ_MAX_CHAN];
    std::memcpy(carry0, c->carry, sizeof carry0);
    std::memcpy(prn0, c->carry_prn, sizeof prn0);
    std::memcpy(handed0, c->handed, sizeof handed0);
    const int slot = c->anext;
    rc = gpsiq_generate_block_async(c, ch, nchan, nsamp, fs, sample_size, dst, carr_phase_out);
    if (rc) return rc;
    if (nsamp > 0) {
        gpsiq_ctx::AsyncSlot &a = c->aslot[slot];
        const hipError_t e = hipEventSynchronize(a.done);
        a.busy = false;
        if (e != hipSuccess) {
            std::memcpy(c->carry, carry0, sizeof carry0);
            std::memcpy(c->carry_prn, prn0, sizeof prn0);
            std::memcpy(c->handed, handed0, sizeof handed0);
            return fail(GPSIQ_E_DEVICE, "block: %s", hipGetErrorString(e));
        }
    }
    return GPSIQ_OK;
}

int gpsiq_generate_block_async(gpsiq_ctx_t *c, const gpsiq_chan_t *ch, int nchan, int nsamp, double fs,
                               int sample_size, void *dst, double *carr_phase_out)
{
    if (!dst) return fail(GPSIQ_E_ARG, "null argument");
    int rc = check_gen_args(c, ch, dst, 1, nchan, nsamp, fs, sample_size);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(c->device));
    const bool reference = c->nco_mode == GPSIQ_NCO_REFERENCE;
    gpsiq_qchan_t q[GPSIQ_MAX_CHAN];
    uint64_t next[GPSIQ_MAX_CHAN] = {};
    std::vector<gpsiq_patch_t> patches;
    double carr_end[GPSIQ_MAX_CHAN] = {};
    int last_prn[GPSIQ_MAX_CHAN] = {};
    const double delt = 1.0 / fs;
    if (reference) {
        // the carrier is the caller's double, walked exactly on the host: everything the device needs (start phases,
        // patches) and the phase to hand out are known before anything is queued
        rc = reference_timeline(ch, 1, nchan, delt, nsamp, q, &patches, carr_end, last_prn);
        if (rc) return rc;
    } else {
        for (int i = 0; i < nchan; ++i) {
            const bool cont = ch[i].prn > 0 && c->carry_prn[i] == ch[i].prn && c->handed[i] == ch[i].carr_phase;
            rc = quantize_one(ch[i], delt, nsamp, cont ? &c->carry[i] : nullptr, &q[i], &next[i]);
            if (rc) return rc;
        }
    }
    gpsiq_ctx::AsyncSlot &a = c->aslot[c->anext];
    if (a.busy) { HIP_TRY(hipEventSynchronize(a.done)); a.busy = false; }     // the ring is full: wait for its oldest block
    // each piece on its own: a call that failed half-way must not leave a slot that looks complete
    if (!a.d) HIP_TRY(hipMalloc((void **) &a.d, GPSIQ_MAX_CHAN * sizeof(gpsiq_qchan_t)));
    if (!a.h) HIP_TRY(hipHostMalloc((void **) &a.h, GPSIQ_MAX_CHAN * sizeof(gpsiq_qchan_t), hipHostMallocDefault));
    if (!a.done) HIP_TRY(hipEventCreateWithFlags(&a.done, hipEventDisableTiming));
    // compact (active channels first) and take the launch parameters, as gpsiq_set_descriptors does for a batch
    int na = 0;
    long amp = 0;
    uint64_t max_step = 0;
    for (int i = 0; i < nchan; ++i) {
        if (!q[i].prn) continue;
        if (!(q[i].gain > -kMaxGain && q[i].gain < kMaxGain)) return fail(GPSIQ_E_RANGE, "gain %g outside the NCO format", q[i].gain);
        if (q[i].code_step > max_step) max_step = q[i].code_step;
        amp += (long) (250.0 * std::fabs(q[i].gain));
        a.h[na++] = q[i];
    }
    for (int i = na; i < nchan; ++i) std::memset(&a.h[i], 0, sizeof(gpsiq_qchan_t));
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) sample_size;
    const size_t stride = (blk_bytes + 15) & ~(size_t) 15;
    if (stride > a.out_cap) {
        if (a.out) HIP_TRY(hipFree(a.out));
        a.out = nullptr; a.out_cap = 0;
        HIP_TRY(hipMalloc(&a.out, stride ? stride : 16));
        a.out_cap = stride ? stride : 16;
    }
    if (!patches.empty() && patches.size() > a.patch_cap) {      // the slot is idle here (its event was waited for above)
        // each pointer is forgotten before its free can fail: a second call (or gpsiq_destroy) must not free it again
        { gpsiq_patch_t *dp = a.d_patch, *hp = a.h_patch; a.d_patch = a.h_patch = nullptr; a.patch_cap = 0;
          const hipError_t f0 = dp ? hipFree(dp) : hipSuccess, f1 = hp ? hipHostFree(hp) : hipSuccess;
          if (f0 != hipSuccess || f1 != hipSuccess) return fail(GPSIQ_E_DEVICE, "patch staging: %s", hipGetErrorString(f0 != hipSuccess ? f0 : f1)); }
        const size_t cap = patches.size() < 256 ? 256 : patches.size();
        HIP_TRY(hipMalloc((void **) &a.d_patch, cap * sizeof(gpsiq_patch_t)));
        HIP_TRY(hipHostMalloc((void **) &a.h_patch, cap * sizeof(gpsiq_patch_t), hipHostMallocDefault));
        a.patch_cap = cap;
    }
    if (nsamp > 0) {
        const int v = max_step <= kRowsMaxCodeStep ? kSeg : max_step <= kHalfRowsMaxCodeStep ? kSegHalf : kGeneric;
        // once the first copy is queued a failure must not return with work in flight on the slot's page-locked staging (the next
        // call would rewrite it under the copy): the stream is drained first
        hipError_t e = hipMemcpyAsync(a.d, a.h, (size_t) nchan * sizeof(gpsiq_qchan_t), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = launch_variant(v, a.d, nchan, nsamp, sample_size, a.out, stride, 0, 1, c->d_tab, c->stream, na, amp, nullptr);
        if (e == hipSuccess && !patches.empty()) {
            std::memcpy(a.h_patch, patches.data(), patches.size() * sizeof(gpsiq_patch_t));
            e = hipMemcpyAsync(a.d_patch, a.h_patch, patches.size() * sizeof(gpsiq_patch_t), hipMemcpyHostToDevice, c->stream);
            if (e == hipSuccess) e = launch_patches(a.d, nchan, nsamp, sample_size, a.out, stride, 0, 1, c->d_tab, a.d_patch, (int) patches.size(), c->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(dst, a.out, blk_bytes, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipEventRecord(a.done, c->stream);
        if (e != hipSuccess) {
            (void) hipStreamSynchronize(c->stream);
            return fail(GPSIQ_E_DEVICE, "block: %s", hipGetErrorString(e));
        }
        a.busy = true;
        c->anext = (c->anext + 1) & 3;
    }
    for (int i = 0; i < nchan; ++i) {
        if (reference) {
            if (carr_phase_out) carr_phase_out[i] = last_prn[i] ? carr_end[i] : ch[i].carr_phase;
            continue;
        }
        c->carry_prn[i] = ch[i].prn > 0 ? ch[i].prn : 0;
        c->carry[i] = next[i];
        c->handed[i] = ch[i].prn > 0 ? carr_phase_to_double(next[i]) : 0.0;
        if (carr_phase_out) carr_phase_out[i] = ch[i].prn > 0 ? c->handed[i] : ch[i].carr_phase;
    }
    return GPSIQ_OK;
}

int gpsiq_wait(gpsiq_ctx_t *c)
{
    if (!c) return fail(GPSIQ_E_ARG, "null context");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    for (auto &a : c->aslot) a.busy = false;
    return GPSIQ_OK;
}

int gpsiq_generate_quantized(gpsiq_ctx_t *c, const gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp,
                             int sample_size, void *dst, int dst_is_device)
{
    int rc = check_gen_args(c, q, dst, nblocks, nchan, nsamp, 1.0, sample_size);
    if (rc) return rc;
    if (nblocks == 0) return GPSIQ_OK;
    return run_to_host_or_device(c, q, nblocks, nchan, nsamp, sample_size, dst, dst_is_device);
}

int gpsiq_generate_batch(gpsiq_ctx_t *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp,
                         double fs, int sample_size, void *dst, int dst_is_device, double *carr_phase_out)
{
    int rc = check_gen_args(c, ch, dst, nblocks, nchan, nsamp, fs, sample_size);
    if (rc) return rc;
    if (nblocks == 0) return GPSIQ_OK;                    // an empty batch leaves the carried phases alone
    {   // descriptors quantised / evaluated on the device (gpsiq_evaldev.cpp) where that path takes the call
        int handled = 0;
        rc = gpsiq_generate_device(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, carr_phase_out, nullptr, &handled);
        if (handled) return rc;
    }
    if (c->nco_mode == GPSIQ_NCO_REFERENCE)
        return generate_reference(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, carr_phase_out);
    const char *trace_env = std::getenv("GPSIQ_TRACE");
    const bool trace = trace_env != nullptr, trace_pieces = trace && std::atoi(trace_env) >= 2;
    const double t0 = trace ? wall_ms() : 0.0;
    // continue a previous call exactly where the caller hands back what it was given
    bool cont0[GPSIQ_MAX_CHAN];
    for (int i = 0; i < nchan; ++i)
        cont0[i] = ch[i].prn > 0 && c->carry_prn[i] == ch[i].prn && c->handed[i] == ch[i].carr_phase;
    uint64_t carry[GPSIQ_MAX_CHAN] = {};
    int prev_prn[GPSIQ_MAX_CHAN] = {};
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) sample_size;
    const int piece = batch_piece_blocks(nblocks, nsamp);
    if (dst_is_device && !(blk_bytes & 15) && !((uintptr_t) dst & 3) && piece < nblocks && nsamp > 0) {
        // A long batch into device memory: quantise (= range-check), compact and upload piece k+1 on host threads under the
        // kernel of piece k; small pieces at both ends (piece_ends: nothing renders before the first piece is through, and the
        // last kernel is all that is left when the host is).  The carrier prefix goes from piece to piece exactly as it goes
        // from call to call.  A descriptor that fails its range check in a LATE piece fails the call after earlier pieces have
        // been rendered: dst and the resident set are then undefined, the carried phases untouched (include/gpsiq.h says so;
        // checking the whole timeline before the first launch was measured at 15 % of the call, 314 -> 265 G samples/s).
        bool cont[GPSIQ_MAX_CHAN];
        uint64_t seed[GPSIQ_MAX_CHAN];
        for (int i = 0; i < nchan; ++i) { cont[i] = cont0[i]; seed[i] = c->carry[i]; }
        std::vector<int> ends;
        // the quantiser is a fraction of the kernel's time (kernel-bound), but nothing renders before the first piece is through
        // it: the first piece is a sixteenth of the nominal one (64 blocks at 2.6 Msps: on the device 0.1 ms into the call), each
        // later one 2.2 x the one before (GPSIQ_TRACE=2 prints the timeline)
        piece_ends(0, nblocks, piece / 8 > 0 ? piece / 8 : 1, &ends, true);
        int longest = 0;
        for (size_t k = 0; k < ends.size(); ++k) { const int nb = ends[k] - (k ? ends[k - 1] : 0); if (nb > longest) longest = nb; }
        // one piece's worth, the context's own (the set is staged out of it at once; a fresh 7 MB vector per call was 0.25 ms of
        // zero-filling before the first descriptor was looked at); every element is written by the quantiser
        std::vector<gpsiq_qchan_t> &q = c->ref_q;
        if (q.size() < (size_t) longest * (size_t) nchan) q.resize((size_t) longest * (size_t) nchan);
        for (size_t k = 0; k < ends.size() && rc == GPSIQ_OK; ++k) {
            const int b0 = k ? ends[k - 1] : 0, nb = ends[k] - b0;
            const double tq0 = trace_pieces ? wall_ms() : 0.0;
            rc = quantize_timeline(ch + (size_t) b0 * nchan, nb, nchan, 1.0 / fs, nsamp, cont, seed, q.data(), carry, prev_prn);
            const double tq1 = trace_pieces ? wall_ms() : 0.0;
            if (rc == GPSIQ_OK) rc = set_descriptors_impl(c, q.data(), nb, nchan, nullptr, 0, true);
            const double tq2 = trace_pieces ? wall_ms() : 0.0;
            if (rc == GPSIQ_OK) rc = gpsiq_launch(c, 0, nb, nsamp, sample_size, static_cast<uint8_t *>(dst) + (size_t) b0 * blk_bytes, blk_bytes, piece_stream(c, (int) k), kAuto);
            if (trace_pieces)
                std::fprintf(stderr, "[gpsiq trace]   piece %zu, blocks [%d, %d): quantise from %.3f to %.3f ms, descriptors queued at %.3f, launched at %.3f\n",
                             k, b0, b0 + nb, tq0 - t0, tq1 - t0, tq2 - t0, wall_ms() - t0);
            for (int i = 0; i < nchan; ++i) {
                // the next piece continues a slot while it keeps its PRN and re-seeds it otherwise, as inside one timeline
                const gpsiq_chan_t *next = b0 + nb < nblocks ? &ch[(size_t) (b0 + nb) * nchan + i] : nullptr;
                cont[i] = next && next->prn > 0 && prev_prn[i] == next->prn;
                seed[i] = carry[i];
            }
        }
        char err[400] = "";
        if (rc != GPSIQ_OK) std::snprintf(err, sizeof err, "%s", gpsiq_last_error());
        int src = gpsiq_synchronize(c, c->stream);                    // on every path: the kernels write the caller's buffer
        const int src2 = gpsiq_synchronize(c, c->stream2);
        if (src == GPSIQ_OK) src = src2;
        if (src == GPSIQ_OK && hipStreamSynchronize(c->up_stream) != hipSuccess) src = fail(GPSIQ_E_DEVICE, "batch pieces: descriptor upload");
        if (src == GPSIQ_OK) for (auto &b : c->buf) b.upload_pending = false;     // the uploads (also of a set never launched on) are done
        if (rc != GPSIQ_OK) return fail(rc, "%s", err);
        if (src != GPSIQ_OK) return src;
        if (trace) std::fprintf(stderr, "[gpsiq trace] batch %d blocks in pieces of %d: whole call %.2f ms\n", nblocks, piece, wall_ms() - t0);
    } else {
        std::vector<gpsiq_qchan_t> &q = c->ref_q;
        if (q.size() < (size_t) nblocks * (size_t) nchan) q.resize((size_t) nblocks * (size_t) nchan);
        int qrc = quantize_timeline(ch, nblocks, nchan, 1.0 / fs, nsamp, cont0, c->carry, q.data(), carry, prev_prn);
        if (qrc) return qrc;
        const double t2 = trace ? wall_ms() : 0.0;
        rc = run_to_host_or_device(c, q.data(), nblocks, nchan, nsamp, sample_size, dst, dst_is_device);
        if (rc) return rc;
        if (trace)
            std::fprintf(stderr, "[gpsiq trace] batch %d blocks: quantise + carrier prefix %.2f ms, upload+kernel%s %.2f ms\n",
                         nblocks, t2 - t0, dst_is_device ? "" : "+D2H", wall_ms() - t2);
    }
    for (int i = 0; i < nchan; ++i) {
        c->carry_prn[i] = prev_prn[i];
        c->carry[i] = carry[i];
        c->handed[i] = prev_prn[i] ? carr_phase_to_double(carry[i]) : 0.0;
        if (carr_phase_out) carr_phase_out[i] = c->handed[i];
    }
    return GPSIQ_OK;
}

int gpsiq_generate_seeded(gpsiq_ctx_t *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs, int sample_size,
                          const double *carr_start, void *dst, int dst_is_device)
{
    int rc = check_gen_args(c, ch, dst, nblocks, nchan, nsamp, fs, sample_size);
    if (rc) return rc;
    if (!carr_start && nblocks) return fail(GPSIQ_E_ARG, "null start states");
    if (nblocks == 0) return GPSIQ_OK;
    for (size_t k = 0; k < (size_t) nblocks * (size_t) nchan; ++k)
        if (ch[k].prn > 0 && !(carr_start[k] >= 0.0 && carr_start[k] <= 1.0)) return fail(GPSIQ_E_RANGE, "start phase %zu outside [0, 1]", k);
    {
        int handled = 0;
        rc = gpsiq_generate_device(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, nullptr, carr_start, &handled);
        if (handled) return rc;
    }
    return generate_reference(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, nullptr, carr_start);
}

int gpsiq_generate_batch_multi(gpsiq_ctx_t *const *ctx, int ndev, const gpsiq_chan_t *ch, int nblocks, int nchan,
                               int nsamp, double fs, int sample_size, void *host_dst, void *const *dev_dst,
                               double *carr_phase_out)
{
    if (!ctx || ndev < 1 || ndev > 64) return fail(GPSIQ_E_ARG, "bad device list");
    for (int i = 0; i < ndev; ++i) {
        if (!ctx[i]) return fail(GPSIQ_E_ARG, "null context %d", i);
        for (int j = 0; j < i; ++j)
            if (ctx[j] == ctx[i]) return fail(GPSIQ_E_ARG, "context %d listed twice: the ranges are rendered concurrently, one context each", i);
    }
    if (!host_dst && !dev_dst) return fail(GPSIQ_E_ARG, "no destination");
    int rc = check_gen_args(ctx[0], ch, host_dst ? host_dst : (void *) dev_dst, nblocks, nchan, nsamp, fs, sample_size);
    if (rc) return rc;
    if (nblocks == 0) return GPSIQ_OK;
    gpsiq_ctx *c0 = ctx[0];
    const bool reference = c0->nco_mode == GPSIQ_NCO_REFERENCE;
    const size_t blk_bytes = (size_t) 2 * (size_t) nsamp * (size_t) sample_size;
    for (int i = 0; i < ndev; ++i) {
        int b0 = 0, b1 = 0;
        (void) gpsiq_shard_range(nblocks, i, ndev, &b0, &b1);
        if (b1 > b0 && !host_dst && !dev_dst[i]) return fail(GPSIQ_E_ARG, "null destination for range %d", i);
    }
    std::vector<gpsiq_qchan_t> q((size_t) nblocks * (size_t) nchan);
    if (reference) {
        // The walk is serial in time, the devices are not: the walkers (one host thread per channel, RefWalk) go through the
        // whole timeline once, and this thread hands every piece, as soon as all channels are through it, to the device that
        // owns it; device i starts as soon as the walk reaches its range, and renders under the walk of what follows.
        struct Item { const gpsiq_qchan_t *q; int b0, nb; std::vector<gpsiq_patch_t> patches; };
        struct Dev {
            gpsiq_ctx *c; int range_blocks, nchan, nsamp, ss; void *dst; int dst_is_device;
            pthread_mutex_t mu; pthread_cond_t cv; std::deque<Item> items; bool closed;
            int rc; char err[256]; pthread_t th; bool started;
        };
        std::vector<Dev> devs((size_t) ndev);
        auto body = [](void *arg) -> void * {
            Dev &d = *static_cast<Dev *>(arg);
            RefRender r;
            d.rc = r.begin(d.c, d.range_blocks, d.nchan, d.nsamp, d.ss, d.dst, d.dst_is_device);
            if (d.rc != GPSIQ_OK) std::snprintf(d.err, sizeof d.err, "%s", gpsiq_last_error());
            for (;;) {
                pthread_mutex_lock(&d.mu);
                while (d.items.empty() && !d.closed) pthread_cond_wait(&d.cv, &d.mu);
                if (d.items.empty()) { pthread_mutex_unlock(&d.mu); break; }
                Item it = std::move(d.items.front());
                d.items.pop_front();
                pthread_mutex_unlock(&d.mu);
                if (d.rc != GPSIQ_OK) continue;                       // after an error: take the rest off the queue, render nothing
                d.rc = r.piece(it.q, it.b0, it.nb, it.patches);
                if (d.rc != GPSIQ_OK) std::snprintf(d.err, sizeof d.err, "%s", gpsiq_last_error());
            }
            const int frc = r.finish();
            if (d.rc == GPSIQ_OK && frc != GPSIQ_OK) { d.rc = frc; std::snprintf(d.err, sizeof d.err, "%s", gpsiq_last_error()); }
            return nullptr;
        };
        for (int i = 0; i < ndev; ++i) {
            int b0 = 0, b1 = 0;
            (void) gpsiq_shard_range(nblocks, i, ndev, &b0, &b1);
            Dev &d = devs[(size_t) i];
            d.c = ctx[i]; d.range_blocks = b1 - b0; d.nchan = nchan; d.nsamp = nsamp; d.ss = sample_size;
            d.dst = host_dst ? (void *) (static_cast<uint8_t *>(host_dst) + (size_t) b0 * blk_bytes) : dev_dst[i];
            d.dst_is_device = host_dst ? 0 : 1;
            pthread_mutex_init(&d.mu, nullptr); pthread_cond_init(&d.cv, nullptr);
            d.closed = false; d.rc = GPSIQ_OK; d.err[0] = 0;
            d.started = d.range_blocks > 0 && pthread_create(&d.th, nullptr, body, &d) == 0;
        }
        // pieces never straddle two devices' ranges
        std::vector<int> ends, owner;
        for (int i = 0; i < ndev; ++i) {
            int r0 = 0, r1 = 0;
            (void) gpsiq_shard_range(nblocks, i, ndev, &r0, &r1);
            if (r1 == r0) continue;
            const size_t before = ends.size();
            piece_ends(r0, r1 - r0, ref_chunk_blocks(r1 - r0, nsamp), &ends, ref_kernel_bound(nsamp, nchan));
            owner.insert(owner.end(), ends.size() - before, i);
        }
        // the carrier chain: level 1 of the whole timeline on the first device (one launch; a GPU walks an hour of config 5 in
        // milliseconds where sixteen host threads took a tenth of a second), the link inside the walkers' chain tasks
        const bool dev_chain = chain_on_device(nblocks, nsamp, nchan);
        int crc = GPSIQ_OK;
        if (dev_chain) {
            crc = hipSetDevice(c0->device) == hipSuccess ? chain_reserve(c0, (size_t) nblocks * (size_t) nchan) : fail(GPSIQ_E_DEVICE, "hipSetDevice");
            if (crc == GPSIQ_OK) {
                chain_stage_inputs(c0, ch, 0, nblocks, nchan);
                crc = chain_maps_staged(c0, nblocks, nchan, fs, nsamp, nullptr, 0, nullptr);
            }
        }
        RefWalk w(ch, nblocks, nchan, 1.0 / fs, nsamp, q.data(), nullptr, nullptr, ends);
        if (dev_chain && crc == GPSIQ_OK) { w.in = c0->chain.h_in; w.maps = c0->chain.h_maps; }     // (a failed level 1: the serial walk)
        pthread_t wth;
        const bool threaded = pthread_create(&wth, nullptr, run_walk, &w) == 0;
        if (!threaded) w.run();
        int wrc = GPSIQ_OK;
        char werr[400] = "";
        int open_dev = 0;                                             // devices before this one have all their pieces
        for (size_t k = 0; k < w.npieces(); ++k) {
            if (wrc == GPSIQ_OK) {
                wrc = w.wait_piece(k);
                if (wrc != GPSIQ_OK) std::snprintf(werr, sizeof werr, "%s", w.err);
                for (int j = 0; j < ndev && wrc == GPSIQ_OK; ++j)                  // a device that has failed: stop walking for it
                    if (__atomic_load_n(&devs[(size_t) j].rc, __ATOMIC_RELAXED) != GPSIQ_OK) w.abort();
            }
            const int i = owner[k];
            for (; open_dev < i; ++open_dev) {                        // the walk has left device open_dev's range: no more pieces for it
                Dev &p = devs[(size_t) open_dev];
                pthread_mutex_lock(&p.mu); p.closed = true; pthread_cond_signal(&p.cv); pthread_mutex_unlock(&p.mu);
            }
            if (wrc != GPSIQ_OK) continue;
            Dev &d = devs[(size_t) i];
            int r0 = 0, r1 = 0;
            (void) gpsiq_shard_range(nblocks, i, ndev, &r0, &r1);
            Item it;
            const int b0 = k ? w.ends[k - 1] : 0;
            it.nb = w.ends[k] - b0;
            it.b0 = b0 - r0;
            it.q = q.data() + (size_t) b0 * nchan;
            w.take_patches(k, &it.patches, true);
            pthread_mutex_lock(&d.mu);
            d.items.push_back(std::move(it));
            pthread_cond_signal(&d.cv);
            pthread_mutex_unlock(&d.mu);
        }
        for (; open_dev < ndev; ++open_dev) {
            Dev &p = devs[(size_t) open_dev];
            pthread_mutex_lock(&p.mu); p.closed = true; pthread_cond_signal(&p.cv); pthread_mutex_unlock(&p.mu);
        }
        if (threaded) pthread_join(wth, nullptr);
        double carr[GPSIQ_MAX_CHAN];
        int prn[GPSIQ_MAX_CHAN];
        for (int i = 0; i < nchan; ++i) { carr[i] = w.carr_end[i]; prn[i] = w.last_prn[i]; }
        for (int i = 0; i < ndev; ++i) {
            Dev &d = devs[(size_t) i];
            if (d.started) pthread_join(d.th, nullptr);
            else if (d.range_blocks > 0) body(&d);
            pthread_mutex_destroy(&d.mu); pthread_cond_destroy(&d.cv);
        }
        if (wrc != GPSIQ_OK) return fail(wrc, "%s", werr);
        for (int i = 0; i < ndev; ++i)
            if (devs[(size_t) i].rc != GPSIQ_OK) return fail(devs[(size_t) i].rc, "device range %d: %s", i, devs[(size_t) i].err);
        if (carr_phase_out)
            for (int i = 0; i < nchan; ++i) carr_phase_out[i] = prn[i] ? carr[i] : ch[(size_t) (nblocks - 1) * nchan + i].carr_phase;
        return GPSIQ_OK;
    }
    // the fixed-point model: the host side once, for the whole timeline (threaded), then every device its range
    uint64_t carry[GPSIQ_MAX_CHAN] = {};
    int prev_prn[GPSIQ_MAX_CHAN] = {};
    {
        bool cont0[GPSIQ_MAX_CHAN];
        for (int i = 0; i < nchan; ++i)
            cont0[i] = ch[i].prn > 0 && c0->carry_prn[i] == ch[i].prn && c0->handed[i] == ch[i].carr_phase;
        rc = quantize_timeline(ch, nblocks, nchan, 1.0 / fs, nsamp, cont0, c0->carry, q.data(), carry, prev_prn);
    }
    if (rc) return rc;
    // one host thread per context; each renders its own contiguous range
    struct Part { gpsiq_ctx *c; const gpsiq_qchan_t *q; int nb, nchan, nsamp, ss; void *dst; int dst_is_device;
                  int rc; char err[256]; pthread_t th; bool started; };
    std::vector<Part> parts((size_t) ndev);
    for (int i = 0; i < ndev; ++i) {
        int b0 = 0, b1 = 0;
        (void) gpsiq_shard_range(nblocks, i, ndev, &b0, &b1);
        Part &p = parts[(size_t) i];
        p.c = ctx[i]; p.q = q.data() + (size_t) b0 * nchan; p.nb = b1 - b0; p.nchan = nchan; p.nsamp = nsamp; p.ss = sample_size;
        p.dst = host_dst ? (void *) (static_cast<uint8_t *>(host_dst) + (size_t) b0 * blk_bytes) : dev_dst[i];
        p.dst_is_device = host_dst ? 0 : 1;
        p.rc = GPSIQ_OK; p.err[0] = 0; p.started = false;
    }
    auto body = [](void *arg) -> void * {
        Part &p = *static_cast<Part *>(arg);
        if (p.nb > 0) {
            p.rc = run_to_host_or_device(p.c, p.q, p.nb, p.nchan, p.nsamp, p.ss, p.dst, p.dst_is_device);
            if (p.rc != GPSIQ_OK) std::snprintf(p.err, sizeof p.err, "%s", gpsiq_last_error());
        }
        return nullptr;
    };
    for (int i = 1; i < ndev; ++i)
        parts[(size_t) i].started = pthread_create(&parts[(size_t) i].th, nullptr, body, &parts[(size_t) i]) == 0;
    body(&parts[0]);
    for (int i = 1; i < ndev; ++i) {
        if (parts[(size_t) i].started) pthread_join(parts[(size_t) i].th, nullptr);
        else body(&parts[(size_t) i]);                       // could not start a thread: do it here
    }
    for (int i = 0; i < ndev; ++i)
        if (parts[(size_t) i].rc != GPSIQ_OK) return fail(parts[(size_t) i].rc, "device range %d: %s", i, parts[(size_t) i].err);
    for (int i = 0; i < nchan; ++i) {
        c0->carry_prn[i] = prev_prn[i];
        c0->carry[i] = carry[i];
        c0->handed[i] = prev_prn[i] ? carr_phase_to_double(carry[i]) : 0.0;
        if (carr_phase_out) carr_phase_out[i] = c0->handed[i];
    }
    return GPSIQ_OK;
}

}  // extern "C"

// ---- what gpsiq_evaldev.cpp uses of this file ---------------------------------------------------------------------------------
double gpsiq_wall_ms() { return wall_ms(); }
int gpsiq_wait_idle(gpsiq_ctx::DescBuf &b) { return wait_idle(b); }
int gpsiq_mark_use(gpsiq_ctx::DescBuf &b, hipStream_t s) { return mark_use(b, s); }
int gpsiq_ensure_out(gpsiq_ctx *c, size_t bytes) { return ensure_out(c, bytes); }
hipStream_t gpsiq_piece_stream(gpsiq_ctx *c, int k) { return piece_stream(c, k); }
int gpsiq_chain_reserve(gpsiq_ctx *c, size_t n) { return chain_reserve(c, n); }
double gpsiq_rate_kernel() { return rate_kernel(); }
void gpsiq_note_kernel_rate(double channel_samples_per_s)
{
    if (!(channel_samples_per_s > 1e10 && channel_samples_per_s < 1e15)) return;
    const double old = g_rate_kernel_measured.load(std::memory_order_relaxed);
    g_rate_kernel_measured.store(old > 0.0 ? 0.75 * old + 0.25 * channel_samples_per_s : channel_samples_per_s, std::memory_order_relaxed);
}
int gpsiq_generate_reference_host(gpsiq_ctx *c, const gpsiq_chan_t *ch, int nblocks, int nchan, int nsamp, double fs,
                                  int sample_size, void *dst, int dst_is_device, double *carr_phase_out, const double *seeds)
{
    return generate_reference(c, ch, nblocks, nchan, nsamp, fs, sample_size, dst, dst_is_device, carr_phase_out, seeds);
}

// ---- the one exported entry to the plumbing (gpsiq_plumbing.h) --------------------------------------------------------------
extern "C" void *gpsiq_plumbing(const char *name)
{
    static const struct { const char *name; void *fn; } table[] = {
#define GPSIQ_P(f) {#f, reinterpret_cast<void *>(&f)}
        GPSIQ_P(gpsiq_reference_chain), GPSIQ_P(gpsiq_reference_seeded), GPSIQ_P(gpsiq_reference_stats), GPSIQ_P(gpsiq_chain_inputs),
        GPSIQ_P(gpsiq_chain_maps), GPSIQ_P(gpsiq_chain_link), GPSIQ_P(gpsiq_chain_summary), GPSIQ_P(gpsiq_chain_fold), GPSIQ_P(gpsiq_chain_stats),
        GPSIQ_P(gpsiq_chain_maps_device), GPSIQ_P(gpsiq_chain_range), GPSIQ_P(gpsiq_chain_range_fold), GPSIQ_P(gpsiq_time_launches), GPSIQ_P(gpsiq_num_variants), GPSIQ_P(gpsiq_variant_name),
        GPSIQ_P(gpsiq_device_eval_stats), GPSIQ_P(gpsiq_device_eval_host_ms),
        GPSIQ_P(gpsiq_prn_code), GPSIQ_P(gpsiq_carrier_table), GPSIQ_P(gpsiq_generate_seeded),
#undef GPSIQ_P
        // the internals libgpsiq_rows.so runs on (gpsiq_rows_link.cpp): one pool, one quantiser, one error text per thread
        {"set_error", reinterpret_cast<void *>(&gpsiq::set_error)}, {"parallel_for", reinterpret_cast<void *>(&gpsiq::parallel_for)},
        {"quantize_one", reinterpret_cast<void *>(&gpsiq::quantize_one)}, {"chain_carrier", reinterpret_cast<void *>(&gpsiq::chain_carrier)},
    };
    if (!name) return nullptr;
    for (const auto &e : table)
        if (!std::strcmp(e.name, name)) return e.fn;
    return nullptr;
}
