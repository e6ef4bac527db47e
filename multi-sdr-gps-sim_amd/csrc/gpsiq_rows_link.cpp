// gpsiq_rows_link.cpp -- libgpsiq_rows.so's end of its link to libgpsiq.so.  The rows either side of the hot path (gpsiq_refresh.cpp,
// gpsiq_nav.cpp, gpsiq_rinex.cpp: include/gpsiq_rows.h, gpsiq_extras.h) are host code in a library of their own, so that the
// library a maintainer binds for the sample loop exports the boundary and nothing else.  They run on four of that library's
// internals -- the worker pool, the quantiser, the exact carrier prefix, the calling thread's error text -- which libgpsiq.so hands
// out by name through its one plumbing entry (csrc/gpsiq_plumbing.h): one pool, one quantiser and one gpsiq_last_error() for both.
#include "gpsiq_internal.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>

namespace gpsiq {

namespace {
template <class F>
F core(const char *name)
{
    void *p = gpsiq_plumbing(name);
    if (!p) { std::fprintf(stderr, "libgpsiq_rows: libgpsiq.so has no plumbing entry '%s' (libraries of two different builds?)\n", name); std::abort(); }
    return reinterpret_cast<F>(p);
}
}  // namespace

int fail(int code, const char *fmt, ...)
{
    static const auto f = core<int (*)(int, const char *)>("set_error");
    char text[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(text, sizeof text, fmt, ap);
    va_end(ap);
    return f(code, text);
}

void parallel_for(int n, int nthreads, int grain, void (*fn)(void *ctx, int begin, int end), void *ctx)
{
    static const auto f = core<decltype(&parallel_for)>("parallel_for");
    f(n, nthreads, grain, fn, ctx);
}

int quantize_one(const gpsiq_chan_t &ch, double delt, int nsamp, const uint64_t *carry_in, gpsiq_qchan_t *q, uint64_t *carry_out)
{
    static const auto f = core<decltype(&quantize_one)>("quantize_one");
    return f(ch, delt, nsamp, carry_in, q, carry_out);
}

void chain_carrier(gpsiq_qchan_t *q, int nblocks, int nchan, int nsamp, const bool *cont0, const uint64_t *carry0, uint64_t *carry_end, int *last_prn)
{
    static const auto f = core<decltype(&chain_carrier)>("chain_carrier");
    f(q, nblocks, nchan, nsamp, cont0, carry0, carry_end, last_prn);
}

}  // namespace gpsiq
