// gpsiq_refresh.cpp — the per-block host refresh of the reference, batched.
//
// SURVEY.md section 8f rank 1.  Before every pass of the sample loop the reference computes,
// per visible satellite, the pseudorange at the new receiver time and position
// (computeRange, gps.c:1972-2026 -> satpos gps.c:508-611, xyz2llh gps.c:361-406, ltcmat
// gps.c:449-469, ecef2neu gps.c:476-482, neu2azel gps.c:488-499, ionosphericDelay
// gps.c:1893-1964), turns the range difference to the previous block into Doppler and code
// phase (computeCodePhase, gps.c:2033-2064) and derives the signal gain (gps.c:2749-2763).
// At >100 000x real time that loop (10 Hz, tens of microseconds per block on one core) is
// what limits a time-sharded run, so here it runs for a whole batch of blocks, spread over
// host threads.  Every formula keeps the reference's operand order (and this file is built
// with -ffp-contract=off), so with the same libm the descriptors are bit-identical to the
// reference's; tests/test_refresh.py checks that against the reference's own lines.
#include "gpsiq_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

namespace {

// constants as the reference defines them (gps.h:61-65, 93-112)
constexpr double kPi = 3.1415926535898;
constexpr double kR2D = 57.2957795131;
constexpr double kC = 2.99792458e8;
constexpr double kLambdaL1 = 0.190293672798365;
constexpr double kOmegaEarth = 7.2921151467e-5;
constexpr double kWgs84A = 6378137.0;
constexpr double kWgs84E = 0.0818191908426;
constexpr double kSecWeek = 604800.0, kSecHalfWeek = 302400.0, kSecDay = 86400.0;
constexpr double kCodeFreq = 1.023e6;
constexpr double kCarrToCode = 1.0 / 1540.0;

// receiver antenna attenuation in dB per 5 degrees of boresight angle (gps.c:216-221)
const double kAntPatDb[37] = {
    0.00, 0.00, 0.22, 0.44, 0.67, 1.11, 1.56, 2.00, 2.44, 2.89, 3.56, 4.22,
    4.89, 5.56, 6.22, 6.89, 7.56, 8.22, 8.89, 9.78, 10.67, 11.56, 12.44, 13.33,
    14.44, 15.56, 16.67, 17.78, 18.89, 20.00, 21.33, 22.67, 24.00, 25.56, 27.33, 29.33,
    31.56};

struct GpsTime { int week; double sec; };

// gps.c:1105-1124
GpsTime advance(GpsTime g, double dt)
{
    g.sec = g.sec + dt;
    g.sec = std::round(g.sec * 1000.0) / 1000.0;
    while (g.sec >= kSecWeek) { g.sec -= kSecWeek; g.week++; }
    while (g.sec < 0.0) { g.sec += kSecWeek; g.week--; }
    return g;
}

// sin and cos of one argument must come from two separate libm calls: the reference's own
// build (Makefile:5, gcc -Og) does not fuse them, whereas an optimising build may turn the pair
// into one glibc sincos(), whose results are not always bit-identical (one ulp in ~0.2 % of
// the ranges).  Routing the calls through non-inlinable functions keeps any compiler from
// fusing them, so the descriptors match the reference whatever builds this file.
__attribute__((noinline)) double sin_only(double x) { return std::sin(x); }
__attribute__((noinline)) double cos_only(double x) { return std::cos(x); }
inline void sin_cos(double x, double *s, double *c) { *s = sin_only(x); *c = cos_only(x); }

inline double norm3(const double *v) { return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]); }   // gps.c:255-257

// Receiver geometry of one block: geodetic position and the ECEF->NEU rotation
// (gps.c:361-406, gps.c:449-469).  The reference recomputes it per satellite; the inputs
// are the same, so once per block gives the same numbers.
struct Site { double llh[3]; double t[3][3]; };

Site site_from_ecef(const double *xyz)
{
    Site s;
    const double a = kWgs84A, e = kWgs84E, eps = 1.0e-3, e2 = e * e;
    if (norm3(xyz) < eps) {
        s.llh[0] = 0.0; s.llh[1] = 0.0; s.llh[2] = -a;
    } else {
        const double x = xyz[0], y = xyz[1], z = xyz[2];
        const double rho2 = x * x + y * y;
        double dz = e2 * z, zdz, nh, n;
        for (;;) {
            zdz = z + dz;
            nh = std::sqrt(rho2 + zdz * zdz);
            const double slat = zdz / nh;
            n = a / std::sqrt(1.0 - e2 * slat * slat);
            const double dz_new = n * e2 * slat;
            if (std::fabs(dz - dz_new) < eps) break;
            dz = dz_new;
        }
        s.llh[0] = std::atan2(zdz, std::sqrt(rho2));
        s.llh[1] = std::atan2(y, x);
        s.llh[2] = nh - n;
    }
    double slat, clat, slon, clon;
    sin_cos(s.llh[0], &slat, &clat);
    sin_cos(s.llh[1], &slon, &clon);
    s.t[0][0] = -slat * clon; s.t[0][1] = -slat * slon; s.t[0][2] = clat;
    s.t[1][0] = -slon;        s.t[1][1] = clon;         s.t[1][2] = 0.0;
    s.t[2][0] = clat * clon;  s.t[2][1] = clat * slon;  s.t[2][2] = slat;
    return s;
}

struct SvState { double pos[3], vel[3], clk[2]; };

// Broadcast-ephemeris orbit, velocity and clock (gps.c:508-611).
SvState sv_state(const gpsiq_ephem_t &e, double sec)
{
    SvState s;
    double tk = sec - e.toe_sec;
    if (tk > kSecHalfWeek) tk -= kSecWeek;
    else if (tk < -kSecHalfWeek) tk += kSecWeek;

    const double mk = e.m0 + e.n * tk;
    double ek = mk, ekold = ek + 1.0, one_m_ecos = 0;
    while (std::fabs(ek - ekold) > 1.0E-14) {          // Kepler, Newton steps
        ekold = ek;
        double so, co;
        sin_cos(ekold, &so, &co);
        one_m_ecos = 1.0 - e.ecc * co;
        ek = ek + (mk - ekold + e.ecc * so) / one_m_ecos;
    }
    double sek, cek;
    sin_cos(ek, &sek, &cek);
    const double ekdot = e.n / one_m_ecos;
    const double relativistic = -4.442807633E-10 * e.ecc * e.sqrta * sek;

    const double pk = std::atan2(e.sq1e2 * sek, cek - e.ecc) + e.aop;
    const double pkdot = e.sq1e2 * ekdot / one_m_ecos;
    double s2pk, c2pk;
    sin_cos(2.0 * pk, &s2pk, &c2pk);

    const double uk = pk + e.cus * s2pk + e.cuc * c2pk;
    double suk, cuk;
    sin_cos(uk, &suk, &cuk);
    const double ukdot = pkdot * (1.0 + 2.0 * (e.cus * c2pk - e.cuc * s2pk));

    const double rk = e.A * one_m_ecos + e.crc * c2pk + e.crs * s2pk;
    const double rkdot = e.A * e.ecc * sek * ekdot + 2.0 * pkdot * (e.crs * c2pk - e.crc * s2pk);

    const double ik = e.inc0 + e.idot * tk + e.cic * c2pk + e.cis * s2pk;
    double sik, cik;
    sin_cos(ik, &sik, &cik);
    const double ikdot = e.idot + 2.0 * pkdot * (e.cis * c2pk - e.cic * s2pk);

    const double xpk = rk * cuk, ypk = rk * suk;
    const double xpkdot = rkdot * cuk - ypk * ukdot;
    const double ypkdot = rkdot * suk + xpk * ukdot;

    const double ok = e.omg0 + tk * e.omgkdot - kOmegaEarth * e.toe_sec;
    double sok, cok;
    sin_cos(ok, &sok, &cok);

    s.pos[0] = xpk * cok - ypk * cik * sok;
    s.pos[1] = xpk * sok + ypk * cik * cok;
    s.pos[2] = ypk * sik;

    const double tmp = ypkdot * cik - ypk * sik * ikdot;
    s.vel[0] = -e.omgkdot * s.pos[1] + xpkdot * cok - tmp * sok;
    s.vel[1] = e.omgkdot * s.pos[0] + xpkdot * sok + tmp * cok;
    s.vel[2] = ypk * cik * ikdot + ypkdot * sik;

    tk = sec - e.toc_sec;
    if (tk > kSecHalfWeek) tk -= kSecWeek;
    else if (tk < -kSecHalfWeek) tk += kSecWeek;
    s.clk[0] = e.af0 + tk * (e.af1 + tk * e.af2) + relativistic - e.tgd;
    s.clk[1] = e.af1 + 2.0 * tk * e.af2;
    return s;
}

// Klobuchar model (gps.c:1893-1964).
double iono_delay(const gpsiq_iono_t &io, double sec, const double *llh, const double *azel)
{
    if (!io.enable) return 0.0;
    const double E = azel[1] / kPi, phi_u = llh[0] / kPi, lam_u = llh[1] / kPi;
    const double F = 1.0 + 16.0 * std::pow((0.53 - E), 3.0);
    if (!io.vflg) return F * 5.0e-9 * kC;

    const double psi = 0.0137 / (E + 0.11) - 0.022;
    double saz, caz;
    sin_cos(azel[0], &saz, &caz);
    double phi_i = phi_u + psi * caz;
    if (phi_i > 0.416) phi_i = 0.416;
    else if (phi_i < -0.416) phi_i = -0.416;
    const double lam_i = lam_u + psi * saz / cos_only(phi_i * kPi);
    const double phi_m = phi_i + 0.064 * cos_only((lam_i - 1.617) * kPi);
    const double phi_m2 = phi_m * phi_m, phi_m3 = phi_m2 * phi_m;
    double amp = io.alpha[0] + io.alpha[1] * phi_m + io.alpha[2] * phi_m2 + io.alpha[3] * phi_m3;
    if (amp < 0.0) amp = 0.0;
    double per = io.beta[0] + io.beta[1] * phi_m + io.beta[2] * phi_m2 + io.beta[3] * phi_m3;
    if (per < 72000.0) per = 72000.0;
    double t = kSecDay / 2.0 * lam_i + sec;
    while (t >= kSecDay) t -= kSecDay;
    while (t < 0) t += kSecDay;
    const double X = 2.0 * kPi * (t - 50400.0) / per;
    if (std::fabs(X) < 1.57) {
        const double X2 = X * X, X4 = X2 * X2;
        return F * (5.0e-9 + amp * (1.0 - X2 / 2.0 + X4 / 24.0)) * kC;
    }
    return F * 5.0e-9 * kC;
}

struct Range { double prange, d, az, el; };

// Light-time and Earth-rotation corrected pseudorange, azimuth/elevation (gps.c:1972-2026).
Range range_to(const gpsiq_ephem_t &e, const gpsiq_iono_t &io, double sec, const double *xyz, const Site &site)
{
    SvState s = sv_state(e, sec);
    double los[3] = {s.pos[0] - xyz[0], s.pos[1] - xyz[1], s.pos[2] - xyz[2]};
    const double tau = norm3(los) / kC;
    s.pos[0] -= s.vel[0] * tau;
    s.pos[1] -= s.vel[1] * tau;
    s.pos[2] -= s.vel[2] * tau;
    const double xrot = s.pos[0] + s.pos[1] * kOmegaEarth * tau;
    const double yrot = s.pos[1] - s.pos[0] * kOmegaEarth * tau;
    s.pos[0] = xrot;
    s.pos[1] = yrot;
    los[0] = s.pos[0] - xyz[0]; los[1] = s.pos[1] - xyz[1]; los[2] = s.pos[2] - xyz[2];
    Range r;
    r.d = norm3(los);
    r.prange = r.d - kC * s.clk[0];
    double neu[3];
    for (int k = 0; k < 3; ++k)
        neu[k] = site.t[k][0] * los[0] + site.t[k][1] * los[1] + site.t[k][2] * los[2];   // gps.c:476-482
    double azel[2];
    azel[0] = std::atan2(neu[1], neu[0]);                                                  // gps.c:488-499
    if (azel[0] < 0.0) azel[0] += (2.0 * kPi);
    azel[1] = std::atan2(neu[2], std::sqrt(neu[0] * neu[0] + neu[1] * neu[1]));
    r.az = azel[0];
    r.el = azel[1];
    r.prange += iono_delay(io, sec, site.llh, azel);
    return r;
}

// The same for all channels of one block at once, stage by stage.  Per channel the operations and their order are exactly
// those of sv_state / range_to / iono_delay above (this file is built with -ffp-contract=off; nothing is reassociated), so
// every double is bit-identical to the one-channel functions -- tests/test_refresh.py holds both against the reference's
// lines.  The point is the dependency chain: one channel is ~25 libm calls of which each needs the one before (Kepler
// iteration -> true anomaly -> argument of latitude -> ...), ~0.5 us of latency with the core mostly idle; with the
// stages looped over up to 16 independent channels the out-of-order core overlaps the calls of neighbouring channels.
constexpr int kLanes = GPSIQ_MAX_CHAN;

void ranges_for_block(const gpsiq_ephem_t *eph, const int *ch, int n, const gpsiq_iono_t &io, double sec, const double *xyz,
                      const Site &site, Range *out /* indexed by channel */)
{
    double tk[kLanes], mk[kLanes], ek[kLanes], ekold[kLanes], one_m_ecos[kLanes], so[kLanes], co[kLanes];
    bool act[kLanes];
    for (int k = 0; k < n; ++k) {
        const gpsiq_ephem_t &e = eph[ch[k]];
        double t = sec - e.toe_sec;
        if (t > kSecHalfWeek) t -= kSecWeek;
        else if (t < -kSecHalfWeek) t += kSecWeek;
        tk[k] = t;
        mk[k] = e.m0 + e.n * t;
        ek[k] = mk[k]; ekold[k] = ek[k] + 1.0; one_m_ecos[k] = 0;
        act[k] = true;
    }
    for (;;) {                                                       // Kepler: every channel stops on its own criterion
        int live = 0;
        for (int k = 0; k < n; ++k) {
            act[k] = act[k] && std::fabs(ek[k] - ekold[k]) > 1.0E-14;
            live += act[k];
        }
        if (!live) break;
        for (int k = 0; k < n; ++k) if (act[k]) { ekold[k] = ek[k]; so[k] = sin_only(ekold[k]); }
        for (int k = 0; k < n; ++k) if (act[k]) co[k] = cos_only(ekold[k]);
        for (int k = 0; k < n; ++k)
            if (act[k]) {
                const gpsiq_ephem_t &e = eph[ch[k]];
                one_m_ecos[k] = 1.0 - e.ecc * co[k];
                ek[k] = ek[k] + (mk[k] - ekold[k] + e.ecc * so[k]) / one_m_ecos[k];
            }
    }
    double sek[kLanes], cek[kLanes], pk[kLanes], s2pk[kLanes], c2pk[kLanes], uk[kLanes], suk[kLanes], cuk[kLanes];
    double ik[kLanes], sik[kLanes], cik[kLanes], ok[kLanes], sok[kLanes], cok[kLanes];
    // the last Newton step usually leaves ek where it was: sin and cos of the same argument are the loop's own
    for (int k = 0; k < n; ++k) sek[k] = ek[k] == ekold[k] ? so[k] : sin_only(ek[k]);
    for (int k = 0; k < n; ++k) cek[k] = ek[k] == ekold[k] ? co[k] : cos_only(ek[k]);
    for (int k = 0; k < n; ++k) { const gpsiq_ephem_t &e = eph[ch[k]]; pk[k] = std::atan2(e.sq1e2 * sek[k], cek[k] - e.ecc) + e.aop; }
    for (int k = 0; k < n; ++k) s2pk[k] = sin_only(2.0 * pk[k]);
    for (int k = 0; k < n; ++k) c2pk[k] = cos_only(2.0 * pk[k]);
    for (int k = 0; k < n; ++k) { const gpsiq_ephem_t &e = eph[ch[k]]; uk[k] = pk[k] + e.cus * s2pk[k] + e.cuc * c2pk[k]; }
    for (int k = 0; k < n; ++k) suk[k] = sin_only(uk[k]);
    for (int k = 0; k < n; ++k) cuk[k] = cos_only(uk[k]);
    for (int k = 0; k < n; ++k) { const gpsiq_ephem_t &e = eph[ch[k]]; ik[k] = e.inc0 + e.idot * tk[k] + e.cic * c2pk[k] + e.cis * s2pk[k]; }
    for (int k = 0; k < n; ++k) sik[k] = sin_only(ik[k]);
    for (int k = 0; k < n; ++k) cik[k] = cos_only(ik[k]);
    for (int k = 0; k < n; ++k) { const gpsiq_ephem_t &e = eph[ch[k]]; ok[k] = e.omg0 + tk[k] * e.omgkdot - kOmegaEarth * e.toe_sec; }
    for (int k = 0; k < n; ++k) sok[k] = sin_only(ok[k]);
    for (int k = 0; k < n; ++k) cok[k] = cos_only(ok[k]);

    double los[kLanes][3], d[kLanes], prange[kLanes], az[kLanes], el[kLanes], neu2[kLanes], rho[kLanes];
    for (int k = 0; k < n; ++k) {
        const gpsiq_ephem_t &e = eph[ch[k]];
        const double ekdot = e.n / one_m_ecos[k];
        const double relativistic = -4.442807633E-10 * e.ecc * e.sqrta * sek[k];
        const double pkdot = e.sq1e2 * ekdot / one_m_ecos[k];
        const double ukdot = pkdot * (1.0 + 2.0 * (e.cus * c2pk[k] - e.cuc * s2pk[k]));
        const double rk = e.A * one_m_ecos[k] + e.crc * c2pk[k] + e.crs * s2pk[k];
        const double rkdot = e.A * e.ecc * sek[k] * ekdot + 2.0 * pkdot * (e.crs * c2pk[k] - e.crc * s2pk[k]);
        const double ikdot = e.idot + 2.0 * pkdot * (e.cis * c2pk[k] - e.cic * s2pk[k]);
        const double xpk = rk * cuk[k], ypk = rk * suk[k];
        const double xpkdot = rkdot * cuk[k] - ypk * ukdot;
        const double ypkdot = rkdot * suk[k] + xpk * ukdot;
        double pos[3], vel[3];
        pos[0] = xpk * cok[k] - ypk * cik[k] * sok[k];
        pos[1] = xpk * sok[k] + ypk * cik[k] * cok[k];
        pos[2] = ypk * sik[k];
        const double tmp = ypkdot * cik[k] - ypk * sik[k] * ikdot;
        vel[0] = -e.omgkdot * pos[1] + xpkdot * cok[k] - tmp * sok[k];
        vel[1] = e.omgkdot * pos[0] + xpkdot * sok[k] + tmp * cok[k];
        vel[2] = ypk * cik[k] * ikdot + ypkdot * sik[k];
        double tc = sec - e.toc_sec;
        if (tc > kSecHalfWeek) tc -= kSecWeek;
        else if (tc < -kSecHalfWeek) tc += kSecWeek;
        const double clk0 = e.af0 + tc * (e.af1 + tc * e.af2) + relativistic - e.tgd;
        // range_to
        double l0[3] = {pos[0] - xyz[0], pos[1] - xyz[1], pos[2] - xyz[2]};
        const double tau = norm3(l0) / kC;
        pos[0] -= vel[0] * tau;
        pos[1] -= vel[1] * tau;
        pos[2] -= vel[2] * tau;
        const double xrot = pos[0] + pos[1] * kOmegaEarth * tau;
        const double yrot = pos[1] - pos[0] * kOmegaEarth * tau;
        pos[0] = xrot;
        pos[1] = yrot;
        los[k][0] = pos[0] - xyz[0]; los[k][1] = pos[1] - xyz[1]; los[k][2] = pos[2] - xyz[2];
        d[k] = norm3(los[k]);
        prange[k] = d[k] - kC * clk0;
        double neu[3];
        for (int m = 0; m < 3; ++m)
            neu[m] = site.t[m][0] * los[k][0] + site.t[m][1] * los[k][1] + site.t[m][2] * los[k][2];
        los[k][0] = neu[0]; los[k][1] = neu[1];                     // kept for the two atan2 stages
        neu2[k] = neu[2];
        rho[k] = std::sqrt(neu[0] * neu[0] + neu[1] * neu[1]);
    }
    for (int k = 0; k < n; ++k) { az[k] = std::atan2(los[k][1], los[k][0]); if (az[k] < 0.0) az[k] += (2.0 * kPi); }
    for (int k = 0; k < n; ++k) el[k] = std::atan2(neu2[k], rho[k]);
    // iono_delay
    if (io.enable) {
        const double phi_u = site.llh[0] / kPi, lam_u = site.llh[1] / kPi;
        double F[kLanes];
        for (int k = 0; k < n; ++k) F[k] = 1.0 + 16.0 * std::pow((0.53 - el[k] / kPi), 3.0);
        if (!io.vflg) {
            for (int k = 0; k < n; ++k) prange[k] += F[k] * 5.0e-9 * kC;
        } else {
            double saz[kLanes], caz[kLanes], phi_i[kLanes], psi[kLanes], cphi[kLanes], lam_i[kLanes], cl[kLanes];
            for (int k = 0; k < n; ++k) saz[k] = sin_only(az[k]);
            for (int k = 0; k < n; ++k) caz[k] = cos_only(az[k]);
            for (int k = 0; k < n; ++k) {
                psi[k] = 0.0137 / (el[k] / kPi + 0.11) - 0.022;
                double p = phi_u + psi[k] * caz[k];
                if (p > 0.416) p = 0.416;
                else if (p < -0.416) p = -0.416;
                phi_i[k] = p;
            }
            for (int k = 0; k < n; ++k) cphi[k] = cos_only(phi_i[k] * kPi);
            for (int k = 0; k < n; ++k) lam_i[k] = lam_u + psi[k] * saz[k] / cphi[k];
            for (int k = 0; k < n; ++k) cl[k] = cos_only((lam_i[k] - 1.617) * kPi);
            for (int k = 0; k < n; ++k) {
                const double phi_m = phi_i[k] + 0.064 * cl[k];
                const double phi_m2 = phi_m * phi_m, phi_m3 = phi_m2 * phi_m;
                double amp = io.alpha[0] + io.alpha[1] * phi_m + io.alpha[2] * phi_m2 + io.alpha[3] * phi_m3;
                if (amp < 0.0) amp = 0.0;
                double per = io.beta[0] + io.beta[1] * phi_m + io.beta[2] * phi_m2 + io.beta[3] * phi_m3;
                if (per < 72000.0) per = 72000.0;
                double t = kSecDay / 2.0 * lam_i[k] + sec;
                while (t >= kSecDay) t -= kSecDay;
                while (t < 0) t += kSecDay;
                const double X = 2.0 * kPi * (t - 50400.0) / per;
                if (std::fabs(X) < 1.57) {
                    const double X2 = X * X, X4 = X2 * X2;
                    prange[k] += F[k] * (5.0e-9 + amp * (1.0 - X2 / 2.0 + X4 / 24.0)) * kC;
                } else {
                    prange[k] += F[k] * 5.0e-9 * kC;
                }
            }
        }
    } else {
        for (int k = 0; k < n; ++k) prange[k] += 0.0;
    }
    for (int k = 0; k < n; ++k) { Range &r = out[ch[k]]; r.prange = prange[k]; r.d = d[k]; r.az = az[k]; r.el = el[k]; }
}

struct Job {
    const gpsiq_ephem_t *eph; const gpsiq_iono_t *iono; const double *xyz; const GpsTime *t;
    int nchan, gain_x2; const gpsiq_track_t *trk; const double *ant_pat;
    Range *rng;            // [nblocks + 1][nchan], row 0 = the state before the batch
    gpsiq_chan_t *out;
    int pass;
    // several navigation-message epochs in one batch: epoch e covers blocks [first[e], first[e+1]) and reads its
    // word buffer and g0 from trk[e*nchan + c]; prn / rho0 / carr_phase always come from trk[0..nchan)
    const int *first = nullptr; int nepochs = 1;
    // fused with the quantiser (gpsiq_refresh_epochs_quantized): the descriptor of a block is built on the stack and
    // quantised at once; out is not written
    gpsiq_qchan_t *qout = nullptr; double delt = 0.0; int nsamp = 0;
    mutable int rc = GPSIQ_OK; mutable char err[320] = "";
};

void work(void *p, int b0, int b1)
{
    const Job &j = *static_cast<const Job *>(p);
    Site site = {};
    int active[kLanes], nact = 0;
    for (int c = 0; c < j.nchan; ++c)
        if (j.trk[c].prn > 0) active[nact++] = c;
    const bool scalar = std::getenv("GPSIQ_REFRESH_SCALAR") != nullptr;      // A/B knob (read per call): one channel at a time
    for (int b = b0; b < b1; ++b) {
        if (j.pass == 0) {                                   // ranges: independent per block
            // a receiver that has not moved since the block before (every static scenario) keeps its geodetic position
            // and rotation: same input, same output
            const double *pos = j.xyz + 3 * (size_t) b;
            if (b == b0 || pos[0] != pos[-3] || pos[1] != pos[-2] || pos[2] != pos[-1]) site = site_from_ecef(pos);
            if (scalar) {
                for (int c = 0; c < j.nchan; ++c)
                    if (j.trk[c].prn > 0)
                        j.rng[(size_t) (b + 1) * j.nchan + c] = range_to(j.eph[c], *j.iono, j.t[b + 1].sec, pos, site);
            } else if (nact) {
                ranges_for_block(j.eph, active, nact, *j.iono, j.t[b + 1].sec, pos, site, &j.rng[(size_t) (b + 1) * j.nchan]);
            }
        } else {                                             // code phase, Doppler, gain
            int ep = 0;
            if (j.nepochs > 1) { while (ep + 1 < j.nepochs && j.first[ep + 1] <= b) ++ep; }
            const gpsiq_track_t *nav = j.trk + (size_t) ep * j.nchan;      // word buffer and g0 of this block's epoch
            gpsiq_chan_t tmp;
            for (int c = 0; c < j.nchan; ++c) {
                gpsiq_chan_t &o = j.qout ? tmp : j.out[(size_t) b * j.nchan + c];
                std::memset(&o, 0, sizeof o);
                o.prn = j.trk[c].prn > 0 ? j.trk[c].prn : 0;
                if (o.prn == 0) {
                    if (j.qout) std::memset(&j.qout[(size_t) b * j.nchan + c], 0, sizeof(gpsiq_qchan_t));
                    continue;
                }
                const Range &r0 = j.rng[(size_t) b * j.nchan + c], &r1 = j.rng[(size_t) (b + 1) * j.nchan + c];
                // time of the previous range (chan.rho0.g): the carried-in one for the first block
                const GpsTime t0 = b == 0 ? GpsTime{j.trk[c].rho0_week, j.trk[c].rho0_sec} : j.t[b];
                // gps.c:2033-2064
                const double rhorate = (r1.prange - r0.prange) / 0.1;
                o.f_carr = -rhorate / kLambdaL1;
                o.f_code = kCodeFreq + o.f_carr * kCarrToCode;
                double dtg = t0.sec - nav[c].g0_sec;
                dtg += (double) (t0.week - nav[c].g0_week) * kSecWeek;
                const double ms = ((dtg + 6.0) - r0.prange / kC) * 1000.0;
                int ims = (int) ms;
                o.code_phase = (ms - (double) ims) * GPSIQ_CA_SEQ_LEN;
                o.iword = ims / 600; ims -= o.iword * 600;
                o.ibit = ims / 20;   ims -= o.ibit * 20;
                o.icode = ims;
                o.carr_phase = j.trk[c].carr_phase;
                // gps.c:2749-2763
                const double path_loss = 20200000.0 / r1.d;
                const int ibs = (int) ((90.0 - r1.el * kR2D) / 5.0);
                double g = path_loss * j.ant_pat[ibs < 0 ? 0 : ibs > 36 ? 36 : ibs];
                if (j.gain_x2) g *= 2;
                o.gain = g;
                std::memcpy(o.dwrd, nav[c].dwrd, sizeof o.dwrd);
                if (j.qout) {
                    const int rc = gpsiq::quantize_one(o, j.delt, j.nsamp, nullptr, &j.qout[(size_t) b * j.nchan + c], nullptr);
                    if (rc != GPSIQ_OK && __sync_bool_compare_and_swap(&j.rc, GPSIQ_OK, rc))
                        std::snprintf(j.err, sizeof j.err, "block %d: %.280s", b, gpsiq_last_error());
                }
            }
        }
    }
}

}  // namespace

using namespace gpsiq;

extern "C" {

int gpsiq_track_init(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                     const double xyz[3], gpsiq_track_t *trk, int nchan)
{
    if (!eph || !iono || !xyz || !trk) return fail(GPSIQ_E_ARG, "null argument");
    if (nchan < 1 || nchan > GPSIQ_MAX_CHAN) return fail(GPSIQ_E_ARG, "nchan %d outside 1..%d", nchan, GPSIQ_MAX_CHAN);
    const Site site = site_from_ecef(xyz);
    const double origin[3] = {0.0, 0.0, 0.0};
    const Site site0 = site_from_ecef(origin);
    for (int c = 0; c < nchan; ++c) {
        if (trk[c].prn <= 0) continue;
        if (trk[c].prn > 32) return fail(GPSIQ_E_ARG, "prn %d out of range", trk[c].prn);
        const Range r = range_to(eph[c], *iono, sec, xyz, site);            // gps.c:2199-2200
        trk[c].rho0_range = r.prange;
        trk[c].rho0_week = week;
        trk[c].rho0_sec = sec;
        const double r_ref = range_to(eph[c], *iono, sec, origin, site0).prange;   // gps.c:2206-2207
        const double phase_ini = (2.0 * r_ref - r.prange) / kLambdaL1;      // gps.c:2209
        trk[c].carr_phase = phase_ini - std::floor(phase_ini);              // gps.c:2211
    }
    return GPSIQ_OK;
}

void gpsiq_ecef_to_llh(const double xyz[3], double llh[3])
{
    const Site s = site_from_ecef(xyz);                                       // gps.c:361-410
    llh[0] = s.llh[0]; llh[1] = s.llh[1]; llh[2] = s.llh[2];
}

void gpsiq_llh_to_ecef(const double llh[3], double xyz[3])
{
    // gps.c:412-447: prime-vertical radius from e*sin(lat), then the three projections
    const double a = kWgs84A, e = kWgs84E, e2 = e * e;
    const double clat = cos_only(llh[0]), slat = sin_only(llh[0]);
    const double clon = cos_only(llh[1]), slon = sin_only(llh[1]);
    const double d = e * slat;
    const double n = a / std::sqrt(1.0 - d * d);
    const double nph = n + llh[2];
    const double tmp = nph * clat;
    xyz[0] = tmp * clon;
    xyz[1] = tmp * slon;
    xyz[2] = ((1.0 - e2) * n + llh[2]) * slat;
}

void gpsiq_ecef_add_neu(const double llh_ref[3], const double neu[3], double xyz[3])
{
    // ltcmat (gps.c:449-468) of the reference point, then gps.c:2354-2356 / 2726-2728
    const double slat = sin_only(llh_ref[0]), clat = cos_only(llh_ref[0]);
    const double slon = sin_only(llh_ref[1]), clon = cos_only(llh_ref[1]);
    const double t[3][3] = {{-slat * clon, -slat * slon, clat}, {-slon, clon, 0.0}, {clat * clon, clat * slon, slat}};
    xyz[0] += t[0][0] * neu[0] + t[1][0] * neu[1] + t[2][0] * neu[2];
    xyz[1] += t[0][1] * neu[0] + t[1][1] * neu[1] + t[2][1] * neu[2];
    xyz[2] += t[0][2] * neu[0] + t[1][2] * neu[1] + t[2][2] * neu[2];
}

int gpsiq_motion_read_csv(const char *path, double *xyz, int max_points)
{
    if (!path || !xyz || max_points < 0) return fail(GPSIQ_E_ARG, "bad argument");
    FILE *fp = std::fopen(path, "rt");
    if (!fp) { (void) fail(GPSIQ_E_ARG, "cannot open %s", path); return -1; }          // gps.c:2259-2260
    char line[100];                                                                     // MAX_CHAR, gps.h:30
    int n = 0;
    double t = 0.0, x = 0.0, y = 0.0, z = 0.0;
    for (; n < max_points; ++n) {
        if (!std::fgets(line, sizeof line, fp)) break;
        if (std::sscanf(line, "%lf,%lf,%lf,%lf", &t, &x, &y, &z) == EOF) break;         // gps.c:2266: only EOF ends the file;
        xyz[3 * n] = x; xyz[3 * n + 1] = y; xyz[3 * n + 2] = z;                         // a short line keeps the last values, as there
    }
    std::fclose(fp);
    return n;
}

int gpsiq_sat_visibility(const gpsiq_ephem_t *eph, int week, double sec, const double xyz[3],
                         double elv_mask_deg, double azel[2])
{
    (void) week;                                  // satpos() reads only the seconds of the week (gps.c:525-530)
    if (!eph || !xyz) return fail(GPSIQ_E_ARG, "null argument");
    const Site site = site_from_ecef(xyz);                                   // gps.c:2150-2151
    const SvState s = sv_state(*eph, sec);                                   // gps.c:2153: no light-time correction here
    const double los[3] = {s.pos[0] - xyz[0], s.pos[1] - xyz[1], s.pos[2] - xyz[2]};
    double neu[3];
    for (int k = 0; k < 3; ++k)
        neu[k] = site.t[k][0] * los[0] + site.t[k][1] * los[1] + site.t[k][2] * los[2];   // gps.c:476-482
    double az = std::atan2(neu[1], neu[0]);                                   // gps.c:488-499
    if (az < 0.0) az += (2.0 * kPi);
    const double el = std::atan2(neu[2], std::sqrt(neu[0] * neu[0] + neu[1] * neu[1]));
    if (azel) { azel[0] = az; azel[1] = el; }
    return el * kR2D > elv_mask_deg ? 1 : 0;                                 // gps.c:2158-2161
}

static int refresh_impl(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                        const double *xyz, int nblocks, int nchan, int gain_x2,
                        gpsiq_track_t *trk, const int *first, int nepochs, gpsiq_chan_t *out, int nthreads,
                        gpsiq_qchan_t *qout = nullptr, double fs = 0.0, int nsamp = 0);

static int check_epochs(const gpsiq_track_t *trk_epochs, const int *first_block, int nepochs, int nblocks, int nchan)
{
    if (!first_block || nepochs < 1) return fail(GPSIQ_E_ARG, "bad epoch list");
    for (int e = 0; e < nepochs; ++e)
        if (first_block[e] < 0 || first_block[e] > nblocks || (e == 0 ? first_block[0] != 0 : first_block[e] < first_block[e - 1]))
            return fail(GPSIQ_E_ARG, "epoch %d starts at block %d: not an ascending cover of [0, %d)", e, first_block[e], nblocks);
    if (trk_epochs)
        for (int e = 1; e < nepochs; ++e)
            for (int c = 0; c < nchan && c < GPSIQ_MAX_CHAN; ++c)
                if (trk_epochs[(size_t) e * nchan + c].prn != trk_epochs[c].prn)
                    return fail(GPSIQ_E_ARG, "epoch %d slot %d holds another satellite: one call covers one allocation", e, c);
    return GPSIQ_OK;
}

int gpsiq_refresh_batch(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                        const double *xyz, int nblocks, int nchan, int gain_x2,
                        gpsiq_track_t *trk, gpsiq_chan_t *out, int nthreads)
{
    return refresh_impl(eph, iono, week, sec, xyz, nblocks, nchan, gain_x2, trk, nullptr, 1, out, nthreads);
}

int gpsiq_refresh_epochs(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                         const double *xyz, int nblocks, int nchan, int gain_x2,
                         gpsiq_track_t *trk_epochs, const int *first_block, int nepochs,
                         gpsiq_chan_t *out, int nthreads)
{
    const int rc = check_epochs(trk_epochs, first_block, nepochs, nblocks, nchan);
    if (rc) return rc;
    return refresh_impl(eph, iono, week, sec, xyz, nblocks, nchan, gain_x2, trk_epochs, first_block, nepochs, out, nthreads);
}

int gpsiq_refresh_epochs_quantized(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                                   const double *xyz, int nblocks, int nchan, int gain_x2,
                                   gpsiq_track_t *trk_epochs, const int *first_block, int nepochs,
                                   double fs, int nsamp, gpsiq_qchan_t *out, int nthreads)
{
    if (!out) return fail(GPSIQ_E_ARG, "null argument");
    const int rc = check_epochs(trk_epochs, first_block, nepochs, nblocks, nchan);
    if (rc) return rc;
    return refresh_impl(eph, iono, week, sec, xyz, nblocks, nchan, gain_x2, trk_epochs, first_block, nepochs, nullptr, nthreads,
                        out, fs, nsamp);
}

static int refresh_impl(const gpsiq_ephem_t *eph, const gpsiq_iono_t *iono, int week, double sec,
                        const double *xyz, int nblocks, int nchan, int gain_x2,
                        gpsiq_track_t *trk, const int *first, int nepochs, gpsiq_chan_t *out, int nthreads,
                        gpsiq_qchan_t *qout, double fs, int nsamp)
{
    if (!eph || !iono || !xyz || !trk || (!out && !qout)) return fail(GPSIQ_E_ARG, "null argument");
    if (qout && (!(fs > 0.0) || nsamp < 0)) return fail(GPSIQ_E_ARG, "bad fs %g / nsamp %d", fs, nsamp);
    if (nchan < 1 || nchan > GPSIQ_MAX_CHAN || nblocks < 0) return fail(GPSIQ_E_ARG, "bad nblocks %d / nchan %d", nblocks, nchan);
    if (nblocks == 0) return GPSIQ_OK;
    double ant_pat[37];
    for (int i = 0; i < 37; ++i) ant_pat[i] = std::pow(10.0, -kAntPatDb[i] / 20.0);    // gps.c:2688-2689

    // receiver time of every block: grx is advanced before the first block (gps.c:2692)
    // and after each one (gps.c:2932); t[0] is the time of the carried-in range
    std::vector<GpsTime> t((size_t) nblocks + 1);
    GpsTime g = {week, sec};
    for (int b = 1; b <= nblocks; ++b) { g = advance(g, 0.1); t[(size_t) b] = g; }
    std::vector<Range> rng((size_t) (nblocks + 1) * (size_t) nchan);
    for (int c = 0; c < nchan; ++c) {
        if (trk[c].prn > 32) return fail(GPSIQ_E_ARG, "prn %d out of range", trk[c].prn);
        rng[(size_t) c].prange = trk[c].rho0_range;
        rng[(size_t) c].d = rng[(size_t) c].az = rng[(size_t) c].el = 0.0;
    }
    t[0] = GpsTime{week, sec};

    Job job = {eph, iono, xyz, t.data(), nchan, gain_x2, trk, ant_pat, rng.data(), out, 0, first, nepochs};
    job.qout = qout; job.delt = qout ? 1.0 / fs : 0.0; job.nsamp = nsamp;
    // pass 0 costs ~4 us per block (16 satellite positions with light-time iteration), waking the pool ~20 us:
    // 64 blocks per thread keep a 300-block epoch of the run-ahead loop (gpsiq/pipeline.py) on several cores
    parallel_for(nblocks, nthreads, 64, work, &job);
    job.pass = 1;
    parallel_for(nblocks, nthreads, 256, work, &job);
    if (job.rc != GPSIQ_OK) return fail(job.rc, "%s", job.err);
    if (qout) chain_carrier(qout, nblocks, nchan, nsamp, nullptr, nullptr, nullptr, nullptr);

    for (int c = 0; c < nchan; ++c) {                      // chan.rho0 = rho1 (gps.c:2063)
        if (trk[c].prn <= 0) continue;
        trk[c].rho0_range = rng[(size_t) nblocks * nchan + c].prange;
        trk[c].rho0_week = t[(size_t) nblocks].week;
        trk[c].rho0_sec = t[(size_t) nblocks].sec;
    }
    return GPSIQ_OK;
}

}  // extern "C"
