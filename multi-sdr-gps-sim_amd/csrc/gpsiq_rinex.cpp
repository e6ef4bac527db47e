// gpsiq_rinex.cpp — RINEX 2 / 3 GPS navigation file reader.
//
// Same observable behaviour as the reference's readRinex2 (gps.c:1131-1505) and readRinex3
// (gps.c:1512-1891): fixed columns, 'D' exponents accepted, gzip or plain input (zlib),
// records grouped into hourly sets, derived orbit variables A, n, sqrt(1-e^2), Omega-dot
// relative to the Earth (gps.c:1492-1497).  Written table-driven: one description of where
// each field sits, used by both format versions (version 3 shifts every data column by one
// and prefixes the satellite system letter).
#include "gpsiq_internal.h"

#include <cmath>
#include <cstdlib>
#include <cstring>
#include <zlib.h>

namespace {

constexpr int kMaxLine = 100;            // gps.h:30 MAX_CHAR
constexpr double kGM = 3.986005e14, kOmegaEarth = 7.2921151467e-5;   // gps.h:89-90
constexpr double kSecWeek = 604800.0, kSecDay = 86400.0, kSecHour = 3600.0, kSecMinute = 60.0;

// text[col, col+len) as a C string with D/d exponents turned into E (gps.c:1079-1094);
// a line that ends before col yields an empty field
struct Field {
    char s[24];
    Field(const char *line, int col, int len)
    {
        int i = 0;
        if ((int) std::strlen(line) > col)
            for (; i < len && line[col + i]; ++i) s[i] = (line[col + i] == 'D' || line[col + i] == 'd') ? 'E' : line[col + i];
        s[i] = 0;
    }
    double real() const { return std::atof(s); }
    int integer() const { return std::atoi(s); }
};

// like Field but without the exponent rewrite (the reference uses plain atoi on these)
int int_at(const char *line, int col, int len)
{
    char t[24];
    int i = 0;
    if ((int) std::strlen(line) > col)
        for (; i < len && line[col + i]; ++i) t[i] = line[col + i];
    t[i] = 0;
    return std::atoi(t);
}

struct GpsTime { int week; double sec; };

// calendar -> GPS week/second (gps.c:315-337)
GpsTime to_gps(int y, int m, int d, int hh, int mm, double sec)
{
    static const int doy[12] = {0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334};
    const int ye = y - 1980;
    int lpdays = ye / 4 + 1;
    if ((ye % 4) == 0 && m <= 2) lpdays--;
    // a damaged record can carry month 0 or > 12: the reference indexes its table with it
    // (gps.c:331, undefined behaviour); here such a month counts as January
    const int de = ye * 365 + doy[(m >= 1 && m <= 12) ? m - 1 : 0] + d + lpdays - 6;
    GpsTime g;
    g.week = de / 7;
    g.sec = (double) (de % 7) * kSecDay + hh * kSecHour + mm * kSecMinute + sec;
    return g;
}

bool label_is(const char *line, const char *label)
{
    return std::strlen(line) > 60 && std::strncmp(line + 60, label, std::strlen(label)) == 0;
}

}  // namespace

using namespace gpsiq;

extern "C" {

int gpsiq_rinex_read(const char *path, int version, gpsiq_rinex_eph_t *eph, gpsiq_nav_utc_t *utc)
{
    if (!path || !eph || !utc) return fail(GPSIQ_E_ARG, "null argument");
    if (version != 2 && version != 3) return fail(GPSIQ_E_ARG, "RINEX version %d not supported (2 or 3)", version);
    gzFile fp = gzopen(path, "rt");
    if (!fp) return -1;
    const bool v3 = version == 3;
    const int c0 = v3 ? 4 : 3;                     // first data column of the continuation lines
    std::memset(eph, 0, sizeof(gpsiq_rinex_eph_t) * GPSIQ_EPHEM_SETS * GPSIQ_MAX_SAT);
    char line[kMaxLine + 4];
    int flags = 0;

    // ---- header -------------------------------------------------------------------------
    while (gzgets(fp, line, kMaxLine)) {
        if (label_is(line, "COMMENT")) continue;
        if (label_is(line, "END OF HEADER")) break;
        if (label_is(line, "RINEX VERSION / TYPE")) {
            const double ver = Field(line, 0, 9).real();
            if (!v3 && ver > 3.0) { gzclose(fp); return -2; }
            if (v3 && ver < 3.0) { gzclose(fp); return -2; }
            if (!v3 && line[20] != 'N') { gzclose(fp); return -3; }
            if (v3 && line[20] != 'N' && line[40] != 'G') { gzclose(fp); return -3; }
        } else if (!v3 && label_is(line, "ION ALPHA")) {
            for (int k = 0; k < 4; ++k) utc->alpha[k] = Field(line, 2 + 12 * k, 12).real();
            flags |= 1;
        } else if (!v3 && label_is(line, "ION BETA")) {
            for (int k = 0; k < 4; ++k) utc->beta[k] = Field(line, 2 + 12 * k, 12).real();
            flags |= 2;
        } else if (!v3 && label_is(line, "DELTA-UTC")) {
            utc->A0 = Field(line, 3, 19).real();
            utc->A1 = Field(line, 22, 19).real();
            utc->tot = int_at(line, 41, 9);
            utc->wnt = int_at(line, 50, 9);
            if (utc->tot % 4096 == 0) flags |= 4;
        } else if (v3 && label_is(line, "IONOSPHERIC CORR")) {
            if (std::strncmp(line, "GPSA", 4) == 0) {
                for (int k = 0; k < 4; ++k) utc->alpha[k] = Field(line, 5 + 12 * k, 12).real();
                flags |= 1;
            } else if (std::strncmp(line, "GPSB", 4) == 0) {
                for (int k = 0; k < 4; ++k) utc->beta[k] = Field(line, 5 + 12 * k, 12).real();
                flags |= 2;
            }
        } else if (v3 && label_is(line, "TIME SYSTEM CORR") && std::strncmp(line, "GPUT", 4) == 0) {
            utc->A0 = Field(line, 5, 17).real();
            utc->A1 = Field(line, 22, 16).real();
            utc->tot = Field(line, 38, 7).integer();
            utc->wnt = int_at(line, 45, 6);
            if (utc->tot % 4096 == 0) flags |= 4;
        } else if (label_is(line, "LEAP SECONDS")) {
            utc->dtls = int_at(line, 0, 6);
            flags |= 8;
        }
    }
    utc->vflg = flags == 0xF ? 1 : 0;               // all four records present (gps.c:1257-1259)

    // ---- records ------------------------------------------------------------------------
    GpsTime g0 = {-1, 0.0};
    int iset = 0;
    while (gzgets(fp, line, kMaxLine)) {
        if (v3 && line[0] != 'G') continue;         // other constellations in a mixed file
        const int sv = (v3 ? int_at(line, 1, 2) : int_at(line, 0, 2)) - 1;
        int y, m, d, hh, mm;
        double sec;
        if (v3) {
            y = int_at(line, 4, 4); m = int_at(line, 9, 2); d = int_at(line, 12, 2);
            hh = int_at(line, 15, 2); mm = int_at(line, 18, 2); sec = (double) int_at(line, 21, 2);
        } else {
            y = int_at(line, 3, 2) + 2000; m = int_at(line, 6, 2); d = int_at(line, 9, 2);
            hh = int_at(line, 12, 2); mm = int_at(line, 15, 2); sec = Field(line, 18, 2).real();
        }
        const GpsTime g = to_gps(y, m, d, hh, mm, sec);
        if (g0.week == -1) g0 = g;
        const double dt = (g.sec - g0.sec) + (double) (g.week - g0.week) * kSecWeek;   // gps.c:1096-1103
        if (dt > kSecHour) {                         // a new hourly set (gps.c:1305-1311)
            g0 = g;
            if (++iset >= GPSIQ_EPHEM_SETS) break;
        }
        if (sv < 0 || sv >= GPSIQ_MAX_SAT) { gzclose(fp); return fail(GPSIQ_E_RANGE, "satellite number %d in %s", sv + 1, path); }
        gpsiq_rinex_eph_t &e = eph[(size_t) iset * GPSIQ_MAX_SAT + sv];
        e.t_y = y; e.t_m = m; e.t_d = d; e.t_hh = hh; e.t_mm = mm; e.t_sec = sec;
        e.toc_week = g.week;
        e.orbit.toc_sec = e.nav.toc_sec = g.sec;
        const int c1 = c0 + 19, c2 = c0 + 38, c3 = c0 + 57;
        e.orbit.af0 = e.nav.af0 = Field(line, c1, 19).real();
        e.orbit.af1 = e.nav.af1 = Field(line, c2, 19).real();
        e.orbit.af2 = e.nav.af2 = Field(line, c3, 19).real();
        bool eof = false;
        // a record cut short by the end of the file never becomes valid: leave it cleared
        auto next = [&]() { if (!gzgets(fp, line, kMaxLine)) { eof = true; std::memset(&e, 0, sizeof e); } return !eof; };
        if (!next()) break;                          // BROADCAST ORBIT - 1
        e.nav.iode = (int) Field(line, c0, 19).real();
        e.orbit.crs = e.nav.crs = Field(line, c1, 19).real();
        e.nav.deltan = Field(line, c2, 19).real();
        e.orbit.m0 = e.nav.m0 = Field(line, c3, 19).real();
        if (!next()) break;                          // - 2
        e.orbit.cuc = e.nav.cuc = Field(line, c0, 19).real();
        e.orbit.ecc = e.nav.ecc = Field(line, c1, 19).real();
        e.orbit.cus = e.nav.cus = Field(line, c2, 19).real();
        e.orbit.sqrta = e.nav.sqrta = Field(line, c3, 19).real();
        if (!next()) break;                          // - 3
        e.orbit.toe_sec = e.nav.toe_sec = Field(line, c0, 19).real();
        e.orbit.cic = e.nav.cic = Field(line, c1, 19).real();
        e.orbit.omg0 = e.nav.omg0 = Field(line, c2, 19).real();
        e.orbit.cis = e.nav.cis = Field(line, c3, 19).real();
        if (!next()) break;                          // - 4
        e.orbit.inc0 = e.nav.inc0 = Field(line, c0, 19).real();
        e.orbit.crc = e.nav.crc = Field(line, c1, 19).real();
        e.orbit.aop = e.nav.aop = Field(line, c2, 19).real();
        e.nav.omgdot = Field(line, c3, 19).real();
        if (!next()) break;                          // - 5
        e.orbit.idot = e.nav.idot = Field(line, c0, 19).real();
        e.code = (int) Field(line, c1, 19).real();
        e.nav.toe_week = (int) Field(line, c2, 19).real();
        e.flag = (int) Field(line, c3, 19).real();
        if (!next()) break;                          // - 6
        if (!v3) e.sva = (int) Field(line, c0, 19).real();
        e.svh = (int) Field(line, c1, 19).real();
        if (e.svh > 0 && e.svh < 32) e.svh += 32;    // gps.c:1463-1464
        e.orbit.tgd = e.nav.tgd = Field(line, c2, 19).real();
        e.nav.iodc = (int) Field(line, c3, 19).real();
        if (!next()) break;                          // - 7
        e.fit = Field(line, c1, 19).real();
        e.vflg = 1;
        // working variables (gps.c:1492-1497)
        e.orbit.A = e.orbit.sqrta * e.orbit.sqrta;
        e.orbit.n = std::sqrt(kGM / (e.orbit.A * e.orbit.A * e.orbit.A)) + e.nav.deltan;
        e.orbit.sq1e2 = std::sqrt(1.0 - e.orbit.ecc * e.orbit.ecc);
        e.orbit.omgkdot = e.nav.omgdot - kOmegaEarth;
    }
    gzclose(fp);
    // the reference returns ieph + 1 = 14 when it stops at a 14th hourly group (gps.c:1310-1311, 1500), one more than
    // its array holds; callers index eph[nsets - 1] and eph[ieph + 1] with it.  Deliberate deviation: never report
    // more sets than were stored.
    const int nsets = g0.week >= 0 ? iset + 1 : iset;
    return nsets > GPSIQ_EPHEM_SETS ? GPSIQ_EPHEM_SETS : nsets;
}

int gpsiq_rinex_overwrite_time(gpsiq_rinex_eph_t *eph, int nsets, gpsiq_nav_utc_t *utc, int week, double sec)
{
    if (!eph || !utc || nsets < 1 || nsets > GPSIQ_EPHEM_SETS) return fail(GPSIQ_E_ARG, "bad argument");
    // gmin: time of clock of the first valid satellite of the first set (gps.c:2507-2513); all zero if there is none
    int gmin_week = 0;
    double gmin_sec = 0.0;
    for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv)
        if (eph[sv].vflg) { gmin_week = eph[sv].toc_week; gmin_sec = eph[sv].nav.toc_sec; break; }
    // the start time cut to whole two hours, and how far that is from gmin (gps.c:2536-2539)
    const double cut = (double) (((int) sec) / 7200) * 7200.0;
    const double dsec = (cut - gmin_sec) + (double) (week - gmin_week) * kSecWeek;        // subGpsTime, gps.c:1096-1103
    utc->wnt = week;                                                                       // gps.c:2542-2543
    utc->tot = (int) cut;
    auto shifted = [dsec](int w, double s, int *w_out) {                                   // incGpsTime, gps.c:1105-1124
        double t = s + dsec;
        t = std::round(t * 1000.0) / 1000.0;
        while (t >= kSecWeek) { t -= kSecWeek; ++w; }
        while (t < 0.0) { t += kSecWeek; --w; }
        *w_out = w;
        return t;
    };
    for (int i = 0; i < nsets; ++i)
        for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) {
            gpsiq_rinex_eph_t &e = eph[(size_t) i * GPSIQ_MAX_SAT + sv];
            if (!e.vflg) continue;
            int w;
            const double toc = shifted(e.toc_week, e.nav.toc_sec, &w);                     // gps.c:2552-2555
            e.toc_week = w;
            e.nav.toc_sec = e.orbit.toc_sec = toc;
            gpsiq_gps_to_date(w, toc, &e.t_y, &e.t_m, &e.t_d, &e.t_hh, &e.t_mm, &e.t_sec);
            const double toe = shifted(e.nav.toe_week, e.nav.toe_sec, &w);                 // gps.c:2557-2558
            e.nav.toe_week = w;
            e.nav.toe_sec = e.orbit.toe_sec = toe;
        }
    return GPSIQ_OK;
}

void gpsiq_date_to_gps(int year, int month, int day, int hour, int minute, double second, int *week, double *sec)
{
    const GpsTime g = to_gps(year, month, day, hour, minute, second);
    if (week) *week = g.week;
    if (sec) *sec = g.sec;
}

void gpsiq_gps_to_date(int week, double sec, int *year, int *month, int *day, int *hour, int *minute, double *second)
{
    // gps.c:339-355: Julian day number of the GPS day, then the usual calendar arithmetic on it
    const int c = (int) (7 * week + std::floor(sec / 86400.0) + 2444245.0) + 1537;
    const int d = (int) ((c - 122.1) / 365.25);
    const int e = 365 * d + d / 4;
    const int f = (int) ((c - e) / 30.6001);
    const int dd = c - e - (int) (30.6001 * f);
    const int mo = f - 1 - 12 * (f / 14);
    if (day) *day = dd;
    if (month) *month = mo;
    if (year) *year = d - 4715 - ((7 + mo) / 10);
    if (hour) *hour = ((int) (sec / 3600.0)) % 24;
    if (minute) *minute = ((int) (sec / 60.0)) % 60;
    if (second) *second = sec - 60.0 * std::floor(sec / 60.0);
}

int gpsiq_rinex_select(const gpsiq_rinex_eph_t *eph, int nsets, int week, double sec)
{
    if (!eph) return -1;
    for (int i = 0; i < nsets && i < GPSIQ_EPHEM_SETS; ++i)
        for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) {
            const gpsiq_rinex_eph_t &e = eph[(size_t) i * GPSIQ_MAX_SAT + sv];
            if (!e.vflg) continue;
            const double dt = (sec - e.orbit.toc_sec) + (double) (week - e.toc_week) * kSecWeek;
            if (dt >= -kSecHour && dt < kSecHour) return i;
        }
    return -1;
}

}  // extern "C"
