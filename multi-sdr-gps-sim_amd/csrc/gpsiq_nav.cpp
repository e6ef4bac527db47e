// gpsiq_nav.cpp — GPS LNAV words for the sample loop's data bits.
//
// Restates, bit for bit, what the reference does between the broadcast ephemeris and the
// 60-word buffer chan.dwrd[] that gps.c:2811 reads: subframe packing (eph2sbf,
// gps.c:617-884), week/TOW insertion, word-to-word parity chaining and the 30 s roll
// (generateNavMsg, gps.c:2066-2140), the (32,26) Hamming parity with the D29*/D30*
// inversion rule and the "non-information-bearing bits" solve for words 2 and 10
// (computeChecksum, gps.c:1008-1072; IS-GPS-200 20.3.5).  Scale factors are written with
// the same decimal literals as the reference (gps.h:66-86) so the truncations agree.
#include "gpsiq_internal.h"

#include <cmath>
#include <cstring>

namespace {

constexpr double kPi = 3.1415926535898;
// gps.h:66-86 (decimal literals on purpose)
constexpr double P2M5 = 0.03125, P2M19 = 1.907348632812500e-6, P2M29 = 1.862645149230957e-9;
constexpr double P2M31 = 4.656612873077393e-10, P2M33 = 1.164153218269348e-10, P2M43 = 1.136868377216160e-13;
constexpr double P2M55 = 2.775557561562891e-17, P2M50 = 8.881784197001252e-016, P2M30 = 9.313225746154785e-010;
constexpr double P2M27 = 7.450580596923828e-009, P2M24 = 5.960464477539063e-008, P2M21 = 4.76837158203125e-007;
constexpr double P2_12 = 4096, P2M38 = 3.63797880709171e-012, P2M11 = 0.00048828125, P2M23 = 1.19209289550781e-007;
constexpr double P2M20 = 9.5367431640625e-007;

constexpr uint32_t kTlm = 0x8B0000u << 6;                       // preamble word
constexpr uint64_t kEmpty = 0xaaaaaaaaull;                      // gps.h:134 EMPTY_WORD
// subframe 4 page -> SV id (IS-GPS-200 table 20-V; gps.c:224-228), page 25 of subframe 5 = 51
const uint64_t kSbf4SvId[25] = {57, 0, 0, 0, 0, 57, 0, 0, 0, 0, 57, 62, 52, 53, 54, 57, 55, 56, 58, 59, 57, 60, 61, 62, 63};

// parity masks over d1..d24 (bits 29..6), IS-GPS-200 table 20-XIV / gps.c:1036-1039
const uint32_t kMask[6] = {0x3B1F3480u, 0x1D8F9A40u, 0x2EC7CD00u, 0x1763E680u, 0x2BB1F340u, 0x0B7A89C0u};

inline uint32_t odd(uint32_t v) { return (uint32_t) __builtin_parity(v); }

typedef int64_t i64;
typedef uint64_t u64;

void almanac_page(uint32_t *w, uint32_t subframe_id, const gpsiq_nav_alm_sv_t &a, int sv, bool toa_pow2)
{
    const u64 data_id = 1, sv_id = (u64) (sv + 1);
    const u64 ecc = (u64) (a.e / P2M21);
    const u64 toa = toa_pow2 ? (u64) (a.toa_sec / P2_12) : (u64) (a.toa_sec / 4096.0);
    const i64 delta_i = (i64) (a.delta_i / P2M19);
    const i64 omegadot = (i64) (a.omegadot / P2M38);
    const u64 sqrta = (u64) (a.sqrta / P2M11);
    const i64 omega0 = (i64) (a.omega0 / P2M23);
    const i64 aop = (i64) (a.aop / P2M23);
    const i64 m0 = (i64) (a.m0 / P2M23);
    const i64 af0 = (i64) (a.af0 / P2M20);
    const i64 af1 = (i64) (a.af1 / P2M38);
    w[0] = kTlm;
    w[1] = subframe_id << 8;
    w[2] = (uint32_t) ((data_id << 28) | (sv_id << 22) | ((ecc & 0xFFFF) << 6));
    w[3] = (uint32_t) (((toa & 0xFF) << 22) | (((u64) delta_i & 0xFFFF) << 6));
    w[4] = (uint32_t) (((u64) omegadot & 0xFFFF) << 14);
    w[5] = (uint32_t) ((sqrta & 0xFFFFFF) << 6);
    w[6] = (uint32_t) (((u64) omega0 & 0xFFFFFF) << 6);
    w[7] = (uint32_t) (((u64) aop & 0xFFFFFF) << 6);
    w[8] = (uint32_t) (((u64) m0 & 0xFFFFFF) << 6);
    w[9] = (uint32_t) ((((u64) af0 & 0x7F8) << 19) | (((u64) af1 & 0x7FF) << 11) | (((u64) af0 & 0x7) << 8));
}

}  // namespace

using namespace gpsiq;

extern "C" {

uint32_t gpsiq_nav_parity(uint32_t source, int nib)
{
    uint32_t d = source & 0x3FFFFFC0u;
    const uint32_t D29 = (source >> 31) & 1u, D30 = (source >> 30) & 1u;
    if (nib) {                      // choose d24, d23 so that D29 = D30 = 0 (gps.c:1046-1056)
        if ((D30 + odd(kMask[4] & d)) & 1u) d ^= 1u << 6;
        if ((D29 + odd(kMask[5] & d)) & 1u) d ^= 1u << 7;
    }
    uint32_t D = d;
    if (D30) D ^= 0x3FFFFFC0u;      // transmitted data bits are inverted when D30* is set
    D |= ((D29 + odd(kMask[0] & d)) & 1u) << 5;
    D |= ((D30 + odd(kMask[1] & d)) & 1u) << 4;
    D |= ((D29 + odd(kMask[2] & d)) & 1u) << 3;
    D |= ((D30 + odd(kMask[3] & d)) & 1u) << 2;
    D |= ((D30 + odd(kMask[4] & d)) & 1u) << 1;
    D |= ((D29 + odd(kMask[5] & d)) & 1u);
    D &= 0x3FFFFFFFu;
    return D | (source & 0xC0000000u);
}

int gpsiq_nav_subframes(const gpsiq_nav_eph_t *e, const gpsiq_nav_utc_t *u, const gpsiq_nav_alm_sv_t *alm,
                        uint32_t sbf[GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF])
{
    if (!e || !u || !sbf) return fail(GPSIQ_E_ARG, "null argument");
    std::memset(sbf, 0, sizeof(uint32_t) * GPSIQ_N_SBF_PAGE * GPSIQ_N_DWRD_SBF);
    // quantise to the broadcast units (gps.c:663-703)
    const u64 wn = 0, ura = 0, data_id = 1;        // the transmission week is filled in later (gps.c:2115)
    const u64 toe = (u64) (e->toe_sec / 16.0), toc = (u64) (e->toc_sec / 16.0);
    const u64 iode = (u64) e->iode, iodc = (u64) e->iodc;
    const i64 deltan = (i64) (e->deltan / P2M43 / kPi);
    const i64 cuc = (i64) (e->cuc / P2M29), cus = (i64) (e->cus / P2M29);
    const i64 cic = (i64) (e->cic / P2M29), cis = (i64) (e->cis / P2M29);
    const i64 crc = (i64) (e->crc / P2M5), crs = (i64) (e->crs / P2M5);
    const u64 ecc = (u64) (e->ecc / P2M33), sqrta = (u64) (e->sqrta / P2M19);
    const i64 m0 = (i64) (e->m0 / P2M31 / kPi), omega0 = (i64) (e->omg0 / P2M31 / kPi);
    const i64 inc0 = (i64) (e->inc0 / P2M31 / kPi), aop = (i64) (e->aop / P2M31 / kPi);
    const i64 omegadot = (i64) (e->omgdot / P2M43 / kPi), idot = (i64) (e->idot / P2M43 / kPi);
    const i64 af0 = (i64) (e->af0 / P2M31), af1 = (i64) (e->af1 / P2M43), af2 = (i64) (e->af2 / P2M55);
    const i64 tgd = (i64) (e->tgd / P2M31);
    const i64 alpha0 = (i64) std::round(u->alpha[0] / P2M30), alpha1 = (i64) std::round(u->alpha[1] / P2M27);
    const i64 alpha2 = (i64) std::round(u->alpha[2] / P2M24), alpha3 = (i64) std::round(u->alpha[3] / P2M24);
    const i64 beta0 = (i64) std::round(u->beta[0] / 2048.0), beta1 = (i64) std::round(u->beta[1] / 16384.0);
    const i64 beta2 = (i64) std::round(u->beta[2] / 65536.0), beta3 = (i64) std::round(u->beta[3] / 65536.0);
    const i64 A0 = (i64) std::round(u->A0 / P2M30), A1 = (i64) std::round(u->A1 / P2M50);
    const i64 dtls = (i64) u->dtls;
    const u64 tot = (u64) (u->tot / 4096), wnt = (u64) (u->wnt % 256);
    const u64 wnlsf = 1929 % 256, dn = 7;          // scheduled leap second fixed by the reference (gps.c:702-704)
    const i64 dtlsf = 18;
#define U(x) ((u64) (x))
    uint32_t (*s)[GPSIQ_N_DWRD_SBF] = sbf;
    // subframe 1: clock (gps.c:707-716)
    s[0][0] = kTlm; s[0][1] = 0x1u << 8;
    s[0][2] = (uint32_t) (((wn & 0x3FF) << 20) | (ura << 14) | (((iodc >> 8) & 0x3) << 6));
    s[0][6] = (uint32_t) ((U(tgd) & 0xFF) << 6);
    s[0][7] = (uint32_t) (((iodc & 0xFF) << 22) | ((toc & 0xFFFF) << 6));
    s[0][8] = (uint32_t) (((U(af2) & 0xFF) << 22) | ((U(af1) & 0xFFFF) << 6));
    s[0][9] = (uint32_t) ((U(af0) & 0x3FFFFF) << 8);
    // subframe 2: orbit part 1 (gps.c:719-728)
    s[1][0] = kTlm; s[1][1] = 0x2u << 8;
    s[1][2] = (uint32_t) (((iode & 0xFF) << 22) | ((U(crs) & 0xFFFF) << 6));
    s[1][3] = (uint32_t) (((U(deltan) & 0xFFFF) << 14) | (((U(m0 >> 24)) & 0xFF) << 6));
    s[1][4] = (uint32_t) ((U(m0) & 0xFFFFFF) << 6);
    s[1][5] = (uint32_t) (((U(cuc) & 0xFFFF) << 14) | (((ecc >> 24) & 0xFF) << 6));
    s[1][6] = (uint32_t) ((ecc & 0xFFFFFF) << 6);
    s[1][7] = (uint32_t) (((U(cus) & 0xFFFF) << 14) | (((sqrta >> 24) & 0xFF) << 6));
    s[1][8] = (uint32_t) ((sqrta & 0xFFFFFF) << 6);
    s[1][9] = (uint32_t) ((toe & 0xFFFF) << 14);
    // subframe 3: orbit part 2 (gps.c:731-740)
    s[2][0] = kTlm; s[2][1] = 0x3u << 8;
    s[2][2] = (uint32_t) (((U(cic) & 0xFFFF) << 14) | ((U(omega0 >> 24) & 0xFF) << 6));
    s[2][3] = (uint32_t) ((U(omega0) & 0xFFFFFF) << 6);
    s[2][4] = (uint32_t) (((U(cis) & 0xFFFF) << 14) | ((U(inc0 >> 24) & 0xFF) << 6));
    s[2][5] = (uint32_t) ((U(inc0) & 0xFFFFFF) << 6);
    s[2][6] = (uint32_t) (((U(crc) & 0xFFFF) << 14) | ((U(aop >> 24) & 0xFF) << 6));
    s[2][7] = (uint32_t) ((U(aop) & 0xFFFFFF) << 6);
    s[2][8] = (uint32_t) ((U(omegadot) & 0xFFFFFF) << 6);
    s[2][9] = (uint32_t) (((iode & 0xFF) << 22) | ((U(idot) & 0x3FFF) << 8));
    // all 25 pages of subframes 4 and 5 start as "dummy SV" pages of alternating bits (gps.c:743-771)
    for (int i = 0; i < 25; ++i)
        for (int sf = 0; sf < 2; ++sf) {
            uint32_t *w = s[3 + sf + i * 2];
            w[0] = kTlm;
            w[1] = (uint32_t) (4 + sf) << 8;
            w[2] = (uint32_t) ((data_id << 28) | ((kEmpty & 0xFFFF) << 6));
            for (int k = 3; k < 9; ++k) w[k] = (uint32_t) ((kEmpty & 0xFFFFFF) << 6);
            w[9] = (uint32_t) ((kEmpty & 0x3FFFFF) << 8);
        }
    // subframe 4 pages 2-5, 7-10: almanac of PRN 25-32 (gps.c:774-805)
    for (int sv = 24; alm && sv < 32; ++sv)
        if (alm[sv].valid != 0)
            almanac_page(s[3 + (sv <= 27 ? sv - 23 : sv - 22) * 2], 4, alm[sv], sv, true);
    // subframe 4 page 18: ionosphere + UTC (gps.c:808-819)
    if (u->vflg) {
        uint32_t *w = s[3 + 17 * 2];
        w[0] = kTlm; w[1] = 0x4u << 8;
        w[2] = (uint32_t) ((data_id << 28) | (kSbf4SvId[17] << 22) | ((U(alpha0) & 0xFF) << 14) | ((U(alpha1) & 0xFF) << 6));
        w[3] = (uint32_t) (((U(alpha2) & 0xFF) << 22) | ((U(alpha3) & 0xFF) << 14) | ((U(beta0) & 0xFF) << 6));
        w[4] = (uint32_t) (((U(beta1) & 0xFF) << 22) | ((U(beta2) & 0xFF) << 14) | ((U(beta3) & 0xFF) << 6));
        w[5] = (uint32_t) ((U(A1) & 0xFFFFFF) << 6);
        w[6] = (uint32_t) ((U(A0 >> 8) & 0xFFFFFF) << 6);
        w[7] = (uint32_t) (((U(A0) & 0xFF) << 22) | ((tot & 0xFF) << 14) | ((wnt & 0xFF) << 6));
        w[8] = (uint32_t) (((U(dtls) & 0xFF) << 22) | ((wnlsf & 0xFF) << 14) | ((dn & 0xFF) << 6));
        w[9] = (uint32_t) ((U(dtlsf) & 0xFF) << 22);
    }
    // subframe 4 page 25: health of PRN 25-32 (gps.c:822-831)
    {
        uint32_t *w = s[3 + 24 * 2];
        std::memset(w, 0, sizeof(uint32_t) * GPSIQ_N_DWRD_SBF);
        w[0] = kTlm; w[1] = 0x4u << 8;
        w[2] = (uint32_t) ((data_id << 28) | (kSbf4SvId[24] << 22));
    }
    // subframe 5 pages 1-24: almanac of PRN 1-24 (gps.c:834-862)
    for (int sv = 0; alm && sv < 24; ++sv)
        if (alm[sv].svid != 0)
            almanac_page(s[4 + sv * 2], 5, alm[sv], sv, false);
    // subframe 5 page 25: almanac reference time and health of PRN 1-24 (gps.c:865-886)
    u64 wna = (u64) (e->toe_week % 256), toa = (u64) (e->toe_sec / 4096.0);
    for (int sv = 0; alm && sv < 32; ++sv)
        if (alm[sv].svid != 0) {
            wna = (u64) (alm[sv].toa_week % 256);
            toa = (u64) (alm[sv].toa_sec / 4096.0);
            break;
        }
    {
        uint32_t *w = s[4 + 24 * 2];
        std::memset(w, 0, sizeof(uint32_t) * GPSIQ_N_DWRD_SBF);
        w[0] = kTlm; w[1] = 0x5u << 8;
        w[2] = (uint32_t) ((data_id << 28) | ((u64) 51 << 22) | ((toa & 0xFF) << 14) | ((wna & 0xFF) << 6));
    }
#undef U
    return GPSIQ_OK;
}

int gpsiq_nav_message(const uint32_t sbf[GPSIQ_N_SBF_PAGE][GPSIQ_N_DWRD_SBF], int week, double sec,
                      int init, gpsiq_nav_state_t *st)
{
    if (!sbf || !st) return fail(GPSIQ_E_ARG, "null argument");
    if (st->ipage < 0 || st->ipage >= 25) return fail(GPSIQ_E_ARG, "ipage %d outside 0..24", st->ipage);
    // frame-aligned reference time of the buffer (gps.c:2074-2080)
    st->g0_week = week;
    st->g0_sec = (double) (((uint64_t) (sec + 0.5)) / 30u) * 30.0;
    const uint64_t wn = (uint64_t) (st->g0_week % 1024);
    uint64_t tow = ((uint64_t) st->g0_sec) / 6u;
    uint32_t prev = 0;

    auto emit = [&](int slot, const uint32_t *page, int iwrd, bool with_wn) {
        uint32_t w = page[iwrd];
        if (with_wn && iwrd == 2) w |= (uint32_t) ((wn & 0x3FF) << 20);          // gps.c:2114-2115
        if (iwrd == 1) w |= (uint32_t) ((tow & 0x1FFFF) << 13);                   // TOW count into the HOW
        w |= (prev << 30) & 0xC0000000u;                                          // D29*, D30*
        st->dwrd[slot] = gpsiq_nav_parity(w, iwrd == 1 || iwrd == 9);
        prev = st->dwrd[slot];
    };

    if (init) {                                     // the frame "before": subframe 5 of the current page
        for (int i = 0; i < GPSIQ_N_DWRD_SBF; ++i) emit(i, sbf[4 + st->ipage * 2], i, false);
    } else {                                        // roll: last subframe of the old buffer comes first
        for (int i = 0; i < GPSIQ_N_DWRD_SBF; ++i) {
            st->dwrd[i] = st->dwrd[GPSIQ_N_DWRD_SBF * 5 + i];
            prev = st->dwrd[i];
        }
    }
    for (int isbf = 0; isbf < 5; ++isbf) {
        ++tow;
        const uint32_t *page = isbf < 3 ? sbf[isbf] : isbf == 3 ? sbf[3 + st->ipage * 2] : sbf[4 + st->ipage * 2];
        for (int i = 0; i < GPSIQ_N_DWRD_SBF; ++i) emit((isbf + 1) * GPSIQ_N_DWRD_SBF + i, page, i, isbf == 0);
    }
    if (++st->ipage >= 25) st->ipage = 0;
    return GPSIQ_OK;
}

int gpsiq_nav_roll(const uint32_t *sbf, int nchan, int week, double sec, gpsiq_nav_state_t *st)
{
    if (!sbf || !st || nchan < 0) return fail(GPSIQ_E_ARG, "null argument");
    typedef const uint32_t (*pages_t)[GPSIQ_N_DWRD_SBF];
    for (int i = 0; i < nchan; ++i) {
        const int rc = gpsiq_nav_message(reinterpret_cast<pages_t>(sbf + (size_t) i * GPSIQ_N_SBF_PAGE * GPSIQ_N_DWRD_SBF), week, sec, 0, &st[i]);
        if (rc != GPSIQ_OK) return rc;
    }
    return GPSIQ_OK;
}

int gpsiq_almanac_read_sem(const char *path, gpsiq_nav_alm_sv_t alm[GPSIQ_MAX_SAT])
{
    if (!path || !alm) return fail(GPSIQ_E_ARG, "null argument");
    std::memset(alm, 0, sizeof(gpsiq_nav_alm_sv_t) * GPSIQ_MAX_SAT);                  // almanac_init(), almanac.c:29-53
    std::FILE *fp = std::fopen(path, "rt");
    if (!fp) return fail(GPSIQ_E_ARG, "cannot open %s", path);
    char buf[100], title[32];
    unsigned n = 0, week = 0, sec = 0;
    // a line per field, in the file's order (almanac.c:85-147); `bad` = a read or a conversion failed
    bool bad = !std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%u %24s", &n, title) != 2;
    bad = bad || !std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%u %u", &week, &sec) != 2;
    n -= 1;                                            // PRNs count from 1; an announced 0 wraps and is cut to 32 as well
    if (n > 31) n = 31;
    for (unsigned j = 0; !bad && j <= n; ++j) {
        unsigned id = 0;
        if (!std::fgets(buf, sizeof buf, fp)) { bad = true; break; }
        if (buf[0] == '\n' || buf[0] == '\r')          // a blank line between records
            if (!std::fgets(buf, sizeof buf, fp)) { bad = true; break; }
        if (std::sscanf(buf, "%u", &id) != 1) { bad = true; break; }
        if (id == 0) id = 1;
        if (id > 32) id = 32;
        gpsiq_nav_alm_sv_t &a = alm[id - 1];
        a.svid = id;
        unsigned short svn;
        unsigned char code;
        if (!std::fgets(buf, sizeof buf, fp)) { bad = true; break; }                    // SVN: optional, may be a blank line
        if (!(buf[0] == '\n' || buf[0] == '\r') && std::sscanf(buf, "%hu", &svn) != 1) { bad = true; break; }
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%hhu", &code) != 1) { bad = true; break; }              // URA
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%lf %lf %lf", &a.e, &a.delta_i, &a.omegadot) != 3) { bad = true; break; }
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%lf %lf %lf", &a.sqrta, &a.omega0, &a.aop) != 3) { bad = true; break; }
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%lf %lf %lf", &a.m0, &a.af0, &a.af1) != 3) { bad = true; break; }
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%hhu", &code) != 1) { bad = true; break; }              // health
        if (!std::fgets(buf, sizeof buf, fp) || std::sscanf(buf, "%hhu", &code) != 1) { bad = true; break; }              // configuration
        a.toa_week = (int) week + 2048;                // almanac.c:160-163: the full week as read, plus one roll-over
        a.toa_sec = (double) sec;
        a.valid = 1;
    }
    // almanac.c:169-182: a file that simply ends early keeps what was read; anything else drops it all
    if (bad && !std::feof(fp)) std::memset(alm, 0, sizeof(gpsiq_nav_alm_sv_t) * GPSIQ_MAX_SAT);
    std::fclose(fp);
    int valid = 0;
    for (int sv = 0; sv < GPSIQ_MAX_SAT; ++sv) valid += alm[sv].valid ? 1 : 0;
    return valid;
}

}  // extern "C"
