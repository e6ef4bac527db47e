/*
 * gpsiq_extras.h -- [convenience] restatements of reference host / CLI pieces OUTSIDE SURVEY.md section 8 (SEM almanac reader,
 * the -T time overwrite, date conversion, receiver-position inputs, tangent-frame move).  They exist so that
 * host/gpsiq_runahead.c can take the reference's own inputs; they are NOT part of the drop-in boundary (include/gpsiq.h), a port
 * of the reference does not need them (its own C host code stays), and the set is frozen.
 */
#ifndef GPSIQ_EXTRAS_H
#define GPSIQ_EXTRAS_H

#include "gpsiq_rows.h"

#ifdef __cplusplus
extern "C" {
#endif
#if defined(__GNUC__)
#pragma GCC visibility push(default)
#endif

/* [convenience] Where the receiver is: the two inputs gps_thread_ep() turns into its xyz[] array before the block loop.
 * gpsiq_llh_to_ecef = llh2xyz() (gps.c:412-447) for the static position `-l lat,lon,h` (gps.c:2480-2490 converts the
 * degrees to radians first): llh = latitude and longitude in RADIANS, height in metres.  gpsiq_ecef_to_llh = xyz2llh()
 * (gps.c:361-410).  gpsiq_motion_read_csv = readUserMotion() (gps.c:2253-2277): a text file with one line
 * "t,x,y,z" per 0.1 s (ECEF metres, t ignored), at most max_points lines; returns the number of points read, -1 if
 * the file cannot be opened. */
void gpsiq_llh_to_ecef(const double llh[3], double xyz[3]);
void gpsiq_ecef_to_llh(const double xyz[3], double llh[3]);
/* Move an ECEF position by (north, east, up) metres in the local tangent frame of the geodetic point llh_ref (radians,
 * metres): xyz += ltcmat(llh_ref)^T * neu, the three lines the reference uses for its target offset (-T distance, bearing:
 * gps.c:2350-2356, neu = distance*cos, distance*sin, height) and for every step of its interactive mode (gps.c:2720-2728,
 * neu = velocity*0.1*cos, velocity*0.1*sin, vertical_speed*0.1); the frame stays that of the START location, as there. */
void gpsiq_ecef_add_neu(const double llh_ref[3], const double neu[3], double xyz[3]);
int  gpsiq_motion_read_csv(const char *path, double *xyz /* [max_points][3] */, int max_points);


/* [convenience] almanac_read_file() (almanac.c:73-184): a SEM almanac file -> the 32 entries gpsiq_nav_subframes() takes, indexed by
 * PRN - 1.  The reference's rules are kept: ids 0 / > 32 are clamped to 1 / 32, at most 32 records are read whatever the
 * header announces, the week gets + 2048 (the reference's roll-over constant), a file that ends early keeps the records
 * read so far (the last one possibly half filled and not valid), any other damage drops them all.
 * Returns the number of valid entries, or GPSIQ_E_ARG when the file cannot be opened. */
int gpsiq_almanac_read_sem(const char *path, gpsiq_nav_alm_sv_t alm[32] /* GPSIQ_MAX_SAT */);

/* [convenience] The reference's -T option (gps.c:2534-2561): move the times of clock and of ephemeris of every valid record, and its
 * calendar time, by the distance from the first set's first time of clock (gps.c:2507-2513) to the start time cut to
 * whole two hours, and set the UTC reference (wnt, tot) to that cut time: an old broadcast file then serves any start
 * time.  eph is [nsets][GPSIQ_MAX_SAT] as gpsiq_rinex_read() filled it. */
int gpsiq_rinex_overwrite_time(gpsiq_rinex_eph_t *eph, int nsets, gpsiq_nav_utc_t *utc, int week, double sec);
/* [convenience] date2gps() / gps2date() (gps.c:315-355): a calendar date and time of day <-> GPS week and seconds of the week, as the
 * reference converts its -t start time and RINEX epochs (no leap seconds either way; months outside 1..12 count as
 * January where the reference indexes past its table). */
void gpsiq_date_to_gps(int year, int month, int day, int hour, int minute, double second, int *week, double *sec);
void gpsiq_gps_to_date(int week, double sec, int *year, int *month, int *day, int *hour, int *minute, double *second);

#if defined(__GNUC__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif
