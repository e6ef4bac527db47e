"""Run-ahead host pipeline over the C-ABI: everything the reference's gps thread computes on
the host between channel allocation and the sample loop, for a whole scenario at once.

    RINEX ephemeris set -> subframes (eph2sbf) -> nav words (generateNavMsg) -> per-block
    range / code phase / gain (gps.c:2731-2765) -> gpsiq_chan_t[nblocks][nchan]

with the reference's 30 s navigation-message refresh (gps.c:2870, 2878-2885) at the right
blocks.  RunAhead keeps a fixed set of satellites; RunAheadAllocating also restates the
allocation policy of allocateChannel() (gps.c:2164-2235: visible satellites, lowest PRN first,
into the first free channel, at the start and at every 30 s refresh, always seen from the start
position as the reference does) and the switch to the next ephemeris set (gps.c:2889-2906).
Every call below is one C-ABI entry point; this module is only the loop around them.
"""
import numpy as np

from . import (CHAN_DTYPE, QCHAN_DTYPE, EPHEM_DTYPE, IONO_DTYPE, NAV_STATE_DTYPE, TRACK_DTYPE, nav_message, nav_roll, nav_subframes,
               refresh_batch, refresh_epochs, refresh_epochs_quantized, sat_visibility, track_init)


def gps_time_after(sec, steps):
    """incGpsTime(.., 0.1) applied `steps` times to a time that is a whole number of
    milliseconds (gps.c:1105-1124 rounds to the millisecond at every step)."""
    return round(round(sec * 1000.0) + 100.0 * steps) / 1000.0


def epoch_plan(sec, nblocks):
    """Split blocks 0..nblocks-1 (block k is generated at receiver time sec + 0.1*(k+1)) at
    the blocks after which the reference refreshes the navigation message: whenever the
    block's time is a multiple of 30 s (igrx % 300 == 0, gps.c:2870-2878).
    Returns [(first_block, one_past_last_block, roll_after)] ."""
    plan, b = [], 0
    t0 = round(sec * 10.0)                       # tenths of a second
    while b < nblocks:
        to_edge = 300 - (int(t0) + b) % 300      # blocks up to and including the one at the next 30 s edge
        e = min(nblocks, b + to_edge)
        plan.append((b, e, e == b + to_edge))
        b = e
    return plan


class RunAhead:
    def __init__(self, eph_set, utc, svs, week, sec, xyz0, alm=None, iono=None):
        """eph_set: gpsiq_rinex_eph_t[32] (one ephemeris set); svs: satellite indices (0-based)
        to allocate, one channel each; (week, sec): receiver start time; xyz0: ECEF start position."""
        self.week, self.sec = int(week), float(sec)
        self.svs = list(svs)
        n = len(self.svs)
        self.orbit = np.ascontiguousarray(eph_set[self.svs]["orbit"])
        if iono is None:
            iono = np.zeros((), dtype=IONO_DTYPE)
            iono["enable"], iono["vflg"], iono["alpha"], iono["beta"] = 1, utc["vflg"], utc["alpha"], utc["beta"]
        self.iono = iono
        self.sbf = np.zeros((n, 53, 10), dtype=np.uint32)
        self.nav = np.zeros(n, dtype=NAV_STATE_DTYPE)
        self.trk = np.zeros(n, dtype=TRACK_DTYPE)
        for i, sv in enumerate(self.svs):
            self.sbf[i] = nav_subframes(eph_set[sv]["nav"], utc, alm)          # eph2sbf, gps.c:2190
            nav_message(self.sbf[i], self.week, self.sec, True, self.nav[i:i + 1])   # gps.c:2196
            self.trk[i]["prn"] = sv + 1
            self.trk[i]["g0_week"], self.trk[i]["g0_sec"] = self.nav[i]["g0_week"], self.nav[i]["g0_sec"]
            self.trk[i]["dwrd"] = self.nav[i]["dwrd"]
        self.ipage0 = self.nav["ipage"].copy()
        track_init(self.orbit, self.iono, self.week, self.sec, np.asarray(xyz0, dtype=np.float64), self.trk)   # gps.c:2199-2214
        self.carr_phase0 = self.trk["carr_phase"].copy()
        self.blocks_done = 0

    def _roll(self, t_roll):
        nav_roll(self.sbf, self.week, t_roll, self.nav)                       # gps.c:2878-2885, all channels in one call
        self.trk["dwrd"] = self.nav["dwrd"]
        self.trk["g0_week"], self.trk["g0_sec"] = self.nav["g0_week"], self.nav["g0_sec"]

    def seek(self, block, xyz_prev):
        """Put the host state where the loop has it just before block `block` (at or after the current one),
        without refreshing the blocks in between: a rank of a time-sharded run starts here, and skips the other
        ranks' blocks here between its rounds.  Navigation words: a frame depends on its time, the week and the
        page counter only (words 2 and 10 are solved so that every subframe ends on parity bits 00, gps.c:2093,
        2127, so nothing of the chain crosses a subframe), and the word buffer holds the previous frame's
        subframe 5 and the current frame (gps.c:2098-2103) -- so however many 30 s edges are passed, the last
        two rolls with the page counter moved on give the buffer all of them would.  The previous block's
        pseudorange (chan.rho0, gps.c:2063) is recomputed at its time and position xyz_prev -- a block's range
        depends on nothing but time and position, so this is what the refresh would have left."""
        n = block - self.blocks_done
        assert n >= 0
        if n == 0:
            return
        first = 300 - int(round(gps_time_after(self.sec, self.blocks_done) * 10.0)) % 300     # epoch_plan's first edge
        if n >= first:
            edges = 1 + (n - first) // 300
            if edges > 2:
                self.nav["ipage"] = (self.nav["ipage"] + (edges - 2)) % 25
            for k in range(max(0, edges - 2), edges):
                self._roll(gps_time_after(self.sec, self.blocks_done + first + 300 * k))
        carr = self.trk["carr_phase"].copy()
        track_init(self.orbit, self.iono, self.week, gps_time_after(self.sec, block), np.asarray(xyz_prev, dtype=np.float64), self.trk)
        self.trk["carr_phase"] = carr                    # the loop's state, not the host model's (gps.c:2821)
        self.blocks_done = block

    def descriptors(self, xyz, carr_phase=None, gain_x2=False, nthreads=0, out=None):
        """Channel state at gps.c:2766 for the next len(xyz) blocks (xyz[k] = ECEF position of
        block k).  carr_phase: what the previous Context.generate_batch call handed out
        (carr_out) when continuing a run; None on the first call = the allocation's value.
        Only block 0's carr_phase is read by the library, which carries it exactly from there."""
        return self._refresh(xyz, carr_phase, gain_x2, nthreads, out, None)

    def descriptors_quantized(self, xyz, fs, nsamp, carr_phase=None, gain_x2=False, nthreads=0, out=None):
        """quantize_blocks(descriptors(xyz, ...), fs, nsamp)[0] in one pass in C (gpsiq_refresh_epochs_quantized): the
        double-precision descriptors are never written out.  For hosts that feed Context.set_descriptors /
        generate_quantized (the rounds of a time-sharded run)."""
        return self._refresh(xyz, carr_phase, gain_x2, nthreads, out, (float(fs), int(nsamp)))

    def _refresh(self, xyz, carr_phase, gain_x2, nthreads, out, quant):
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        # The satellites are fixed, so the whole call is ONE threaded pass in C (gpsiq_refresh_epochs): the word
        # buffers of the 30 s epochs it crosses are rolled first (cheap), the ranges -- the expensive part -- do
        # not depend on them.
        t_start = gps_time_after(self.sec, self.blocks_done)
        plan = epoch_plan(t_start, len(xyz))
        if not plan:
            return np.zeros((0, len(self.svs)), dtype=CHAN_DTYPE if quant is None else QCHAN_DTYPE)
        trk_ep = np.empty((len(plan), len(self.svs)), dtype=TRACK_DTYPE)
        done = self.blocks_done
        for e, (b0, b1, roll) in enumerate(plan):
            trk_ep[e] = self.trk
            done += b1 - b0
            if roll:                                                          # gps.c:2878-2885
                self._roll(gps_time_after(self.sec, done))
        # the loop's own state (gps.c:2821), not the host model's; only block 0's value is read by the library
        # (it carries the phase itself from there), the reference keeps it in chan[i] between blocks
        carr0 = self.carr_phase0 if carr_phase is None else np.asarray(carr_phase, dtype=np.float64)
        # out: a caller-owned [len(xyz)][nchan] array to fill (a fresh 20 MB array per call costs as much in page
        # faults as the refresh itself)
        first = [p[0] for p in plan]
        if quant is None:
            desc = refresh_epochs(self.orbit, self.iono, self.week, t_start, xyz, trk_ep, first, gain_x2=gain_x2, nthreads=nthreads, out=out)
            if len(desc):
                desc["carr_phase"][0] = carr0
        else:
            trk_ep[0]["carr_phase"] = carr0                                   # the quantiser seeds block 0 from it
            desc = refresh_epochs_quantized(self.orbit, self.iono, self.week, t_start, xyz, trk_ep, first, quant[0], quant[1],
                                            gain_x2=gain_x2, nthreads=nthreads, out=out)
        for f in ("rho0_week", "rho0_sec", "rho0_range"):                     # chan.rho0 = rho1 (gps.c:2063)
            self.trk[f] = trk_ep[0][f]
        self.blocks_done = done
        return desc


class RunAheadAllocating:
    """RunAhead with the reference's channel allocation in the loop: nchan channels, satellites
    allocated and released at the start and at every 30 s refresh as allocateChannel() does."""

    def __init__(self, eph_sets, utc, nchan, week, sec, xyz0, ieph=0, alm=None, iono=None):
        """eph_sets: gpsiq_rinex_eph_t[nsets][32] (or one set [32]); ieph: the set to start with
        (gpsiq.rinex_select)."""
        self.week, self.sec, self.nchan = int(week), float(sec), int(nchan)
        self.eph_sets = eph_sets if eph_sets.ndim == 2 else eph_sets[None, :]
        self.ieph = int(ieph)
        self.eph_set, self.utc, self.alm = self.eph_sets[self.ieph], utc, alm
        self.xyz0 = np.ascontiguousarray(xyz0, dtype=np.float64)
        if iono is None:
            iono = np.zeros((), dtype=IONO_DTYPE)
            iono["enable"], iono["vflg"], iono["alpha"], iono["beta"] = 1, utc["vflg"], utc["alpha"], utc["beta"]
        self.iono = iono
        self.allocated_sat = [-1] * 32                                   # gps.c:2668-2669
        self.orbit = np.zeros(self.nchan, dtype=EPHEM_DTYPE)
        self.sbf = np.zeros((self.nchan, 53, 10), dtype=np.uint32)
        self.nav = np.zeros(self.nchan, dtype=NAV_STATE_DTYPE)
        self.trk = np.zeros(self.nchan, dtype=TRACK_DTYPE)               # prn 0 = free channel (gps.c:2664-2665)
        self.blocks_done = 0
        self.nsat = [self.allocate(self.sec)]                            # gps.c:2675

    def allocate(self, t):
        """allocateChannel() at receiver time t (gps.c:2164-2235); returns the number of visible satellites."""
        nsat = 0
        for sv in range(32):
            e = self.eph_set[sv]
            visible = bool(e["vflg"]) and sat_visibility(e["orbit"], self.week, t, self.xyz0, 0.0)[0]
            if visible:
                nsat += 1
                if self.allocated_sat[sv] == -1:
                    for i in range(self.nchan):
                        if self.trk[i]["prn"] == 0:
                            self.trk[i]["prn"] = sv + 1
                            self.orbit[i] = e["orbit"]
                            self.sbf[i] = nav_subframes(e["nav"], self.utc, self.alm)               # gps.c:2190
                            ipage = self.nav[i]["ipage"]          # chan[i].ipage survives release and re-allocation:
                            self.nav[i] = np.zeros((), dtype=NAV_STATE_DTYPE)      # only generateNavMsg ever touches it
                            self.nav[i]["ipage"] = ipage                          # (gps.c:2137-2139)
                            nav_message(self.sbf[i], self.week, t, True, self.nav[i:i + 1])          # gps.c:2193
                            self.trk[i]["g0_week"], self.trk[i]["g0_sec"] = self.nav[i]["g0_week"], self.nav[i]["g0_sec"]
                            self.trk[i]["dwrd"] = self.nav[i]["dwrd"]
                            track_init(self.orbit[i:i + 1], self.iono, self.week, t, self.xyz0, self.trk[i:i + 1])   # gps.c:2196-2214
                            self.allocated_sat[sv] = i
                            break
            elif self.allocated_sat[sv] >= 0:                            # gps.c:2224-2231
                self.trk[self.allocated_sat[sv]]["prn"] = 0
                self.allocated_sat[sv] = -1
        return nsat

    def refresh_ephemeris(self, t):
        """gps.c:2889-2906: when the first valid satellite of the next set has its time of clock less
        than an hour ahead, that set takes over (for the orbits immediately, for the navigation words
        from the next refresh on: the subframes are rebuilt here, the word buffer is not)."""
        if self.ieph + 1 >= len(self.eph_sets):
            return
        nxt = self.eph_sets[self.ieph + 1]
        for sv in range(32):
            if nxt[sv]["vflg"]:
                dt = (int(nxt[sv]["toc_week"]) - self.week) * 604800.0 + (float(nxt[sv]["nav"]["toc_sec"]) - t)   # subGpsTime, gps.c:1096-1103
                if dt < 3600.0:
                    self.ieph += 1
                    self.eph_set = self.eph_sets[self.ieph]
                    for i in range(self.nchan):
                        if self.trk[i]["prn"] != 0:
                            e = self.eph_set[self.trk[i]["prn"] - 1]
                            self.sbf[i] = nav_subframes(e["nav"], self.utc, self.alm)
                            self.orbit[i] = e["orbit"]
                break

    def descriptors(self, xyz, carr_phase=None, gain_x2=False, nthreads=0):
        """As RunAhead.descriptors.  carr_phase of every block is the value allocateChannel() gave the
        channel's current satellite: the library re-seeds a slot from it whenever the slot's PRN changes
        and carries the phase itself otherwise.  carr_phase: when a scenario is rendered in several
        descriptors() + generate_batch() calls, what the previous generate_batch handed out (carr_out); it
        goes into block 0 for the slots that still hold the satellite they held in the previous call's last
        block, so the run continues exactly (slots re-allocated in between start from their allocation)."""
        xyz = np.ascontiguousarray(xyz, dtype=np.float64).reshape(-1, 3)
        first_prn = self.trk["prn"].copy()
        out = []
        for b0, b1, roll in epoch_plan(gps_time_after(self.sec, self.blocks_done), len(xyz)):
            t = gps_time_after(self.sec, self.blocks_done)
            if self.trk["prn"].max() > 0:
                d = refresh_batch(self.orbit, self.iono, self.week, t, xyz[b0:b1], self.trk, gain_x2=gain_x2, nthreads=nthreads)
            else:
                d = np.zeros((b1 - b0, self.nchan), dtype=CHAN_DTYPE)
            d["carr_phase"] = np.where(self.trk["prn"] > 0, self.trk["carr_phase"], 0.0)[None, :]
            out.append(d)
            self.blocks_done += b1 - b0
            if roll:
                t_roll = gps_time_after(self.sec, self.blocks_done)
                for i in range(self.nchan):                              # gps.c:2878-2885
                    if self.trk[i]["prn"] > 0:
                        nav_message(self.sbf[i], self.week, t_roll, False, self.nav[i:i + 1])
                        self.trk[i]["dwrd"] = self.nav[i]["dwrd"]
                        self.trk[i]["g0_week"], self.trk[i]["g0_sec"] = self.nav[i]["g0_week"], self.nav[i]["g0_sec"]
                self.refresh_ephemeris(t_roll)                           # gps.c:2889-2906
                self.nsat.append(self.allocate(t_roll))                  # gps.c:2909
        desc = np.concatenate(out) if out else np.zeros((0, self.nchan), dtype=CHAN_DTYPE)
        if len(desc):
            if carr_phase is not None and getattr(self, "_prn_at_end", None) is not None:
                keep = (first_prn > 0) & (first_prn == self._prn_at_end)
                desc["carr_phase"][0] = np.where(keep, np.asarray(carr_phase, dtype=np.float64), desc["carr_phase"][0])
            self._prn_at_end = desc["prn"][-1].copy()
        return desc
