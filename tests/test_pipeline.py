"""The run-ahead host pipeline (gpsiq/pipeline.py: RINEX set -> subframes -> nav words -> per-block
refresh, with the 30 s navigation-message refresh at the right blocks) against the reference's
own loop with its own refresh lines in it (oracle/_ref: gps.c:2731-2765, 2870, 2878-2885), over
several 30 s epochs, and on to IQ bytes on the GPU."""
import numpy as np
import pytest

import gpsiq
from gpsiq.abi import NAV_STATE_DTYPE, SC08, TRACK_DTYPE
from gpsiq.pipeline import RunAhead, epoch_plan, gps_time_after
from gpsiq.scenario import circle_track, llh_to_ecef, synth_rinex_records, write_rinex_nav

TOKYO = llh_to_ecef(35.681298, 139.766247, 10.0)
WEEK = 2190
UTC = dict(alpha=[0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06], beta=[0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07],
           A0=-0.931322574615e-09, A1=-0.355271367880e-14, tot=233472, wnt=2190, dtls=18)


def test_epoch_plan_is_the_reference_rule():
    """igrx % 300 == 0 after the block generated at a multiple of 30 s (gps.c:2870-2878)."""
    for sec, nb in ((270000.0, 700), (270020.0, 700), (270029.9, 5), (270000.1, 299), (270010.0, 0)):
        plan = epoch_plan(sec, nb)
        if nb == 0:
            assert plan == []
            continue
        assert [p[0] for p in plan] == [0] + [p[1] for p in plan[:-1]] and plan[-1][1] == nb
        for b0, b1, roll in plan:
            times = [int(round(gps_time_after(sec, k + 1) * 10.0 + 0.0)) for k in range(b0, b1)]
            assert all(t % 300 != 0 for t in times[:-1])
            assert (times[-1] % 300 == 0) == roll


def scenario(tmp_path, sec, nblocks, moving=False):
    recs = synth_rinex_records(12, TOKYO, WEEK, 270000.0, seed=35, sets=2)
    path = write_rinex_nav(str(tmp_path / "run.21n"), recs, UTC, 2)
    eph, utc, n = gpsiq.rinex_read(path, 2)
    ieph = gpsiq.rinex_select(eph, n, WEEK, sec)
    svs = [sv for sv in range(32) if eph[ieph, sv]["vflg"]]
    if moving:
        xyz = circle_track(TOKYO, nblocks, radius_m=200.0, period_s=45.0)
    else:
        xyz = np.repeat(TOKYO[None, :], nblocks + 1, axis=0)
    return path, eph, utc, ieph, svs, xyz


@pytest.mark.parametrize("sec,moving", [(270020.0, False), (270000.0, True)])
def test_multi_epoch_descriptors_match_the_reference_loop(ref, tmp_path, sec, moving):
    nblocks = 700
    path, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving)
    ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
    desc = ra.descriptors(xyz[1:])

    reph, rutc, rn = ref.read_rinex(path, 2)
    n = len(svs)
    rtrk = np.zeros(n, dtype=TRACK_DTYPE)
    rsbf = np.zeros((n, 53, 10), dtype=np.uint32)
    ipage = np.zeros(n, dtype=np.int32)
    for i, sv in enumerate(svs):
        rsbf[i] = ref.nav_subframes(reph[ieph, sv]["nav"], rutc)
        st = np.zeros(1, dtype=NAV_STATE_DTYPE)
        ref.nav_message(rsbf[i], WEEK, sec, True, st)
        rtrk[i]["prn"], rtrk[i]["g0_week"], rtrk[i]["g0_sec"], rtrk[i]["dwrd"] = sv + 1, st[0]["g0_week"], st[0]["g0_sec"], st[0]["dwrd"]
        ipage[i] = st[0]["ipage"]
    want, carr = ref.refresh_epochs(np.ascontiguousarray(reph[ieph, svs]["orbit"]), ra.iono, WEEK, sec, xyz, rtrk, rsbf, ipage)
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd"):
        assert desc[f].tobytes() == want[f].tobytes(), f
    assert desc["carr_phase"][0].tobytes() == carr.tobytes()
    # the navigation words really were refreshed where the reference refreshes them
    edges = [e for _, e, roll in epoch_plan(sec, nblocks) if roll and e < nblocks]
    assert len(edges) >= 2
    for e in edges:
        assert desc["dwrd"][e].tobytes() != desc["dwrd"][e - 1].tobytes()
        assert desc["dwrd"][e - 1].tobytes() == desc["dwrd"][max(e - 300, 0)].tobytes()
        assert (desc["iword"][e] < desc["iword"][e - 1]).all()          # the word counter starts over
    assert len(svs) == 12 and (desc["iword"] < 60).all()


@pytest.mark.parametrize("sec,moving", [(270020.0, False), (270000.0, True)])
def test_a_rank_can_start_anywhere(tmp_path, sec, moving):
    """Host-side sharding: RunAhead.seek(b0) + descriptors(its range) == the same rows of the full run,
    for starts before, on and after the 30 s navigation-message refreshes."""
    nblocks = 700
    path, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving)
    full = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    edges = [e for _, e, roll in epoch_plan(sec, nblocks) if roll]
    for b0 in (0, 1, 57, edges[0] - 1, edges[0], edges[0] + 1, edges[1], 650):
        b1 = min(nblocks, b0 + 40)
        ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
        ra.seek(b0, xyz[b0])                       # xyz[b0] is the position of block b0-1 (xyz[0]: the start position)
        part = ra.descriptors(xyz[1 + b0:1 + b1])
        assert part.tobytes() == full[b0:b1].tobytes(), b0


def test_seek_jumps_any_number_of_epochs_and_goes_on_from_anywhere(tmp_path):
    """seek() rolls the navigation words only twice however many 30 s edges it passes (the page counter is moved
    on instead) and may be called again after descriptors(): the rounds of a time-sharded run.  8 100 blocks = 27
    edges, past the wrap of the 25-page counter."""
    sec, nblocks = 270020.0, 8100
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving=False)
    full = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    for b0 in (1500, 7790, 7800, 8050):
        ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
        ra.seek(b0, xyz[b0])
        assert ra.descriptors(xyz[1 + b0:1 + b0 + 50]).tobytes() == full[b0:b0 + 50].tobytes(), b0
    ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
    for b0, n in ((0, 120), (120, 1), (400, 350), (2900, 100), (3000, 20), (7600, 500)):
        ra.seek(b0, xyz[b0])
        assert ra.descriptors(xyz[1 + b0:1 + b0 + n]).tobytes() == full[b0:b0 + n].tobytes(), b0


def test_rounds_of_a_time_sharded_run_quantise_like_one_timeline(tmp_path):
    """quantize_own_shard(history=...): round m gives rank r the blocks after rank r-1's of round m and after all
    of round m-1; every rank's rows equal the rows of the whole timeline quantised in one go."""
    from gpsiq.shard import quantize_own_shard
    sec, world, nb, rounds, fs, ns = 270000.0, 3, 70, 4, 2.6e6, 260000
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, world * nb * rounds, moving=True)
    full = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    want, _ = gpsiq.quantize_blocks(full, fs, ns)
    ras = [RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]) for _ in range(world)]
    hist = [[] for _ in range(world)]
    for m in range(rounds):
        own, carries = [], []
        for r in range(world):
            b0 = (m * world + r) * nb
            ras[r].seek(b0, xyz[b0])
            own.append(ras[r].descriptors(xyz[1 + b0:1 + b0 + nb]))
            carries.append(gpsiq.shard_carry(gpsiq.quantize_blocks(own[r], fs, ns)[0], ns).tobytes())
        for r in range(world):
            b0 = (m * world + r) * nb
            got = quantize_own_shard(own[r], fs, ns, r, world, lambda mine: carries, history=hist[r])
            assert got.tobytes() == want[b0:b0 + nb].tobytes(), (m, r)
    assert all(len(h) == world * rounds for h in hist)


def test_golden_multi_epoch_capture(tmp_path):
    """Committed capture of the reference loop incl. its nav refresh (runs without /root/reference):
    every block's descriptors by SHA-256, the blocks either side of each refresh byte for byte."""
    import hashlib
    import os
    from gpsiq.abi import CHAN_DTYPE
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "epochs_circle.npz"))
    sec, nblocks = float(z["sec"]), int(z["nblocks"])
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving=True)
    assert svs == list(z["svs"])
    desc = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    want = np.ascontiguousarray(z["desc_keep"]).view(CHAN_DTYPE).reshape(len(z["keep"]), len(svs))
    for k, b in enumerate(z["keep"]):
        assert desc[b].tobytes() == want[k].tobytes(), b
    for b in range(nblocks):
        assert hashlib.sha256(desc[b].tobytes()).digest() == z["sha256"][b].tobytes(), b


def test_descriptors_in_pieces_equal_one_call(tmp_path):
    """Calling the pipeline epoch by epoch (or in odd pieces) gives the same blocks as one call."""
    sec, nblocks = 270010.0, 450
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving=True)
    whole = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
    parts = [ra.descriptors(xyz[1 + a:1 + b]) for a, b in ((0, 7), (7, 200), (200, 201), (201, 450))]
    assert np.concatenate(parts).tobytes() == whole.tobytes()


@pytest.mark.gpu
def test_multi_epoch_scenario_to_samples_on_gpu(oracle, tmp_path):
    """RINEX -> two navigation refreshes -> IQ bytes: the blocks either side of each refresh
    (and the first and last) equal the oracle, batch call == epoch-by-epoch calls."""
    sec, nblocks, fs, ns = 270020.0, 420, 2.6e6, 260000
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving=True)
    desc = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors(xyz[1:])
    import torch
    ctx = gpsiq.Context(0)
    buf = torch.empty(nblocks * 2 * ns, dtype=torch.int8, device="cuda")
    ctx.generate_batch(desc, ns, fs, SC08, device_ptr=buf.data_ptr())
    q = oracle.quantize_blocks(desc, fs, ns)
    for b in (0, 99, 100, 101, 399, 400, 419):
        got = buf[b * 2 * ns:(b + 1) * 2 * ns].cpu().numpy()
        assert np.array_equal(got, oracle.block_fixed(q[b], ns, SC08)), b
    # epoch by epoch with the carrier handed over, as INTEGRATION.md section 3 describes
    ctx2 = gpsiq.Context(0)
    ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
    buf2 = torch.empty_like(buf)
    carr = None
    for b0, b1, _ in epoch_plan(sec, nblocks):
        d = ra.descriptors(xyz[1 + b0:1 + b1], carr_phase=carr)
        carr = np.zeros(len(svs))
        ctx2.generate_batch(d, ns, fs, SC08, device_ptr=buf2.data_ptr() + b0 * 2 * ns, carr_out=carr)
    assert torch.equal(buf, buf2)
    ctx.close(); ctx2.close()


def horizon_scenario(tmp_path, nblocks, seed=5, sec=270000.0):
    """9 satellites well above the horizon and 9 within +-0.35 degrees of it at the start time, so
    that allocateChannel() has satellites rising and setting at the 30 s refreshes."""
    from gpsiq.scenario import _elevation_deg, synth_constellation
    high = synth_constellation(9, TOKYO, sec, seed=seed, min_elev_deg=15.0)
    near = synth_constellation(60, TOKYO, sec, seed=seed + 100, min_elev_deg=-0.45, max_elev_deg=0.45)
    el0 = np.array([_elevation_deg(e, sec, TOKYO) for e in near])
    el1 = np.array([_elevation_deg(e, sec + 60.0, TOKYO) for e in near])
    setting = near[(el0 > 0.05) & (el1 < el0 - 0.15)][:4]        # up now, below the horizon within a minute or two
    rising = near[(el0 < -0.05) & (el1 > el0 + 0.15)][:5]        # the other way round
    assert len(setting) == 4 and len(rising) == 5
    orbits = np.concatenate([high, setting, rising])[np.random.default_rng(seed).permutation(18)]
    recs = synth_rinex_records(len(orbits), TOKYO, WEEK, sec, seed=seed, sets=2, eph=orbits)
    path = write_rinex_nav(str(tmp_path / "horizon.21n"), recs, UTC, 2)
    eph, utc, n = gpsiq.rinex_read(path, 2)
    ieph = gpsiq.rinex_select(eph, n, WEEK, sec)                 # the set gps_thread_ep would start with (gps.c:2588-2608)
    assert ieph >= 0
    xyz = circle_track(TOKYO, nblocks, radius_m=150.0, period_s=50.0)
    return path, eph[:n], ieph, utc, xyz, sec


def test_visibility_matches_reference(ref, tmp_path):
    """gpsiq_sat_visibility == checkSatVisibility (gps.c:2142-2162), through the allocation it drives:
    the number of visible satellites allocateChannel() reports at every call."""
    from gpsiq.pipeline import RunAheadAllocating
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, 1300)
    ra = RunAheadAllocating(eph, utc, 12, WEEK, sec, xyz[0], ieph=ieph)
    ra.descriptors(xyz[1:])
    _, nsat, _ = ref.run_host(eph, ieph, utc, WEEK, sec, xyz, 12)
    assert list(nsat) == ra.nsat and len(nsat) == 5
    vis, azel = gpsiq.sat_visibility(eph[ieph, 0]["orbit"], WEEK, sec, TOKYO)
    assert 0.0 <= azel[0] < 2 * np.pi and -np.pi / 2 <= azel[1] <= np.pi / 2 and vis == (azel[1] > 0.0)


@pytest.mark.parametrize("nchan,seed", [(12, 5), (16, 6), (8, 7)])
def test_allocating_pipeline_matches_the_reference_loop(ref, tmp_path, nchan, seed):
    """Channel allocation, release and re-allocation at the 30 s refreshes: every descriptor field of
    every block equals the reference's own allocateChannel + refresh + nav-refresh lines."""
    from gpsiq.pipeline import RunAheadAllocating
    nblocks = 1300
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=seed)
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    desc = ra.descriptors(xyz[1:])
    want, nsat, ieph_end = ref.run_host(eph, ieph, utc, WEEK, sec, xyz, nchan)
    assert list(nsat) == ra.nsat and ieph_end == ra.ieph
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd", "carr_phase"):
        assert desc[f].tobytes() == want[f].tobytes(), f
    prn = desc["prn"]
    changes = int((prn[1:] != prn[:-1]).sum())
    assert changes >= 1, "the scenario should exercise release / allocation"
    if nchan == 16:          # room for everything: satellites are released and others allocated
        assert ((prn[1:] == 0) & (prn[:-1] > 0)).any() and ((prn[1:] > 0) & (prn[:-1] == 0)).any()
    # changes only happen right after a 30 s refresh
    for b in np.nonzero((prn[1:] != prn[:-1]).any(axis=1))[0]:
        assert (b + 1) % 300 == 0


@pytest.mark.gpu
def test_allocating_scenario_to_samples_on_gpu(oracle, tmp_path):
    """Satellites released and allocated at a 30 s refresh: a slot that changes PRN is re-seeded
    from the new satellite's carrier phase, every other slot is carried exactly; IQ == oracle."""
    from gpsiq.abi import SC16
    from gpsiq.pipeline import RunAheadAllocating
    nblocks, fs, ns = 330, 2.6e6, 260000
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=6)
    desc = RunAheadAllocating(eph, utc, 16, WEEK, sec, xyz[0], ieph=ieph).descriptors(xyz[1:])
    assert (desc["prn"][300] != desc["prn"][299]).any()
    import torch
    ctx = gpsiq.Context(0)
    buf = torch.empty(nblocks * 2 * ns, dtype=torch.int16, device="cuda")
    ctx.generate_batch(desc, ns, fs, SC16, device_ptr=buf.data_ptr())
    q = oracle.quantize_blocks(desc, fs, ns)
    for b in (0, 298, 299, 300, 301, 329):
        got = buf[b * 2 * ns:(b + 1) * 2 * ns].cpu().numpy()
        assert np.array_equal(got, oracle.block_fixed(q[b], ns, SC16)), b
    ctx.close()


def test_switch_to_the_next_ephemeris_set_matches_reference(ref, tmp_path):
    """Start 10 s before the hour mark of the next set: the second 30 s refresh switches sets
    (gps.c:2889-2906) — new orbits at once, new subframes in the words from the refresh after."""
    from gpsiq.pipeline import RunAheadAllocating
    nblocks, nchan = 1000, 14
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=9, sec=269990.0)
    assert ieph == 0 and len(eph) == 2
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    desc = ra.descriptors(xyz[1:])
    want, nsat, ieph_end = ref.run_host(eph, ieph, utc, WEEK, sec, xyz, nchan)
    assert ieph_end == 1 and ra.ieph == 1 and list(nsat) == ra.nsat
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd", "carr_phase"):
        assert desc[f].tobytes() == want[f].tobytes(), f


def test_golden_allocation_capture(tmp_path):
    """Committed capture of the reference's allocating host loop (runs without /root/reference)."""
    import hashlib
    import os
    from gpsiq.pipeline import RunAheadAllocating
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "alloc_horizon.npz"))
    nblocks, nchan = int(z["nblocks"]), int(z["nchan"])
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=int(z["seed"]), sec=float(z["sec"]))
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    desc = ra.descriptors(xyz[1:])
    assert ra.nsat == list(z["nsat"])
    assert np.array_equal(desc["prn"], z["prn"])
    for b in range(nblocks):
        assert hashlib.sha256(desc[b].tobytes()).digest() == z["sha256"][b].tobytes(), b


@pytest.mark.gpu
def test_c_runahead_program_equals_the_python_pipeline(oracle, tmp_path):
    """host/gpsiq_runahead.c (RINEX -> allocation -> refresh -> IQ -> file, every step a C-ABI call,
    30 s epochs with nav refresh and re-allocation) writes the bytes the Python pipeline + one
    gpsiq_generate_batch produce, and those equal the oracle."""
    import os
    import subprocess
    from gpsiq.pipeline import RunAheadAllocating
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-sdr-gps-sim_amd", "host")
    subprocess.run(["make", "-s", "-C", host], check=True)
    nblocks, nchan, fs, ns = 430, 16, 2.6e6, 260000
    path, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=6, sec=269990.0)
    xyz.tofile(str(tmp_path / "xyz.bin"))
    out = str(tmp_path / "iq.bin")
    r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", str(WEEK), repr(sec), str(tmp_path / "xyz.bin"),
                        str(nblocks), str(nchan), repr(fs), "1", out], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(out, dtype=np.int8).reshape(nblocks, 2 * ns)
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    desc = ra.descriptors(xyz[1:])
    assert ra.ieph == ieph + 1, "the run should cross the switch to the next ephemeris set"
    ctx = gpsiq.Context(0)
    want = ctx.generate_batch(desc, ns, fs, SC08)
    ctx.close()
    assert np.array_equal(got, want)
    q = oracle.quantize_blocks(desc, fs, ns)
    for b in (0, 99, 100, 399, 400, 429):
        assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, SC08)), b


@pytest.mark.gpu
def test_c_runahead_position_argument_forms(tmp_path):
    """gpsiq_runahead's position argument: the reference's user-motion CSV (-m) and a static "lat,lon,h" (-l)
    give the stream the same positions give as an xyz.bin; a short CSV shortens the run (numd, gps.c:2502-2505)."""
    import os
    import subprocess
    host = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "multi-sdr-gps-sim_amd", "host")
    subprocess.run(["make", "-s", "-C", host], check=True)
    nblocks, nchan, fs, ns = 6, 8, 2.6e6, 260000
    path, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=8, sec=270026.0)   # two seconds into subframe 5: the data words of an almanac page
    xyz = np.round(xyz, 4)                                        # what the CSV's %.4f keeps
    def run(position, n=nblocks):
        out = str(tmp_path / "o.bin")
        r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", str(WEEK), repr(sec), position, str(n), str(nchan),
                            repr(fs), "1", out], capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        return np.fromfile(out, dtype=np.int8)
    xyz.tofile(str(tmp_path / "xyz.bin"))
    want = run(str(tmp_path / "xyz.bin"))
    assert want.size == nblocks * 2 * ns
    with open(tmp_path / "m.csv", "w") as f:
        for k, p in enumerate(xyz):
            f.write("%.1f,%.4f,%.4f,%.4f\n" % (0.1 * k, p[0], p[1], p[2]))
    assert np.array_equal(run(str(tmp_path / "m.csv")), want)
    assert np.array_equal(run(str(tmp_path / "m.csv"), n=50), want), "a 7-point file gives 6 blocks whatever was asked"
    llh = (35.681298, 139.766247, 10.0)
    static = np.tile(gpsiq.llh_to_ecef(llh[0] / 57.2957795131, llh[1] / 57.2957795131, llh[2]), (nblocks + 1, 1))
    static.tofile(str(tmp_path / "s.bin"))
    assert np.array_equal(run("%r,%r,%r" % llh), run(str(tmp_path / "s.bin")))
    # a SEM almanac file as the optional last argument: the almanac pages of subframes 4/5 get filled (other words, other bits)
    from gpsiq.pipeline import RunAheadAllocating
    from test_nav import _sem_text
    rng = np.random.default_rng(4)
    recs = [dict(id=i, e=rng.uniform(0, 0.02), di=rng.uniform(-0.01, 0.01), od=-2.5e-9, sq=5153.6, o0=rng.uniform(-1, 1), w=rng.uniform(-1, 1),
                 m0=rng.uniform(-1, 1), af0=rng.uniform(-1e-4, 1e-4), af1=0.0) for i in range(1, 32)]
    (tmp_path / "almanac.sem").write_text(_sem_text(recs, week=WEEK - 2048, sec=405504))
    alm, nalm = gpsiq.almanac_read_sem(tmp_path / "almanac.sem")
    assert nalm == 31
    out = str(tmp_path / "a.bin")
    r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", str(WEEK), repr(sec), str(tmp_path / "xyz.bin"), str(nblocks), str(nchan),
                        repr(fs), "1", out, str(tmp_path / "almanac.sem")], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    with_alm = np.fromfile(out, dtype=np.int8)
    assert not np.array_equal(with_alm, want), "the almanac page of subframe 5 should differ from the empty one"
    ctx = gpsiq.Context(0)
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph, alm=alm)
    assert np.array_equal(with_alm, ctx.generate_batch(ra.descriptors(xyz[1:]), ns, fs, SC08).reshape(-1))
    ra0 = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    assert np.array_equal(want, ctx.generate_batch(ra0.descriptors(xyz[1:]), ns, fs, SC08).reshape(-1))
    ctx.close()
    (tmp_path / "old.sem").write_text(_sem_text(recs, week=WEEK - 2048 - 6, sec=405504))
    r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", str(WEEK), repr(sec), str(tmp_path / "xyz.bin"), "2", str(nchan),
                        repr(fs), "1", out, str(tmp_path / "old.sem")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "time of almanac" in r.stderr
    # the start time in the reference's -t form (2021/12/29 03:00:26 = week 2190, 270 026 s)
    y, mo, d, hh, mi, s = gpsiq.gps_to_date(WEEK, sec)
    out = str(tmp_path / "t.bin")
    r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", "%d/%d/%d,%d:%d:%g" % (y, mo, d, hh, mi, s), "-", str(tmp_path / "xyz.bin"),
                        str(nblocks), str(nchan), repr(fs), "1", out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert np.array_equal(np.fromfile(out, dtype=np.int8), want)
    r = subprocess.run([os.path.join(host, "gpsiq_runahead"), path, "2", str(WEEK), repr(sec), str(tmp_path / "none.csv"), "2", "8",
                        repr(fs), "1", str(tmp_path / "o.bin")], capture_output=True, text=True, timeout=60)
    assert r.returncode == 1 and "motion file" in r.stderr


def late_scenario(tmp_path, nblocks, sec=270000.0, seed=11):
    """Nine satellites that stay up, one that sets between the refreshes at 450 s and 480 s and one that rises
    between those at 510 s and 540 s: with ten channels the riser can only take the slot the setter gave up,
    and it takes it when that slot's page counter stands at 17, i.e. at the iono/UTC page 18 of subframe 4."""
    from gpsiq.scenario import _elevation_deg, synth_constellation
    high = synth_constellation(9, TOKYO, sec, seed=seed, min_elev_deg=20.0)
    cand = synth_constellation(1500, TOKYO, sec, seed=seed + 100, min_elev_deg=-4.0, max_elev_deg=4.0)
    def up(e, t):                                                 # what allocateChannel() will see at the refresh at sec + t
        return gpsiq.sat_visibility(e, WEEK, sec + t, TOKYO)[0]
    near = [e for e in cand if abs(_elevation_deg(e, sec + 500.0, TOKYO)) < 1.0]
    setter = [e for e in near if up(e, 0) and up(e, 450) and not up(e, 480) and not up(e, 600)]
    riser = [e for e in near if not up(e, 0) and not up(e, 510) and up(e, 540) and up(e, 600)]
    assert setter and riser
    orbits = np.concatenate([high, np.array(setter[:1]), np.array(riser[:1])])
    recs = synth_rinex_records(len(orbits), TOKYO, WEEK, sec, seed=seed, sets=2, eph=orbits)
    path = write_rinex_nav(str(tmp_path / "late.21n"), recs, UTC, 2)
    eph, utc, n = gpsiq.rinex_read(path, 2)
    ieph = gpsiq.rinex_select(eph, n, WEEK, sec)
    xyz = np.repeat(TOKYO[None, :], nblocks + 1, axis=0)
    return path, eph[:n], ieph, utc, xyz, sec


def test_subframe_page_counter_survives_reallocation(ref, tmp_path):
    """chan[i].ipage is only ever advanced by generateNavMsg (gps.c:2137-2139): a slot that is released after
    16 refreshes and re-allocated later continues with the subframe 4/5 page it had reached -- here the
    iono/UTC page 18 -- it does not start over at page 1.  6 000 blocks = 20 refreshes, ten channels, all in use."""
    from gpsiq.pipeline import RunAheadAllocating
    nblocks, nchan = 6000, 10
    _, eph, ieph, utc, xyz, sec = late_scenario(tmp_path, nblocks)
    assert utc["vflg"] == 1
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    desc = ra.descriptors(xyz[1:])
    want, nsat, _ = ref.run_host(eph, ieph, utc, WEEK, sec, xyz, nchan)
    prn = desc["prn"]
    reused = [(b, c) for b in range(1, nblocks) for c in range(nchan) if prn[b, c] > 0 and prn[b - 1, c] == 0]
    released = [(b, c) for b in range(1, nblocks) for c in range(nchan) if prn[b, c] == 0 and prn[b - 1, c] > 0]
    assert released == [(4800, reused[0][1])] and len(reused) == 1 and reused[0][0] > 4800, (released, reused)   # gone at the 16th refresh
    assert list(nsat) == ra.nsat
    for f in ("prn", "iword", "ibit", "icode", "f_carr", "f_code", "code_phase", "gain", "dwrd", "carr_phase"):
        assert desc[f].tobytes() == want[f].tobytes(), f
    # and the words really depend on where the counter stands: a fresh counter gives other words
    b, c = reused[0]
    fresh = RunAheadAllocating(eph, utc, nchan, WEEK, gps_time_after(sec, b), xyz[0], ieph=ieph)
    slot = [i for i in range(nchan) if fresh.trk[i]["prn"] == prn[b, c]]
    assert slot and fresh.trk[slot[0]]["dwrd"].tobytes() != desc["dwrd"][b, c].tobytes()


def test_split_batches_continue_the_carrier(tmp_path):
    """RunAheadAllocating.descriptors(carr_phase=...): a scenario rendered in several calls hands the phase
    the previous generate_batch returned to the slots that kept their satellite, the allocation's phase to
    the slots (re-)allocated at the refresh in between; everything else equals the single call."""
    from gpsiq.pipeline import RunAheadAllocating
    nblocks, nchan = 900, 16
    _, eph, ieph, utc, xyz, sec = horizon_scenario(tmp_path, nblocks, seed=6)
    whole = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph).descriptors(xyz[1:])
    ra = RunAheadAllocating(eph, utc, nchan, WEEK, sec, xyz[0], ieph=ieph)
    cut = 300                                                   # right at a refresh: slots change hands here
    assert (whole["prn"][cut] != whole["prn"][cut - 1]).any()
    a = ra.descriptors(xyz[1:1 + cut])
    handed = np.arange(nchan) * 0.01 + 0.005                    # stands for generate_batch's carr_out
    b = ra.descriptors(xyz[1 + cut:], carr_phase=handed)
    assert a.tobytes() == whole[:cut].tobytes()
    kept = (whole["prn"][cut] == whole["prn"][cut - 1]) & (whole["prn"][cut] > 0)
    assert kept.any() and not kept.all()
    assert np.array_equal(b["carr_phase"][0][kept], handed[kept])
    assert np.array_equal(b["carr_phase"][0][~kept], whole["carr_phase"][cut][~kept])
    b["carr_phase"][0] = whole["carr_phase"][cut]
    assert b.tobytes() == whole[cut:].tobytes()


@pytest.mark.parametrize("moving", [False, True])
def test_fused_refresh_and_quantise_equals_the_two_calls(tmp_path, moving):
    """gpsiq_refresh_epochs_quantized == gpsiq_refresh_epochs followed by gpsiq_quantize_batch, byte for byte, over
    several navigation-message epochs, from the start and after a seek, with a carrier handed in, Pluto gain, a
    slot left unused -- and it reports what the quantiser reports."""
    sec, nblocks, fs, ns = 270012.0, 900, 2.6e6, 260000
    _, eph, utc, ieph, svs, xyz = scenario(tmp_path, sec, nblocks, moving)
    for b0, n, carr, x2 in ((0, 900, None, False), (123, 500, np.linspace(0.01, 0.95, len(svs)), True), (600, 1, None, False)):
        a, b = (RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]) for _ in range(2))
        a.seek(b0, xyz[b0]); b.seek(b0, xyz[b0])
        want, _ = gpsiq.quantize_blocks(a.descriptors(xyz[1 + b0:1 + b0 + n], carr_phase=carr, gain_x2=x2), fs, ns)
        got = b.descriptors_quantized(xyz[1 + b0:1 + b0 + n], fs, ns, carr_phase=carr, gain_x2=x2)
        assert got.tobytes() == want.tobytes(), (b0, n)
        assert a.trk.tobytes() == b.trk.tobytes() and a.blocks_done == b.blocks_done
    ra = RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0])
    ra.trk["prn"][2] = 0                                           # an unused slot stays an all-zero descriptor
    q = ra.descriptors_quantized(xyz[1:41], fs, ns)
    assert not q[:, 2].tobytes().strip(b"\0") and (q["prn"][:, 3] > 0).all()
    with pytest.raises(gpsiq.GpsiqError):                          # 0.3 Msps: more than two chips per sample
        RunAhead(eph[ieph], utc, svs, WEEK, sec, xyz[0]).descriptors_quantized(xyz[1:5], 0.3e6, 30000)
