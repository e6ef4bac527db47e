"""Run the reference PROGRAM headless (oracle/_ref/gps-sim-ref: the reference's own sources; or
oracle/_ref/gps-sim-gpsiq: the same with its sample loop replaced by gpsiq_generate_block, see
oracle/Makefile) and collect the iqdata.bin it writes.  TEST INFRASTRUCTURE."""
import os
import signal
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFDIR = os.path.join(ROOT, "oracle", "_ref")
RINEX = os.path.join(ROOT, "tests", "golden", "synth_static.21n")
RINEX16 = os.path.join(ROOT, "tests", "golden", "synth_static16.21n")      # 16 satellites in view: for the MAX_CHAN 16 builds
LLH = "35.681298,139.766247,10.0"          # BASELINE config 1: static position
FS = 3000000                                # the reference's TX_SAMPLERATE as shipped (sdr.h:21)


def program(name):
    p = os.path.join(REFDIR, name)
    return p if os.path.exists(p) else None


def write_circle_motion(path, seconds=30):
    """A user-motion file in the reference's format (readUserMotion, gps.c:2253-2277: one line 't,x,y,z' per 0.1 s,
    ECEF metres): a 150 m circle around the static BASELINE position, 50 s per lap -- BASELINE config 4's kind of
    scenario (per-block range/Doppler refresh on the host), written by this repository's own track generator."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "multi-sdr-gps-sim_amd"))
    from gpsiq.scenario import circle_track, llh_to_ecef
    lat, lon, h = (float(v) for v in LLH.split(","))
    xyz = circle_track(llh_to_ecef(lat, lon, h), seconds * 10, radius_m=150.0, period_s=50.0)
    with open(path, "w") as f:
        for k, (x, y, z) in enumerate(xyz):
            f.write("%5.1f,%.3f,%.3f,%.3f\n" % (0.1 * k, x, y, z))
    return path


def run_program(binary, workdir, seconds=30, iq16=False, env_extra=None, timeout=300, motion=None, fs=FS, rinex=RINEX):
    """-> bytes of iqdata.bin.  The program has no batch mode: it draws its ncurses screen (LINES/COLUMNS
    given so that it does not ask about the window size), generates `seconds` of signal through the fifo
    into iqdata.bin in the working directory and then idles in its key loop until it is told to stop."""
    nblocks = seconds * 10 - 1                                   # the block loop starts at 1 (gps.c:2703)
    expect = nblocks * (fs // 10) * 2 * (2 if iq16 else 1)
    where = ["-m", motion] if motion else ["-l", LLH]
    args = [binary, "-e", rinex] + where + ["-r", "iqfile", "-d", str(seconds), "--disable-almanac"]
    if iq16:
        args.append("--iq16")
    env = dict(os.environ, LINES="50", COLUMNS="160", TERM="xterm")
    env.update(env_extra or {})
    out = os.path.join(workdir, "iqdata.bin")
    if os.path.exists(out):
        os.remove(out)
    p = subprocess.Popen(args, cwd=workdir, env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL,
                         stderr=subprocess.DEVNULL)
    try:
        t0, last, stable = time.time(), -1, 0
        while time.time() - t0 < timeout:
            time.sleep(0.25)
            if p.poll() is not None:
                raise RuntimeError(f"{binary} exited with {p.returncode} before the run was complete")
            size = os.path.getsize(out) if os.path.exists(out) else 0
            stable = stable + 1 if size == last else 0
            last = size
            if size > expect - (1 << 20) and stable >= 3:          # all written but the stdio tail
                break
        else:
            raise RuntimeError(f"{binary}: {last} of {expect} bytes after {timeout} s")
    finally:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)                          # main loop: signal_handler -> cleanup_and_exit (gps-sim.c:216-247)
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    data = open(out, "rb").read()
    assert len(data) == expect, (len(data), expect)
    return data
