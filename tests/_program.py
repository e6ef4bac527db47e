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


def run_program_lossy(binary, workdir, seconds=30, iq16=False, timeout=300, fs=FS, rinex=RINEX):
    """-> bytes of iqdata.bin of a program whose FIFO may lose blocks (the reference's own fifo.c, fifo.c:166-168): the
    file's final size is not known in advance, so the run is over when the file has stopped growing for three seconds."""
    args = [binary, "-e", rinex, "-l", LLH, "-r", "iqfile", "-d", str(seconds), "--disable-almanac"] + (["--iq16"] if iq16 else [])
    env = dict(os.environ, LINES="50", COLUMNS="160", TERM="xterm")
    out = os.path.join(workdir, "iqdata.bin")
    if os.path.exists(out):
        os.remove(out)
    p = subprocess.Popen(args, cwd=workdir, env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        t0, last, stable = time.time(), -1, 0
        while time.time() - t0 < timeout and stable < 6:
            time.sleep(0.5)
            if p.poll() is not None:
                raise RuntimeError(f"{binary} exited with {p.returncode} before the run was complete")
            size = os.path.getsize(out) if os.path.exists(out) else 0
            stable = stable + 1 if (size == last and size > 0) else 0
            last = size
    finally:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return open(out, "rb").read()


# ---- BASELINE config 4: the reference's own circle.csv -------------------------------------------------------
CONFIG4 = os.path.join(ROOT, "tests", "golden", "program_config4_circle.npz")


def write_motion_csv(path, xyz_mm):
    """A user-motion file in the very format of the reference's circle.csv ('%5.1f,%.3f, %.3f, %.3f' per 0.1 s) from
    positions in whole millimetres (the file holds three decimals): rows 0..2999 of the fixture written this way are
    byte-identical to /root/reference/circle.csv (checked where the reference is present), so readUserMotion()
    (gps.c:2253-2277) parses the same doubles."""
    def dec(mm):
        mm = int(mm)
        return "%s%d.%03d" % ("-" if mm < 0 else "", abs(mm) // 1000, abs(mm) % 1000)
    with open(path, "w") as f:
        for k, (x, y, z) in enumerate(xyz_mm):
            f.write("%5.1f,%s, %s, %s\n" % (k / 10.0, dec(x), dec(y), dec(z)))
    return path


def stream_blocks(args, workdir, out_name, blk_bytes, nblocks, env_extra=None, timeout=900, idles=False, on_block=None):
    """Run a program that writes nblocks blocks of blk_bytes to `out_name` in workdir and hand every block to
    on_block(index, bytes) as it arrives.  out_name is a named pipe: a 600 s run at 2.6 Msps int16 is 6.2 GB, which
    never has to exist on a disk.  idles: the program does not exit by itself (the reference program sits in its key
    loop after the run and keeps the last few KB in its stdio buffer): it gets SIGTERM once everything but that tail
    has arrived, which makes it close the file (gps-sim.c:216-247).  Returns the number of blocks seen."""
    import select
    fifo = os.path.join(workdir, out_name)
    if os.path.exists(fifo):
        os.remove(fifo)
    os.mkfifo(fifo)
    fd = os.open(fifo, os.O_RDONLY | os.O_NONBLOCK)               # the reader first: the program's fopen() then never blocks
    try:
        import fcntl
        fcntl.fcntl(fd, 1031, 1 << 20)                            # F_SETPIPE_SZ: 1 MiB instead of 64 KiB, fewer wake-ups per 10 MB block
    except OSError:
        pass
    env = dict(os.environ, LINES="50", COLUMNS="160", TERM="xterm")
    env.update(env_extra or {})
    p = subprocess.Popen(args, cwd=workdir, env=env, stdin=subprocess.DEVNULL, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    expect = blk_bytes * nblocks
    st = {"got": 0, "seen": 0}
    pend = bytearray()

    def take(chunk):
        st["got"] += len(chunk)
        pend.extend(chunk)
        while len(pend) >= blk_bytes:
            if on_block:
                on_block(st["seen"], bytes(pend[:blk_bytes]))
            st["seen"] += 1
            del pend[:blk_bytes]

    def read_some():
        """bytes, b"" (no writer has the pipe open) or None (a writer has, nothing to read yet)"""
        try:
            return os.read(fd, 1 << 22)
        except BlockingIOError:
            return None

    t0 = last_data = time.time()
    termed = False
    try:
        while True:
            if time.time() - t0 > timeout:
                raise RuntimeError(f"{args[0]}: {st['got']} of {expect} bytes after {timeout} s")
            select.select([fd], [], [], 0.25)
            chunk = read_some()
            if chunk:
                take(chunk)
                last_data = time.time()
                continue
            if p.poll() is not None:                                # the program is gone: whatever is left is in the pipe
                while True:
                    chunk = read_some()
                    if not chunk:
                        break
                    take(chunk)
                if st["got"] == 0:
                    raise RuntimeError(f"{args[0]} exited with {p.returncode} without writing")
                break
            if chunk == b"":
                time.sleep(0.02)                                    # not opened yet, or closed and about to exit
            if idles and not termed and st["got"] >= expect - (1 << 16) and time.time() - last_data > 0.75:
                p.send_signal(signal.SIGTERM)                       # main loop: signal_handler -> cleanup_and_exit
                termed = True
    finally:
        if p.poll() is None:
            p.send_signal(signal.SIGTERM)
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        os.close(fd)
        os.remove(fifo)
    got, seen = st["got"], st["seen"]
    assert got == expect and not pend, (got, expect, len(pend))
    return seen


def program_block_digests(binary, workdir, motion, seconds, nblocks, iq16=True, fs=2600000, rinex=RINEX16, env_extra=None,
                          keep=(), timeout=900):
    """The reference program (patched or not) on a user-motion file (motion=None: at the static BASELINE position): SHA-256
    of every block of its iqdata.bin, and the first 4096 elements of the blocks listed in `keep`."""
    import hashlib
    import numpy as np
    sha, heads = [], {}
    ss = 2 if iq16 else 1

    def on_block(i, b):
        sha.append(hashlib.sha256(b).hexdigest())
        if i in keep:
            heads[i] = np.frombuffer(b[:4096 * ss], dtype=np.int16 if iq16 else np.int8).copy()
    where = ["-m", motion] if motion else ["-l", LLH]             # a user-motion file, or the static BASELINE position
    args = [binary, "-e", rinex] + where + ["-r", "iqfile", "-d", str(seconds), "--disable-almanac"] + (["--iq16"] if iq16 else [])
    n = stream_blocks(args, workdir, "iqdata.bin", (fs // 10) * 2 * ss, nblocks, env_extra, timeout, idles=True, on_block=on_block)
    assert n == nblocks
    return sha, heads
