"""The C host side (multi-sdr-gps-sim_amd/host): the fifo.h-compatible FIFO under stress
(and under ThreadSanitizer when the toolchain has it), and — on the GPU box — the
gpsiq_play program, which is the reference's generator-thread / fifo / sink-thread structure
with libgpsiq in place of the sample loop."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest

from gpsiq.abi import CHAN_DTYPE, SC08, SC16, SINK_HACKRF, SINK_IQFILE
from gpsiq.scenario import synth_blocks

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HOST = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "host")


@pytest.fixture(scope="module")
def host_built():
    subprocess.run(["make", "-s", "-C", HOST], check=True)
    return HOST


def run_sanitized(cmd, **kw):
    """Run a sanitizer build.  The sanitizer runtimes reserve fixed address ranges and on kernels with wide address
    space randomisation now and then die at start-up (killed by a signal, or 'unexpected memory mapping', before main
    runs): such a run is repeated with randomisation off (setarch -R) and, if the runtime still cannot start, skipped.
    A run that reaches main and fails, or reports anything, is returned as it is."""
    r = subprocess.run(cmd, capture_output=True, text=True, **kw)
    startup = ("FATAL: ThreadSanitizer", "unexpected memory mapping", "Shadow memory range interleaves", "ReserveShadowMemoryRange failed")
    died_early = lambda q: (q.returncode < 0 and not q.stdout and "Sanitizer:" not in q.stderr.replace("FATAL: ThreadSanitizer", "")) \
        or any(m in q.stderr for m in startup)
    if died_early(r) and shutil.which("setarch"):
        r = subprocess.run(["setarch", os.uname().machine, "-R", *cmd], capture_output=True, text=True, **kw)
    if died_early(r):
        pytest.skip("the sanitizer runtime cannot start in this container: rc %d %s" % (r.returncode, r.stderr.strip()[-160:]))
    return r


def test_fifo_is_drop_free_and_ordered(host_built):
    r = subprocess.run([os.path.join(host_built, "fifo_selftest"), "20000"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip() == "ok", r.stdout + r.stderr


def test_fifo_under_thread_sanitizer(tmp_path):
    exe = str(tmp_path / "fifo_tsan")
    b = subprocess.run(["gcc", "-O1", "-g", "-std=c11", "-D_GNU_SOURCE", "-pthread", "-fsanitize=thread", "-o", exe,
                        os.path.join(HOST, "fifo_selftest.c"), os.path.join(HOST, "fifo.c")], capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("no ThreadSanitizer runtime here: " + b.stderr[-200:])
    r = run_sanitized([exe, "3000"], timeout=600)
    assert r.returncode == 0 and "WARNING: ThreadSanitizer" not in r.stderr, "rc %d\n%s" % (r.returncode, (r.stdout + r.stderr)[-2000:])


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_host_half_under_sanitizers(tmp_path, san):
    """The host-only sources of the library (quantiser + worker pool, threaded refresh, nav words,
    RINEX readers incl. every truncation of a file, fifo hand-off) under ASan + UBSan, and under
    ThreadSanitizer for the worker pool."""
    from gpsiq.scenario import llh_to_ecef, synth_rinex_records, write_rinex_nav
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "sanitize_host")
    flags = ["-O1", "-g", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"), "-I" + csrc]
    if "undefined" in san:
        flags.append("-fno-sanitize-recover=undefined")
    objs = []
    for src in ("gpsiq_host.cpp", "gpsiq_exact.cpp", "gpsiq_chain.cpp", "gpsiq_refresh.cpp", "gpsiq_nav.cpp", "gpsiq_rinex.cpp"):
        o = str(tmp_path / (src + ".o"))
        b = subprocess.run(["g++", "-std=c++17", *flags, "-c", os.path.join(csrc, src), "-o", o], capture_output=True, text=True)
        if b.returncode != 0:
            pytest.skip("no sanitizer toolchain here: " + b.stderr[-200:])
        objs.append(o)
    o = str(tmp_path / "drv.o")
    subprocess.run(["gcc", "-std=c11", *flags, "-c", os.path.join(root, "tests", "sanitize_host.c"), "-o", o], check=True)
    b = subprocess.run(["g++", "-fsanitize=" + san, "-o", exe, o, *objs, "-lpthread", "-lz", "-lm"], capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("no sanitizer runtime here: " + b.stderr[-200:])
    pos = llh_to_ecef(35.681298, 139.766247, 10.0)
    utc = dict(alpha=[0.1118e-07, -0.7451e-08, -0.5961e-07, 0.1192e-06], beta=[0.1167e+06, -0.2294e+06, -0.1311e+06, 0.1049e+07],
               A0=-0.931322574615e-09, A1=-0.355271367880e-14, tot=233472, wnt=2190, dtls=18)
    recs = synth_rinex_records(10, pos, 2190, 270000.0, seed=13, sets=2)
    for version in (2, 3):
        path = write_rinex_nav(str(tmp_path / f"in.v{version}"), recs, utc, version)
        r = run_sanitized([exe, path, str(tmp_path / "cut"), str(version)], timeout=600, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0"))
        assert r.returncode == 0 and r.stdout.strip() == "ok" and "WARNING: ThreadSanitizer" not in r.stderr, (r.stdout + r.stderr)[-3000:]


@pytest.mark.parametrize("san", ["address,undefined", "thread"])
def test_reference_walker_consumed_in_pieces_under_sanitizers(tmp_path, san):
    """The one-pass walker of GPSIQ_NCO_REFERENCE (one host thread per channel through the whole timeline) with a consumer
    that takes every piece as soon as all channels are through it -- what the device side does while it renders -- under
    ThreadSanitizer and ASan + UBSan: the pieces add up to the single call's descriptors, patches and end state."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "sanitize_refwalk")
    flags = ["-O1", "-g", "-std=c++17", "-ffp-contract=off", "-fsanitize=" + san, "-fno-omit-frame-pointer", "-I" + os.path.join(root, "include"), "-I" + csrc]
    if "undefined" in san:
        flags.append("-fno-sanitize-recover=undefined")
    b = subprocess.run(["g++", *flags, "-o", exe, os.path.join(root, "tests", "sanitize_refwalk.cpp"), os.path.join(csrc, "gpsiq_host.cpp"),
                        os.path.join(csrc, "gpsiq_exact.cpp"), os.path.join(csrc, "gpsiq_chain.cpp"), "-lpthread", "-lm"], capture_output=True, text=True)
    if b.returncode != 0:
        pytest.skip("no sanitizer toolchain / runtime here: " + b.stderr[-300:])
    for threads in (None, "2"):           # the default pool; fewer threads than channels (piece-major)
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")
        if threads:
            env["GPSIQ_THREADS"] = threads
        r = run_sanitized([exe], timeout=900, env=env)
        assert r.returncode == 0 and r.stdout.strip() == "ok" and "WARNING: ThreadSanitizer" not in r.stderr, (threads, (r.stdout + r.stderr)[-3000:])


def test_drift_enclosure_holds_the_walked_accumulator(tmp_path):
    """The decision GPSIQ_NCO_REFERENCE takes WITHOUT walking an accumulator (Drift in csrc/gpsiq_exact.cpp: an enclosure of
    the reference's double phase at sample n from the block's start state alone) against the accumulator walked exactly
    (Nco::advance, itself pinned to the plain loop of gps.c:2789-2792 / 2821-2826 by test_reference_nco_host.py): 400 000
    random and adversarial cases per seed -- carrier both signs and code, four sample rates, addends with trailing zero
    mantissas (exact-tie binades), tiny and binade-edge start states -- every one inside the enclosure, no more than 0.7 of
    its half-width used, and the enclosure two orders of magnitude narrower than the a-priori window."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "drift_enclosure")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "include"), "-I" + csrc, "-o", exe,
                    os.path.join(root, "tests", "drift_enclosure.cpp"), os.path.join(csrc, "gpsiq_host.cpp"), os.path.join(csrc, "gpsiq_chain.cpp"), "-lpthread", "-lm"], check=True)
    for seed in ("1",):
        r = subprocess.run([exe, seed], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout + r.stderr
        f = dict(kv.split("=") for kv in r.stdout.split())
        assert int(f["bad"]) == 0 and int(f["checked"]) > 300000
        assert float(f["max_use"]) < 0.7 and float(f["mean_width"]) < 0.01 and float(f["worst_width"]) < 0.1, r.stdout


def test_eight_cycles_at_once_equal_the_scalar_walk(tmp_path):
    """The wrap-to-wrap table of a block is built eight carrier cycles at a time where the host has AVX-512 (NcoWalk::walk8_up /
    walk8_down): lane for lane the vector walk gives the scalar walk's end state, sample count and validity range, or says
    not-ok (470 000 lanes per seed: random and exact-tie addends, both signs, four sample rates, edge and round start states)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    exe = str(tmp_path / "batch_walk")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-I" + os.path.join(root, "include"), "-I" + csrc, "-o", exe,
                    os.path.join(root, "tests", "batch_walk.cpp"), os.path.join(csrc, "gpsiq_host.cpp"), os.path.join(csrc, "gpsiq_chain.cpp"), "-lpthread", "-lm"], check=True)
    r = subprocess.run([exe, "3"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    f = dict(kv.split("=") for kv in r.stdout.split())
    if f.get("skipped"):
        pytest.skip("no AVX-512 on this host: the table is built by the scalar walk alone")
    assert int(f["bad"]) == 0 and int(f["lanes_ok"]) > 400000 and int(f["lanes_ok"]) <= int(f["scalar_ok"])


def test_fifo_header_matches_reference_api():
    """Same nine entry points and the same struct fields as the reference's fifo.h:19-63."""
    txt = open(os.path.join(HOST, "fifo.h")).read()
    for fn in ("fifo_create", "fifo_destroy", "fifo_wait_next", "fifo_wait_full", "fifo_halt", "fifo_acquire",
               "fifo_enqueue", "fifo_dequeue", "fifo_release"):
        assert fn + "(" in txt
    for field in ("signed char  *data8", "signed short *data16", "unsigned int  totalLength", "unsigned int  validLength",
                  "struct iq_buf *next"):
        assert field in txt
    ref = "/root/reference/fifo.h"
    if os.path.exists(ref):      # compile a translation unit that includes BOTH declarations
        # identical prototypes may be repeated in C; conflicting ones are an error
        code = '#include "%s"\n' % os.path.join(HOST, "fifo.h") + "\n".join(
            l for l in open(ref).read().splitlines() if l.strip().startswith(("bool fifo_", "void fifo_", "struct iq_buf *fifo_"))
            and "()" not in l) + "\nint main(void){return 0;}\n"
        r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-x", "c", "-"], input=code, capture_output=True, text=True)
        assert r.returncode == 0, r.stderr


def write_descriptors(path, desc, fs, nsamp, ss):
    nb, nc = desc.shape
    with open(path, "wb") as f:
        f.write(struct.pack("<8sIIIId", b"GPSIQD1\0", nb, nc, ss, nsamp, fs))
        f.write(np.ascontiguousarray(desc, dtype=CHAN_DTYPE).tobytes())


@pytest.mark.gpu
@pytest.mark.parametrize("sink,name,ss", [(SINK_IQFILE, "iqfile", SC08), (SINK_HACKRF, "hackrf", SC08), (SINK_IQFILE, "pluto", SC16)])
def test_gpsiq_play_end_to_end(host_built, oracle, tmp_path, sink, name, ss):
    """C generator thread -> pinned fifo -> C sink thread -> file == the oracle's blocks in
    the reference's enqueue order (HackRF: 262144-element chunks, trailing partial chunk kept
    back exactly as gps.c:2847-2856 does)."""
    fs, ns, nb, nc = 2.6e6, 260000, 5, 12
    d = synth_blocks(nb, nc, seed=91)
    d["prn"][3:, 2] = 0
    dpath, opath = str(tmp_path / "desc.bin"), str(tmp_path / "out.bin")
    write_descriptors(dpath, d, fs, ns, ss)
    r = subprocess.run([os.path.join(host_built, "gpsiq_play"), dpath, opath, name], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    got = np.fromfile(opath, dtype=np.int8 if ss == SC08 else np.int16)
    q = oracle.quantize_blocks(d, fs, ns)
    want = np.concatenate([oracle.block_fixed(q[b], ns, ss, seq=True) for b in range(nb)])
    plan = oracle.chunk_plan(sink, 2 * ns, nb)
    assert len(got) == plan.sum()
    assert np.array_equal(got, want[: plan.sum()])


@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3])
def test_gpsiq_shard_parts_concatenate_to_the_single_run(host_built, oracle, tmp_path, world):
    """C host, time-sharded: every rank quantises the whole timeline, renders its own block
    range (gpsiq_shard_range + gpsiq_generate_quantized) and writes its part; the parts
    concatenate to the stream the one-block-at-a-time program (gpsiq_play, iqfile sink) writes,
    which is the oracle's.  All ranks share device 0 here; on a node each has its own GPU."""
    fs, ns, nb, nc, ss = 2.6e6, 260000, 10, 12, SC16
    d = synth_blocks(nb, nc, seed=17)
    d["prn"][4:, 5] = 0
    d["prn"][7:, 5] = 29           # slot re-allocated inside the last shard
    dpath = str(tmp_path / "desc.bin")
    write_descriptors(dpath, d, fs, ns, ss)
    parts = []
    for r in range(world):
        out = str(tmp_path / f"part{r}.bin")
        p = subprocess.run([os.path.join(host_built, "gpsiq_shard"), dpath, out, str(r), str(world), "0"],
                           capture_output=True, text=True, timeout=300)
        assert p.returncode == 0, p.stdout + p.stderr
        parts.append(np.fromfile(out, dtype=np.int16))
    got = np.concatenate(parts)
    q = oracle.quantize_blocks(d, fs, ns)
    want = np.concatenate([oracle.block_fixed(q[b], ns, ss, seq=True) for b in range(nb)])
    assert np.array_equal(got, want)
    single = str(tmp_path / "single.bin")
    p = subprocess.run([os.path.join(host_built, "gpsiq_play"), dpath, single, "iqfile"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    assert np.array_equal(np.fromfile(single, dtype=np.int16), got)


@pytest.mark.gpu
def test_gpsiq_shard_in_the_reference_nco_model(host_built, oracle, tmp_path):
    """C host, time-sharded, GPSIQ_NCO_REFERENCE: every rank walks the carrier chain (gpsiq_reference_chain) and renders ONLY its
    own block range from the start states (gpsiq_generate_seeded); the three parts concatenate to the float loop's stream."""
    fs, ns, nb, nc, ss = 10.0e6, 200000, 13, 9, SC16
    d = synth_blocks(nb, nc, seed=29)
    rng = np.random.default_rng(3)
    d["carr_phase"][:] = (rng.integers(0, 512, (nb, nc)) + 1e-12) / 512.0          # candidates and patches at the block starts
    d["code_phase"][:] = rng.integers(0, 1023, (nb, nc)) + 1e-9
    d["prn"][4:, 5] = 0
    d["prn"][9:, 5] = 29
    dpath = str(tmp_path / "desc.bin")
    write_descriptors(dpath, d, fs, ns, ss)
    parts = []
    for r in range(3):
        out = str(tmp_path / f"part{r}.bin")
        p = subprocess.run([os.path.join(host_built, "gpsiq_shard"), dpath, out, str(r), "3", "0", "reference"], capture_output=True, text=True, timeout=300)
        assert p.returncode == 0 and "GPSIQ_NCO_REFERENCE" in p.stdout, p.stdout + p.stderr
        parts.append(np.fromfile(out, dtype=np.int16))
    got = np.concatenate(parts).reshape(nb, 2 * ns)
    carr, prev = None, None
    for b in range(nb):
        db = d[b].copy()
        if b:
            db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
        want, carr = oracle.block_float(db, ns, fs, ss)
        prev = db["prn"].copy()
        assert np.array_equal(got[b], want), b


@pytest.mark.gpu
@pytest.mark.parametrize("contexts,mode", [(0, "fixed"), (3, "fixed"), (2, "reference")])
def test_gpsiq_render_one_process_all_devices(host_built, oracle, tmp_path, contexts, mode):
    """C host, one process: gpsiq_generate_batch_multi over `contexts` contexts (0 = one per visible GPU), the
    timeline in slices of 256 blocks chained through carr_phase == the one-block-at-a-time program's stream; in
    GPSIQ_NCO_REFERENCE == the float loop."""
    fs, ns, nb, nc, ss = 2.6e6, 26000, 600, 7, SC08
    d = synth_blocks(nb, nc, seed=23)
    d["prn"][300:, 2] = 0
    d["prn"][450:, 2] = 31
    dpath, out = str(tmp_path / "desc.bin"), str(tmp_path / "render.bin")
    write_descriptors(dpath, d, fs, ns, ss)
    args = [os.path.join(host_built, "gpsiq_render"), dpath, out, str(contexts)] + (["reference"] if mode == "reference" else [])
    p = subprocess.run(args, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    got = np.fromfile(out, dtype=np.int8).reshape(nb, 2 * ns)
    if mode == "fixed":
        q = oracle.quantize_blocks(d, fs, ns)
        for b in (0, 1, 255, 256, 257, 299, 300, 449, 450, 511, 512, 599):
            assert np.array_equal(got[b], oracle.block_fixed(q[b], ns, ss, seq=True)), b
    else:
        carr, prev = None, None
        for b in range(nb):
            db = d[b].copy()
            if b:
                db["carr_phase"] = np.where((prev == db["prn"]) & (db["prn"] > 0), carr, db["carr_phase"])
            want, carr = oracle.block_float(db, ns, fs, ss)
            prev = db["prn"].copy()
            assert np.array_equal(got[b], want), b


def test_c_hosts_fail_loudly_without_a_gpu(host_built, tmp_path):
    """No CPU fallback anywhere: on a box without a GPU both C programs stop at gpsiq_create."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    d = synth_blocks(2, 4, seed=1)
    dpath = str(tmp_path / "desc.bin")
    write_descriptors(dpath, d, 2.6e6, 260000, SC08)
    from gpsiq.scenario import llh_to_ecef, synth_rinex_records, write_rinex_nav
    pos = llh_to_ecef(35.681298, 139.766247, 10.0)
    utc = dict(alpha=[0.0] * 4, beta=[0.0] * 4, A0=0.0, A1=0.0, tot=233472, wnt=2190, dtls=18)
    rinex = write_rinex_nav(str(tmp_path / "r.21n"), synth_rinex_records(6, pos, 2190, 270000.0, seed=3, sets=2), utc, 2)
    np.repeat(pos[None, :], 3, axis=0).tofile(str(tmp_path / "xyz.bin"))
    for argv in (["gpsiq_play", dpath, str(tmp_path / "o.bin")], ["gpsiq_shard", dpath, str(tmp_path / "p.bin"), "0", "1"],
                 ["gpsiq_render", dpath, str(tmp_path / "r.bin")],
                 ["gpsiq_runahead", rinex, "2", "2190", "270000", str(tmp_path / "xyz.bin"), "2", "8", "2600000", "1", str(tmp_path / "q.bin")]):
        p = subprocess.run([os.path.join(host_built, argv[0])] + argv[1:], capture_output=True, text=True, timeout=120)
        assert p.returncode != 0
        assert "gpsiq" in p.stderr and "device" in p.stderr.lower(), p.stderr
