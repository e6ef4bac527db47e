"""The C-ABI shared library: loads on a box without a GPU, exports every entry point
include/gpsiq.h declares, struct layouts match, and there is no silent CPU fallback."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import gpsiq
from gpsiq.abi import CHAN_DTYPE, QCHAN_DTYPE

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", h) for h in ("gpsiq.h", "gpsiq_rows.h", "gpsiq_extras.h")]     # boundary, section 8f rows, frozen extras


def declared_functions(headers=HEADERS):
    txt = "".join(open(h).read() for h in headers)
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(gpsiq_[a-z0-9_]+)\s*\(", txt)))


def exported(path):
    out = subprocess.run(["nm", "-D", "--defined-only", path], check=True, capture_output=True, text=True).stdout
    return sorted(l.split()[2] for l in out.splitlines() if " T gpsiq_" in l)


def test_every_declared_symbol_is_exported():
    """libgpsiq.so exports the boundary (include/gpsiq.h) and its one plumbing entry, libgpsiq_rows.so the rows either side of the
    path (gpsiq_rows.h, gpsiq_extras.h) -- each exactly what its headers declare, and the boundary stays thin."""
    lib, rows = C.CDLL(gpsiq.LIB_PATH), C.CDLL(gpsiq.ROWS_PATH)
    names = declared_functions()
    assert len(names) >= 17, names
    # the boundary header stays the boundary: the drop-in calls + sharding, nothing of the host model
    boundary = re.sub(r"/\*.*?\*/", "", open(HEADERS[0]).read(), flags=re.S)
    assert len(open(HEADERS[0]).read().splitlines()) <= 450 and "gpsiq_rinex" not in boundary and "gpsiq_almanac" not in boundary
    core_names, rows_names = declared_functions(HEADERS[:1]), declared_functions(HEADERS[1:])
    for n in core_names:
        assert hasattr(lib, n), f"{n} declared in include/gpsiq.h but not exported by libgpsiq.so"
    for n in rows_names:
        assert hasattr(rows, n), f"{n} declared in include/gpsiq_rows.h / gpsiq_extras.h but not exported by libgpsiq_rows.so"
    assert exported(gpsiq.LIB_PATH) == sorted(core_names + ["gpsiq_plumbing"]), "libgpsiq.so exports something its header does not declare"
    assert exported(gpsiq.ROWS_PATH) == rows_names, "libgpsiq_rows.so exports something its headers do not declare"
    assert len(exported(gpsiq.LIB_PATH)) <= 30


def test_plumbing_resolves_every_name_its_header_declares():
    """csrc/gpsiq_plumbing.h: hidden symbols behind gpsiq_plumbing(name), plus the four internals the rows library runs on."""
    lib = C.CDLL(gpsiq.LIB_PATH)
    lib.gpsiq_plumbing.restype = C.c_void_p
    lib.gpsiq_plumbing.argtypes = [C.c_char_p]
    hdr = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc", "gpsiq_plumbing.h")
    names = [n for n in declared_functions([hdr]) if n != "gpsiq_plumbing" and not n.startswith("gpsiq_p_")]
    assert len(names) >= 20, names
    for n in names + ["set_error", "parallel_for", "quantize_one", "chain_carrier"]:
        assert lib.gpsiq_plumbing(n.encode()), f"gpsiq_plumbing has no entry {n}"
        assert not hasattr(lib, n), f"{n} is plumbing and exported as well"
    assert not lib.gpsiq_plumbing(b"no_such_entry")


def test_rows_library_shares_the_error_text_of_the_boundary_library():
    """One gpsiq_last_error() for both libraries: a rows call that fails leaves its text where the boundary library's accessor reads it."""
    from gpsiq.abi import EPHEM_DTYPE, IONO_DTYPE, TRACK_DTYPE
    with pytest.raises(gpsiq.GpsiqError) as e:
        gpsiq.refresh_batch(np.zeros(32, EPHEM_DTYPE), np.zeros(1, IONO_DTYPE), 2190, 0.0, np.zeros((2, 3)), np.zeros(17, TRACK_DTYPE))
    assert "bad nblocks 2 / nchan 17" in str(e.value), str(e.value)       # formatted in libgpsiq_rows.so, read back from libgpsiq.so


def test_header_compiles_as_c_and_struct_sizes_match(tmp_path):
    src = tmp_path / "t.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "gpsiq_extras.h"\n'
                   'int main(void){printf("%zu %zu %zu %zu %zu %zu\\n", sizeof(gpsiq_chan_t), sizeof(gpsiq_qchan_t),'
                   ' offsetof(gpsiq_chan_t, dwrd), offsetof(gpsiq_qchan_t, nav_bits), offsetof(gpsiq_qchan_t, prn),'
                   ' sizeof(gpsiq_iq_buf_t)); return 0;}\n')
    exe = tmp_path / "t"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)],
                   check=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.split()
    assert int(out[0]) == CHAN_DTYPE.itemsize == 296
    assert int(out[1]) == QCHAN_DTYPE.itemsize == 48
    assert int(out[2]) == CHAN_DTYPE.fields["dwrd"][1]
    assert int(out[3]) == QCHAN_DTYPE.fields["nav_bits"][1]
    assert int(out[4]) == QCHAN_DTYPE.fields["prn"][1]
    assert int(out[5]) == 32       # struct iq_buf on LP64: 2 pointers, 2 unsigned, 1 pointer (fifo.h:19-25)


def test_iq_buf_layout_is_the_references_field_for_field(tmp_path):
    """gpsiq_iq_buf_t is cast to and from struct iq_buf in the binding (the two fifo
    callbacks): every field must sit where the reference's header (fifo.h:19-25) puts it, and the fifo.h this
    repository ships for C hosts (host/fifo.h) must agree as well.  Compile-time assertions against the
    reference's own header where it is present."""
    fields = ["data8", "data16", "totalLength", "validLength", "next"]
    headers = [os.path.join(ROOT, "multi-sdr-gps-sim_amd", "host")]
    if os.path.exists("/root/reference/fifo.h"):
        headers.append("/root/reference")
    for k, inc in enumerate(headers):
        src = tmp_path / f"layout{k}.c"
        src.write_text('#include <stddef.h>\n#include "fifo.h"\n#include "gpsiq.h"\n' +
                       "".join(f'_Static_assert(offsetof(struct iq_buf, {f}) == offsetof(gpsiq_iq_buf_t, {f}), "{f} offset");\n'
                               f'_Static_assert(sizeof(((struct iq_buf *) 0)->{f}) == sizeof(((gpsiq_iq_buf_t *) 0)->{f}), "{f} size");\n'
                               for f in fields) +
                       '_Static_assert(sizeof(struct iq_buf) == sizeof(gpsiq_iq_buf_t), "size");\nint main(void) { return 0; }\n')
        subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-fsyntax-only", "-I", inc, "-I", os.path.join(ROOT, "include"), str(src)], check=True)


def test_library_is_hip_code_for_gfx950():
    """The product is the HIP library: it must carry a gfx950 code object and link the HIP
    runtime, and must not link anything from oracle/."""
    out = subprocess.run(["readelf", "-d", gpsiq.LIB_PATH], capture_output=True, text=True, check=True).stdout
    assert "libamdhip64" in out
    assert "oracle" not in out
    blob = open(gpsiq.LIB_PATH, "rb").read()
    assert b"gfx950" in blob
    for kern in (b"synth_tile", b"synth_generic"):
        assert kern in blob


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(gpsiq.GpsiqError) as e:
        gpsiq.Context(0)
    assert e.value.code == -3 and "no CPU path" in str(e.value)


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "multi-sdr-gps-sim_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", ".c")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "_oracle" not in txt and "liboracle" not in txt and "oracle/" not in txt.replace(
                    "anything under oracle/", "").replace("on anything under oracle", ""), os.path.join(dirpath, f)


def test_replayed_counters_are_bound_to_the_loaded_kernel(tmp_path, monkeypatch):
    """bench.py's `counters` / `roofline.traffic` are rocprofv3 counts committed under profiles/, divided by the live
    launch time: they may only be replayed next to the library they were taken from.  gpsiq_kernels_id() is the SHA-256
    of the device sources (every .hip file and the headers the kernels share with the host) the loaded library was built from; scripts/prof_summary.py stores the same id with every
    profile; a profile of any other kernel is reported as {"stale_profile": true} and its counts are withheld."""
    import hashlib
    import json
    import sys
    csrc = os.path.join(ROOT, "multi-sdr-gps-sim_amd", "csrc")
    devsrc = [l for l in open(os.path.join(csrc, "Makefile")) if l.startswith("DEVSRC")][0].split(":=")[1].split()     # every device source
    assert "gpsiq_kernels.hip" in devsrc and "gpsiq_chain_kernels.hip" in devsrc and "gpsiq_eval_kernels.hip" in devsrc and "gpsiq_lane.h" in devsrc
    blob = b"".join(open(os.path.join(csrc, f), "rb").read() for f in devsrc)
    assert gpsiq.kernels_id() == hashlib.sha256(blob).hexdigest()[:16], "libgpsiq.so is older than its device sources: rebuild"
    sys.path.insert(0, ROOT)
    import bench
    prof = tmp_path / "profiles"
    prof.mkdir()
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    good = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    key = sorted(good)[0]
    for kid, stale in ((gpsiq.kernels_id(), False), ("0123456789abcdef", True), (None, True)):
        c = dict(good[key])
        c.pop("kernels_id", None)
        if kid:
            c["kernels_id"] = kid
        json.dump({key: c}, open(prof / "pmc_counters.json", "w"))
        json.dump({key: 123, key + "_detail": {"kernels_id": kid} if kid else {}}, open(prof / "pmc_traffic.json", "w"))
        out = bench.counters_obj(key, 2.85)
        assert out["stale_profile"] is stale
        assert ("valu_issue_frac" in out) is (not stale)
        assert bench.replayed_traffic(key) == ((None, True) if stale else (123, False))
    assert bench.replayed_traffic("no_such_workload") == (None, False)
    # the committed profiles belong to the committed kernel
    for name in ("pmc_counters.json", "pmc_traffic.json"):
        d = json.load(open(os.path.join(ROOT, "profiles", name)))
        ids = {(v.get("kernels_id") if isinstance(v, dict) else None) for k, v in d.items() if isinstance(v, dict)}
        assert ids == {gpsiq.kernels_id()}, (name, ids)
