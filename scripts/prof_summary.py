#!/usr/bin/env python3
"""Summarise rocprofv3 (rocpd sqlite) outputs under gpurun_out/prof into profiles/<tag>_*.txt."""
import glob
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
os.makedirs("profiles", exist_ok=True)


def kernels_id():
    """What libgpsiq's gpsiq_kernels_id() returns for a library built from the tree these profiles were taken in: the
    first 16 hex digits of the SHA-256 over the device sources (DEVSRC of csrc/Makefile, in that order).  Stored next to every
    replayed counter so that bench.py can tell a profile of another kernel from one of the library it has loaded."""
    import hashlib
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    csrc = os.path.join(root, "multi-sdr-gps-sim_amd", "csrc")
    devsrc = [l for l in open(os.path.join(csrc, "Makefile")) if l.startswith("DEVSRC")][0].split(":=")[1].split()
    return hashlib.sha256(b"".join(open(os.path.join(csrc, f), "rb").read() for f in devsrc)).hexdigest()[:16]


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


lines = []
for db in sorted(glob.glob(os.path.join(src, "kt*", "*.db"))):
    lines.append(f"== rocprofv3 --kernel-trace --stats ({db}) ==")
    lines.append(f"{'kernel':<70} {'calls':>6} {'total_ns':>14} {'avg_ns':>14} {'pct':>7}")
    for name, calls, tot, avg, pct in q(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{name[:70]:<70} {calls:>6} {tot:>14.0f} {avg:>14.1f} {pct:>7.2f}")
    lines.append("per dispatch: kernel, grid, wg, vgpr, sgpr, lds, duration_ns (the first 40 and the last 10 dispatches)")
    rows = q(db, "select name,grid_x,workgroup_x,vgpr_count,sgpr_count,lds_size,duration from kernels order by start")
    for i, r in enumerate(rows):
        if i < 40 or i >= len(rows) - 10:
            lines.append("  " + " ".join(str(x)[:60] for x in r))
        elif i == 40:
            lines.append(f"  ... {len(rows) - 50} more dispatches ...")
    # the steady state of the dominant kernel: all dispatches of its most frequent grid shape after the first 20
    main = [r for r in rows if "synth_tile" in r[0]]
    if main:
        from collections import Counter
        shape = Counter((r[0], r[1]) for r in main).most_common(1)[0][0]
        d = [r[6] for r in main if (r[0], r[1]) == shape]
        steady = d[20:] if len(d) > 40 else d
        lines.append(f"dominant kernel {shape[0][:60]} grid {shape[1]}: {len(d)} dispatches, mean {sum(d) / len(d):.0f} ns; "
                     f"after the first 20: mean {sum(steady) / len(steady):.0f} ns, min {min(steady)} ns, max {max(steady)} ns")
open(f"profiles/{tag}_kernel_trace_stats.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

lines = []
for db in sorted(glob.glob(os.path.join(src, "pmc*", "*.db"))):
    lines.append(f"== rocprofv3 --pmc ({db}) ==")
    rows = q(db, "select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
                 "from counters_collection group by kernel_name, counter_name order by kernel_name, counter_name")
    lines.append(f"{'kernel':<50} {'counter':<24} {'n':>3} {'avg':>18} {'min':>18} {'max':>18} {'avg_dur_ns':>12}")
    for k, c, n, a, mn, mx, d in rows:
        lines.append(f"{k[:50]:<50} {c:<24} {n:>3} {a:>18.1f} {mn:>18.1f} {mx:>18.1f} {d:>12.0f}")
open(f"profiles/{tag}_pmc_counters.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

# HBM traffic per launch for bench.py's roofline.traffic (MI355X_MICROARCH.md section HBM:
# WRITE_SIZE and FETCH_SIZE in KiB from separate passes; FETCH_SIZE doubled on gfx950).
import json
w = f = None
for db in sorted(glob.glob(os.path.join(src, "pmc_write", "*.db"))):
    r = q(db, "select avg(value) from counters_collection where counter_name='WRITE_SIZE' and kernel_name like '%synth_%'")
    w = r[0][0]
for db in sorted(glob.glob(os.path.join(src, "pmc_fetch", "*.db"))):
    r = q(db, "select avg(value) from counters_collection where counter_name='FETCH_SIZE' and kernel_name like '%synth_%'")
    f = r[0][0]
if w is not None and f is not None:
    key = os.environ.get("PMC_KEY", "2600000_16_1_4130")
    path = "profiles/pmc_traffic.json"
    d = json.load(open(path)) if os.path.exists(path) else {}
    d[key] = int(w * 1024 + 2 * f * 1024)
    d[key + "_detail"] = {"kernels_id": kernels_id(), "WRITE_SIZE_KiB": w, "FETCH_SIZE_KiB_raw": f, "fetch_correction": "x2 (gfx950)", "source": f"profiles/{tag}_pmc_counters.txt"}
    json.dump(d, open(path, "w"), indent=1)
    print("traffic bytes/launch:", d[key])

# SQ counters per launch of the dominant kernel for bench.py's `counters` object (valu_issue_frac, lds_busy,
# lds_conflict_ratio): averages over the dispatches of the synth kernel, from the two SQ passes.
sq = {}
for sub in ("pmc_sq1", "pmc_sq2"):
    for db in sorted(glob.glob(os.path.join(src, sub, "*.db"))):
        for name, val, dur in q(db, "select counter_name, avg(value), avg(duration) from counters_collection "
                                    "where kernel_name like '%synth_%' group by counter_name"):
            sq[name] = val
            sq.setdefault("_avg_duration_ns_" + sub, dur)
if sq:
    key = os.environ.get("PMC_KEY", "2600000_16_1_4130")
    path = "profiles/pmc_counters.json"
    d = json.load(open(path)) if os.path.exists(path) else {}
    sq["source"] = f"profiles/{tag}_pmc_counters.txt"
    sq["kernels_id"] = kernels_id()
    d[key] = sq
    json.dump(d, open(path, "w"), indent=1, sort_keys=True)
    print("SQ counters per launch:", {k: v for k, v in sq.items() if not k.startswith("_")})
