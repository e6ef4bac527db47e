#!/usr/bin/env python3
"""rocprofv3 outputs of scripts/exact_call_prof.py (rocpd sqlite under gpurun_out/exact_prof) -> profiles/<tag>_exact_call_*.txt:
the kernel-trace statistics, the timeline of the LAST call of each batch size (every dispatch: start offset, duration, queue),
and the HBM traffic of one call from the WRITE_SIZE / FETCH_SIZE passes (profiles/pmc_traffic.json, key exact_<fs>_<nchan>_<ss>_<blocks>)."""
import glob
import json
import os
import sqlite3
import sys

src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/exact_prof"
tag = sys.argv[2] if len(sys.argv) > 2 else "r06"
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))


def q(db, sql):
    con = sqlite3.connect(db)
    try:
        return con.execute(sql).fetchall()
    finally:
        con.close()


def short(name):
    name = name.split("(")[0]
    return name.replace("gpsiq::", "").replace("void ", "")[:44]


lines = []
for db in sorted(glob.glob(os.path.join(src, "kt*", "*.db"))):
    lines.append(f"== rocprofv3 --kernel-trace --stats -- python scripts/exact_call_prof.py ({os.path.basename(db)}) ==")
    lines.append(f"{'kernel':<46} {'calls':>6} {'total_us':>12} {'avg_us':>10} {'pct':>7}")
    for name, calls, tot, avg, pct in q(db, "select name,total_calls,total_duration,average,percentage from top_kernels"):
        lines.append(f"{short(name):<46} {calls:>6} {tot / 1e3:>12.1f} {avg / 1e3:>10.2f} {pct:>7.2f}")
    cols = [r[1] for r in q(db, "pragma table_info(kernels)")]
    qcol = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else None)
    sel = "name, start, end, grid_x, workgroup_x" + (", " + qcol if qcol else "")
    rows = q(db, f"select {sel} from kernels order by start")
    # a call ends with its apply_patches (every call of the profiled script has patches)
    calls_, cur = [], []
    for r in rows:
        cur.append(r)
        if "apply_patches" in r[0]:
            calls_.append(cur); cur = []
    seen = set()
    for call in reversed(calls_):
        nsynth = sum(1 for r in call if "synth_tile" in r[0])
        grid = sum(r[3] for r in call if "synth_tile" in r[0])
        key = (nsynth, grid)
        if not nsynth or key in seen or not any("eval_blocks" in r[0] for r in call):
            continue
        seen.add(key)
        t0 = call[0][1]
        lines.append(f"-- timeline of one call ({len(call)} dispatches, {(max(r[2] for r in call) - t0) / 1e3:.1f} us from the first dispatch's start to the last one's end; "
                     f"busy union {sum(r[2] - r[1] for r in call) / 1e3:.1f} us summed) --")
        lines.append(f"   {'start_us':>9} {'dur_us':>9} {'grid':>9} {'wg':>4} {qcol or '':>8}  kernel")
        for r in call:
            lines.append(f"   {(r[1] - t0) / 1e3:>9.1f} {(r[2] - r[1]) / 1e3:>9.1f} {r[3]:>9} {r[4]:>4} {str(r[5]) if qcol else '':>8}  {short(r[0])}")
open(f"profiles/{tag}_exact_call_kernel_trace.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))

lines = []
tot = {}
for sub, counter in (("pmc_write", "WRITE_SIZE"), ("pmc_fetch", "FETCH_SIZE")):
    for db in sorted(glob.glob(os.path.join(src, sub, "*.db"))):
        lines.append(f"== rocprofv3 --pmc {counter} ({os.path.basename(db)}) ==")
        rows = q(db, f"select kernel_name, count(*), sum(value), avg(value), avg(duration) from counters_collection where counter_name='{counter}' group by kernel_name order by sum(value) desc")
        lines.append(f"{'kernel':<46} {'n':>5} {'sum_KiB':>16} {'avg_KiB':>14} {'avg_dur_us':>11}")
        for k, n, s, a, d in rows:
            lines.append(f"{short(k):<46} {n:>5} {s:>16.1f} {a:>14.1f} {d / 1e3:>11.2f}")
        tot[counter] = sum(r[2] for r in rows)
        tot[counter + "_dispatches"] = sum(r[1] for r in rows)
open(f"profiles/{tag}_exact_call_pmc.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
meta = os.path.join(src, "meta.json")
if "WRITE_SIZE" in tot and "FETCH_SIZE" in tot and os.path.exists(meta):
    from prof_summary_id import kernels_id
    m = json.load(open(meta))            # {"calls": c, "blocks": [2000, 4129], "nsamp": .., "ss": ..}
    # both sizes were run `calls` times each in one process: split the totals by their share of the algorithmic bytes
    alg = {nb: nb * m["nsamp"] * 2 * m["ss"] for nb in m["blocks"]}
    whole = (tot["WRITE_SIZE"] + 2 * tot["FETCH_SIZE"]) * 1024 / m["calls"]
    path = "profiles/pmc_traffic.json"
    d = json.load(open(path)) if os.path.exists(path) else {}
    for nb in m["blocks"]:
        key = f"exact_2600000_16_{m['ss']}_{nb}"
        d[key] = int(whole * alg[nb] / sum(alg.values()))
        d[key + "_detail"] = {"kernels_id": kernels_id(), "what": "HBM bytes of one GPSIQ_NCO_REFERENCE gpsiq_generate_batch call, every kernel of it (pack/upload excluded: copies are not kernels): "
                              "WRITE_SIZE + 2 x FETCH_SIZE over all dispatches of scripts/exact_call_prof.py, shared between the two batch sizes by their algorithmic bytes",
                              "algorithmic_bytes": alg[nb], "ratio": round(d[key] / alg[nb], 4), "source": f"profiles/{tag}_exact_call_pmc.txt"}
    json.dump(d, open(path, "w"), indent=1)
    print({k: v for k, v in d.items() if k.startswith("exact_") and not k.endswith("_detail")})
