#!/usr/bin/env python3
"""From a rocprofv3 --kernel-trace of `bench.py --no-cpu-baseline --no-extra --rounds R`: the gaps between the kernels of the
streamed end-to-end leg (the last R dispatches of the run = its second pass), i.e. how much of that leg the GPU was busy.
usage: streamed_gaps.py <dir with kt*/ *.db> <rounds> <tag>"""
import glob
import os
import sqlite3
import sys

src, rounds, tag = sys.argv[1], int(sys.argv[2]), sys.argv[3]
db = sorted(glob.glob(os.path.join(src, "kt*", "*.db")))[0]
con = sqlite3.connect(db)
rows = con.execute("select name, start, duration from kernels order by start").fetchall()
con.close()
main = [(s, d) for n, s, d in rows if "synth_" in n]
out = []
for name, leg in (("first pass", main[-2 * rounds:-rounds]), ("second pass", main[-rounds:])):
    out.append(f"== streamed end-to-end leg, {name}: {rounds} consecutive synthesis dispatches of {db} ==")
    out.append(f"{'round':>5} {'kernel_us':>10} {'gap_before_us':>14}")
    busy = 0
    gaps = []
    for i, (s, d) in enumerate(leg):
        gap = (s - (leg[i - 1][0] + leg[i - 1][1])) / 1e3 if i else float("nan")
        if i:
            gaps.append(gap)
        out.append(f"{i:>5} {d / 1e3:>10.1f} {gap:>14.1f}")
        busy += d
    span = leg[-1][0] + leg[-1][1] - leg[0][0]
    gaps.sort()
    out.append(f"first kernel start to last kernel end: {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms = {100.0 * busy / span:.1f} % busy; "
               f"gaps: median {gaps[len(gaps) // 2]:.1f} us, max {gaps[-1]:.1f} us")
js = [ln for ln in open(os.path.join(src, "kt.log")) if ln.startswith("{")]
if js:
    import json
    e = json.loads(js[-1])["end_to_end"]["streamed"]
    out.append(f"bench.py's own clock in this (profiled) run: {e['rounds']} rounds in {e['seconds'] * 1e3:.2f} ms (best of the two passes), {e['value']} Msamples/s")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_streamed_leg_gaps.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
