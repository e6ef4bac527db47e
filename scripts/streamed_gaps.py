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
leg = main[-rounds:]
out = [f"== streamed end-to-end leg, second pass: the last {rounds} synthesis dispatches of {db} =="]
out.append(f"{'round':>5} {'kernel_us':>10} {'gap_before_us':>14}")
busy = 0
for i, (s, d) in enumerate(leg):
    gap = (s - (leg[i - 1][0] + leg[i - 1][1])) / 1e3 if i else float("nan")
    out.append(f"{i:>5} {d / 1e3:>10.1f} {gap:>14.1f}")
    busy += d
span = leg[-1][0] + leg[-1][1] - leg[0][0]
out.append(f"first kernel start to last kernel end: {span / 1e6:.3f} ms, kernels {busy / 1e6:.3f} ms = {100.0 * busy / span:.1f} % busy; "
           f"gaps: mean {(span - busy) / (rounds - 1) / 1e3:.1f} us")
os.makedirs("profiles", exist_ok=True)
open(f"profiles/{tag}_streamed_leg_gaps.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
