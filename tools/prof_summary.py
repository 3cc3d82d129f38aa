#!/usr/bin/env python3
"""Summarise a `rocprofv3 --kernel-trace --stats -d DIR -- python3 bench.py ...` run into the table kept under
profiles/:  python tools/prof_summary.py DIR STEPS > profiles/rNN_bench_kernel_stats.txt   (STEPS = warm-up + timed)"""
import csv, glob, re, sys

d, steps = sys.argv[1], int(sys.argv[2])
rows = []
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((r["Name"], int(r["Calls"]), float(r["TotalDurationNs"])))
if not rows:      # rocprofv3's default output is a rocpd SQLite database
    import sqlite3
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(f)
        rows += [(n, int(k), float(t)) for n, k, t in c.execute("select name, count(*), sum(duration) from kernels group by name")]
if not rows:
    sys.exit("no kernel statistics under " + d)
tot = sum(r[2] for r in rows)


def short(n):
    n = n.replace("vqa::", "").replace("void ", "")
    n = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, (\d+), \d+>", lambda m: f"T{m.group(1)}x{m.group(2)}" + ("L8" if m.group(3) == "8" else ""), n)
    n = re.sub(r"\(.*$", "", n)
    return n[:86]


print(f"# total kernel time {tot/1e6:.1f} ms over {steps} steps = {tot/1e6/steps:.2f} ms/step")
print(f"{'kernel':88s}{'calls':>6s}{'ms/step':>9s}{'avg_us':>11s}{'pct':>7s}")
for n, c, t in sorted(rows, key=lambda r: -r[2])[:45]:
    print(f"{short(n):88s}{c:6d}{t/1e6/steps:9.3f}{t/1e3/c:11.1f}{100*t/tot:7.2f}")
