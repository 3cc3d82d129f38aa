#!/usr/bin/env python3
"""Effective clock per kernel name from one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` run:
clock ~= GRBM_GUI_ACTIVE / 8 / kernel duration (MI355X_MICROARCH.md, DVFS give-back: rocprofv3 reports the sum over the 8
XCDs; reads high on dispatches shorter than ~0.3 ms).   python tools/pmc_clock_by_name.py DIR > profiles/rNN_clock.txt"""
import glob, re, sqlite3, sys


def short(n):
    n = n.replace("vqa::", "").replace("void ", "")
    n = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, (\d+), \d+>", lambda m: f"T{m.group(1)}x{m.group(2)}" + ("L8" if m.group(3) == "8" else ""), n)
    return re.sub(r"\(.*$", "", n)[:86]


rows = {}
for f in glob.glob(sys.argv[1] + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    cols = [r[1] for r in c.execute("pragma table_info(pmc_events)")]
    dcol = "dispatch_id" if "dispatch_id" in cols else None
    if dcol is None:
        sys.exit("pmc_events has no dispatch_id column: " + ", ".join(cols))
    kcols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    kid = "dispatch_id" if "dispatch_id" in kcols else "id"
    # pmc_events holds one row per XCD and dispatch: sum the counter over a dispatch's rows, take its duration once
    q = (f"select k.name, sum(p.counter_value), max(k.duration) from pmc_events p join kernels k on k.{kid} = p.{dcol} "
         f"where p.counter_name = 'GRBM_GUI_ACTIVE' group by p.{dcol}")
    for name, v, dur in c.execute(q):
        a = rows.setdefault(short(name), [0, 0.0, 0.0])
        a[0] += 1; a[1] += float(v); a[2] += float(dur)
print("# effective clock per kernel name = GRBM_GUI_ACTIVE / 8 / duration (kernels of at least 0.3 ms only)")
print(f"{'kernel':88s}{'launches':>9s}{'avg_us':>10s}{'GHz':>7s}")
for n, (k, v, d) in sorted(rows.items(), key=lambda kv: -kv[1][2]):
    if d / k < 3e5:
        continue
    print(f"{n:88s}{k:9d}{d / k / 1e3:10.1f}{v / 8.0 / d:7.2f}")
