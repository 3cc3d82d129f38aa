#!/usr/bin/env python3
"""Average of one PMC counter per kernel name over a whole run:
    rocprofv3 --pmc MfmaUtil -d gpurun_out/pmc_step -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline
    python tools/pmc_by_name.py gpurun_out/pmc_step MfmaUtil > profiles/rNN_step_mfma_util.txt"""
import glob, re, sqlite3, sys

d, counter = sys.argv[1], sys.argv[2]
acc = {}
for f in glob.glob(d + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    for name, v in c.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
        n = name.replace("vqa::", "").replace("void ", "")
        n = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, (\d+), \d+>", lambda m: f"T{m.group(1)}x{m.group(2)}" + ("L8" if m.group(3) == "8" else ""), n)
        n = re.sub(r"\(.*$", "", n)[:80]
        a = acc.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(v)
print(f"# {counter} per kernel name (average over launches; rocprofv3 --pmc {counter})")
print(f"{'kernel':82s}{'launches':>9s}{counter:>10s}")
for n, (k, t) in sorted(acc.items(), key=lambda kv: -kv[1][1]):
    if t / k >= 0.05:
        print(f"{n:82s}{k:9d}{t / k:10.2f}")
