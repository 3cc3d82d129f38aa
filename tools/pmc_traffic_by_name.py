#!/usr/bin/env python3
"""HBM-side bytes per launch, per kernel name, from two rocprofv3 --pmc passes over the same command:
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_f -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_w -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0
    python tools/pmc_traffic_by_name.py gpurun_out/pmc_f gpurun_out/pmc_w > profiles/rNN_step_traffic.json
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts a wide coalesced 128-byte request as 64 bytes -> x2
(calibrated for 16-byte-per-lane streams; narrower access patterns are uncalibrated); WRITE_SIZE is exact.
Both counters are reported in KiB by rocprofv3."""
import glob, json, re, sqlite3, sys


def short(n):
    n = n.replace("vqa::", "").replace("void ", "")
    n = re.sub(r"TileCfg<(\d+), (\d+), \d+, \d+, (\d+), \d+>", lambda m: f"T{m.group(1)}x{m.group(2)}" + ("L8" if m.group(3) == "8" else ""), n)
    return re.sub(r"\(.*$", "", n)[:90]


def per_name(d, counter):
    acc = {}
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(f)
        for name, v in c.execute("select name, counter_value from pmc_events where counter_name = ?", (counter,)):
            a = acc.setdefault(short(name), [0, 0.0])
            a[0] += 1
            a[1] += float(v)
    return acc


fetch, write = per_name(sys.argv[1], "FETCH_SIZE"), per_name(sys.argv[2], "WRITE_SIZE")
out = {}
for k in sorted(set(fetch) | set(write), key=lambda k: -(2 * fetch.get(k, [0, 0])[1] + write.get(k, [0, 0])[1])):
    nf, f = fetch.get(k, [0, 0.0])
    nw, w = write.get(k, [0, 0.0])
    n = max(nf, nw, 1)
    out[k] = {"launches": n, "FETCH_SIZE_KiB_per_launch": f / n, "WRITE_SIZE_KiB_per_launch": w / n,
              "hbm_bytes_corrected_per_launch": (2 * f + w) * 1024 / n}
print(json.dumps(out, indent=1))
