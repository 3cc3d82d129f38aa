#!/usr/bin/env python3
"""HBM-side bytes per launch of the conv kernels from two rocprofv3 --pmc passes (see tools/pmc_conv1_run.py).
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests as 64 bytes -> x2; WRITE_SIZE exact."""
import glob, json, sqlite3, sys


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(f)
        q = "select name, counter_value from pmc_events where counter_name = ?"
        try:
            rows = list(c.execute(q, (counter,)))
        except sqlite3.OperationalError:
            cols = [r[1] for r in c.execute("pragma table_info('pmc_events')")]
            sys.exit(f"unexpected pmc_events schema {cols}")
        for name, v in rows:
            for key in ("conv_fwd", "conv_wgrad", "conv_dgrad"):
                if key + "_kernel" in name:
                    out.setdefault(key, []).append(float(v))
    return {k: sum(v) / len(v) for k, v in out.items()}


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in fetch:
    res[k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0),
              "hbm_bytes_corrected": 2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024}
print(json.dumps(res, indent=1))
