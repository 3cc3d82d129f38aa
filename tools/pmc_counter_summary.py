#!/usr/bin/env python3
"""Per-launch average of one PMC counter for the conv kernels of tools/pmc_conv_run.py:
    rocprofv3 --pmc MfmaUtil -d gpurun_out/pmc_mfma -- python3 tools/pmc_conv_run.py
    python tools/pmc_counter_summary.py gpurun_out/pmc_mfma MfmaUtil > profiles/rNN_mfma_util.json
Events of a family alternate layer 1, layer 2 in dispatch order (see the workload)."""
import glob, json, sqlite3, sys

d, counter = sys.argv[1], sys.argv[2]
out = {}
for f in glob.glob(d + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    q = "select name, counter_value, dispatch_id from pmc_events where counter_name = ? order by dispatch_id"
    for name, v, _ in c.execute(q, (counter,)):
        for key in ("conv_fwd", "conv_wgrad", "conv_dgrad"):
            if key + "_kernel" in name:
                out.setdefault(key, []).append(float(v))
res = {}
for k, v in sorted(out.items()):
    for layer in (1, 2):
        vals = v[layer - 1::2]
        if vals:
            res[f"{k}:{layer}"] = {counter: round(sum(vals) / len(vals), 3), "launches": len(vals)}
print(json.dumps(res, indent=1))
