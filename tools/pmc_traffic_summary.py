#!/usr/bin/env python3
"""HBM-side bytes per launch of the conv kernels from two rocprofv3 --pmc passes (see tools/pmc_conv_run.py).
gfx950 corrections (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 128-byte requests as 64 bytes -> x2; WRITE_SIZE exact.
The workload launches every family alternately for layer 1 and layer 2, so a family's events in dispatch order
split into the two layers by parity."""
import glob, json, re, sqlite3, sys


NAMES = {}


def per_kernel(d, counter):
    out = {}
    for f in glob.glob(d + "/**/*_results.db", recursive=True):
        c = sqlite3.connect(f)
        q = "select name, counter_value, dispatch_id from pmc_events where counter_name = ? order by dispatch_id"
        for name, v, _ in c.execute(q, (counter,)):
            short = name.replace("vqa::", "").replace("void ", "").split("(")[0][:80]
            key = None
            for k in ("conv_fwd", "conv_wgrad", "conv_dgrad"):
                if short.startswith(k + "_"):     # conv_fwd_kernel, conv_fwd_bf16_kernel, ...
                    key = k
            # bf16 path (csrc/conv_patch_bf16.hip): pconv_kernel<WM, EPI> is the forward for EPI 0/1 (pooled output) and the
            # backward-data for EPI 2/3/4; the weight gradient is pconv_wgrad_kernel (+ its split reduce, listed beside it)
            m = re.match(r"pconv_kernel<\d+, (\d)[,>]", short)
            if m:
                key = "conv_fwd" if m.group(1) in "01" else "conv_dgrad"
            elif short.startswith("pconv_wgrad_kernel"):
                key = "conv_wgrad"
            if key:
                out.setdefault(key, []).append(float(v))
                NAMES.setdefault(key, []).append(short)
    res = {}
    for k, v in out.items():
        for layer in (1, 2):
            vals = v[layer - 1::2]
            if vals:
                res[f"{k}:{layer}"] = sum(vals) / len(vals)
    return res


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
res = {}
for k in sorted(fetch):
    fam, layer = k.split(":")
    res[k] = {"FETCH_SIZE_KiB": fetch[k], "WRITE_SIZE_KiB": write.get(k, 0.0),
              "hbm_bytes_corrected": 2 * fetch[k] * 1024 + write.get(k, 0.0) * 1024,
              "kernel_names": sorted(set(NAMES.get(fam, [])[int(layer) - 1::2]))}
print(json.dumps(res, indent=1))
