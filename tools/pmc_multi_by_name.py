#!/usr/bin/env python3
"""Per-kernel-name averages of several PMC counters from one rocprofv3 --pmc pass:
    rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES \\
        -d gpurun_out/pmc_x -- python3 tools/kbench_pconv.py --iters 2
    python tools/pmc_multi_by_name.py gpurun_out/pmc_x [name-filter]"""
import glob, re, sqlite3, sys

d = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
acc, counters = {}, []
for f in glob.glob(d + "/**/*_results.db", recursive=True):
    c = sqlite3.connect(f)
    for name, cn, v in c.execute("select name, counter_name, counter_value from pmc_events"):
        n = re.sub(r"\(.*$", "", name.replace("vqa::", "").replace("void ", ""))[:70]
        if flt and flt not in n:
            continue
        if cn not in counters:
            counters.append(cn)
        a = acc.setdefault(n, {}).setdefault(cn, [0, 0.0])
        a[0] += 1
        a[1] += float(v)
print(f"{'kernel':72s}" + "".join(f"{c[-22:]:>24s}" for c in counters))
for n, row in sorted(acc.items()):
    print(f"{n:72s}" + "".join(f"{(row[c][1] / row[c][0]) if c in row else float('nan'):24.4g}" for c in counters))
