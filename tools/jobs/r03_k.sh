#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
T="timeout -k 10"
rm -rf $O/r03_pmc_*
for c in FETCH_SIZE WRITE_SIZE; do
  $T 300 rocprofv3 --pmc $c -d $O/r03_pmc_bf16_$c -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pmc_bf16_$c.log 2>&1 || exit 1
done
cd $R
python3 tools/pmc_traffic_by_name.py $O/r03_pmc_bf16_FETCH_SIZE $O/r03_pmc_bf16_WRITE_SIZE > $O/r03_step_traffic_bf16_448.json
rm -rf $O/r03_pmc_bf16_FETCH_SIZE $O/r03_pmc_bf16_WRITE_SIZE
python3 - <<'PY'
import json
d=json.load(open('gpurun_out/r03_step_traffic_bf16_448.json'))
rows=d if isinstance(d,list) else d.get('kernels',d)
if isinstance(rows,dict):
    rows=[dict(kernel=k,**v) for k,v in rows.items() if isinstance(v,dict)]
for r in rows[:40]: print(r)
PY
