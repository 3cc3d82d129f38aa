#!/bin/bash
# round-2 final evidence run: tests, bench lines (fp32 headline + fp32x3 object, fp32x3 full line, bf16 configs[3], stress
# configs[4] in both fp32 modes, RCCL ws=1), rocprofv3 kernel stats and PMC passes.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T="timeout -k 10"
$T 300 python bench.py --dtype fp32x3 --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline > $O/r02_bench_stress_fp32x3.json 2> $O/r02_bench_stress_fp32x3.err; head -c 300 $O/r02_bench_stress_fp32x3.json; echo
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r02_pmc_*
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  $T 300 rocprofv3 --pmc $c -d $O/r02_pmc_fp32_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r02_pmc_fp32_$c.log 2>&1
  $T 300 rocprofv3 --pmc $c -d $O/r02_pmc_x3_$c -- python3 $R/bench.py --dtype fp32x3 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_pmc_x3_$c.log 2>&1
  $T 300 rocprofv3 --pmc $c -d $O/r02_pmc_bf16_$c -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_pmc_bf16_$c.log 2>&1
  echo "pmc $c done"
done
cd $R
for m in fp32 x3 bf16; do
  python3 tools/pmc_traffic_by_name.py $O/r02_pmc_${m}_FETCH_SIZE $O/r02_pmc_${m}_WRITE_SIZE > $O/r02_step_traffic_$m.json
  python3 tools/pmc_by_name.py $O/r02_pmc_${m}_MfmaUtil MfmaUtil > $O/r02_step_mfma_util_$m.txt
done
# the raw PMC databases are large: keep the summaries only
rm -rf $O/r02_pmc_*
ls -la $O | grep r02 | head -40
