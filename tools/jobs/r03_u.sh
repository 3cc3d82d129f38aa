#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputests_u.log 2>&1; rc=$?; tail -3 $O/r03_gputests_u.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_bench_fp32_u.json 2> $O/r03_bench_fp32_u.err; head -c 250 $O/r03_bench_fp32_u.json; echo
timeout -k 10 300 python bench.py --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_bench_stress_u.json 2> $O/r03_bench_stress_u.err; head -c 250 $O/r03_bench_stress_u.json; echo
