#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/r03_tests_m.log 2>&1; rc=$?; tail -3 $O/r03_tests_m.log
[ $rc -eq 0 ] || exit 1
VQA_TALL_BM=128 timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "tall" > $O/r03_tests_m2.log 2>&1; rc=$?; tail -2 $O/r03_tests_m2.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_m.json 2> $O/r03_bench_bf16_448_m.err; head -c 250 $O/r03_bench_bf16_448_m.json; echo
