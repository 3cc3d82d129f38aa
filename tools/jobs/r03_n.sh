#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/r03_tests_n.log 2>&1; rc=$?; tail -3 $O/r03_tests_n.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_n.json 2> $O/r03_bench_bf16_448_n.err; head -c 250 $O/r03_bench_bf16_448_n.json; echo
