#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python bench.py --dtype bf16 --batch 512 --size 448 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_bf16_448.json 2> $O/r03_bench_bf16_448.err; head -c 200 $O/r03_bench_bf16_448.json; echo
