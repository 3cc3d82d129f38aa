#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/r03_bf16_tests.log 2>&1; echo rc=$? >> $O/r03_bf16_tests.log; tail -6 $O/r03_bf16_tests.log
timeout -k 10 200 python tools/kbench_pconv.py --skip-old --iters 3 --dbg 0,1,2,3,4,6 2>&1 | grep "pconv"
timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_a.json 2> $O/r03_bench_bf16_448_a.err; head -c 400 $O/r03_bench_bf16_448_a.json; echo
VQA_PCONV=0 timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_old.json 2> $O/r03_bench_bf16_448_old.err; head -c 400 $O/r03_bench_bf16_448_old.json; echo
