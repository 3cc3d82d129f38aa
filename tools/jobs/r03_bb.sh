#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > $O/r03_tests_bb.log 2>&1; rc=$?; tail -3 $O/r03_tests_bb.log
[ $rc -eq 0 ] || exit 1
for m in 0 1 0 1; do
  VQA_GEMM_WIDE=$m timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_wide$m.json 2> $O/r03_bench_bf16_wide$m.err; echo "VQA_GEMM_WIDE=$m $(head -c 170 $O/r03_bench_bf16_wide$m.json | tail -c 60)"
done
bash tools/jobs/r03_p.sh 40 | grep -E "total kernel|gemm_bf16"
