#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputests_aa.log 2>&1; rc=$?; tail -3 $O/r03_gputests_aa.log
[ $rc -eq 0 ] || exit 1
VQA_PDGRAD=2 timeout -k 10 600 python -m pytest tests/test_model_gpu.py -m gpu -x -q > $O/r03_gputests_aa2.log 2>&1; rc=$?; tail -2 $O/r03_gputests_aa2.log
[ $rc -eq 0 ] || exit 1
for m in 0 1 2 1 0; do
  VQA_PDGRAD=$m timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_bench_fp32_pd$m.json 2> $O/r03_bench_fp32_pd$m.err; echo "VQA_PDGRAD=$m $(head -c 170 $O/r03_bench_fp32_pd$m.json | tail -c 70)"
done
