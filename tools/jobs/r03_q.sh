#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputests_q.log 2>&1; rc=$?; tail -3 $O/r03_gputests_q.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_q.json 2> $O/r03_bench_bf16_448_q.err; head -c 250 $O/r03_bench_bf16_448_q.json; echo
timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_bench_fp32_q.json 2> $O/r03_bench_fp32_q.err; head -c 250 $O/r03_bench_fp32_q.json; echo
bash tools/jobs/r03_p.sh 30 | grep -E "total kernel|att_score|pconv_wgrad|l2norm"
