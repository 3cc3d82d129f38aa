#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputests_w.log 2>&1; rc=$?; tail -3 $O/r03_gputests_w.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 200 python tools/kbench.py --only conv0_fast_fwd,conv0_fast_wgrad --iters 10 2>&1 | grep -v amdgpu.ids
timeout -k 10 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_bench_fp32_w.json 2> $O/r03_bench_fp32_w.err; head -c 250 $O/r03_bench_fp32_w.json; echo
