#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 150 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "tall" > $O/r03_tests_dd.log 2>&1; rc=$?; tail -3 $O/r03_tests_dd.log
[ $rc -eq 0 ] || exit 1
VQA_TALL_RS=0 timeout -k 10 150 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "tall" > $O/r03_tests_dd2.log 2>&1; rc=$?; tail -2 $O/r03_tests_dd2.log
[ $rc -eq 0 ] || exit 1
( timeout -k 10 100 python tools/kbench_tall.py --iters 8; VQA_TALL_RS=0 timeout -k 10 100 python tools/kbench_tall.py --iters 8; timeout -k 10 100 python tools/kbench_tall.py --iters 8 ) 2>&1 | grep -v amdgpu.ids
