#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "pconv or patch or configs3" > $O/r03_tests_e.log 2>&1; rc=$?; echo rc=$rc >> $O/r03_tests_e.log; tail -25 $O/r03_tests_e.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/kbench_pconv.py --skip-old > $O/r03_kbench_pconv_e.txt 2>&1; cat $O/r03_kbench_pconv_e.txt
