#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "tall" > $O/r03_tests_l.log 2>&1; rc=$?; tail -3 $O/r03_tests_l.log
[ $rc -eq 0 ] || exit 1
VQA_TALL_BM=256 timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "tall" > $O/r03_tests_l2.log 2>&1; rc=$?; tail -3 $O/r03_tests_l2.log
[ $rc -eq 0 ] || exit 1
( VQA_TALL_BM=256 VQA_LIB=build_var/libvqa_talldiag.so timeout -k 10 200 python tools/kbench_tall.py --dbg 0,1,8,16,0
  echo BM128;  VQA_LIB=build_var/libvqa_talldiag.so timeout -k 10 200 python tools/kbench_tall.py --dbg 0,1,8,16,0 ) > $O/r03_kbench_tall4.txt 2>&1; cat $O/r03_kbench_tall4.txt
