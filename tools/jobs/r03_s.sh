#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 300 python -m pytest tests/test_bf16_gpu.py -m gpu -x -q -k "pconv or patch" > $O/r03_tests_s.log 2>&1; rc=$?; tail -2 $O/r03_tests_s.log
[ $rc -eq 0 ] || exit 1
for rep in 1 2; do
for st in 0 1; do
  echo "VQA_PCONV_STAGGER=$st"
  VQA_PCONV_STAGGER=$st timeout -k 10 200 python tools/kbench_pconv.py --skip-old --iters 8 2>&1 | grep -v amdgpu.ids
done
done > $O/r03_kbench_stagger.txt 2>&1
cat $O/r03_kbench_stagger.txt
