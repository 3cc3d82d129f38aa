#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
( for rep in 1 2; do
  echo "== shipped"; timeout -k 10 200 python tools/kbench.py --only conv1_fwd,conv1_dgrad,conv1_wgrad,conv2_fwd,conv2_dgrad,conv2_wgrad --iters 8 2>&1 | grep -v amdgpu.ids
  echo "== -DVQA_EXP_NOROUTE (results wrong: timing only)"; VQA_LIB=build_var/libvqa_noroute.so timeout -k 10 200 python tools/kbench.py --only conv1_fwd,conv1_dgrad,conv1_wgrad,conv2_fwd,conv2_dgrad,conv2_wgrad --iters 8 2>&1 | grep -v amdgpu.ids
done ) > $O/r03_kbench_noroute.txt 2>&1; cat $O/r03_kbench_noroute.txt
