#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 600 python -m pytest tests/test_bf16_gpu.py tests/test_model_gpu.py -m gpu -x -q > $O/r03_tests_cc.log 2>&1; rc=$?; tail -3 $O/r03_tests_cc.log
[ $rc -eq 0 ] || exit 1
bash tools/jobs/r03_p.sh 40 | grep -E "total kernel|conv0"
