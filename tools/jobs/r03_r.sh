#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r03_gputests_r.log 2>&1; rc=$?; tail -3 $O/r03_gputests_r.log
[ $rc -eq 0 ] || exit 1
bash tools/jobs/r03_p.sh 40 | grep -E "total kernel|att_score|l2norm|conv0"
