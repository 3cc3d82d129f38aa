#!/bin/bash
# kernel statistics of the fp32x3 step (question branch not overlapped: isolated kernel durations)
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/x3_prof
VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/x3_prof -o p --output-format csv -- python3 $R/bench.py --dtype fp32x3 --steps 5 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/x3_prof.log 2>&1
tail -1 $O/x3_prof.log | cut -c1-200
python3 $R/tools/prof_summary.py $O/x3_prof 7 > $O/x3_kernel_stats.txt; head -40 $O/x3_kernel_stats.txt
