#!/bin/bash
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd /tmp; export TMPDIR=/tmp
rm -rf $O/r03_prof_bf16
VQA_STREAMS=1 timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/r03_prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_prof_bf16.log 2>&1
cd $R
python3 tools/prof_summary.py $O/r03_prof_bf16 4 > $O/r03_bf16_448_kernel_stats_serial.txt; head -${1:-26} $O/r03_bf16_448_kernel_stats_serial.txt
rm -rf $O/r03_prof_bf16
