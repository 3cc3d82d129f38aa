#!/bin/bash
# refresh of the configs[3] records after a change to the bf16 path: bench line + isolated kernel durations
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T="timeout -k 10"
$T 300 python bench.py --dtype bf16 --batch 512 --size 448 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_bf16_448.json 2> $O/r03_bench_bf16_448.err; head -c 300 $O/r03_bench_bf16_448.json; echo
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r03_prof_bf16
VQA_STREAMS=1 $T 300 rocprofv3 --kernel-trace --stats -d $O/r03_prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_prof_bf16.log 2>&1
cd $R
python3 tools/prof_summary.py $O/r03_prof_bf16 4 > $O/r03_bf16_448_kernel_stats_serial.txt
rm -rf $O/r03_prof_bf16
head -12 $O/r03_bf16_448_kernel_stats_serial.txt
