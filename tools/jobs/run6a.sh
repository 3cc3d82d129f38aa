#!/bin/bash
# round-2 final evidence run: tests, bench lines (fp32 headline + fp32x3 object, fp32x3 full line, bf16 configs[3], stress
# configs[4] in both fp32 modes, RCCL ws=1), rocprofv3 kernel stats and PMC passes.  Every step under its own timeout.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -x -q > $O/r02_gputests.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputests.log; tail -3 $O/r02_gputests.log
$T 300 python bench.py --steps 20 --warmup 3 > $O/r02_bench_fp32.json 2> $O/r02_bench_fp32.err; head -c 300 $O/r02_bench_fp32.json; echo
$T 300 python bench.py --dtype fp32x3 --steps 20 --warmup 3 --no-cpu-baseline > $O/r02_bench_fp32x3.json 2> $O/r02_bench_fp32x3.err; head -c 300 $O/r02_bench_fp32x3.json; echo
$T 300 python bench.py --dtype bf16 --batch 512 --size 448 --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_bf16_448.json 2> $O/r02_bench_bf16_448.err; head -c 300 $O/r02_bench_bf16_448.json; echo
$T 300 python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench_bf16_224.json 2> $O/r02_bench_bf16_224.err; head -c 300 $O/r02_bench_bf16_224.json; echo
$T 300 python bench.py --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline > $O/r02_bench_stress.json 2> $O/r02_bench_stress.err; head -c 300 $O/r02_bench_stress.json; echo
$T 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --no-cpu-baseline > $O/r02_bench_rccl_ws1.json 2> $O/r02_bench_rccl_ws1.err; head -c 300 $O/r02_bench_rccl_ws1.json; echo
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r02_prof_* $O/r02_pmc_*
VQA_STREAMS=1 $T 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_serial -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r02_prof_serial.log 2>&1
$T 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_fp32 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r02_prof_fp32.log 2>&1
VQA_STREAMS=1 $T 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_x3 -o p --output-format csv -- python3 $R/bench.py --dtype fp32x3 --steps 5 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r02_prof_x3.log 2>&1
$T 300 rocprofv3 --kernel-trace --stats -d $O/r02_prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_prof_bf16.log 2>&1
echo "kernel traces done"
cd $R
python3 tools/prof_summary.py $O/r02_prof_serial 7 > $O/r02_bench_kernel_stats_serial.txt
python3 tools/prof_summary.py $O/r02_prof_fp32 7 > $O/r02_bench_kernel_stats.txt
python3 tools/prof_summary.py $O/r02_prof_x3 7 > $O/r02_x3_kernel_stats.txt
python3 tools/prof_summary.py $O/r02_prof_bf16 4 > $O/r02_bf16_448_kernel_stats.txt
ls -la $O | grep r02 | head -40
