#!/bin/bash
# round-3 evidence run: tests, bench lines (fp32 headline, bf16 configs[3], stress configs[4], RCCL ws=1), rocprofv3 kernel
# stats and PMC passes.  Every step under its own timeout.  tools/jobs/r03_collect.sh copies the summaries to profiles/.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
T="timeout -k 10"
$T 900 python -m pytest tests -m gpu -q > $O/r03_gputests.log 2>&1; echo "pytest rc=$?" >> $O/r03_gputests.log; tail -3 $O/r03_gputests.log
$T 400 python bench.py --steps 20 --warmup 3 > $O/r03_bench_fp32.json 2> $O/r03_bench_fp32.err; head -c 300 $O/r03_bench_fp32.json; echo
$T 300 python bench.py --dtype bf16 --batch 512 --size 448 --steps 5 --warmup 2 --no-cpu-baseline > $O/r03_bench_bf16_448.json 2> $O/r03_bench_bf16_448.err; head -c 300 $O/r03_bench_bf16_448.json; echo
$T 300 python bench.py --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline > $O/r03_bench_stress.json 2> $O/r03_bench_stress.err; head -c 300 $O/r03_bench_stress.json; echo
$T 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --no-cpu-baseline > $O/r03_bench_rccl_ws1.json 2> $O/r03_bench_rccl_ws1.err; head -c 300 $O/r03_bench_rccl_ws1.json; echo
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r03_prof_* $O/r03_pm_* $O/r03_pmcc_*
VQA_STREAMS=1 $T 300 rocprofv3 --kernel-trace --stats -d $O/r03_prof_serial -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_prof_serial.log 2>&1
$T 300 rocprofv3 --kernel-trace --stats -d $O/r03_prof_fp32 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_prof_fp32.log 2>&1
VQA_STREAMS=1 $T 300 rocprofv3 --kernel-trace --stats -d $O/r03_prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_prof_bf16.log 2>&1
echo "kernel traces done"
for c in FETCH_SIZE WRITE_SIZE; do
  $T 200 rocprofv3 --pmc $c -d $O/r03_pmcc_fp32_$c -- python3 $R/tools/pmc_conv_run.py > $O/r03_pmcc_fp32_$c.log 2>&1
  $T 200 rocprofv3 --pmc $c -d $O/r03_pmcc_bf16_$c -- python3 $R/tools/pmc_conv_run.py --dtype bf16 --batch 512 --size 448 > $O/r03_pmcc_bf16_$c.log 2>&1
  $T 300 rocprofv3 --pmc $c -d $O/r03_pm_bf16_$c -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pm_bf16_$c.log 2>&1
  $T 300 rocprofv3 --pmc $c -d $O/r03_pm_fp32_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_pm_fp32_$c.log 2>&1
done
$T 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/r03_pm_bf16_mfma -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pm_bf16_mfma.log 2>&1
$T 300 rocprofv3 --pmc MfmaUtil --kernel-trace -d $O/r03_pm_fp32_mfma -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-x3 --stream-steps 0 > $O/r03_pm_fp32_mfma.log 2>&1
$T 300 rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace -d $O/r03_pm_bf16_clk -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r03_pm_bf16_clk.log 2>&1
echo "pmc done"
cd $R
python3 tools/prof_summary.py $O/r03_prof_serial 7 > $O/r03_bench_kernel_stats_serial.txt
python3 tools/prof_summary.py $O/r03_prof_fp32 7 > $O/r03_bench_kernel_stats.txt
python3 tools/prof_summary.py $O/r03_prof_bf16 4 > $O/r03_bf16_448_kernel_stats_serial.txt
python3 tools/pmc_traffic_summary.py $O/r03_pmcc_fp32_FETCH_SIZE $O/r03_pmcc_fp32_WRITE_SIZE > $O/r03_conv_traffic_fp32_224_256.json
python3 tools/pmc_traffic_summary.py $O/r03_pmcc_bf16_FETCH_SIZE $O/r03_pmcc_bf16_WRITE_SIZE > $O/r03_conv_traffic_bf16_448_512.json
python3 tools/pmc_traffic_by_name.py $O/r03_pm_bf16_FETCH_SIZE $O/r03_pm_bf16_WRITE_SIZE > $O/r03_step_traffic_bf16_448.json
python3 tools/pmc_traffic_by_name.py $O/r03_pm_fp32_FETCH_SIZE $O/r03_pm_fp32_WRITE_SIZE > $O/r03_step_traffic_fp32.json
python3 tools/pmc_by_name.py $O/r03_pm_bf16_mfma MfmaUtil > $O/r03_step_mfma_util_bf16_448.txt
python3 tools/pmc_by_name.py $O/r03_pm_fp32_mfma MfmaUtil > $O/r03_step_mfma_util_fp32.txt
python3 tools/pmc_clock_by_name.py $O/r03_pm_bf16_clk > $O/r03_clock_bf16_448.txt
rm -rf $O/r03_prof_serial $O/r03_prof_fp32 $O/r03_prof_bf16 $O/r03_pm_* $O/r03_pmcc_*_SIZE
ls $O | grep r03_ | head -60
