#!/bin/bash
# copy the summaries that tools/jobs/r03_evidence.sh merged into gpurun_out/ to profiles/ (tracked), with the command lines
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out; P=profiles
for n in fp32 bf16_448 stress rccl_ws1; do tail -1 $O/r03_bench_$n.json > $P/r03_bench_line_$n.json; done
( echo "# VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0   (question branch NOT overlapped with the convolutions: isolated kernel durations)"; cat $O/r03_bench_kernel_stats_serial.txt ) > $P/r03_bench_kernel_stats_serial.txt
( echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0   (default schedule: question branch under the convolutions)"; cat $O/r03_bench_kernel_stats.txt ) > $P/r03_bench_kernel_stats.txt
( echo "# VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0   (isolated kernel durations)"; cat $O/r03_bf16_448_kernel_stats_serial.txt ) > $P/r03_bf16_448_kernel_stats.txt
cp $O/r03_conv_traffic_fp32_224_256.json $O/r03_conv_traffic_bf16_448_512.json $O/r03_step_traffic_fp32.json $O/r03_step_traffic_bf16_448.json $P/
cp $O/r03_step_mfma_util_fp32.txt $O/r03_step_mfma_util_bf16_448.txt $O/r03_clock_bf16_448.txt $P/
tail -4 $O/r03_gputests.log > $P/r03_gputests_tail.txt
for f in r03_kbench_tall.txt r03_kbench_tall2.txt r03_kbench_tall3.txt r03_kbench_tall4.txt; do [ -f $O/$f ] && cp $O/$f $P/ ; done
python3 - <<'PY'
import json
for n in ('fp32','bf16_448','stress','rccl_ws1'):
    d=json.load(open(f'profiles/r03_bench_line_{n}.json'))
    r=d.get('roofline') or {}
    print(n, d['value'], d['ms_per_step'], d.get('step_mfma_frac'), r.get('kernel'), r.get('frac'), r.get('traffic'), r.get('algorithmic_bytes_per_launch'))
PY
