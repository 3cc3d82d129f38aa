#!/bin/bash
# copy the summaries that tools/jobs/run6a.sh + run6b.sh merged into gpurun_out/ to profiles/ (tracked), with the command lines
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out; P=profiles
for n in fp32 fp32x3 bf16_448 bf16_224 stress stress_fp32x3 rccl_ws1; do tail -1 $O/r02_bench_$n.json > $P/r02_bench_line_$n.json; done
( echo "# VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0   (question branch NOT overlapped with the convolutions: isolated kernel durations)"; cat $O/r02_bench_kernel_stats_serial.txt ) > $P/r02_bench_kernel_stats_serial.txt
( echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-x3 --stream-steps 0   (default schedule: question branch under the convolutions)"; cat $O/r02_bench_kernel_stats.txt ) > $P/r02_bench_kernel_stats.txt
( echo "# VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype fp32x3 --steps 5 --warmup 2 --no-cpu-baseline --stream-steps 0   (fp32x3 mode, isolated kernel durations)"; cat $O/r02_x3_kernel_stats.txt ) > $P/r02_x3_kernel_stats.txt
( echo "# rocprofv3 --kernel-trace --stats -- python3 bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0"; cat $O/r02_bf16_448_kernel_stats.txt ) > $P/r02_bf16_448_kernel_stats.txt
cp $O/r02_step_traffic_fp32.json $P/r02_step_traffic_fp32.json; cp $O/r02_step_traffic_x3.json $P/r02_step_traffic_fp32x3.json; cp $O/r02_step_traffic_bf16.json $P/r02_step_traffic_bf16_448.json
cp $O/r02_step_mfma_util_fp32.txt $P/r02_step_mfma_util_fp32.txt; cp $O/r02_step_mfma_util_x3.txt $P/r02_step_mfma_util_fp32x3.txt; cp $O/r02_step_mfma_util_bf16.txt $P/r02_step_mfma_util_bf16_448.txt
tail -4 $O/r02_gputests.log > $P/r02_gputests_tail.txt
python3 - <<'PY'
import json
for n in ('fp32','fp32x3','bf16_448','bf16_224','stress','stress_fp32x3','rccl_ws1'):
    d=json.load(open(f'profiles/r02_bench_line_{n}.json'))
    x=d.get('fp32x3',{})
    print(n, d['value'], d['ms_per_step'], d.get('step_mfma_frac'), (d.get('roofline') or {}).get('frac'), x.get('value'), x.get('ms_per_step'), x.get('step_frac_of_fp32_mfma_peak'))
PY
