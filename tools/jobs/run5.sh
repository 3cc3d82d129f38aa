#!/bin/bash
# round-2 evidence run: tests, benches, rocprofv3 kernel stats and PMC passes (fp32 headline, bf16 configs[3], stress configs[4])
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
python -m pytest tests -m gpu -x -q > $O/r02_gputests.log 2>&1; echo "pytest rc=$?" >> $O/r02_gputests.log; tail -3 $O/r02_gputests.log
python bench.py --steps 20 --warmup 3 > $O/r02_bench_fp32.json 2> $O/r02_bench_fp32.err; head -c 300 $O/r02_bench_fp32.json; echo
python bench.py --dtype bf16 --batch 512 --size 448 --steps 5 --warmup 2 --no-cpu-baseline > $O/r02_bench_bf16_448.json 2> $O/r02_bench_bf16_448.err; head -c 300 $O/r02_bench_bf16_448.json; echo
python bench.py --dtype bf16 --steps 10 --warmup 3 --no-cpu-baseline > $O/r02_bench_bf16_224.json 2> $O/r02_bench_bf16_224.err; head -c 300 $O/r02_bench_bf16_224.json; echo
python bench.py --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline > $O/r02_bench_stress.json 2> $O/r02_bench_stress.err; head -c 300 $O/r02_bench_stress.json; echo
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 10 --warmup 3 --force-dist --no-cpu-baseline > $O/r02_bench_rccl_ws1.json 2> $O/r02_bench_rccl_ws1.err; head -c 300 $O/r02_bench_rccl_ws1.json; echo
cd /tmp; export TMPDIR=/tmp
rm -rf $O/r02_prof_fp32 $O/r02_prof_bf16 $O/r02_prof_serial $O/r02_pmc_*
VQA_STREAMS=1 rocprofv3 --kernel-trace --stats -d $O/r02_prof_serial -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r02_prof_serial.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r02_prof_fp32 -o p --output-format csv -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r02_prof_fp32.log 2>&1
rocprofv3 --kernel-trace --stats -d $O/r02_prof_bf16 -o p --output-format csv -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_prof_bf16.log 2>&1
for c in FETCH_SIZE WRITE_SIZE MfmaUtil; do
  rocprofv3 --pmc $c -d $O/r02_pmc_fp32_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_pmc_fp32_$c.log 2>&1
  rocprofv3 --pmc $c -d $O/r02_pmc_bf16_$c -- python3 $R/bench.py --dtype bf16 --batch 512 --size 448 --steps 2 --warmup 1 --no-cpu-baseline --stream-steps 0 > $O/r02_pmc_bf16_$c.log 2>&1
  echo "pmc $c done"
done
ls $O | grep r02_pmc | head -20
