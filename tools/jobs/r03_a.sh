#!/bin/bash
# round-3 checkpoint: scratch guard on the reconstructed hang variant, full GPU suite, bench lines
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
echo "== guard on the reconstructed variant (expect a clean refusal, not a hang)"
VQA_LIB=build_var/libvqa_hangrepro.so timeout -k 5 40 python tools/diag_rg.py 5408 1024 256 676 relu 2>&1 | tail -3
echo "== GPU tests"
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/r03_gputests_b.log 2>&1; echo rc=$? >> $O/r03_gputests_b.log; tail -6 $O/r03_gputests_b.log
echo "== bench"
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-x3 > $O/r03_bench_fp32_b.json 2> $O/r03_bench_fp32_b.err; head -c 250 $O/r03_bench_fp32_b.json; echo
timeout -k 10 200 python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 > $O/r03_bench_bf16_448_c.json 2> $O/r03_bench_bf16_448_c.err; head -c 250 $O/r03_bench_bf16_448_c.json; echo
