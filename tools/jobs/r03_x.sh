#!/bin/bash
# rehearsal of the world_size = 2 branches of bench.py on ONE GPU (gloo through the host): not a measurement
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
VQA_BENCH_DEVICE=0 VQA_BENCH_BACKEND=gloo timeout -k 10 500 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29531 bench.py --gpus 2 --batch 64 --steps 4 --warmup 2 --no-cpu-baseline > $O/r03_bench_ws2_gloo.json 2> $O/r03_bench_ws2_gloo.err; echo rc=$?; tail -c 1500 $O/r03_bench_ws2_gloo.json; tail -5 $O/r03_bench_ws2_gloo.err
