python -m pytest tests/test_bf16_gpu.py -m gpu -x -q > gpurun_out/r02_t8.log 2>&1; tail -3 gpurun_out/r02_t8.log
python bench.py --dtype bf16 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bf16_224.json 2> gpurun_out/r02_bf16_224.err; head -c 600 gpurun_out/r02_bf16_224.json; echo
python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline > gpurun_out/r02_bf16_448.json 2> gpurun_out/r02_bf16_448.err; head -c 600 gpurun_out/r02_bf16_448.json; echo
tail -3 gpurun_out/r02_bf16_448.err
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof_bf16 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --dtype bf16 --batch 512 --size 448 --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof_bf16.log 2>&1
