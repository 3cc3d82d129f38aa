set -e
python -m pytest tests -m gpu -x -q -k "lstm or attention or model" > gpurun_out/r02_t4.log 2>&1 || true
tail -3 gpurun_out/r02_t4.log
for m in 1 2 1 2; do
  VQA_STREAMS=$m python bench.py --steps 10 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('VQA_STREAMS=$m', d['ms_per_step'], d['value'])"
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r02_prof3 -o p --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r02_prof3.log 2>&1
