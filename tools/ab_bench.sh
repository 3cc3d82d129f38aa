#!/bin/bash
# same-box A/B of two library builds: interleaved bench runs (overlapped and serial schedules)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in dl_vqa_amd/libvqa_hip.so $1; do
    for st in 2 1; do
      VQA_LIB=$lib VQA_STREAMS=$st timeout -k 10 120 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-x3 --stream-steps 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('$lib streams=$st', d['ms_per_step'], [round(k['avg_launch_ms'],3) for k in d['kernels'][:2]])"
    done
  done
done
