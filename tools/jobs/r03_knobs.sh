#!/bin/bash
# the A/B knobs documented in DESIGN.md still run: module-level parity tests under each fallback setting
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; cd $R
for kv in VQA_PCONV=0 VQA_FC16=0 VQA_JOIN_DVN=0 VQA_PDGRAD=0 VQA_PDGRAD=2 VQA_TALL_BM=128 VQA_TALL_GEMM=0 VQA_STREAMS=1 VQA_STREAMS_BWD=1 VQA_PERSISTENT=1 VQA_PERSISTENT=0 VQA_LSTM16=0; do
  env $kv timeout -k 10 600 python -m pytest tests/test_model_gpu.py tests/test_bf16_gpu.py -m gpu -x -q -k "module or reference or configs3 or distinct or fp16" > $O/r03_knob_$kv.log 2>&1; rc=$?
  echo "$kv rc=$rc $(tail -1 $O/r03_knob_$kv.log)"
  [ $rc -eq 0 ] || exit 1
done
