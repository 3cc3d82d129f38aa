#!/bin/bash
# Diagnostic variant of the library with the tall-GEMM timing experiments compiled in (-DVQA_TALL_DIAG: VQA_TALL_DBG 1 = no
# epilogue, 2 = no stage traffic after the prologue, 4 = no MFMA) -> build_var/libvqa_talldiag.so.
# Use:  VQA_LIB=build_var/libvqa_talldiag.so python tools/kbench_tall.py --dbg 0,1,2,4,3,5,6
set -e
cd "$(dirname "$0")/.."
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w -DVQA_TALL_DIAG -c dl_vqa_amd/csrc/gemm_tall_bf16.hip -o build_var/gemm_tall_bf16_diag.o
others=$(ls dl_vqa_amd/csrc/*.o | grep -v "/gemm_tall_bf16.o$" | grep -v "_diag.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_talldiag.so build_var/gemm_tall_bf16_diag.o $others
rm -f build_var/gemm_tall_bf16_diag.o
echo build_var/libvqa_talldiag.so
