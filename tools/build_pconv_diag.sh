#!/bin/bash
# Diagnostic variant of the library with the patch-convolution timing experiments compiled in (-DVQA_PCONV_DIAG:
# VQA_PCONV_DBG 1 = no epilogue stores, 2 = no DMA after the first stage, 4 = no MFMA) -> build_var/libvqa_pconvdiag.so.
# Use:  VQA_LIB=build_var/libvqa_pconvdiag.so python tools/kbench_pconv.py --skip-old --dbg 0,1,2,3,4
# Only conv_patch_bf16.hip is rebuilt; every other object comes from the in-tree build.
set -e
cd "$(dirname "$0")/.."
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w -DVQA_PCONV_DIAG -c dl_vqa_amd/csrc/conv_patch_bf16.hip -o build_var/conv_patch_bf16_diag.o
others=$(ls dl_vqa_amd/csrc/*.o | grep -v "/conv_patch_bf16.o$" | grep -v "_diag.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_pconvdiag.so build_var/conv_patch_bf16_diag.o $others
rm -f build_var/conv_patch_bf16_diag.o
echo build_var/libvqa_pconvdiag.so
