#!/bin/bash
# Experimental variant of the 3 x bf16 kernels only: tools/build_x3_variant.sh NAME [hipcc flags...]
# -> build_var/libvqa_NAME.so (the other translation units are the regular in-tree objects).  VQA_LIB selects it.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var
# (the convolution kernels only: the split GEMM keeps the in-tree objects)
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w "$@" -c dl_vqa_amd/csrc/conv_x3.hip -o build_var/conv_x3_$name.o
others=$(ls dl_vqa_amd/csrc/*.o | grep -v "/conv_x3.o$" | grep -v "_diag.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_$name.so build_var/conv_x3_$name.o $others
rm -f build_var/conv_x3_$name.o
echo build_var/libvqa_$name.so
