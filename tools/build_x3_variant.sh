#!/bin/bash
# Experimental variant of the 3 x bf16 kernels only: tools/build_x3_variant.sh NAME [hipcc flags...]
# -> build_var/libvqa_NAME.so (the other translation units are the regular in-tree objects).  VQA_LIB selects it.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var
for f in conv_x3 gemm_x3; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w "$@" -c dl_vqa_amd/csrc/$f.hip -o build_var/${f}_$name.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_$name.so build_var/conv_x3_$name.o build_var/gemm_x3_$name.o \
  dl_vqa_amd/csrc/{gemm,conv,conv0,lstm,elementwise,bf16,conv_bf16}.o
rm -f build_var/conv_x3_$name.o build_var/gemm_x3_$name.o
echo build_var/libvqa_$name.so
