#!/bin/bash
# Experimental variant of the library: tools/build_variant.sh NAME [extra hipcc flags...] -> build_var/libvqa_NAME.so
# (run kernels against it with VQA_LIB=build_var/libvqa_NAME.so).  build_var/ is git-ignored but travels with gpurun.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var/$name
for f in gemm conv conv0 lstm elementwise; do
  /opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w "$@" -c dl_vqa_amd/csrc/$f.hip -o build_var/$name/$f.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_$name.so build_var/$name/*.o
rm -rf build_var/$name
echo build_var/libvqa_$name.so
