#!/bin/bash
# Experimental variant of the library: tools/build_variant.sh NAME [extra hipcc flags...] -> build_var/libvqa_NAME.so
# (run kernels against it with VQA_LIB=build_var/libvqa_NAME.so).  build_var/ is git-ignored but travels with gpurun.
# The fp32 engine's translation units are rebuilt with the flags (gemm.hip in its parts, as dl_vqa_amd/build.py does); the
# bf16 / fp32x3 units are taken from the in-tree build.
set -e
cd "$(dirname "$0")/.."
name=$1; shift
mkdir -p build_var/$name
CC="/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w"
for f in conv conv0 lstm elementwise; do
  $CC "$@" -c dl_vqa_amd/csrc/$f.hip -o build_var/$name/$f.o &
done
for k in 0 1 2 3 4 5; do
  $CC "$@" -DVQA_GEMM_PART=$k -c dl_vqa_amd/csrc/gemm.hip -o build_var/$name/gemm_p$k.o &
done
wait
others=$(ls dl_vqa_amd/csrc/*.o | grep -E "/(bf16|conv_bf16|conv_patch_bf16|conv_patch_f32|conv_generic|gemm_tall_bf16|conv_x3|gemm_x3(_p[0-9])?)\.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_$name.so build_var/$name/*.o $others
rm -rf build_var/$name
echo build_var/libvqa_$name.so
