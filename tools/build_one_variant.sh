#!/bin/bash
# One translation unit rebuilt with extra flags, the rest taken from the in-tree build:
#   tools/build_one_variant.sh NAME SRC.hip [OBJ_TO_REPLACE.o] [hipcc flags...]  ->  build_var/libvqa_NAME.so
# e.g.  tools/build_one_variant.sh NOSTORE_AM conv.hip conv.o -DVQA_EXP_NOSTORE_AM
#       tools/build_one_variant.sh GEMM_NOSTORE gemm.hip gemm_p2.o -DVQA_GEMM_PART=2 -DVQA_EXP_NOSTORE_ALL
# (run with VQA_LIB=build_var/libvqa_NAME.so; build_var/ is git-ignored but travels with gpurun).  Timing experiments only.
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; obj=$3; shift 3
mkdir -p build_var
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -w "$@" -c dl_vqa_amd/csrc/$src -o build_var/$name.o
others=$(ls dl_vqa_amd/csrc/*.o | grep -v "/$obj$" | grep -v "_diag.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o build_var/libvqa_$name.so build_var/$name.o $others
rm -f build_var/$name.o
echo build_var/libvqa_$name.so
