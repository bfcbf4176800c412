#!/bin/bash
# Builds multigrid_amd/libmgx_<tag>.so: the production objects with the fp64 macro-element kernel
# recompiled with extra flags (kernel tuning experiments; selected at run time with MGX_LIB_PATH).
# usage: tools/build_variant.sh <tag> [-DMGX_MACRO_CHUNK=7 ...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../multigrid_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result "$@" \
  -DMGX_MACRO_T=double -DMGX_MACRO_SUFFIX=f64 -DMGX_MACRO_IS_F64=1 -c mgx_macro.hip -o build/mgx_macro_f64_$tag.o
g++ -O3 -std=c++17 -fPIC -fopenmp -march=x86-64-v3 -Wall "$@" -c mgx_bricks.cpp -o build/mgx_bricks_$tag.o
objs="build/mgx_kernels.o build/mgx_brick.o build/mgx_macro_f32.o build/mgx_transfer.o build/mgx_vector.o build/mgx_api.o build/mgx_cube.o build/mgx_bricks_$tag.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmgx_$tag.so $objs build/mgx_macro_f64_$tag.o -lgomp
echo built libmgx_$tag.so
