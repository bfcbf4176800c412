#!/bin/bash
# Builds multigrid_amd/libmgx_<tag>.so: the production objects with mgx_kernels.hip (per-cell kernels:
# general / variable-coefficient branch, diagonal, DG <-> FE_Q transfer) recompiled with extra flags; selected
# at run time with MGX_LIB_PATH.   usage: tools/build_kernels_variant.sh <tag> [-DMGX_GENERAL_PREFETCH=0 ...]
set -e
tag=$1; shift
cd "$(dirname "$0")/../multigrid_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -DMGX_CELLS_FORM=0 "$@" \
  -c mgx_kernels.hip -o build/mgx_kernels_$tag.o
objs="build/mgx_brick.o build/mgx_macro_f64.o build/mgx_macro_f32.o build/mgx_transfer.o build/mgx_vector.o build/mgx_dg.o build/mgx_api.o build/mgx_cube.o build/mgx_bricks.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmgx_$tag.so $objs build/mgx_kernels_$tag.o -lgomp
echo built libmgx_$tag.so
