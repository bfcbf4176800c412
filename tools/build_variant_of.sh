#!/bin/bash
# Builds multigrid_amd/libmgx_<tag>.so: the production objects with ONE .hip translation unit recompiled with
# extra flags (kernel experiments; selected at run time with MGX_LIB_PATH).
# usage: tools/build_variant_of.sh <mgx_dg|mgx_kernels|mgx_transfer|mgx_vector|mgx_macro_f64|mgx_macro_f32|mgx_macro2_f64|mgx_macro2_f32> <tag> [-DFLAG=... ...]
set -e
unit=$1; tag=$2; shift 2
cd "$(dirname "$0")/../multigrid_amd/csrc"
src=$unit.hip; extra=""
case $unit in
  mgx_macro_f64) src=mgx_macro.hip; extra="-DMGX_MACRO_T=double -DMGX_MACRO_SUFFIX=f64 -DMGX_MACRO_IS_F64=1" ;;
  mgx_macro_f32) src=mgx_macro.hip; extra="-DMGX_MACRO_T=float -DMGX_MACRO_SUFFIX=f32" ;;
  mgx_macro2_f64) src=mgx_macro2.hip; extra="-DMGX_MACRO_T=double -DMGX_MACRO_SUFFIX=f64 -DMGX_MACRO_IS_F64=1" ;;
  mgx_macro2_f32) src=mgx_macro2.hip; extra="-DMGX_MACRO_T=float -DMGX_MACRO_SUFFIX=f32" ;;
esac
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-result -DMGX_CELLS_FORM=0 $extra "$@" \
  -c $src -o build/${unit}_$tag.o
objs=""
for o in mgx_kernels mgx_brick mgx_macro_f64 mgx_macro_f32 mgx_macro2_f64 mgx_macro2_f32 mgx_transfer mgx_vector mgx_dg mgx_api mgx_cube mgx_bricks; do
  if [ $o = $unit ]; then objs="$objs build/${unit}_$tag.o"; else objs="$objs build/$o.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmgx_$tag.so $objs -lgomp
echo built libmgx_$tag.so
