#!/bin/bash
# Merged DG Chebyshev step (tools/matvec_dg_cheby.py, Hermite-like basis) for several degrees and library
# variants (tools/build_variant_of.sh mgx_dg <tag> ...): DoFs/s per (variant, degree, number type).
# usage: tools/dg_variants.sh "<tags>" "<degrees>" [f32|f64] [steps]     (tag prod = the production library)
tags=$1; degs=$2; num=${3:-f32}; steps=${4:-18}
for tag in $tags; do
  if [ $tag = prod ]; then unset MGX_LIB_PATH; else export MGX_LIB_PATH=$GRAFT_REPO_ROOT/multigrid_amd/libmgx_$tag.so; fi
  for p in $degs; do
    r=$(python3 $GRAFT_REPO_ROOT/tools/matvec_dg_cheby.py $p $steps 10 --number $num --bases 0 --outer 3 2>&1 | grep "Best MF" | head -1)
    echo "$tag p=$p $num: $r"
  done
done
