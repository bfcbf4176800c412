#!/bin/bash
# A/B of library variants on one box: tools/ab_forms.sh "<tag> ..." [bench.py arguments]   (tag "prod" = libmgx.so)
# alternates the variants twice so that a drift of the box shows
tags=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(dirname "$0")/..
for rep in 1 2; do
  for t in $tags; do
    if [ $t = prod ]; then lib=$R/multigrid_amd/libmgx.so; else lib=$R/multigrid_amd/libmgx_$t.so; fi
    echo "== $t"; MGX_LIB_PATH=$lib python3 $R/tools/bench_forms.py "$@"
  done
done
