#!/bin/bash
# Per-kernel time of the finest-level brick kernels in both forms (macro-element / cell-by-cell):
# rocprofv3 kernel trace of tools/matvec_loop.py; summaries end up in gpurun_out/forms_<tag>/.
# usage: tools/forms_profile.sh <tag> [cells] [extra env assignments ...]
set -e
tag=$1; cells=${2:-128}; shift; shift || true
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/forms_$tag
mkdir -p $out
for form in macro cells; do
  for mode in vmult cheb; do
    ( export MGX_BRICK_FORM=$form; for kv in "$@"; do export "$kv"; done
      rocprofv3 --kernel-trace --stats --output-format csv -d $out/${form}_$mode -o t -- python3 $GRAFT_REPO_ROOT/tools/matvec_loop.py $cells 10 $mode > $out/${form}_$mode.log 2>&1 )
    f=$(find $out/${form}_$mode -name "*kernel_stats.csv" | head -1)
    echo "== $form $mode"; head -8 $f | cut -c1-200
  done
done
