#!/bin/bash
# rocprofv3 kernel trace of tools/matvec_loop.py (finest-level operator forms only), summarised per
# (kernel, grid): usage: tools/trace_forms.sh <out-file> <cells> <iterations> <vmult|cheb> [degree] [env assignments ...]
set -e
out=$(realpath -m $1); cells=$2; n=$3; mode=$4; deg=${5:-4}; shift 4; shift || true
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
d=$(mktemp -d /tmp/ktXXXX)
( for kv in "$@"; do export "$kv"; done
  rocprofv3 --kernel-trace --stats --output-format csv -d $d -o t -- python3 $R/tools/matvec_loop.py $cells $n $mode $deg > $d/log.txt 2>&1 )
python3 $R/tools/summarize_trace.py $(find $d -name "*kernel_trace.csv" | head -1) 0.0 > $out
rm -rf $d
