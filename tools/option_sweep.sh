#!/bin/bash
# V-cycle time of bench.py's default workload under context options / scheduling thresholds, each twice, alternating
# with the default so that a drift of the box shows.  usage: tools/option_sweep.sh "name=value ..." [bench.py arguments]
opts=$1; shift
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(dirname "$0")/..
for rep in 1 2; do
  for o in default $opts; do
    if [ $o = default ]; then a=""; else a="--option $o"; fi
    echo -n "[$o] "; python3 $R/tools/bench_forms.py $a "$@" | cut -c1-62
  done
done
