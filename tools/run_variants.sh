#!/bin/bash
# Times the finest-level brick kernels of several library variants (tools/build_variant_of.sh) on the GPU box.
# usage: tools/run_variants.sh <cells> <mode> tag1 tag2 ...   (tag "prod" = the production library)
cells=$1; mode=$2; shift; shift
cd /tmp && export TMPDIR=/tmp
for tag in "$@"; do
  out=$GRAFT_REPO_ROOT/gpurun_out/var_$tag
  rm -rf $out; mkdir -p $out
  if [ $tag = prod ]; then unset MGX_LIB_PATH; else export MGX_LIB_PATH=$GRAFT_REPO_ROOT/multigrid_amd/libmgx_$tag.so; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -o t -- python3 $GRAFT_REPO_ROOT/tools/matvec_loop.py $cells 10 $mode > $out/log.txt 2>&1
  echo "== $tag"
  python3 - $out <<'PY'
import csv, sys, re, glob
f = glob.glob(sys.argv[1] + '/**/*kernel_stats.csv', recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]:
    m = re.search(r'(brick_\w+)<(\d), (\w+), (\d+)', r['Name'])
    if m: print("  %-28s calls %5s avg %8.1f us  min %8.1f  max %8.1f" % (m.group(1) + ' mode ' + m.group(4), r['Calls'], float(r['AverageNs'])/1e3, float(r['MinNs'])/1e3, float(r['MaxNs'])/1e3))
PY
done
