#!/bin/bash
set -e
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/emu; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf $O/kt
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o t -- python3 $R/tools/rank_emulation.py 8 128 6 > $O/log.txt 2>&1
f=$(find $O/kt -name "*kernel_trace.csv" | head -1)
python3 - "$f" <<'PY'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# find last AllReduce kernel and print the neighbourhood
idx=[i for i,r in enumerate(rows) if "AllReduce" in r["Kernel_Name"] or "ncclDevKernel" in r["Kernel_Name"] and int(r["Grid_Size_X"])>0]
ar=[i for i,r in enumerate(rows) if "AllReduce" in r["Kernel_Name"]]
print("n allreduce kernels", len(ar))
i0=ar[-2] if len(ar)>1 else ar[-1]
t0=int(rows[i0-6]["Start_Timestamp"])
for r in rows[i0-6:i0+14]:
    print("%9.1f %8.1f  %s" % ((int(r["Start_Timestamp"])-t0)/1e3,(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3, r["Kernel_Name"][:90]))
# time from allreduce start to the gather (k_pack) after graph
PY
rm -rf $O/kt
