"""Reads a rocprofv3 kernel trace CSV and reports how much of the RCCL send/recv kernels' run time
falls inside brick-kernel run time (other stream).  usage: overlap_trace.py <dir>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
nccl = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "nccl" in r["Kernel_Name"].lower()]
brick = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "brick_macro_kernel" in r["Kernel_Name"])
tot = ov = 0
n_ov = 0
for a, b in nccl:
    tot += b - a
    o = sum(max(0, min(b, e) - max(a, s)) for s, e in brick if s < b and e > a)
    ov += o
    n_ov += o > 0
print("RCCL kernels: %d launches, %.1f us average; %d of them ran concurrently with a brick kernel; "
      "%.0f %% of their run time overlapped" % (len(nccl), tot / max(1, len(nccl)) / 1e3, n_ov, 100. * ov / max(1, tot)))
for a, b in nccl[len(nccl) // 2:len(nccl) // 2 + 3]:
    inside = [(s, e) for s, e in brick if s < b and e > a]
    print("  nccl kernel [%d, %d] ns (%.1f us) concurrent with brick kernels %s" %
          (a, b, (b - a) / 1e3, ["[%d, %d]" % se for se in inside[:3]]))
