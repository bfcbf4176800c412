"""Static instruction counts per kernel from the compiler's assembly: vector instructions, lane moves (v_readlane /
v_writelane = scalar operands parked in vector registers), LDS, barriers, scratch, vector memory, scalar instructions.
usage:  hipcc --offload-arch=gfx950 -O3 -std=c++17 <flags of the Makefile> --cuda-device-only -S -o out.s file.hip
        python tools/asm_stats.py out.s <kernel name fragment> [rows]        (sorted by v_readlane count)"""
import re
import subprocess
import sys

lines = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
cur, stats = None, {}
for ln in lines:
    m = re.match(r"^(_Z\w+):\s*(;.*)?$", ln)
    if m:
        cur = m.group(1)
        stats[cur] = dict(valu=0, readlane=0, writelane=0, lds=0, barrier=0, scratch=0, vmem=0, salu=0)
        continue
    if cur is None:
        continue
    s = ln.strip()
    if s.startswith("v_readlane"):
        stats[cur]["readlane"] += 1
    elif s.startswith("v_writelane"):
        stats[cur]["writelane"] += 1
    if s.startswith("v_"):
        stats[cur]["valu"] += 1
    elif s.startswith("ds_"):
        stats[cur]["lds"] += 1
    elif s.startswith("s_barrier"):
        stats[cur]["barrier"] += 1
    elif s.startswith("scratch_"):
        stats[cur]["scratch"] += 1
    elif s.startswith("global_") or s.startswith("buffer_"):
        stats[cur]["vmem"] += 1
    elif s.startswith("s_"):
        stats[cur]["salu"] += 1
names = [n for n in stats if stats[n]["valu"] > 0]
dem = subprocess.run(["c++filt"] + names, capture_output=True).stdout.decode().split("\n")
rows = []
for n, d in zip(names, dem):
    if pat in d:
        m = re.search(pat + r"<(.*?)>\(", d)
        rows.append((m.group(1) if m else d[:60], stats[n]))
for k, st in sorted(rows, key=lambda r: -r[1]["readlane"])[:int(sys.argv[3]) if len(sys.argv) > 3 else 40]:
    print(k, st)
print(len(rows), "kernels")
