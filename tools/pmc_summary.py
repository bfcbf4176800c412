"""Per-kernel averages of arbitrary rocprofv3 --pmc counters.
usage: pmc_summary.py <counter_collection.csv> [...more passes] [--min-blocks N] [--match substr]"""
import csv, sys, collections

def main():
    files = [a for a in sys.argv[1:] if not a.startswith("--")]
    min_blocks, match = 2048, "brick_sep"
    for i, a in enumerate(sys.argv):
        if a == "--min-blocks":
            min_blocks = int(sys.argv[i + 1]); files.remove(sys.argv[i + 1])
        if a == "--match":
            match = sys.argv[i + 1]; files.remove(sys.argv[i + 1])
    acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    for path in files:
        with open(path) as f:
            for r in csv.DictReader(f):
                blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
                if blocks < min_blocks or match not in r["Kernel_Name"]:
                    continue
                k = r["Kernel_Name"].split("(")[0].replace("void mgx::", "")
                a = acc[k][r["Counter_Name"]]
                a[0] += float(r["Counter_Value"]); a[1] += 1
    for k in sorted(acc):
        print(k)
        for c in sorted(acc[k]):
            s, n = acc[k][c]
            print("    %-28s %16.1f   (%d launches)" % (c, s / n, n))

if __name__ == "__main__":
    main()
