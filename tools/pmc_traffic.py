"""Per-kernel averages of FETCH_SIZE / WRITE_SIZE from rocprofv3 --pmc counter_collection.csv files
(one counter per pass).  usage: pmc_traffic.py <fetch.csv> <write.csv> [min_grid_blocks]
Prints bytes per launch with the gfx950 correction (FETCH_SIZE x2 for 8-B-per-lane loads, calibrated
on streaming kernels in profiles/r01e_pmc_traffic_128cube_p4.json)."""
import csv, sys, collections

def load(path, min_blocks):
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            wg = int(r.get("Workgroup_Size", 256) or 256)
            blocks = int(r["Grid_Size"]) // max(wg, 1)
            if blocks < min_blocks:
                continue
            k = (r["Kernel_Name"].split("(")[0], r["Counter_Name"])
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    return acc

def main():
    min_blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
    fe, wr = load(sys.argv[1], min_blocks), load(sys.argv[2], min_blocks)
    names = sorted({k[0] for k in fe} | {k[0] for k in wr})
    print("%-60s %8s %14s %14s %14s" % ("kernel", "launches", "fetch MB", "write MB", "total MB"))
    for n in names:
        f = fe.get((n, "FETCH_SIZE"), [0, 0]); w = wr.get((n, "WRITE_SIZE"), [0, 0])
        if not f[1] or not w[1]:
            continue
        fb = 2 * f[0] / f[1] * 1024; wb = w[0] / w[1] * 1024
        print("%-60s %8d %14.1f %14.1f %14.1f" % (n[:60], f[1], fb / 1e6, wb / 1e6, (fb + wb) / 1e6))

if __name__ == "__main__":
    main()
