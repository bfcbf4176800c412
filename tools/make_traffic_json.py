"""profiles/*_pmc_traffic_*.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of
tools/matvec_loop.py <cells> <n> all [degree] (every finest-level form of the brick loop: operator, residual,
Chebyshev forms, and the V-cycle's fused transfer forms).
usage: make_traffic_json.py fetch.csv write.csv cells out.json [degree=4] [calib_fetch.csv calib_write.csv]

Streaming kernels with known byte counts calibrate the counters (gfx950: FETCH_SIZE reports 1/2 for 8-B-per-lane
reads, WRITE_SIZE is exact): the two calibration files are passes of `tools/matvec_loop.py <cells> <n> calib [degree]`,
which runs copy / scaled add / dot product on finest-level vectors ONLY, so that every sampled launch has the known
size (in the `all` run the same kernels also serve the coarser levels of the V-cycle with the same grid)."""
import csv, hashlib, json, os, sys, collections

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KERNEL_SOURCES = ["multigrid_amd/csrc/mgx_macro.hip", "multigrid_amd/csrc/mgx_macro2.hip", "multigrid_amd/csrc/mgx_macro_device.hpp",
                  "multigrid_amd/csrc/mgx_brick_device.hpp", "multigrid_amd/csrc/mgx_brick.hip", "multigrid_amd/csrc/mgx_bricks.cpp"]


def kernel_source_sha():
    """fingerprint of the brick-kernel sources: bench.py attaches the measured traffic to its
    roofline only while the kernels are the ones that were measured"""
    h = hashlib.sha256()
    for f in KERNEL_SOURCES:
        h.update(open(os.path.join(ROOT, f), "rb").read())
    return h.hexdigest()[:16]


def load(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            blocks = int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1)
            acc[r["Kernel_Name"].split("(")[0]][blocks].append(float(r["Counter_Value"]))
    return acc


def main():
    fe, wr, cells, out = load(sys.argv[1]), load(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    p = int(sys.argv[5]) if len(sys.argv) > 5 else 4
    cal_fe, cal_wr = (load(sys.argv[6]), load(sys.argv[7])) if len(sys.argv) > 7 else (fe, wr)
    n = (cells * p + 1) ** 3
    res = {"command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -- python3 tools/matvec_loop.py %d 3 all %d"
                      "   (and a second pass with --pmc WRITE_SIZE)" % (cells, p),
           "workload": "poisson_cube FE_Q(%d) %d^3 cells, %d DoFs, fp64, finest level, per colour launch" % (p, cells, n),
           "units": "FETCH_SIZE / WRITE_SIZE in KiB as reported; traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 B "
                    "(gfx950: FETCH_SIZE counts 64 B per 128-B request, MI355X_MICROARCH.md 'HBM')",
           "kernel_source_sha16": kernel_source_sha(), "kernel_sources": KERNEL_SOURCES,
           "calibration": {}, "kernels": {}}
    known = {"void mgx::k_copy_cast<double, double>": (8 * n, 8 * n), "void mgx::k_xpby<double>": (16 * n, 8 * n),
             "void mgx::k_sadd<double>": (16 * n, 8 * n), "void mgx::k_dot_partial<double>": (16 * n, 0)}
    res["calibration_source"] = ("tools/matvec_loop.py %d <n> calib %d: finest-level vectors only" % (cells, p)
                                 if len(sys.argv) > 7 else "the profiled run itself (launches of several level sizes mixed)")
    for k, (rb, wb) in known.items():
        if k in cal_fe and k in cal_wr:
            big = max(cal_fe[k].keys())
            f = sum(cal_fe[k][big]) / len(cal_fe[k][big])
            w = sum(cal_wr[k][big]) / len(cal_wr[k][big])
            res["calibration"][k.replace("void mgx::", "")] = {
                "known_read_bytes": rb, "FETCH_SIZE_KiB": f, "reported/known": f * 1024 / rb,
                "known_write_bytes": wb, "WRITE_SIZE_KiB": w, "write reported/known": (w * 1024 / wb) if wb else None}
    names = {0: ("kPlain", 16), 1: ("kResidual", 24), 2: ("kCheb", 40), 3: ("kChebFirst", 32), 4: ("kChebZeroOld", 32),
             5: ("kChebInit", 24), 6: ("kChebOldInit", 32), 7: ("kResidualRestrict", 18), 9: ("kChebFirstProlong", 33)}
    for mode, (nm, alg) in names.items():
        # macro-element form (production; fused Chebyshev forms with the inverse diagonal in registers
        # when the diagonal is uniform), else the cell-by-cell form
        # (last template argument: the eight-colour schedule of the finest level, not a reduced-colour one)
        # (second pipeline, mgx_macro2.hip: plain, residual, the two start forms of the smoother, residual + restriction)
        for k in ("void mgx::brick_macro2_kernel<%d, double, %d>" % (p, mode),
                  "void mgx::brick_macro_kernel<%d, double, %d, true, false>" % (p, mode),
                  "void mgx::brick_macro_kernel<%d, double, %d, false, false>" % (p, mode),
                  "void mgx::brick_sep_kernel<%d, double, %d, false>" % (p, mode)):
            if k in fe and k in wr:
                break
        else:
            continue
        big = max(fe[k].keys())
        f = sum(fe[k][big]) / len(fe[k][big])
        w = sum(wr[k][big]) / len(wr[k][big])
        traffic = (2 * f + w) * 1024
        res["kernels"][nm] = {"kernel": k.replace("void ", ""), "blocks_per_launch": big,
                              "launches_sampled": [len(fe[k][big]), len(wr[k][big])],
                              "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "traffic_bytes_per_launch": traffic,
                              "algorithmic_bytes_per_launch": alg * n / 8.0,
                              "traffic/algorithmic": traffic / (alg * n / 8.0)}
    json.dump(res, open(out, "w"), indent=1)
    for k, v in res["kernels"].items():
        print(k, "%.1f MB" % (v["traffic_bytes_per_launch"] / 1e6), "x%.3f" % v["traffic/algorithmic"])


if __name__ == "__main__":
    main()
