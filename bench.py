#!/usr/bin/env python3
"""bench.py -- headline benchmark of the matrix-free multigrid Laplace path on MI355X.

Metric (BASELINE.json): DoFs/s for the fp64 Laplace matvec + V-cycle of poisson_cube, FE_Q(4).
One *step* = the two operator applications one PCG iteration of the reference performs on the
finest level (multigrid_solver.h:483-510): one fp64 `LaplaceOperator::vmult` (matrix_dp) followed
by one fp64 V-cycle (`MultigridSolver::vmult`: Chebyshev(3) pre/post smoothing, residual,
restriction, coarse solve, prolongation).  `value` = global DoFs / time per step; the matvec-only
and V-cycle-only rates are reported next to it (`matvec_dofs_per_s`, `vcycle_dofs_per_s`).

N = 1: BASELINE config 1 -- 128^3 cells, 135 005 697 DoFs, one MI355X, inputs resident in HBM.
N > 1: domain decomposition, one process per GPU over RCCL (torch.distributed "nccl"), interface
       DoFs exchanged and summed once per operator application.
       --scaling strong (default): the SAME 128^3-cell mesh, block-split 2x1x1 / 2x2x1 / 2x2x2
       (n_subdiv = 2 coarse cells per direction so that every level splits evenly:
       the finest level is identical to N = 1, the hierarchy is one level shorter).
       --scaling weak: the reference's "doubling" family (program.cc:509-529), one 128^3-cell
       coarse cube per rank.   --replicas: N independent copies.
`python bench.py --gpus N` starts the N ranks itself (fresh child processes with RANK / LOCAL_RANK /
WORLD_SIZE / MASTER_ADDR / MASTER_PORT set) and relays rank 0's JSON line; under
torch.distributed.run the ranks are already there and --gpus must equal WORLD_SIZE.

Usage: python bench.py [--gpus N] [--steps K] [--warmup W] [--cells 128] [--degree 4]
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FP64_PEAK_TFLOPS = 78.6   # MI355X fp64 vector peak (datasheet); tools/microbench.hip sustains 60.5
# executed by the macro-element brick kernel at p = 4: 289 lines x 628 fp64 instructions (ISA count of
# the three sweeps: 158 + 279 + 191, ~80 % of them FMA = 2 flop) per 4096-DoF brick (dense 12-sweep form: 270)
FLOP_PER_DOF_P4 = 80.0
TRAFFIC_FILES = {(4, 128): "r04_pmc_traffic_128cube_p4.json", (8, 64): "r04_pmc_traffic_64cube_p8.json"}
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy rate)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--cells", type=int, default=128, help="cells per direction of the finest level")
    ap.add_argument("--degree", type=int, default=4)
    ap.add_argument("--smoother-degree", type=int, default=3)
    ap.add_argument("--vcycle-number", choices=["f64", "f32"], default="f64")
    ap.add_argument("--smoother-polynomial", choices=["first_kind", "fourth_kind", "reference"], default="first_kind",
                    help="first_kind: the smoother SURVEY.md 8d defines the metric with (multigrid_solver.h:277-278, the "
                         "README run); reference: what the reference instantiates for the V-cycle number type -- "
                         "fourth_kind for MultigridSolver<dim,p,double,double> (:951-952), first_kind for fp32; the work "
                         "per V-cycle is the same")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the end-to-end check after the timed region (PCG iteration count and L2 error of the "
                         "manufactured problem, README.md:135-159)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true",
                    help="skip the secondary configurations measured after the headline (N = 1, default workload only): "
                         "BASELINE configs 3-5 and the reference's default fp32 V-cycle, each with its own time, roofline "
                         "fraction and check")
    ap.add_argument("--option", action="append", default=[], metavar="NAME=VALUE",
                    help="context option for A/B timings of numerically equivalent code paths (include/mgx.h, "
                         "mgx_context_set_option); recorded in the JSON line")
    ap.add_argument("--host-rhs", action="store_true",
                    help="assemble the right-hand sides on the host instead of on the GPU (set-up only: not in the timed region)")
    ap.add_argument("--replicas", action="store_true", help="N>1: independent replicas instead of domain decomposition")
    ap.add_argument("--scaling", choices=["strong", "weak"], default="strong",
                    help="N>1: split the --cells^3 mesh over the ranks (strong) or give every rank a --cells^3 cube (weak)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launch check without a GPU: ranks rendezvous (gloo), build their part of the mesh on the host "
                         "and exchange one interface vector; prints the JSON line with value null")
    ap.add_argument("--cpu-cells", type=int, default=0,
                    help="finest level of the CPU baseline sample (0: the bench mesh itself if the host has the memory "
                         "and the cores for it, else 64)")
    return ap.parse_args()


def split_size(cells):
    """poisson_cube/program.cc:532-539: strip factors of two -> (n_subdiv, n_refine)"""
    n_subdiv, n_refine = cells, 0
    if n_subdiv > 1:
        while n_subdiv % 2 == 0:
            n_refine += 1
            n_subdiv //= 2
    return n_subdiv, n_refine


def cpu_baseline(args):
    """The oracle (CPU restatement, kind "port") timed on the host cores on a bounded sample of
    the same workload: same element/algorithm on a smaller finest level (DoFs/s of a matrix-free
    operator is size independent once out of cache)."""
    from oracle import Oracle
    cpu_cells = args.cpu_cells
    if cpu_cells == 0:
        # the bench mesh itself (about 25 fp64 vectors of 1.08 GB at 128^3 cells, p = 4) where the host can hold it
        # and has the cores to finish the sample within a minute; otherwise one level coarser
        avail = 0
        try:
            avail = [int(l.split()[1]) for l in open("/proc/meminfo") if l.startswith("MemAvailable")][0] * 1024
        except (OSError, IndexError, ValueError):
            pass
        big = args.cells <= 128 and avail > 160e9 and (os.cpu_count() or 1) >= 16
        cpu_cells = args.cells if big else min(args.cells, 64)
    ns, nr = split_size(cpu_cells)
    t0 = time.time()
    orc = Oracle(args.degree, ns, nr, degree=args.smoother_degree, n_cycles=1, vfloat=False)
    n = orc.n_dofs(orc.max_level)
    orc.time_vmult(orc.max_level, 1)  # warm-up
    n_mv = 40 if cpu_cells <= 64 else 10
    t_mv = orc.time_vmult(orc.max_level, n_mv) / n_mv
    orc.time_vcycle(1)
    n_vc = 16 if cpu_cells <= 64 else 3
    t_vc = orc.time_vcycle(n_vc) / n_vc
    threads = orc.num_threads()
    # one core on its own (the per-core rate of the threaded run includes what the cores cost each other)
    orc.set_num_threads(1)
    n_one = 2 if cpu_cells > 64 else 8
    t_one = orc.time_vmult(orc.max_level, n_one) / n_one
    orc.set_num_threads(0)
    orc.close()
    return {
        "value": n / (t_mv + t_vc), "unit": "DoFs/s", "cores": threads, "kind": "port",
        "sample": "FE_Q(%d) %d^3 cells (%d DoFs): %d fp64 matvecs + %d fp64 V-cycles of the oracle "
                  "(oracle/mg_oracle.c: OpenMP over cells, SIMD over batches of 8 cells), %.1f s incl. setup"
                  % (args.degree, cpu_cells, n, n_mv, n_vc, time.time() - t0),
        "matvec_dofs_per_s": n / t_mv, "vcycle_dofs_per_s": n / t_vc,
        # per core, next to the only genuine deal.II figures there are (README.md:127, 12 Broadwell cores,
        # AVX2-vectorised over cells with even-odd sweeps; the oracle batches 8 cells per SIMD lane group with dense sweeps)
        "matvec_dofs_per_s_per_core": n / t_mv / max(1, threads),
        "matvec_dofs_per_s_one_core_alone": n / t_one,
        "reference_readme_12c_broadwell": {"matvec_dofs_per_s": 8.74e8, "matvec_dofs_per_s_per_core": 8.74e8 / 12,
                                           "vcycle_mixed_precision_dofs_per_s": 9.7e7},
    }


# README.md:135-159 (mixed precision, 2 V-cycles per level, deal.II 9.1): cells per direction -> (PCG iterations,
# L2 error after PCG) of poisson_cube at p = 4; the iteration count also holds for the fp64 V-cycle
README_P4 = {8: (8, 3.822e-4), 16: (8, 1.319e-5), 32: (8, 4.220e-7), 64: (8, 1.327e-8), 128: (8, 4.207e-10)}


def verify(args, solver, cube, world, decomposed, transport, native):
    """End-to-end check of the run that was just timed (in particular of a multi-GPU run, whose first bytes over
    xGMI move inside this script): the V-cycle-preconditioned CG of the program (multigrid_solver.h:483-493) on
    the manufactured problem must converge in the README's number of iterations to the README's L2 error.  Every
    rank runs it (the solve is collective); a wrong answer makes the run fail."""
    t0 = time.time()
    its, red = solver.solve_cg()
    hist = solver.cg_history()
    l2 = solver.compute_l2_error()
    strong = not decomposed or args.scaling == "strong"
    expect = README_P4.get(args.cells) if (args.degree == 4 and strong) else None
    # (CG residual norms need not fall monotonically: no increase beyond 10x the previous entry)
    ok = bool(its <= 12 and hist[-1] <= 1e-9 * hist[0] and all(hist[i + 1] < 10. * hist[i] for i in range(len(hist) - 1)))
    if expect:
        ok = ok and its == expect[0] and abs(l2 - expect[1]) <= 0.03 * expect[1]
    res = {"cg_its": its, "cg_reduction_rate": red, "l2_error": l2, "ok": ok,
           "expected": {"cg_its": expect[0], "l2_error": expect[1], "source": "README.md:135-159"} if expect else None,
           "rccl_ranks": world if decomposed else 1, "transport": transport, "seconds": time.time() - t0}
    if not ok:
        sys.stderr.write("bench.py: verification FAILED: %s\n" % json.dumps(res))
    return res



def secondary(mg, ctx, cube128):
    """BASELINE configs 3-5 and the reference's default number type, driver-timed next to the headline (after its timed
    region, verification and CPU baseline: the headline numbers do not depend on their presence).  One GPU, inputs resident
    in HBM, each entry with its own `ms`, `roofline` (algorithmic bytes of SURVEY.md 8d over the time, against 8 TB/s) and
    `verify`.  cube128: the headline's mesh, reused by the fp32-V-cycle entry."""
    import numpy as np
    out = {}

    def timed(fn, reps, warm=2):
        for _ in range(warm):
            fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    def roof(bytes_per_dof, n, seconds):
        ach = bytes_per_dof * n / seconds / 1e9
        return {"bound": "hbm", "algorithmic_bytes_per_dof": bytes_per_dof, "achieved": ach, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": ach / HBM_PEAK_GBS}

    def cube_entry(cube, degree, cells, vnum, expect):
        t0 = time.time()
        solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, vnum, polynomial="first_kind", device_rhs=True)
        l = cube.max_level
        n = cube.n_dofs(l)
        x, y, z = ctx.vector(n, data=cube.seeded_vector(l, 42)), ctx.vector(n), ctx.vector(n)
        rhs = solver.get_vector(l, "rhs")
        A = solver.matrix_dp(l)
        setup = time.time() - t0
        solver.matrix(l).set_profiled(True)
        ctx.profile_enable(True)
        t_step = timed(lambda: (A.vmult(y, x), solver.vmult(z, rhs)), 5)
        launches, ms = ctx.profile_read(2)  # fused Chebyshev iteration of the finest level, HIP events per application
        ctx.profile_enable(False)
        t_mv = timed(lambda: A.vmult(y, x), 10)
        t_vc = timed(lambda: solver.vmult(z, rhs), 5)
        its, red = solver.solve_cg()
        hist = solver.cg_history()
        l2 = solver.compute_l2_error()
        ok = bool(its <= 12 and hist[-1] <= 1e-9 * hist[0] and its == expect[0] and abs(l2 - expect[1]) <= expect[2] * expect[1])
        nb = 8.0 if vnum == mg.F64 else 4.0
        res = {"workload": "poisson_cube FE_Q(%d) %d^3 cells, %d DoFs, step = 1 fp64 vmult + 1 %s V-cycle (Chebyshev degree 3, "
                           "first kind)" % (degree, cells, n, "fp64" if vnum == mg.F64 else "fp32"),
               "ms": 1e3 * t_step, "dofs_per_s": n / t_step, "matvec_ms": 1e3 * t_mv, "vcycle_ms": 1e3 * t_vc, "setup_s": setup,
               # the V-cycle against the fully fused model of SURVEY.md 8d, the fused Chebyshev launch against its 5 accesses
               "roofline": roof((10.0 * 3 + 4.25) * nb * 8.0 / 7.0, n, t_vc),
               "roofline_cheb": (dict(roof(5 * nb, n / 8, ms / launches * 1e-3), launches=launches, avg_launch_ms=ms / launches)
                                 if launches else None),
               "roofline_matvec": roof(16.0, n, t_mv),
               "verify": {"cg_its": its, "l2_error": l2, "ok": ok,
                          "expected": {"cg_its": expect[0], "l2_error": expect[1], "source": expect[3]}}}
        for v in (x, y, z):
            v.free()
        solver.close()
        return res

    # C3: p = 8, the same 135 M DoFs (expected: this repository's own round-3 run, profiles/r03_bench_64cube_p8.json -- the
    # reference holds no number for p = 8)
    c8 = mg.Cube(8, 1, 6)
    out["poisson_cube_p8"] = cube_entry(c8, 8, 64, mg.F64, (8, 8.76e-11, 0.05, "profiles/r03_bench_64cube_p8.json"))
    c8.close()
    # the reference's default number types: fp64 CG around an fp32 V-cycle (poisson_cube/program.cc:76-77), README row
    out["poisson_cube_p4_f32_vcycle"] = cube_entry(cube128, 4, 128, mg.F32, (8, 4.207e-10, 0.03, "README.md:159"))

    # config 4: hyper_shell(6), variable coefficient, per-point tensor (poisson_shell/program.cc:159-169, 425-431)
    t0 = time.time()
    shell = mg.Cube(4, n_refine=5, shell=6, problem="shell")
    l = shell.max_level
    n = shell.n_dofs(l)
    solver = mg.MultigridSolver(ctx, shell, 3, 3, 1, mg.F64)
    A = solver.matrix_dp(l)
    x, y, z = ctx.vector(n, data=shell.seeded_vector(l, 1)), ctx.vector(n), ctx.vector(n)
    rhs = solver.get_vector(l, "rhs")
    setup = time.time() - t0
    t_mv = timed(lambda: A.vmult(y, x), 20, warm=3)
    t_vc = timed(lambda: solver.vmult(z, rhs), 5)
    its, red = solver.solve_cg()
    hist = solver.cg_history()
    l2 = solver.compute_l2_error()
    # the reference publishes nothing for this program: the check is convergence of the V-cycle-preconditioned CG to the
    # tolerance and a discretisation error of the size this repository measured before (profiles/r03_poisson_shell_p4.txt)
    ok = bool(its <= 40 and hist[-1] <= 1e-9 * hist[0] and l2 < 1e-6)
    out["poisson_shell_p4"] = {
        "workload": "poisson_shell FE_Q(4) on hyper_shell(6) refined 5 times, %d cells, %d DoFs, coefficient 1 + 1e6 prod cos^2: "
                    "fp64 vmult (general tensor branch) and fp64 V-cycle" % (shell.n_cells(l), n),
        "ms": 1e3 * t_mv, "dofs_per_s": n / t_mv, "vcycle_ms": 1e3 * t_vc, "setup_s": setup,
        "roofline": roof(16.0 + 48.0 * (5.0 / 4.0) ** 3, n, t_mv),
        "verify": {"cg_its": its, "l2_error": l2, "ok": ok,
                   "expected": {"cg_its": "<= 40", "l2_error": "< 1e-6", "source": "profiles/r03_poisson_shell_p4.txt"}}}
    for v in (x, y, z):
        v.free()
    solver.close()
    shell.close()

    # config 5: DG-SIP matvec merged with one Chebyshev update, fp32, Hermite-like basis (matvec_dg_cheby/program.cc)
    rng = np.random.default_rng(0)
    for degree, steps in ((4, 18), (8, 15)):
        t0 = time.time()
        cells, jac = mg.dg_cheby_mesh(steps)
        nb, _ = mg.dg_box_neighbours(cells)
        op = mg.DGLaplaceOperator(ctx, degree, 0, nb, jac, mg.F32, 0, None)
        n = int(np.prod(cells)) * (degree + 1) ** 3
        r_h, x_h, xo_h = (rng.random(n).astype(np.float32) for _ in range(3))
        rhs, sol, old = (op.initialize_dof_vector(v) for v in (r_h, x_h, xo_h))
        setup = time.time() - t0
        # check: the merged kernel against its unmerged parts on the same inputs (operator, block-Jacobi, update as three
        # kernels + host arithmetic), laplace_operator_dg.h:910-955: x_new = x + f1 (x - x_old) + f2 P^-1 (rhs - A x)
        ax, tmp = op.initialize_dof_vector(), op.initialize_dof_vector()
        op.vmult(ax, sol)
        tmp.upload((r_h.astype(np.float64) - ax.download().astype(np.float64)).astype(np.float32))
        op.jacobi_vmult(ax, tmp)
        ref = x_h.astype(np.float64) + 0.6 * (x_h.astype(np.float64) - xo_h) + 0.2 * ax.download().astype(np.float64)
        op.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, sol, old)
        err = float(np.abs(sol.download().astype(np.float64) - ref).max() / np.abs(ref).max())

        def stepfn():
            nonlocal sol, old
            op.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, sol, old)
            sol, old = old, sol
        t_step = min(timed(stepfn, 20, warm=3) for _ in range(3))
        out["matvec_dg_cheby_p%d_f32" % degree] = {
            "workload": "matvec_dg_cheby FE_DGQHermite(%d), %d x %d x %d cells, %d DoFs, fp32: DG-SIP vmult merged with the "
                        "Chebyshev update (block Jacobi in the eigenvector basis)" % (degree, *cells, n),
            "ms": 1e3 * t_step, "dofs_per_s": n / t_step, "setup_s": setup,
            # the reference's own model (matvec_dg_cheby/program.cc:178): 5 accesses per DoF; the kernel moves 4
            "roofline": dict(roof(5 * 4.0, n, t_step), frac_at_the_4_accesses_moved=4 * 4.0 * n / t_step / 1e9 / HBM_PEAK_GBS),
            "verify": {"merged_vs_unmerged_rel_max": err, "ok": bool(err < 2e-5),
                       "expected": {"merged_vs_unmerged_rel_max": "< 2e-5 (fp32)",
                                    "source": "the merged step against operator + JacobiTransformed + update as separate "
                                              "kernels, laplace_operator_dg.h:910-955"}}}
        for v in (rhs, sol, old, ax, tmp):
            v.free()
        op.clear()
    return out


def spawn(args):
    """`python bench.py --gpus N` outside a launcher: start N fresh child processes (this parent never
    touches the GPU and never replaces itself), give each its rank environment, relay rank 0's JSON
    line, fail if any rank fails."""
    import subprocess
    import tempfile
    # The ranks meet through a file (torch.distributed "file://" rendezvous, MGX_BENCH_INIT_FILE): no port is picked by
    # a process that does not keep it (a socket bound and closed here could be taken by someone else before rank 0 binds
    # it).  MASTER_ADDR / MASTER_PORT are set for code that looks at them, but nothing listens there.
    rendezvous = os.path.join(tempfile.mkdtemp(prefix="mgx_bench_"), "rendezvous")
    one_gpu = os.environ.get("MGX_BENCH_BACKEND", "nccl") != "nccl" or args.dry_run
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK="0" if one_gpu else str(r), WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT="29500", MGX_BENCH_INIT_FILE=rendezvous)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # host-side table construction is OpenMP code: every rank gets its share of the cores
        env.setdefault("OMP_NUM_THREADS", str(max(1, min(16, (os.cpu_count() or 8) // args.gpus))))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # a rank that dies leaves the others waiting in a collective: once one has failed, the rest get
    # 30 s to finish and are then ended (exactly the processes started here), so the run fails instead of hanging
    import time
    failed_at, out = None, None
    while True:
        if procs[0].poll() is None or out is None:
            try:
                out, _ = procs[0].communicate(timeout=1.0)  # drains the pipe while waiting; a timed-out call loses nothing
            except subprocess.TimeoutExpired:
                pass
        else:
            time.sleep(0.25)  # rank 0 is done: wait for the others without spinning
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if failed_at is None and any(c not in (None, 0) for c in codes):
            failed_at = time.time()
        if failed_at is not None and time.time() - failed_at > 30.0:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            for p in procs:
                try:
                    p.wait(timeout=10.0)
                except subprocess.TimeoutExpired:
                    p.kill()
            break
    codes = [p.wait() for p in procs]
    import shutil
    shutil.rmtree(os.path.dirname(rendezvous), ignore_errors=True)
    if out is None:
        out, _ = procs[0].communicate()
    sys.stdout.write(out.decode())
    sys.stdout.flush()
    if any(codes):
        raise SystemExit("bench.py: rank exit codes %s" % codes)


def dry_run(args, rank, world, dist):
    """No GPU: decomposition tables on the host + one interface exchange over gloo, checked against
    the multiplicity of every interface DoF."""
    import numpy as np
    import torch
    import multigrid_amd as mg
    procs = mg.process_grid(world)
    ns, nr = split_size(args.cells)
    if world > 1 and args.scaling == "strong":
        cube = mg.Cube(args.degree, n_refine=nr - 1, box=(2, 2, 2), procs=procs, rank=rank, origin=-0.9, h0=0.95)
    elif world > 1:
        cube = mg.Cube(args.degree, n_refine=nr, box=procs, procs=procs, rank=rank)
    else:
        cube = mg.Cube(args.degree, ns, nr)
    l = cube.max_level
    ok = True
    if world > 1:
        v = np.ones(cube.n_dofs(l))
        ops, recvs = [], []
        for (rk, idx) in cube.neighbors(l):
            st, rt = torch.ones(idx.size, dtype=torch.float64), torch.empty(idx.size, dtype=torch.float64)
            recvs.append((idx, rt))
            ops += [dist.P2POp(dist.isend, st, rk), dist.P2POp(dist.irecv, rt, rk)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        for idx, rt in recvs:
            v[idx] += rt.numpy()
        # every interface DoF now counts the ranks that share it: 2 on faces, 4 on edges, 8 at corners
        shared = cube.shared(l)
        ok = bool(np.isin(v[shared], (2., 4., 8.)).all() and (np.delete(v, shared) == 1.).all())
        t = torch.tensor([1 if ok else 0])
        dist.all_reduce(t, op=dist.ReduceOp.MIN)
        ok = bool(t.item())
    g = np.array(cube.cells_per_dim3(l)[1], dtype=np.int64) * args.degree + 1
    if rank == 0:
        print(json.dumps({"metric": "DoFs/s for Laplace matvec + V-cycle, poisson_cube p=%d fp64" % args.degree,
                          "value": None, "unit": "DoFs/s", "n_gpus": world, "dry_run": True, "exchange_ok": ok,
                          "scaling": args.scaling if world > 1 else "weak",
                          # what the real run checks after its timed region (verify()), unless --no-verify
                          "verify": None if args.no_verify else
                          {"expected": ({"cg_its": README_P4[args.cells][0], "l2_error": README_P4[args.cells][1]}
                                        if args.degree == 4 and args.cells in README_P4 and
                                        (world == 1 or args.scaling == "strong") else None), "ok": None},
                          "config": {"global_dofs": int(g.prod()), "n_dofs_per_gpu": cube.n_dofs(l),
                                     "process_grid": list(procs)}}))
    cube.close()
    if not ok:
        raise SystemExit("bench.py --dry-run: interface exchange check failed")


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (start it as `python bench.py --gpus N`, or under "
                         "torch.distributed.run with --nproc-per-node equal to --gpus)" % (args.gpus, world))
    dist = None
    # started by spawn(): file rendezvous; under torch.distributed.run: the launcher's environment
    init_file = os.environ.get("MGX_BENCH_INIT_FILE")
    init_kw = dict(init_method="file://" + init_file, rank=rank, world_size=world) if init_file else {}
    if args.dry_run:
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo", **init_kw)
        dry_run(args, rank, world, dist)
        if dist is not None:
            dist.destroy_process_group()
        return
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")  # "gloo": functional test on one GPU
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank), **init_kw)
        else:
            local_rank = 0
            dist.init_process_group(backend, **init_kw)

    import numpy as np

    if world > 1 and os.environ.get("OMP_NUM_THREADS", "1") == "1":
        # torch.distributed.run pins OMP_NUM_THREADS=1; the host-side table construction (mesh
        # tables, brick schedule, rhs assembly) is OpenMP code: give every rank its share of the cores
        os.environ["OMP_NUM_THREADS"] = str(max(1, min(16, (os.cpu_count() or 8) // world)))

    import multigrid_amd as mg

    ctx = mg.Context(local_rank, options={k: float(v) for k, v in (o.split("=") for o in args.option)})
    ns, nr = split_size(args.cells)
    t_setup = time.time()
    vnum = mg.F64 if args.vcycle_number == "f64" else mg.F32
    decomposed = world > 1 and not args.replicas
    native = False
    polynomial = args.smoother_polynomial
    if polynomial == "reference":
        polynomial = "fourth_kind" if vnum == mg.F64 else "first_kind"
    if decomposed:
        if ns != 1 or nr < 1:
            raise SystemExit("--cells must be a power of two (>= 2) for N > 1")
        procs = mg.process_grid(world)
        comm = mg.Communicator(ctx, dist)
        if args.scaling == "strong":
            # the square mesh [-0.9,1]^3 with n_subdiv = 2, block-split: every rank owns
            # (2/procs[d]) coarse cells per direction on every level
            cube = mg.Cube(args.degree, n_refine=nr - 1, box=(2, 2, 2), procs=procs, rank=rank, origin=-0.9, h0=0.95)
        else:
            cube = mg.Cube(args.degree, n_refine=nr, box=procs, procs=procs, rank=rank)
        solver = mg.MultigridSolver(ctx, cube, args.smoother_degree, args.smoother_degree, 1, vnum, comm=comm,
                                    polynomial=polynomial, device_rhs=not args.host_rhs)
        # RCCL send/recv issued by the library on its own stream (no host round trip per exchange),
        # switched on only after one exchange + one reduction agree bitwise with the torch transport
        native = comm.verify_and_enable_native(solver.matrix_dp(cube.max_level), cube.n_dofs(cube.max_level))
        global_dofs = 1
        for d in range(3):
            global_dofs *= (1 if args.scaling == "strong" else procs[d]) * args.cells * args.degree + 1
    else:
        procs = (1, 1, 1)
        cube = mg.Cube(args.degree, ns, nr)
        solver = mg.MultigridSolver(ctx, cube, args.smoother_degree, args.smoother_degree, 1, vnum, polynomial=polynomial,
                                    device_rhs=not args.host_rhs)
    lmax = cube.max_level
    n_dofs = cube.n_dofs(lmax)
    if not decomposed:
        global_dofs = n_dofs * world
    # inputs resident in HBM: seeded vector for the matvec, rhs as V-cycle defect
    x = ctx.vector(n_dofs, data=cube.seeded_vector(lmax, 42))
    y = ctx.vector(n_dofs)
    rhs = solver.get_vector(lmax, "rhs")
    z = ctx.vector(n_dofs)
    A = solver.matrix_dp(lmax)
    t_setup = time.time() - t_setup

    def barrier():
        ctx.sync()
        if dist is not None:
            import torch
            if dist.get_backend() == "nccl":
                torch.cuda.synchronize()
            dist.barrier()

    def step():
        A.vmult(y, x)          # LaplaceOperator::vmult, fp64
        solver.vmult(z, rhs)   # MultigridSolver::vmult = one V-cycle

    for _ in range(args.warmup):
        step()
    # ---- the timed region: exactly K steps, bracketed by barrier + synchronize ----
    A.set_profiled(True)
    solver.matrix(lmax).set_profiled(True)
    ctx.profile_enable(True)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    prof = {form: ctx.profile_read(form) for form in range(10)}
    ctx.profile_enable(False)
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # secondary figures (outside the timed region): matvec-only and V-cycle-only
    def timed(fn, reps):
        fn()
        ctx.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / reps

    t_mv = timed(lambda: A.vmult(y, x), 20)
    t_vc = timed(lambda: solver.vmult(z, rhs), 5)

    ms_per_step = 1e3 * elapsed / args.steps
    total_dofs = global_dofs
    # Roofline of the dominant kernel.  A step runs 7 finest-level cell loops: the plain matvec
    # (form 0, 16 B/DoF algorithmic), the residual fused with the restriction (form 7:
    # x, b read, coarse sums read+written = 16 + 2 B/DoF; form 1 is the plain residual), and 5 Chebyshev
    # iterations -- post-smoothing: first step (form 3) + 2 full iterations (form 2: x, x_old, b,
    # D^-1 read + x_new written = 5 accesses = 40 B/DoF, the reference's own 5-access model,
    # matvec_dg_cheby/program.cc:178); pre-smoothing from a zero guess: forms 5 and 6, which
    # recompute x_1 = D^-1 b / theta instead of storing it.  Form 2 has the most launches and the
    # most traffic.  One application = n_colours launches, each over n_dofs / n_colours DoFs.
    # form 9 (first post-smoothing step with the prolongation fused in): x, b and 1/8 coarse value read, the
    # corrected x and x_new written = 33 B/DoF
    ALG = {0: 16.0, 1: 24.0, 2: 40.0, 3: 32.0, 4: 32.0, 5: 24.0, 6: 32.0, 7: 18.0, 9: 33.0}
    NAMES = {0: "kPlain", 1: "kResidual", 2: "kCheb", 3: "kChebFirst", 4: "kChebZeroOld", 5: "kChebInit",
             6: "kChebOldInit", 7: "kResidualRestrict", 9: "kChebFirstProlong"}
    if vnum != mg.F64:
        ALG = {k: (v / 2 if k != 0 else v) for k, v in ALG.items()}  # the V-cycle forms run in fp32, the matvec in fp64

    def roof(form):
        launches, ms = prof[form]
        if not launches:
            return None
        n_col = 8
        avg = ms / launches
        # HBM traffic from the PMC counters is collected in separate rocprofv3 --pmc passes (they
        # cannot run inside this timed process) and committed under profiles/; it is reported here
        # for the configuration it was measured on
        traffic, traffic_note = None, None
        TRAFFIC_FILE = TRAFFIC_FILES.get((args.degree, args.cells), "")
        pmc_file = os.path.join(ROOT, "profiles", TRAFFIC_FILE)
        if TRAFFIC_FILE and vnum == mg.F64 and world == 1 and os.path.exists(pmc_file):
            pmc = json.load(open(pmc_file))
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            from make_traffic_json import kernel_source_sha
            k = pmc["kernels"].get(NAMES[form])
            if pmc.get("kernel_source_sha16") != kernel_source_sha():
                traffic_note = "profiles/%s was measured on other kernel sources: not attached" % TRAFFIC_FILE
            elif k:
                traffic = k["traffic_bytes_per_launch"]
        per_launch_bytes = ALG[form] * n_dofs / n_col
        ach = per_launch_bytes / (avg * 1e-3) / 1e9
        return {"bound": "hbm", "kernel": "mgx::brick_macro%s_kernel<%d,%s,%s> (finest level, per colour launch)"
                % ("2" if (form in (0, 1, 6) or (form in (5, 7) and args.degree <= 4)) and not any(o.startswith("no_macro_v2") for o in args.option) else "", args.degree, "double" if (vnum == mg.F64 or form == 0) else "float", NAMES[form]), "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": traffic,
                "traffic_source": ("profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, FETCH_SIZE x2 "
                                   "per the gfx950 correction; same kernel sources as this build)" % TRAFFIC_FILE)
                if traffic else traffic_note,
                "launches": launches, "avg_launch_ms": avg,
                "algorithmic_bytes_per_launch": per_launch_bytes, "algorithmic_bytes_per_dof": ALG[form],
                # co-limiter asked for by SURVEY.md 8d: fp64 vector throughput of the cell loop.  Executed
                # flops of the separable even-odd form at p = 4 (ISA count of the round loop: 159 fp64
                # VALU instructions per lane and round, ~110 of them FMA, 25 active lanes per cell)
                "fp64_valu": ({"flop_per_dof": FLOP_PER_DOF_P4, "achieved_tflops": FLOP_PER_DOF_P4 * (n_dofs / n_col)
                               / (avg * 1e-3) / 1e12, "peak_tflops": FP64_PEAK_TFLOPS,
                               "frac": FLOP_PER_DOF_P4 * (n_dofs / n_col) / (avg * 1e-3) / 1e12 / FP64_PEAK_TFLOPS}
                              if args.degree == 4 else None)}
    transport = None
    if decomposed:
        transport = ("RCCL ncclSend/ncclRecv issued by the library on the solver stream" if native else
                     "torch.distributed (%s) point-to-point batches" % dist.get_backend())
    out = {
        "metric": "DoFs/s for Laplace matvec + V-cycle, poisson_cube p=%d fp64" % args.degree,
        "value": total_dofs / (elapsed / args.steps),
        "unit": "DoFs/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": args.scaling if decomposed else "weak", "vs_baseline": None,
        "dtype": "f64" if vnum == mg.F64 else "f64 outer / f32 V-cycle", "data": "synthetic",
        "config": {"workload": "poisson_cube FE_Q(%d) %d^3 cells%s, %d DoFs per GPU, %d levels, "
                               "step = 1 fp64 vmult + 1 V-cycle (Chebyshev degree %d, %s)" %
                               (args.degree, args.cells, " in total" if decomposed and args.scaling == "strong" else "", n_dofs,
                                cube.n_levels, args.smoother_degree, polynomial.replace("_", " ")),
                   "cells_per_dim": args.cells, "degree": args.degree, "n_dofs_per_gpu": n_dofs,
                   "global_dofs": total_dofs,
                   "parallelism": "1 GPU" if world == 1 else
                   ("domain decomposition %dx%dx%d, %s, interface exchange over %s" %
                    (procs + (("the %d^3-cell mesh block-split over the GPUs" if args.scaling == "strong" else
                               "one %d^3-cell cube per GPU") % args.cells, transport)) if decomposed
                    else "%d independent replicas" % world),
                   "transport": transport if decomposed else None},
        "matvec_dofs_per_s": total_dofs / t_mv, "vcycle_dofs_per_s": total_dofs / t_vc,
        "matvec_ms": 1e3 * t_mv, "vcycle_ms": 1e3 * t_vc, "setup_s": t_setup,
        "rhs_assembly": "host" if args.host_rhs else "device", "options": args.option or None,
        "roofline": roof(2) or roof(0),
        "roofline_matvec": roof(0),
        # the other finest-level forms of the step (HIP events around every application, as above)
        "roofline_forms": {NAMES[f]: {k: r[k] for k in ("achieved", "frac", "avg_launch_ms", "launches",
                                                         "algorithmic_bytes_per_dof", "traffic")}
                           for f in (1, 3, 5, 6, 7, 9) for r in [roof(f)] if r},
        # the V-cycle as a whole against the fully fused model of SURVEY.md 8d: (10 n + 4.25) accesses per level DoF
        # x 8/7 (level sum) = 313 B per fine DoF at n = 3 in fp64
        "roofline_vcycle": (lambda b: {"bound": "hbm", "algorithmic_bytes_per_fine_dof": b,
                                       "achieved": b * n_dofs / t_vc / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                       "frac": b * n_dofs / t_vc / 1e9 / HBM_PEAK_GBS})(
            (10.0 * args.smoother_degree + 4.25) * (8.0 if vnum == mg.F64 else 4.0) * 8.0 / 7.0),
    }
    if not args.no_verify:
        out["verify"] = verify(args, solver, cube, world, decomposed, transport, native)
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args)
    failed = "verify" in out and not out["verify"]["ok"]
    solver.close()
    default_workload = (world == 1 and args.degree == 4 and args.cells == 128 and vnum == mg.F64 and not args.option and
                        polynomial == "first_kind" and args.smoother_degree == 3)
    if default_workload and not args.no_secondary:
        for v in (x, y, z):
            v.free()
        t_sec = time.time()
        out["secondary"] = secondary(mg, ctx, cube)
        out["secondary_seconds"] = time.time() - t_sec
        failed = failed or not all(e["verify"]["ok"] for e in out["secondary"].values())
    if rank == 0:
        print(json.dumps(out))
    cube.close()
    ctx.close()
    if dist is not None:
        dist.destroy_process_group()
    if failed:
        raise SystemExit("bench.py: a verification gave a wrong answer (see the verify objects)")


if __name__ == "__main__":
    main()
