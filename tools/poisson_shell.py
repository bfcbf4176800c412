#!/usr/bin/env python3
"""poisson_shell -- the reference program's command line, run protocol and output on the MI355X path
(poisson_shell/program.cc): variable-coefficient Laplace problem (coefficient 1 + 1e6 prod cos^2(2 pi x_e +
0.1 e), :157-169; solution sin(2 pi (x + y)), :97-137) on GridGenerator::hyper_shell(0, 0.5, 1.0, 6 | 12) with
curved cells, solved by the mixed-precision multigrid solver (V-cycle in float, :67-68).

    poisson_shell.py degree maxsize [n_mg_cycles n_pre_smooth n_post_smooth] [--vcycle f32|f64] [--cycles A:B] [--gpus N]

--gpus N: N ranks, one per GPU (started here, or already there under torch.distributed.run), the coarse cells of the
shell dealt out to them (contiguous shares; 8 ranks hold 1, 2, 1, 2, ... of the 12-cell shell and at most one cell
each of the 6-cell one -- cycles whose shell has fewer cells than ranks are skipped).

as `./program degree maxsize n_mg_cycles n_pre_smooth n_post_smooth` (:520-546).  run() (:413-446): cycle c uses
the 6-cell shell for even c and the 12-cell one for odd c, refined c / 2 times, until the number of DoFs
exceeds maxsize; per cycle (solve(), :315-381) 5 x solve(false), solve(true), L2 error, solve_cg(), L2 error,
5 batches of do_matvec() and of do_matvec_smoother(), the "Best timings" and "L2 error" lines, and the
convergence table at the end."""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("degree", type=int)
    ap.add_argument("maxsize", type=int, nargs="?", default=2000000)
    ap.add_argument("n_mg_cycles", type=int, nargs="?", default=1)
    ap.add_argument("n_pre_smooth", type=int, nargs="?", default=3)
    ap.add_argument("n_post_smooth", type=int, nargs="?", default=3)
    ap.add_argument("--vcycle", choices=["f32", "f64"], default="f32", help="vcycle_number (program.cc:67: float)")
    ap.add_argument("--cycles", default="0:35", help="first:last cycle of run() (default: all, as the program)")
    ap.add_argument("--gpus", type=int, default=1)
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        codes = mg.spawn_ranks(__file__, sys.argv[1:], a.gpus, one_gpu=os.environ.get("MGX_BENCH_BACKEND", "nccl") != "nccl")
        if any(codes):
            raise SystemExit("poisson_shell.py: rank exit codes %s" % codes)
        return
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dist = comm = None
    if world > 1:
        import torch
        import torch.distributed as dist
        if os.environ.get("MGX_BENCH_BACKEND", "nccl") == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(os.environ["MGX_BENCH_BACKEND"])
    out = print
    if rank != 0:
        def out(*x, **k):
            return None
    _main(a, out, dist, rank, world, local_rank)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def _main(a, print, dist, rank, world, local_rank):  # noqa: A002 (rank 0 prints)
    def slowest(v):
        if dist is None:
            return v
        import torch
        t = torch.tensor([v], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    print("Settings of parameters: ")
    print("Polynomial degree:              %d" % a.degree)
    print("Maximum size:                   %d" % a.maxsize)
    print("Number of MG cycles in V-cycle: %d" % a.n_mg_cycles)
    print("Number of pre-smoother iters:   %d" % a.n_pre_smooth)
    print("Number of post-smoother iters:  %d" % a.n_post_smooth)
    print()
    print("Testing FE_Q<3>(%d)" % a.degree)
    ctx_ranks = mg.Context(local_rank)
    ctx_alone = None   # a context without communicator, for the meshes every rank solves as a whole
    comm = mg.Communicator(ctx_ranks, dist) if dist is not None else None
    vnum = mg.F32 if a.vcycle == "f32" else mg.F64
    c0, c1 = (int(v) for v in a.cycles.split(":"))
    rows = []
    for cycle in range(c0, min(c1, 35)):
        print("Cycle %d" % cycle)
        n_coarse, n_refine = (6 if cycle % 2 == 0 else 12), cycle // 2          # :425-431
        N = a.degree * 2 ** n_refine
        n_dofs = (n_coarse * N * N + 2) * (N + 1)
        print("Number of degrees of freedom: %d" % n_dofs)
        if n_dofs > a.maxsize:
            print("Max size reached, terminating.")
            print()
            break
        # The ranks share the mesh in equal parts: coarse cells where the rank count divides them, else the cells of level
        # 1 (8 ranks: 6 of 48 / 12 of 96).  An unrefined shell of 6 or 12 cells that cannot be dealt out evenly is solved
        # by every rank as a whole (the reference's partition of such a mesh leaves ranks without cells).
        whole = world > 1 and n_coarse % world != 0 and (n_refine == 0 or (8 * n_coarse) % world != 0)
        if whole:
            print("(%d cells on %d ranks: every rank solves the whole mesh)" % (n_coarse, world))
        if whole and ctx_alone is None:
            ctx_alone = mg.Context(local_rank)
        ctx = ctx_alone if whole else ctx_ranks
        t0 = time.time()
        cube = mg.Cube(a.degree, n_refine=n_refine, shell=n_coarse, problem="shell", procs=(1 if whole else world, 1, 1),
                       rank=0 if whole else rank)
        assert (world > 1 and not whole) or cube.n_dofs(cube.max_level) == n_dofs
        solver = mg.MultigridSolver(ctx, cube, a.n_pre_smooth, a.n_post_smooth, a.n_mg_cycles, vnum, comm=None if whole else comm)
        print("Total setup time:      %gs" % (time.time() - t0))
        best_time, tot_time = 1e10, 0.
        for _ in range(5):                                                        # :334-343
            ctx.sync()
            t = time.perf_counter()
            solver.solve(False)
            ctx.sync()
            dt = slowest(time.perf_counter() - t)
            best_time, tot_time = min(best_time, dt), tot_time + dt
            print("Time solve   (CPU/wall)    %gs/%gs" % (dt, dt))
        reduction, _ = solver.solve(True)                                         # :344
        print("All solver time %g [p0] %g %g [p0]" % (tot_time, tot_time, tot_time))
        l2_error = solver.compute_l2_error()                                      # :351
        ctx.sync()
        t = time.perf_counter()
        cg_its, cg_red = solver.solve_cg()                                        # :354
        ctx.sync()
        time_cg = slowest(time.perf_counter() - t)
        l2_error_cg = solver.compute_l2_error()
        n_mv = 200 if n_dofs < 10000000 else 50
        best = {}
        for name, fn in (("mv", solver.do_matvec), ("mvs", solver.do_matvec_smoother)):  # :359-384
            best[name] = 1e10
            for _ in range(5):
                ctx.sync()
                t = time.perf_counter()
                for _ in range(n_mv):
                    fn()
                ctx.sync()
                dt = slowest((time.perf_counter() - t) / n_mv)
                best[name] = min(best[name], dt)
                if name == "mv":
                    print("matvec time dp %g [p0] %g %g [p0] DoFs/s: %g" % (dt, dt, dt, n_dofs / dt))
        print("Best timings for ndof = %d   mv %g    mv smooth %g   mg %g" % (n_dofs, best["mv"], best["mvs"], best_time))
        print("L2 error with ndof = %d  %g  with CG %g" % (n_dofs, l2_error, l2_error_cg))
        print()
        rows.append((n_coarse * 8 ** n_refine, n_dofs, best["mv"], best["mvs"], reduction, l2_error, best_time, l2_error_cg,
                     time_cg, cg_its, cg_red))
        solver.close()
        cube.close()
    print(" cells    dofs    mv_outer  mv_inner  reduction  fmg_L2error   fmg_time    cg_L2error    cg_time  cg_its cg_reduction")
    for r in rows:
        print("%-8d %-9d %.3e %.3e %.3e %.3e %.3e %.3e %.3e %-6d %.3e" % r)
    if ctx_alone is not None:
        ctx_alone.close()
    ctx_ranks.close()


if __name__ == "__main__":
    main()
