"""Harness of BASELINE config 5, mirroring matvec_dg_cheby/program.cc: DG-SIP Laplace matvec merged
with one Chebyshev update (block Jacobi in the eigenvector basis), three local bases, fp32.

    python tools/matvec_dg_cheby.py [degree=3] [n_refinement_steps=15] [nsteps=100] [--number f32|f64] [--json]
                                    [--gpus N]

--gpus N: the mesh is block-decomposed over N ranks (one process per GPU, started by this script or by
torch.distributed.run), ghost cells exchanged over RCCL (MGX_BENCH_BACKEND=gloo: ranks share one GPU,
functional test); rates are those of the whole job (global DoFs over the slowest rank's time).

Same positional arguments as the reference program (program.cc:289-296) and the same result lines
("Best MF Chebyshev update <basis> n_dof= ... DoFs/s ... GFlop/s ... GB/s ... ops/dof", :171-187, and
"Best preconditioner", :246-253).  GB/s uses the reference's own 5-access model (:178); the kernel
here moves 4 vector accesses per DoF (the inverse diagonal is a 64-entry table, not a stream), which
is what the roofline fraction is computed from: 4 * sizeof(Number) B per DoF over 8 TB/s."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg  # noqa: E402

NAMES = {0: "Hermite", 1: "DGQ_GL ", 2: "DGQ_G  "}


def ops_approx(n_cells, degree, kind, dim=3):
    """flop model of matvec_dg_cheby/program.cc:140-169 (JACOBI_TRANSFORMATION_TYPE = 0)"""
    p = degree
    interp = 2 * ((p + 1) // 2) * 2 + p + 1 + 2 * ((p - 1) * (p + 1) // 2)
    n1 = p + 1
    per_cell = ((4 if kind < 2 else 2) * dim * interp * n1 ** (dim - 1) + dim * 2 * dim * n1 ** dim
                + (2 * dim * ((5 * (dim - 1) if kind < 2 else 3 * (dim - 1)) * interp * n1 ** (dim - 2)
                              + (4 * dim - 1 + 2 + 2 + 3 + 2 * dim) * n1 ** (dim - 1))
                   + ((dim + 2 if kind == 0 else 2 * dim) * (p + 1 + 2 * (p - 1) + 2) * 2
                      + (((dim - 2) * 4 + 2 * dim * 2) if kind == 0 else 4 * dim * (2 * p + 1))) * n1 ** (dim - 1)
                   + 2 * dim * interp * n1 ** (dim - 1)
                   + (1 + 1 + 5) * n1 ** dim))
    return n_cells * per_cell


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("degree", nargs="?", type=int, default=3)
    ap.add_argument("n_refinement_steps", nargs="?", type=int, default=15)
    ap.add_argument("nsteps", nargs="?", type=int, default=100)
    ap.add_argument("--number", choices=["f32", "f64"], default="f32")
    ap.add_argument("--bases", default="0,1,2")
    ap.add_argument("--outer", type=int, default=5)
    ap.add_argument("--json", action="store_true")
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--cpu-baseline", action="store_true",
                    help="also time the numpy restatement (oracle/dg_oracle.py, the checker of tests/) on a bounded sample "
                         "on the host: a reported baseline, not the target")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn(a.gpus)
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit("--gpus %d but WORLD_SIZE=%d" % (a.gpus, world))
    dist = comm = None
    if world > 1:
        import torch
        import torch.distributed as dist
        backend = os.environ.get("MGX_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            local_rank = 0
            dist.init_process_group(backend)
    number = mg.F32 if a.number == "f32" else mg.F64
    nbytes = 4 if number == mg.F32 else 8
    ctx = mg.Context(local_rank)
    cells, jac = mg.dg_cheby_mesh(a.n_refinement_steps)
    if world > 1:
        comm = mg.Communicator(ctx, dist)
        if comm.native_ready:  # RCCL send/recv issued by the library on its own stream
            mg.check(ctx.lib.mgx_context_use_rccl(ctx.h, 1))
        part = mg.dg_box_partition(cells, mg.process_grid(world), rank)
        nb, n_ghost, exchange = part["neighbours"], part["n_ghost"], part["exchange"]
    else:
        nb, _ = mg.dg_box_neighbours(cells)
        n_ghost, exchange = 0, None
    n_cells = int(np.prod(cells))
    say = print if rank == 0 else (lambda *args, **kw: None)

    def slowest(t):
        if dist is None:
            return t
        import torch
        v = torch.tensor([t], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(v, op=dist.ReduceOp.MAX)
        return float(v.item())

    say("Number of GPUs:                 %d%s" % (world, "" if world == 1 else " (%s)" % ("native RCCL" if comm.native_ready
                                                                                      else dist.get_backend())))
    say("Degree of element:              %d" % a.degree)
    say("Cells:                          %d x %d x %d, number type %s\n" % (*cells, a.number))
    results = []
    rng = np.random.default_rng(rank)
    for kind in [int(k) for k in a.bases.split(",")]:
        op = mg.DGLaplaceOperator(ctx, a.degree, kind, nb, jac, number, n_ghost, exchange)
        n_own = op.m()
        n = n_cells * (a.degree + 1) ** 3
        if kind == 0:
            say("Number of DoFs: %d" % n)
        rhs = op.initialize_dof_vector(rng.random(n_own))  # program.cc:106-107
        inp, out = op.initialize_dof_vector(), op.initialize_dof_vector()
        best = 1e10
        for o in range(a.outer):
            ctx.sync()
            t = time.perf_counter()
            for _ in range(a.nsteps):
                # the binding trades the storage of the two vectors (program.cc:119-121 swaps twice)
                op.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, out, inp)
                out, inp = inp, out
            ctx.sync()
            avg = slowest((time.perf_counter() - t) / a.nsteps)
            say("MF Chebyshev update %12.4e" % avg)
            best = min(best, avg)
        ops = ops_approx(n_cells, a.degree, kind)
        frac = 4 * nbytes * n / best / 8e12
        say("Best MF Chebyshev update %s n_dof= %-12d%-12.4e   DoFs/s %.5e    GFlop/s %.1f    GB/s %.1f    ops/dof %.1f"
              "    [4 accesses: %.1f GB/s = %.3f of 8 TB/s]\n"
              % (NAMES[kind], n, best, n / best, 1e-9 * ops / best, 1e-9 * n * nbytes * 5 / best, ops / n,
                 1e-9 * 4 * nbytes * n / best, frac))
        res = dict(basis=NAMES[kind].strip(), degree=a.degree, n_dofs=n, seconds=best, dofs_per_s=n / best,
                   gb_per_s_5_access_model=1e-9 * n * nbytes * 5 / best, roofline_frac_4_accesses=frac, dtype=a.number)
        # plain operator application (matvec_dg/program.cc:204: 3 accesses in the reference's model)
        ctx.sync()
        t = time.perf_counter()
        for _ in range(a.nsteps):
            op.vmult(out, inp)
        ctx.sync()
        tv = slowest((time.perf_counter() - t) / a.nsteps)
        say("MF vmult             %s n_dof= %-12d%-12.4e   DoFs/s %.5e    [2 accesses: %.3f of 8 TB/s]"
              % (NAMES[kind], n, tv, n / tv, 2 * nbytes * n / tv / 8e12))
        res.update(vmult_seconds=tv, vmult_dofs_per_s=n / tv)
        if kind == 2:
            ctx.sync()
            t = time.perf_counter()
            for _ in range(a.nsteps):
                op.jacobi_vmult(out, inp)
            ctx.sync()
            tj = slowest((time.perf_counter() - t) / a.nsteps)
            say("Best preconditioner  n_dof= %-12d%-12.4e   DoFs/s %.5e   GB/s %.1f\n"
                  % (n, tj, n / tj, 1e-9 * 4 * n * nbytes / tj))
            res.update(jacobi_seconds=tj)
        results.append(res)
        for v in (rhs, inp, out):
            v.free()
        op.clear()
    cpu = None
    if a.cpu_baseline and rank == 0:
        from oracle import dg_oracle as dgo
        ccells = (16, 16, 8)
        orc = dgo.DGOracle(a.degree, 0, ccells, dgo.cheby_mesh(11)[1])
        shape = orc.shape
        r, x, xo = (rng.random(shape) for _ in range(3))
        orc.vmult_with_chebyshev_update(r, 2, 0.6, 0.2, x, xo)  # builds the cached inverse diagonals
        t, reps = time.perf_counter(), 0
        while time.perf_counter() - t < 10.0:
            x, xo = orc.vmult_with_chebyshev_update(r, 2, 0.6, 0.2, x, xo)
            reps += 1
        dt = (time.perf_counter() - t) / reps
        nd = int(np.prod(shape))
        try:
            from threadpoolctl import threadpool_info
            cores = max([i.get("num_threads", 1) for i in threadpool_info()] or [1])
        except Exception:  # noqa: BLE001
            cores = os.cpu_count() or 1
        cpu = dict(value=nd / dt, unit="DoFs/s", cores=cores, kind="port",
                   sample="FE_DGQHermite(%d) on %dx%dx%d cells (%d DoFs), %d merged Chebyshev steps of the dense numpy "
                          "restatement (fp64), %.1f s" % (a.degree, *ccells, nd, reps, dt * reps))
        print("CPU baseline (numpy restatement, %d BLAS threads): %.3e DoFs/s  [%s]" % (cores, cpu["value"], cpu["sample"]))
    if a.json and rank == 0:
        print(json.dumps(dict(metric="DoFs/s, DG-SIP matvec merged with a Chebyshev update", n_gpus=world, results=results,
                              cpu_baseline=cpu)))
    ctx.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def spawn(n):
    """start n ranks of this script (the parent never touches the GPU)"""
    codes = mg.spawn_ranks(__file__, sys.argv[1:], n, one_gpu=os.environ.get("MGX_BENCH_BACKEND", "nccl") != "nccl")
    if any(codes):
        raise SystemExit("matvec_dg_cheby.py: rank exit codes %s" % codes)


if __name__ == "__main__":
    main()
