"""What one rank of `tools/matvec_dg_cheby.py --gpus N` does, run on ONE GPU: rank 0's block of the
decomposed DG mesh with its ghost cells, and a one-rank RCCL communicator in which every neighbour
is the rank itself (context option "rccl_selftest") -- the kernels, the pack launches and the RCCL send/recv
groups of a real run; only the links are missing (and the ghost values are the rank's own).

    python tools/dg_rank_emulation.py [N=8] [degree=4] [n_refinement_steps=21] [nsteps=20]

Prints the time of the merged Chebyshev step (a) with the ghost exchange under the interior cells
(default), (b) with the exchange in front of all cells (option "dg_no_overlap", second context) and
(c) of the same number of cells without any neighbour rank."""
import ctypes as C
import os
import sys
import time

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
degree = int(sys.argv[2]) if len(sys.argv) > 2 else 4
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 21
nsteps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import multigrid_amd as mg  # noqa: E402


def rccl_context(option=None):
    ctx = mg.Context(0, options=dict({"rccl_selftest": 1}, **({option: 1} if option else {})))
    buf = (C.c_uint8 * 128)()
    mg.check(ctx.lib.mgx_rccl_unique_id(buf))
    mg.check(ctx.lib.mgx_context_set_rccl(ctx.h, 0, 1, buf))
    return ctx


def cheb_time(ctx, op):
    rng = np.random.default_rng(0)
    rhs = op.initialize_dof_vector(rng.random(op.m()))
    inp, out = op.initialize_dof_vector(rng.random(op.m())), op.initialize_dof_vector()
    best = 1e10
    for _ in range(3):
        ctx.sync()
        t = time.perf_counter()
        for _ in range(nsteps):
            op.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, out, inp)
            out, inp = inp, out
        ctx.sync()
        best = min(best, (time.perf_counter() - t) / nsteps)
    for v in (rhs, inp, out):
        v.free()
    return best


cells, jac = mg.dg_cheby_mesh(steps)
procs = mg.process_grid(N)
part = mg.dg_box_partition(cells, procs, 0)
n_own = len(part["neighbours"])
n3 = (degree + 1) ** 3
print("mesh %dx%dx%d cells over %s ranks: rank 0 owns %d cells (%d DoFs), %d ghost cells from %d neighbours"
      % (*cells, "x".join(map(str, procs)), n_own, n_own * n3, part["n_ghost"], len(part["exchange"])))
times = {}
for label, option in (("overlapped", None), ("exchange first", "dg_no_overlap")):
    ctx = rccl_context(option)
    op = mg.DGLaplaceOperator(ctx, degree, 0, part["neighbours"], jac, mg.F32, part["n_ghost"], part["exchange"])
    times[label] = cheb_time(ctx, op)
    op.clear()
    ctx.close()
ctx = mg.Context(0)
blk = tuple(int(c) // int(p) for c, p in zip(cells, procs))
nb, _ = mg.dg_box_neighbours(blk)
op = mg.DGLaplaceOperator(ctx, degree, 0, nb, jac, mg.F32)
times["same cells, no neighbour rank"] = cheb_time(ctx, op)
op.clear()
ctx.close()
for label, t in times.items():
    print("%-32s %.4e s per merged Chebyshev step   %.4e DoFs/s per rank" % (label, t, n_own * n3 / t))
