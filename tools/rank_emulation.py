"""What one rank of `bench.py --gpus N` (strong scaling of the 128^3-cell problem) does, run on ONE GPU:
the rank-0 part of the block-split mesh with a one-rank RCCL communicator in which every neighbour
is the rank itself (context option "rccl_selftest").  The sums are wrong (the rank adds copies of its own
interface values), but the launch sequence, the kernels, the RCCL send/recv groups and the
reductions are those of a real run; only the link is missing.  Compared with the same number of
cells as an undecomposed cube this shows what the decomposition costs on the device.
usage: rank_emulation.py [N=8] [cells=128] [reps=10]"""
import ctypes as C, os, sys, time
N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 10
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg


def timed(ctx, fn, k):
    fn(); ctx.sync()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / k * 1e3


nr = int(np.log2(cells))
procs = mg.process_grid(N)
# further context options: RANK_EMULATION_OPTS="name=value,..."
ctx = mg.Context(0, options=dict({"rccl_selftest": 1}, **{k: float(v) for k, v in (o.split("=") for o in os.environ.get("RANK_EMULATION_OPTS", "").split(",") if o)}))
buf = (C.c_uint8 * 128)()
mg.check(ctx.lib.mgx_rccl_unique_id(buf))
mg.check(ctx.lib.mgx_context_set_rccl(ctx.h, 0, 1, buf))
cube = mg.Cube(4, n_refine=nr - 1, box=(2, 2, 2), procs=procs, rank=0, origin=-0.9, h0=0.95)
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64, comm=object())
l = cube.max_level
n = cube.n_dofs(l)
x, y, z = ctx.vector(n, data=cube.seeded_vector(l, 42)), ctx.vector(n), ctx.vector(n)
rhs = solver.get_vector(l, "rhs")
A = solver.matrix_dp(l)
t_mv = timed(ctx, lambda: A.vmult(y, x), reps)
t_vc = timed(ctx, lambda: solver.vmult(z, rhs), reps)
print("emulated rank 0 of %d (%s): %d DoFs, %d neighbours: vmult %.3f ms, V-cycle %.3f ms -> job rate %.3e DoFs/s "
      "(%.2f of %d x the one-GPU rate 1.26e10)"
      % (N, "x".join(map(str, procs)), n, len(cube.neighbors(l)), t_mv, t_vc, (cells * 4 + 1) ** 3 / (t_mv + t_vc) * 1e3,
         (cells * 4 + 1) ** 3 / (t_mv + t_vc) * 1e3 / 1.26e10 / N, N))
if len(sys.argv) > 4 and sys.argv[4] == "levels":
    solver.enable_timings(True)
    for _ in range(reps):
        solver.vmult(z, rhs)
    ctx.sync()
    t = solver.wall_times() / reps * 1e3
    for lev in range(t.shape[0]):
        print(lev, cube.n_dofs(lev), np.round(t[lev], 3))
