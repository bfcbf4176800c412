"""Cost / benefit of launching the interface bricks first and overlapping the exchange with the
interior bricks, on ONE GPU: a single-rank RCCL communicator whose only neighbour is the rank itself
(context option "rccl_selftest"), the "interface" being the unconstrained DoFs of the mid plane x = G/2 of the
cube (2 brick layers touch it, as for a rank with two interface faces).  The sums are wrong (the
rank adds its own copy), the launch sequence, the RCCL send/recv on the side stream and the timing
are those of a real decomposed run.  usage: overlap_probe.py on|off [cells] [reps]
Under `rocprofv3 --kernel-trace` the trace shows ncclDevKernel_* concurrent with brick_macro_kernel."""
import ctypes as C, os, sys, time
mode = sys.argv[1] if len(sys.argv) > 1 else "on"
cells = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 20
os.environ["MGX_OVERLAP_MIN_BRICKS"] = "1" if mode == "on" else "4000000000"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
from multigrid_amd import _lib

nr = int(np.log2(cells))
ctx = mg.Context(0, options={"rccl_selftest": 1})
lib = ctx.lib
buf = (C.c_uint8 * 128)()
mg.check(lib.mgx_rccl_unique_id(buf))
mg.check(lib.mgx_context_set_rccl(ctx.h, 0, 1, buf))
cube = mg.Cube(4, 1, nr)
l = cube.max_level
n = cube.n_dofs(l)
d = cube.operator_desc(l)
G = cells * 4 + 1
grid = cube.dof_grid(l)
nfree = n - cube.n_constrained(l)
plane = np.nonzero((grid[:nfree] % G) == (G // 2))[0].astype(np.uint32)
plane = plane[np.argsort(grid[plane])]
ex = _lib.ExchangeDesc()
ranks = (C.c_int * 1)(0)
counts = (C.c_uint32 * 1)(plane.size)
idxp = (_lib.u32p * 1)(plane.ctypes.data_as(_lib.u32p))
shared = np.sort(plane).astype(np.uint32)
ex.plan_id, ex.n_neighbors = 1, 1
ex.neighbor_rank = C.cast(ranks, C.POINTER(C.c_int))
ex.count = C.cast(counts, _lib.u32p)
ex.index = C.cast(idxp, C.POINTER(_lib.u32p))
ex.shared, ex.n_shared = shared.ctypes.data_as(_lib.u32p), shared.size
ex.not_owned, ex.n_not_owned = None, 0
ex.send_buf, ex.recv_buf = None, None
d.exchange = C.pointer(ex)
op = mg.LaplaceOperator(ctx, d)
x = ctx.vector(n, data=cube.seeded_vector(l, 7))
y = ctx.vector(n)


def timed(fn, k):
    fn(); ctx.sync()
    t = time.perf_counter()
    for _ in range(k):
        fn()
    ctx.sync()
    return (time.perf_counter() - t) / k * 1e3


t_mv = timed(lambda: op.vmult(y, x), reps)
sm = mg.Chebyshev(op, 20., 3, 15)
t_step = timed(lambda: sm.step(y, x), max(2, reps // 4))
print("overlap %s: %d^3 cells, interface %d DoFs (%.2f MB): vmult %.3f ms, Chebyshev step (3 fused iterations) %.3f ms"
      % (mode, cells, plane.size, plane.size * 8e-6, t_mv, t_step))
