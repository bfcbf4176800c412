"""Runs N finest-level fp64 matvecs (and optionally Chebyshev steps) -- profiling target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10
mode = sys.argv[3] if len(sys.argv) > 3 else "vmult"
deg = int(sys.argv[4]) if len(sys.argv) > 4 else 4
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
# mode "all": the level below the finest one runs on the one-launch schedule with separate transfer kernels, so that
# every launch of the fused transfer forms (and every 512-block launch of the eight-colour kernels) in the trace
# belongs to the finest level
# (no_restrict_scratch: otherwise the fused residual + restriction would run there as well, as one launch)
opts = {"free_one_max": 8192, "no_restrict_scratch": 1} if mode == "all" else {}
# MATVEC_OPTS="name=value,...": further context options (A/B of code paths under the profiler)
for kv in filter(None, os.environ.get("MATVEC_OPTS", "").split(",")):
    opts[kv.split("=")[0]] = float(kv.split("=")[1])
ctx = mg.Context(0, options=opts or None)
cube = mg.Cube(deg, ns, nr)
l = cube.max_level
op = mg.LaplaceOperator.from_cube(ctx, cube, l)
x = ctx.vector(cube.n_dofs(l), data=cube.seeded_vector(l, 42))
y = ctx.vector(cube.n_dofs(l))
if mode == "calib":
    # streaming kernels with known byte counts on finest-level vectors only: the calibration of the traffic counters
    # (tools/make_traffic_json.py)
    lib, nd = ctx.lib, cube.n_dofs(l)
    for _ in range(n):
        mg.check(lib.mgx_copy_cast(ctx.h, y.ptr, mg.F64, x.ptr, mg.F64, nd))
        mg.check(lib.mgx_sadd(ctx.h, mg.F64, y.ptr, 0.5, 0.25, x.ptr, nd))
        ctx.dot(x, y)
elif mode == "vmult":
    for _ in range(n):
        op.vmult(y, x)
elif mode == "all":
    # every finest-level form: operator (0), residual (1), and V-cycles -- pre-smoothing from a zero start (5, 6),
    # residual + restriction (7), post-smoothing with the prolongation fused in (9, 2, 2)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    A = solver.matrix_dp(l)
    r = ctx.vector(cube.n_dofs(l))
    for _ in range(n):
        A.vmult(y, x)
        A.vmult_residual(x, y, r)
        solver.vmult(y, x)
        ctx.sync()
else:
    sm = mg.Chebyshev(op, 20., 3, 15)
    for _ in range(n):
        sm.step(y, x)   # forms 3, 2, 2
        sm.vmult(y, x)  # forms 5, 6 (zero initial guess)
ctx.sync()
print("done", cube.n_dofs(l))
