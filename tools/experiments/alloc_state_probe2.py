"""Is the 141-144 / 151 us state of the fused Chebyshev launch a matter of what runs between the V-cycles?  One process,
one solver: V-cycles alone, V-cycles alternating with the fp64 matvec (the bench step), and again."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigrid_amd as mg
ctx = mg.Context(0)
cube = mg.Cube(4, 1, 7)
l = cube.max_level
n = cube.n_dofs(l)
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
x = ctx.vector(n, data=cube.seeded_vector(l, 42)); y = ctx.vector(n); z = ctx.vector(n)
rhs = solver.get_vector(l, "rhs")
A = solver.matrix_dp(l)
solver.matrix(l).set_profiled(True)
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for mode in ("vcycle only", "matvec + vcycle"):
        for _ in range(2):
            solver.vmult(z, rhs)
        ctx.profile_enable(True)
        ctx.sync(); t = time.perf_counter()
        for _ in range(10):
            if mode != "vcycle only":
                A.vmult(y, x)
            solver.vmult(z, rhs)
        ctx.sync(); dt = (time.perf_counter() - t) / 10
        prof = {f: ctx.profile_read(f) for f in (2, 6, 9)}
        ctx.profile_enable(False)
        print("%-16s %.3f ms per step | " % (mode, 1e3 * dt) + "  ".join("form %d %.1f us" % (f, 1e3 * ms / max(1, k)) for f, (k, ms) in prof.items()), flush=True)
