"""Do the per-form launch times of the finest level depend on where the vectors of a solver happen to lie?  Several
solver instances in ONE process, with allocations of varying size in between; per instance the HIP-event averages of
the fused Chebyshev form (2), the old-from-rhs form (6) and the prolongation form (9), and the V-cycle time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import multigrid_amd as mg
ctx = mg.Context(0)
cube = mg.Cube(4, 1, 7)
l = cube.max_level
n = cube.n_dofs(l)
keep = []
for inst in range(int(sys.argv[1]) if len(sys.argv) > 1 else 6):
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    solver.set_polynomial_type("first") if hasattr(solver, "set_polynomial_type") else None
    z, rhs = ctx.vector(n), solver.get_vector(l, "rhs")
    for _ in range(3):
        solver.vmult(z, rhs)
    solver.matrix(l).set_profiled(True)
    ctx.profile_enable(True)
    ctx.sync(); t = time.perf_counter()
    for _ in range(10):
        solver.vmult(z, rhs)
    ctx.sync(); dt = (time.perf_counter() - t) / 10
    prof = {f: ctx.profile_read(f) for f in (2, 6, 9)}
    ctx.profile_enable(False)
    print("instance %d: V-cycle %.3f ms | " % (inst, 1e3 * dt) + "  ".join("form %d %.1f us" % (f, 1e3 * ms / max(1, k)) for f, (k, ms) in prof.items()), flush=True)
    solver.close()
    keep.append(ctx.vector((inst + 1) * 37_000_000))  # shifts where the next instance's vectors go
