"""Time of the V-cycle on the coarsest levels alone (the replayed graph of the big problems): cubes with 1 ... 5 levels."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4
ctx = mg.Context(0)
for nr in range(0, 6):
    cube = mg.Cube(p, 1, nr)
    l = cube.max_level
    n = cube.n_dofs(l)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    z, rhs = ctx.vector(n), solver.get_vector(l, "rhs")
    for _ in range(20):
        solver.vmult(z, rhs)
    ctx.sync(); t = time.perf_counter()
    for _ in range(200):
        solver.vmult(z, rhs)
    ctx.sync(); dt = (time.perf_counter() - t) / 200
    print("levels 0..%d (%d cells, %d DoFs on the finest): V-cycle %.1f us" % (nr, cube.n_cells(l), n, 1e6 * dt), flush=True)
    solver.close()
