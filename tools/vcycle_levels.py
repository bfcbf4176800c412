"""Per-level wall times of the V-cycle (MultigridSolver::print_wall_times), synchronised timers."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import multigrid_amd as mg
cells = int(sys.argv[1]) if len(sys.argv) > 1 else 128
deg = int(sys.argv[2]) if len(sys.argv) > 2 else 4
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
# further arguments NAME=VALUE: context options (include/mgx.h mgx_context_set_option)
ctx = mg.Context(0, options={k: float(v) for k, v in (o.split("=") for o in sys.argv[3:])})
cube = mg.Cube(deg, ns, nr)
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
n = cube.n_dofs(cube.max_level)
z = ctx.vector(n); rhs = solver.get_vector(cube.max_level, "rhs")
for _ in range(3): solver.vmult(z, rhs)
ctx.sync(); t = time.perf_counter()
for _ in range(5): solver.vmult(z, rhs)
ctx.sync(); print("V-cycle ms (untimed levels) %.3f" % ((time.perf_counter() - t) / 5 * 1e3))
solver.enable_timings(True); solver.wall_times()
for _ in range(5): solver.vmult(z, rhs)
t = solver.wall_times() / 5 * 1e3
np.set_printoptions(precision=3, suppress=True, linewidth=150)
print("level: mg_mv restrict prolongate inhomBC mg_vec smoother  [ms per V-cycle]")
for l in range(cube.n_levels): print(l, cube.n_dofs(l), t[l])
