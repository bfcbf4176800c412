"""Variable-coefficient / curved-geometry matvec (poisson_shell slice, BASELINE config 4): DoFs/s and the
fraction of the HBM roofline at 16 + 48 ((p+1)/p)^3 B per DoF (109.75 B at p = 4)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
p = int(sys.argv[1]) if len(sys.argv) > 1 else 4
nr = int(sys.argv[2]) if len(sys.argv) > 2 else 6
n_coarse = int(sys.argv[3]) if len(sys.argv) > 3 else 0  # 6 | 12: the hyper_shell mesh; 0: one structured sector
t = time.time()
ctx = mg.Context(0)
if n_coarse:
    cube = mg.Cube(p, n_refine=nr, shell=n_coarse, problem="shell")
else:
    cube = mg.Cube(p, n_refine=nr, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry="shell_sector", problem="shell")
l = cube.max_level
n = cube.n_dofs(l)
op = mg.LaplaceOperator.from_cube(ctx, cube, l)
print("setup %.1f s, %d cells, %d DoFs" % (time.time() - t, cube.n_cells(l), n))
x, y = ctx.vector(n, data=cube.seeded_vector(l, 1)), ctx.vector(n)
for _ in range(3):
    op.vmult(y, x)
ctx.sync()
t = time.perf_counter()
reps = 20
for _ in range(reps):
    op.vmult(y, x)
ctx.sync()
dt = (time.perf_counter() - t) / reps
bpd = 16 + 48 * ((p + 1) / p) ** 3
print("shell p=%d: vmult %.3f ms, %.3e DoFs/s, %.1f B/DoF algorithmic -> %.2f TB/s = %.3f of 8 TB/s"
      % (p, dt * 1e3, n / dt, bpd, bpd * n / dt / 1e12, bpd * n / dt / 8e12))
# the V-cycle on the same mesh (Chebyshev degree 3): every operator application is the general branch
t = time.time()
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
z = ctx.vector(n)
rhs = solver.get_vector(l, "rhs")
for _ in range(2):
    solver.vmult(z, rhs)
ctx.sync()
t1 = time.perf_counter()
for _ in range(5):
    solver.vmult(z, rhs)
ctx.sync()
dtv = (time.perf_counter() - t1) / 5
print("shell p=%d: V-cycle %.3f ms, %.3e DoFs/s (solver setup %.1f s)" % (p, dtv * 1e3, n / dtv, time.time() - t - 7 * dtv))
its, red = solver.solve_cg()
print("shell p=%d: PCG %d iterations, reduction %.3e per iteration, L2 error %.3e" % (p, its, red, solver.compute_l2_error()))
