"""Time of one full-multigrid cycle (MultigridSolver::solve, the program's "fmg" column) and of the stand-alone transfer
kernels it is made of.  usage: [MGX_LIB_PATH=...] python tools/fmg_time.py <degree> <cells per direction> [f32]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multigrid_amd as mg
p, cells = int(sys.argv[1]), int(sys.argv[2])
num = mg.F32 if len(sys.argv) > 3 and sys.argv[3] == "f32" else mg.F64
ns, nr = cells, 0
while ns % 2 == 0 and ns > 1:
    ns //= 2; nr += 1
ctx = mg.Context(0)
cube = mg.Cube(p, ns, nr)
solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, num)
best = 1e9
for _ in range(6):
    ctx.sync(); t = time.perf_counter(); red = solver.solve(False)[0]; ctx.sync()
    best = min(best, time.perf_counter() - t)
print("p=%d %d^3 cells %d DoFs: FMG %.3f ms (reduction %.4g), L2 error %.4g" % (p, cells, cube.n_dofs(cube.max_level), 1e3 * best, red, solver.compute_l2_error()))
