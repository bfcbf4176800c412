import faulthandler, sys, time
faulthandler.dump_traceback_later(90, exit=True)
sys.path.insert(0, ".")
import numpy as np
import multigrid_amd as mg
from oracle import Oracle
t0 = time.time()
def log(*a):
    print("[%.2fs]" % (time.time() - t0), *a, flush=True)
ctx = mg.Context(0)
for (p, ns, nr) in [(2, 1, 2), (4, 1, 2)]:
    cube = mg.Cube(p, ns, nr); log("cube", p, ns, nr)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64); log("solver built")
    orc = Oracle(p, ns, nr, degree=3)
    for l in range(cube.n_levels):
        sm = solver.smoother(l); log(l, sm.info(), orc.cheb_info(l))
        b = cube.seeded_vector(l, 7)
        bd, xd = ctx.vector(b.size, data=b), ctx.vector(b.size)
        sm.vmult(xd, bd); log("vmult done", np.abs(xd.download() - orc.cheb_vmult(l, b)).max())
    x = cube.seeded_vector(cube.max_level, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    solver.vmult(dst, src); log("vcycle", np.abs(dst.download() - orc.vcycle(x)).max())
    log(solver.solve(True)); log(orc.solve(True)); log(solver.compute_l2_error(), orc.l2_error())
    log(solver.solve_cg(), orc.solve_cg())
    solver.close()
log("done")
