import sys, numpy as np
import os; R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import multigrid_amd as mg
from oracle import Oracle
ctx = mg.Context(0)
for n_coarse, p, nr in ((12, 4, 1), (6, 5, 1), (6, 4, 2)):
    cube = mg.Cube(p, n_refine=nr, shell=n_coarse, problem="shell")
    orc = Oracle(p, degree=3, n_cycles=1, mesh=cube, problem="shell")
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    its, red = solver.solve_cg(); oits, ored = orc.solve_cg()
    h, oh = solver.cg_history(), orc.cg_history()
    n = min(len(h), len(oh))
    print(n_coarse, p, nr, its, oits, solver.compute_l2_error(), orc.l2_error())
    print(" rel hist", np.array2string(oh[:n]/oh[0], precision=2))
    print(" err", np.array2string(np.abs(h[:n]-oh[:n])/oh[:n], precision=1))
    solver.close(); orc.close(); cube.close()
