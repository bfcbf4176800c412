import sys, os
os.environ['MGX_BRICK_MIN'] = '1'; os.environ['MGX_RESTRICT_COLOUR_MIN'] = '8'
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, multigrid_amd as mg
from oracle_view import oracle_for
ctx = mg.Context(0)
def rel(a,b): return np.abs(a-b).max()/max(np.abs(b).max(),1e-300)
for env in ({}, {"MGX_NO_FUSED_RESTRICT": "1"}, {"MGX_NO_DIAG_TABLE": "1"}, {"MGX_BRICK_FORM": "cells"}, {"MGX_NO_GRAPH": "1"}):
    for k in ("MGX_NO_FUSED_RESTRICT", "MGX_NO_DIAG_TABLE", "MGX_BRICK_FORM", "MGX_NO_GRAPH"):
        os.environ.pop(k, None)
    os.environ.update(env)
    for p,ns,nr in [(4,1,3),(3,3,2)]:
        for num in (mg.F32, mg.F64):
            cube = mg.Cube(p, ns, nr)
            orc = oracle_for(cube, p, ns, nr, degree=3, n_cycles=1, vfloat=(num == mg.F32))
            solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, num)
            lmax = cube.max_level
            x = cube.seeded_vector(lmax, 5)
            src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
            solver.vmult(dst, src)
            e = rel(dst.download(), orc.vcycle(x))
            solver.vmult(dst, src)
            e2 = rel(dst.download(), orc.vcycle(x))
            print(env, "p=%d number=%s vcycle err %.3e (second call %.3e)" % (p, "f32" if num == mg.F32 else "f64", e, e2))
            solver.close(); cube.close(); orc.close()
