import sys, os
os.environ['MGX_BRICK_MIN'] = '1'; os.environ['MGX_RESTRICT_COLOUR_MIN'] = '8'
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np, multigrid_amd as mg
from oracle_view import oracle_for
ctx = mg.Context(0)
def rel(a,b): return np.abs(a-b).max()/max(np.abs(b).max(),1e-300)
for p,ns,nr in [(4,1,2),(3,3,2),(2,1,2)]:
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=3, n_cycles=1, vfloat=True)
    l = cube.max_level
    x = cube.seeded_vector(l, 5); b = cube.seeded_vector(l, 6)
    x32 = x.astype(np.float32).astype(np.float64); b32 = b.astype(np.float32).astype(np.float64)
    for env in ("0", "1"):
        os.environ.pop("MGX_NO_DIAG_TABLE", None)
        if env == "1": os.environ["MGX_NO_DIAG_TABLE"] = "1"
        A = mg.LaplaceOperator.from_cube(ctx, cube, l, mg.F32)
        s, r, d = ctx.vector(x.size, mg.F32, x), ctx.vector(x.size, mg.F32, b), ctx.vector(x.size, mg.F32)
        A.vmult(d, s); e1 = rel(d.download().astype(np.float64), orc.vmult(l, x32))
        A.vmult_residual(r, s, d); e2 = rel(d.download().astype(np.float64), orc.vmult_residual(l, b32, x32))
        sm = mg.Chebyshev(A, 20., 3, 15)
        sm.vmult(d, r); ref = orc.cheb_vmult(l, b32); e3 = rel(d.download().astype(np.float64), ref)
        d.upload(ref.astype(np.float32)); sm.step(d, r); e4 = rel(d.download().astype(np.float64), orc.cheb_step(l, ref.astype(np.float32).astype(np.float64), b32))
        print("p=%d notable=%s vmult %.2e residual %.2e cheb_vmult %.2e cheb_step %.2e" % (p, env, e1, e2, e3, e4))
