import sys, numpy as np
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import multigrid_amd as mg
from test_gpu_dg_multigrid import Pair, rel
ctx = mg.Context(0)
for (p,nr,basis,num) in [(1,3,0,mg.F64),(6,1,0,mg.F64),(2,2,1,mg.F32),(5,2,2,mg.F64),(1,2,2,mg.F32),(7,1,0,mg.F64)]:
    P = Pair(ctx,p,nr,basis,num)
    rng=np.random.default_rng(1)
    x=rng.standard_normal(P.dgo.shape)
    src,dst=ctx.vector(P.solver.m(),data=P.to_product(x)),ctx.vector(P.solver.m())
    P.solver.vmult(dst,src)
    e=rel(P.to_oracle(dst.download()),P.orc.v_cycle(x))
    rhs=rng.standard_normal(P.dgo.shape)
    b,sol=ctx.vector(P.solver.m(),data=P.to_product(rhs)),ctx.vector(P.solver.m())
    its,red=P.solver.solve_cg(b,sol,1e-9)
    xo,oits,ored=P.orc.solve_cg(rhs,1e-9)
    print(p,nr,basis,num,"vcycle err %.2e its %d/%d sol err %.2e"%(e,its,oits,rel(P.to_oracle(sol.download()),xo)))
    P.close()
