"""Worker of the decomposed DG multigrid test (one process per rank over gloo, all on one GPU).
argv: degree n_refine basis number(f32|f64)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run(int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], dist, rank, world)
    dist.barrier()
    dist.destroy_process_group()


def run(p, nr, basis, number, dist, rank, world, say=print):
    """`dist`: torch.distributed or an object with its interface (tests/thread_ranks.py)"""
    import multigrid_amd as mg
    from oracle import Oracle, dg_oracle as dg

    num = mg.F64 if number == "f64" else mg.F32
    procs = mg.process_grid(world)
    ctx = mg.Context(0)
    comm = mg.Communicator(ctx, dist)
    cube = mg.Cube(p, n_refine=nr, box=procs, procs=procs, rank=rank)
    solver = mg.DGMultigridSolver(ctx, cube, basis, 3, num, comm=comm)
    cells = tuple(int(c) for c in cube.cells_per_dim3(cube.max_level)[1])
    dgo = dg.DGOracle(p, basis, cells, np.eye(3) * cube.cell_size(cube.max_level))
    fe = Oracle(p, n_refine=nr, degree=3, box=procs, vfloat=(num == mg.F32))
    n_global = int(np.prod(dgo.shape))
    start = (np.arange(n_global) % 11).astype(float)
    start -= start.mean()
    orc = dg.DGMultigridOracle(dgo, fe, 3, start.reshape(dgo.shape))
    ijk = solver.cell_ijk
    mine = lambda a: a[ijk[:, 2], ijk[:, 1], ijk[:, 0]].ravel()   # noqa: E731
    n = solver.m()
    rel = lambda a, b: abs(a - b).max() / abs(b).max()            # noqa: E731
    tol = 1e-8 if num == mg.F64 else 5e-4
    info = solver.smoother_info()
    assert info["cg_its"] == orc.cg_its
    assert abs(info["lambda_max"] - orc.lambda_max) < (1e-8 if num == mg.F64 else 1e-4) * orc.lambda_max, (info, orc.lambda_max)
    rng = np.random.default_rng(3)
    x, rhs = rng.standard_normal(dgo.shape), rng.standard_normal(dgo.shape)
    src, dst = solver.initialize_dof_vector(mine(x)), solver.initialize_dof_vector()
    solver.vmult(dst, src)
    assert rel(dst.download()[:n], mine(orc.v_cycle(x))) < tol, "v-cycle"
    b, sol = solver.initialize_dof_vector(mine(rhs)), solver.initialize_dof_vector()
    its, red = solver.solve_cg(b, sol, 1e-9)
    xo, oits, ored = orc.solve_cg(rhs, 1e-9)
    assert abs(its - oits) <= (0 if num == mg.F64 else 1), (its, oits)
    assert rel(sol.download()[:n], mine(xo)) < (1e-7 if num == mg.F64 else 1e-6), "solution"
    say("rank %d dg multigrid ok: %d iterations, lambda_max %.6f" % (rank, its, info["lambda_max"]), flush=True)
    solver.close()
    ctx.close()


if __name__ == "__main__":
    main()
