"""Worker of the decomposed DG tests (one process per rank over gloo, all ranks on the one GPU):
the DG operator and the merged Chebyshev update on a block-decomposed box against the single-domain
face-based oracle on the same global mesh.  argv: degree basis steps number(f32|f64)"""
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


def run(p, basis, steps, number, dist, rank, world, say=print):
    """`dist`: torch.distributed or an object with its interface (tests/thread_ranks.py)"""
    import multigrid_amd as mg
    from oracle import dg_oracle as dg

    num = mg.F64 if number == "f64" else mg.F32
    tol = 2e-11 if num == mg.F64 else 5e-5
    cells, jac = mg.dg_cheby_mesh(steps)
    procs = mg.process_grid(world)
    part = mg.dg_box_partition(cells, procs, rank)
    ctx = mg.Context(0)
    comm = mg.Communicator(ctx, dist)
    op = mg.DGLaplaceOperator(ctx, p, basis, part["neighbours"], jac, num, part["n_ghost"], part["exchange"])
    o = dg.DGOracle(p, basis, cells, jac)
    ijk = part["ijk"]
    rng = np.random.default_rng(5)   # the same global vectors on every rank
    x, xo, rhs = (rng.standard_normal(o.shape) for _ in range(3))
    mine = lambda a: a[ijk[:, 2], ijk[:, 1], ijk[:, 0]].ravel()   # noqa: E731
    n = op.m()

    def owned(v):
        return v.download()[:n].astype(float)

    def rel(a, b):
        return abs(a - b).max() / abs(b).max()

    src, dst = op.initialize_dof_vector(mine(x)), op.initialize_dof_vector()
    op.vmult(dst, src)
    assert rel(owned(dst), mine(o.vmult(x))) < tol, "vmult"
    b = op.initialize_dof_vector(mine(rhs))
    op.vmult_residual(dst, b, src)
    assert rel(owned(dst), mine(rhs - o.vmult(x))) < tol, "residual"
    for idx in (0, 1, 2):
        sol, old = op.initialize_dof_vector(mine(x)), op.initialize_dof_vector(mine(xo))
        op.vmult_with_chebyshev_update(b, idx, 0.6, 0.2, sol, old)
        new_ref, old_ref = o.vmult_with_chebyshev_update(rhs, idx, 0.6, 0.2, x, xo)
        assert rel(owned(sol), mine(new_ref)) < 5 * tol, ("cheb", idx)
        assert rel(owned(old), mine(old_ref)) < 5 * tol, ("cheb old", idx)
    # repeated updates: the ghosts are refreshed from the new iterate every time
    out, inp = op.initialize_dof_vector(mine(x)), op.initialize_dof_vector(mine(xo))
    ro, ri = x, xo
    for _ in range(3):
        op.vmult_with_chebyshev_update(b, 2, 0.6, 0.2, out, inp)
        ro, ri = o.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, ro, ri)
    assert rel(owned(out), mine(ro)) < 50 * tol, "loop"
    # merged CG iteration (action 2): the sums cover every rank's owned cells
    r, q, pv = (rng.standard_normal(o.shape) for _ in range(3))
    R, Q, Pv, X = (op.initialize_dof_vector(mine(a)) for a in (r, q, pv, x))
    sums = op.vmult_with_cg_update(0.37, 0.81, R, Q, Pv, X)
    p_ref = 0.81 * pv + q
    q_ref = o.vmult(p_ref)
    assert rel(owned(Q), mine(q_ref)) < 5 * tol and rel(owned(X), mine(x + 0.37 * pv)) < 5 * tol, "cg update"
    ref = np.array([(q_ref * p_ref).sum(), (r * r).sum(), (q_ref * r).sum(), (q_ref * q_ref).sum()])
    assert np.allclose(sums, ref, rtol=100 * tol, atol=100 * tol * abs(ref).max()), ("cg sums", sums, ref)
    say("rank %d dg ok: %d owned cells, %d ghost cells, %d neighbours" % (rank, len(ijk), part["n_ghost"], len(part["exchange"])),
          flush=True)
    op.clear()
    ctx.close()


if __name__ == "__main__":
    main()
