"""Worker of the multi-process tests of the distributed hyper_shell mesh (one process per rank, RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in the environment): the coarse cells of hyper_shell(6 | 12) dealt out to the ranks,
against the single-domain oracle on the whole mesh.
mode "host": gloo, CPU only -- partition tables and the exchange protocol on host arrays.
mode "gpu" : gloo transport, every rank computes on the (same, single) GPU through the C ABI."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5] if len(sys.argv) > 5 else "shell", dist, rank,
        world)
    dist.barrier()
    dist.destroy_process_group()


def run(mode, n_coarse, p, nr, problem, dist, rank, world, say=print):
    """`dist`: torch.distributed or an object with its interface (tests/thread_ranks.py)"""
    import torch
    import multigrid_amd as mg
    from oracle import Oracle

    whole = mg.Cube(p, n_refine=nr, shell=n_coarse, problem=problem)
    cube = mg.Cube(p, n_refine=nr, shell=n_coarse, problem=problem, procs=(world, 1, 1), rank=rank)
    orc = Oracle(p, degree=3, n_cycles=1, mesh=whole, problem=problem)
    # level i of the rank's mesh is level i + off of the whole mesh (off = 1 where the cells of level 1 are dealt out:
    # 8 ranks on either shell, 4 ranks on the six-cell one)
    off = cube.level_offset
    assert cube.n_levels + off == whole.n_levels
    # local DoF -> DoF of the whole mesh through the run-independent id of a DoF
    l2g = []
    for lev in range(cube.n_levels):
        gid = whole.dof_grid(lev + off)
        order = np.argsort(gid)
        pos = np.searchsorted(gid[order], cube.dof_grid(lev))
        assert np.array_equal(gid[order][pos], cube.dof_grid(lev))
        l2g.append(order[pos])

    def exchange_add_host(level, v):
        nbs, shared = cube.neighbors(level), cube.shared(level)
        own = v[shared].copy()
        ops, recvs = [], []
        for (rk, idx) in nbs:
            st, rt = torch.from_numpy(v[idx].copy()), torch.empty(idx.size, dtype=torch.float64)
            recvs.append(rt)
            ops += [dist.P2POp(dist.isend, st, rk), dist.P2POp(dist.irecv, rt, rk)]
        for r in dist.batch_isend_irecv(ops):
            r.wait()
        v[shared] = 0
        done_self = False
        for (rk, idx), rt in zip(nbs, recvs):
            if rk > rank and not done_self:
                v[shared] += own
                done_self = True
            np.add.at(v, idx, rt.numpy())
        if not done_self:
            v[shared] += own
        return v

    if mode == "host":
        for lev in range(cube.n_levels):
            nc = torch.tensor([float(cube.n_cells(lev))])
            dist.all_reduce(nc)
            assert int(nc.item()) == whole.n_cells(lev + off)
            units = n_coarse * 8 ** off           # what is dealt out: coarse cells, or the cells of level 1
            if units % world == 0:                # ... in equal shares wherever that is possible
                assert cube.n_cells(lev) * world == whole.n_cells(lev + off)
            rhs = exchange_add_host(lev, cube.rhs(lev).copy())
            ref = orc.rhs(lev + off)[l2g[lev]]
            assert np.abs(rhs - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30), (lev, np.abs(rhs - ref).max())
            owned = np.ones(cube.n_dofs(lev))
            owned[cube.not_owned(lev)] = 0
            t = torch.tensor([owned.sum()])
            dist.all_reduce(t)
            assert int(t.item()) == whole.n_dofs(lev + off), (int(t.item()), whole.n_dofs(lev + off))
            # the local index tables address the same DoFs of the whole mesh as the whole mesh's own tables
            c0 = (rank * units) // world * (whole.n_cells(lev + off) // units)   # first cell of this rank's first unit
            gi, li = whole.idx27(lev + off)[c0:c0 + cube.n_cells(lev)], cube.idx27(lev)
            ok = li != 0xFFFFFFFF
            assert np.array_equal(ok, gi != 0xFFFFFFFF) and np.array_equal(l2g[lev][li[ok]], gi[ok])
        say("rank %d host ok" % rank, flush=True)
    else:
        from oracle_view import assert_same_cg

        class View:  # the oracle seen through this rank's DoFs
            def __init__(self, o):
                self.o = o

            def solve_cg(self):
                return self.o.solve_cg()

            def cg_history(self):
                return self.o.cg_history()

        ctx = mg.Context(0)
        comm = mg.Communicator(ctx, dist)
        solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64, comm=comm)
        rel = lambda a, b: np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)
        for lev in range(cube.n_levels):
            m = l2g[lev]
            A = solver.matrix_dp(lev)
            x, b = cube.seeded_vector(lev, 1), cube.seeded_vector(lev, 2)
            xg, bg = whole.seeded_vector(lev + off, 1), whole.seeded_vector(lev + off, 2)
            assert np.array_equal(x, xg[m])
            src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
            A.vmult(dst, src)
            assert rel(dst.download(), orc.vmult(lev + off, xg)[m]) < 1e-12, ("vmult", lev)
            A.vmult_residual(rhs, src, dst)
            assert rel(dst.download(), orc.vmult_residual(lev + off, bg, xg)[m]) < 1e-12, ("residual", lev)
            assert rel(A.get_matrix_diagonal_inverse().download(), orc.inv_diag(lev + off)[m]) < 1e-12, ("diagonal", lev)
            assert abs(ctx.l2_norm(src) - np.linalg.norm(xg)) < 1e-12 * np.linalg.norm(xg)
            if off > 0 and lev == 0:
                continue  # (level 0 of the rank's hierarchy is a smoothed level of the whole mesh: it runs on the undecomposed copy)
            gi, oi = solver.smoother(lev).info(), orc.cheb_info(lev + off)
            assert gi["degree"] == oi["degree"] and abs(gi["cg_its"] - oi["cg_its"]) <= (2 if lev + off == 0 else 0), (lev, gi, oi)
            assert abs(gi["lambda_max"] - oi["lambda_max"]) < 1e-8 * oi["lambda_max"], (lev, gi, oi)
        l = cube.max_level
        m = l2g[l]
        x, xg = cube.seeded_vector(l, 5), whole.seeded_vector(l + off, 5)
        src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
        for _ in range(2):
            solver.vmult(dst, src)
            assert rel(dst.download(), orc.vcycle(xg)[m]) < 1e-9, "vcycle"
        rate, trace = solver.solve(True)
        orate, otrace = orc.solve(True)
        assert abs(rate - orate) < 1e-6 * orate, (rate, orate)
        l2 = solver.compute_l2_error()
        assert abs(l2 - orc.l2_error()) < 2e-6 * l2, (l2, orc.l2_error())
        its, _ = assert_same_cg(solver, View(orc))
        l2 = solver.compute_l2_error()
        assert abs(l2 - orc.l2_error()) < 2e-6 * l2, (l2, orc.l2_error())
        agg = (", coarse levels <= %d agglomerated" % solver.coarse_level) if solver.coarse is not None else ""
        say("rank %d gpu ok: FMG rate %.4f, cg its %d, L2 %.6e%s" % (rank, rate, its, l2, agg), flush=True)
        solver.close()
        ctx.close()


if __name__ == "__main__":
    main()
