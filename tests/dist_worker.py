"""Worker of the multi-process tests (one process per rank, launched by test_decomposition.py /
test_gpu_distributed.py with RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment).

mode "host": gloo, CPU only -- exercises the decomposition tables and the exchange protocol on host
             arrays (the interface sum a rank performs is emulated in numpy).
mode "gpu" : gloo transport, every rank computes on the (same, single) GPU through the C ABI;
             results are compared with the single-domain CPU oracle on the same global mesh.
mode "nccl": one rank, backend nccl (= RCCL): the torch transport and the library's own RCCL
             communicator (mgx_context_set_rccl) live in one process; the native transport is
             cross-checked and enabled as bench.py does, then the same comparisons as "gpu".
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def lex_of(o, l, v):
    out = np.empty(v.size)
    out[o.dof_grid(l)] = v
    return out


def main():
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    import torch
    if sys.argv[1] == "nccl":
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    run(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4:], dist, rank, world)
    dist.barrier()
    dist.destroy_process_group()


def run(mode, p, nr, flags, dist, rank, world, say=print):
    """the checks of one rank; `dist`: torch.distributed or an object with its interface (tests/thread_ranks.py)"""
    import torch
    vfloat = "f32" in flags  # V-cycle number type (reference default: float)
    strong = "strong" in flags  # block-split of the square mesh with n_subdiv = 2 (bench.py --scaling strong)
    # poisson_shell slice: shell sector (or sheared box) with the variable coefficient, block-split like "strong"
    mapped = [g for g in ("shell_sector", "sheared") if g in flags]

    import multigrid_amd as mg
    from oracle import Oracle

    procs = mg.process_grid(world)
    if mapped:
        problem = "shell" if mapped[0] == "shell_sector" else "cube"
        cube = mg.Cube(p, n_refine=nr, box=(2, 2, 2), procs=procs, rank=rank, origin=-0.9, h0=0.95, geometry=mapped[0],
                       problem=problem)
        orc = Oracle(p, n_subdiv=2, n_refine=nr, degree=3, n_cycles=1, vfloat=vfloat, geometry=mapped[0], problem=problem,
                     origin=-0.9, h0=0.95)
    elif strong:
        cube = mg.Cube(p, n_refine=nr, box=(2, 2, 2), procs=procs, rank=rank, origin=-0.9, h0=0.95)
        orc = Oracle(p, n_subdiv=2, n_refine=nr, degree=3, n_cycles=1, vfloat=vfloat)
    else:
        cube = mg.Cube(p, n_refine=nr, box=procs, procs=procs, rank=rank)
        orc = Oracle(p, n_refine=nr, degree=3, n_cycles=1, box=procs, vfloat=vfloat)
    l = cube.max_level
    gid = cube.dof_grid(l)

    def exchange_add_host(level, v):
        """numpy emulation of mgx_exchange_add over gloo (ascending-rank summation)"""
        nbs = cube.neighbors(level)
        shared = cube.shared(level)
        own = v[shared].copy()
        ops, recvs = [], []
        for (rk, idx) in nbs:
            st = torch.from_numpy(v[idx].copy())
            rt = torch.empty(idx.size, dtype=torch.float64)
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
            g = cube.dof_grid(lev)
            rhs = exchange_add_host(lev, cube.rhs(lev).copy())
            ref = lex_of(orc, lev, orc.rhs(lev))[g]
            assert np.abs(rhs - ref).max() <= 1e-12 * max(np.abs(ref).max(), 1e-30), (lev, np.abs(rhs - ref).max())
            # ownership: every global DoF is owned exactly once
            owned = np.ones(cube.n_dofs(lev))
            owned[cube.not_owned(lev)] = 0
            t = torch.tensor([owned.sum()])
            dist.all_reduce(t)
            assert int(t.item()) == orc.n_dofs(lev), (int(t.item()), orc.n_dofs(lev))
        say("rank %d host ok" % rank, flush=True)
    else:
        ctx = mg.Context(0)
        comm = mg.Communicator(ctx, dist)
        device_rhs = os.environ.get("MGX_TEST_DEVICE_RHS", "0") == "1"
        solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F32 if vfloat else mg.F64, comm=comm, device_rhs=device_rhs)
        if device_rhs:  # assembled per rank on the GPU and summed over the interface: the oracle's rhs of the whole mesh
            for lev in range(cube.n_levels):
                ref = lex_of(orc, lev, orc.rhs(lev))[cube.dof_grid(lev)]
                got = solver.get_vector(lev, "rhs").download()
                assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), ("device rhs", lev, np.abs(got - ref).max())
        tol_v, tol_r, tol_l2 = (2e-4, 2e-3, 1e-4) if vfloat else (1e-9, 1e-6, 1e-8)
        if mode == "nccl":
            assert comm.native_ready, "library-side RCCL communicator was not created"
            assert comm.verify_and_enable_native(solver.matrix_dp(l), cube.n_dofs(l))
        # matvec / residual on every level against the oracle on the global mesh
        for lev in range(cube.n_levels):
            g = cube.dof_grid(lev)
            A = solver.matrix_dp(lev)
            x = cube.seeded_vector(lev, 1)
            b = cube.seeded_vector(lev, 2)
            xo, bo = np.empty(orc.n_dofs(lev)), np.empty(orc.n_dofs(lev))
            og = orc.dof_grid(lev)
            xg, bg = np.zeros(orc.n_dofs(lev)), np.zeros(orc.n_dofs(lev))
            # the seeded vector depends on the global grid id only: build the oracle's copy the same way
            import tests.golden.make_golden as mk
            lexx, lexb = mk.seeded_lex(orc.n_dofs(lev), 1), mk.seeded_lex(orc.n_dofs(lev), 2)
            xo[:], bo[:] = lexx[og], lexb[og]
            assert np.array_equal(x, lexx[g])
            src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
            A.vmult(dst, src)
            ref = lex_of(orc, lev, orc.vmult(lev, xo))[g]
            err = np.abs(dst.download() - ref).max() / np.abs(ref).max()
            assert err < 1e-12, ("vmult", lev, err)
            A.vmult_residual(rhs, src, dst)
            ref = lex_of(orc, lev, orc.vmult_residual(lev, bo, xo))[g]
            err = np.abs(dst.download() - ref).max() / np.abs(ref).max()
            assert err < 1e-12, ("residual", lev, err)
            assert abs(ctx.l2_norm(src) - np.linalg.norm(xo)) < 1e-12 * np.linalg.norm(xo)
            gi, oi = solver.smoother(lev).info(), orc.cheb_info(lev)
            # (level 0 iterates to convergence: in fp32 the count moves with the order of the interface sums)
            assert gi["degree"] == oi["degree"] and abs(gi["cg_its"] - oi["cg_its"]) <= (2 if vfloat and lev == 0 else 0), (lev, gi, oi)
            assert abs(gi["lambda_max"] - oi["lambda_max"]) < (1e-4 if vfloat else 1e-8) * oi["lambda_max"], (lev, gi, oi)
        # V-cycle, FMG, PCG
        x = cube.seeded_vector(l, 5)
        import tests.golden.make_golden as mk
        xo = mk.seeded_lex(orc.n_dofs(l), 5)[orc.dof_grid(l)]
        src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
        solver.vmult(dst, src)
        ref = lex_of(orc, l, orc.vcycle(xo))[gid]
        err = np.abs(dst.download() - ref).max() / np.abs(ref).max()
        assert err < tol_v, ("vcycle", err)
        rate, trace = solver.solve(True)
        orate, otrace = orc.solve(True)
        assert abs(rate - orate) < max(tol_r, 1e-6) * orate * (100 if vfloat else 1), (rate, orate)
        assert np.allclose(trace[1:, 0], otrace[1:, 1], rtol=1e-3 if vfloat else 1e-9)
        l2 = solver.compute_l2_error()
        assert abs(l2 - orc.l2_error()) < tol_l2 * l2, (l2, orc.l2_error())
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from oracle_view import assert_same_cg
        its, red = assert_same_cg(solver, orc, rtol=0.05 if vfloat else 1e-6)
        l2 = solver.compute_l2_error()
        assert abs(l2 - orc.l2_error()) < tol_l2 * l2
        agg = (", coarse levels <= %d agglomerated" % solver.coarse_level) if solver.coarse is not None else ""
        say("rank %d gpu ok: FMG L2 %.6e, cg its %d%s%s" % (rank, l2, its, ", native RCCL" if comm.native_enabled else "", agg),
            flush=True)
        solver.close()
        ctx.close()


if __name__ == "__main__":
    main()
