"""Domain decomposition, host side (no GPU): the per-rank tables of the provider (mgx_cube box
meshes = the reference's "doubling" mesh family, program.cc:509-529) against the single-domain
oracle on the same global mesh, within one process and with world_size-2/4 gloo processes."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import multigrid_amd as mg
from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def launch(mode, world, p, nr, timeout=600, extra=(), worker="dist_worker.py"):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
        argv = [mode, str(p), str(nr)] if worker == "dist_worker.py" else []
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", worker)] + argv + list(extra), env=env,
                                      cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    for pr in procs:
        try:
            out, _ = pr.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(out.decode())
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0, "rank %d failed:\n%s" % (r, out[-3000:])
    return outs


@pytest.mark.parametrize("procs", [(2, 1, 1), (2, 2, 1), (2, 2, 2)])
@pytest.mark.parametrize("p", [2, 4])
def test_partition_tables(procs, p):
    nr = 2
    size = procs[0] * procs[1] * procs[2]
    cubes = [mg.Cube(p, n_refine=nr, box=procs, procs=procs, rank=r) for r in range(size)]
    orc = Oracle(p, n_refine=nr, box=procs)
    whole = mg.Cube(p, n_refine=nr, box=procs, numbering="cell")  # the same mesh on one rank
    for l in range(nr + 1):
        n_global = orc.n_dofs(l)
        assert whole.n_dofs(l) == n_global
        assert np.array_equal(whole.dof_grid(l), orc.dof_grid(l))
        np.testing.assert_allclose(whole.rhs(l), orc.rhs(l), rtol=1e-12, atol=1e-14)
        gids = [c.dof_grid(l) for c in cubes]
        # coverage and unique ownership
        count = np.zeros(n_global, dtype=int)
        owner = np.zeros(n_global, dtype=int)
        for r, c in enumerate(cubes):
            assert len(set(gids[r])) == gids[r].size
            count[gids[r]] += 1
            own = np.ones(c.n_dofs(l), dtype=int)
            own[c.not_owned(l)] = 0
            owner[gids[r]] += own
        assert (count >= 1).all() and (owner == 1).all()
        # neighbour lists: symmetric and identically ordered
        for r, c in enumerate(cubes):
            shared = set(c.shared(l).tolist())
            dup = set(np.nonzero(count[gids[r]] > 1)[0].tolist()) - set(c.constrained(l).tolist())
            assert shared == dup
            for (q, idx) in c.neighbors(l):
                back = dict(cubes[q].neighbors(l))[r]
                assert np.array_equal(gids[r][idx], gids[q][back])
        # the rhs assembled per rank sums to the global rhs
        ref_lex = np.empty(n_global)
        ref_lex[orc.dof_grid(l)] = orc.rhs(l)
        total = np.zeros(n_global)
        for r, c in enumerate(cubes):
            np.add.at(total, gids[r], c.rhs(l))
        np.testing.assert_allclose(total, ref_lex, rtol=1e-12, atol=1e-13)
        # boundary values and constrained sets agree with the single-domain provider
        ref_bc = np.zeros(n_global)
        bi, bv = whole.bc(l)
        ref_bc[whole.dof_grid(l)[bi]] = bv
        for r, c in enumerate(cubes):
            bi, bv = c.bc(l)
            loc = np.zeros(c.n_dofs(l))
            loc[bi] = bv
            assert np.array_equal(loc, ref_bc[gids[r]])
    for c in cubes:
        c.close()
    whole.close()
    orc.close()


def test_box_mesh_matches_oracle_single_rank():
    """the doubling-mesh family on one rank (2x1x1 coarse cells) against the oracle"""
    c = mg.Cube(3, n_refine=2, box=(2, 1, 1), numbering="cell")
    o = Oracle(3, n_refine=2, box=(2, 1, 1))
    for l in range(3):
        assert np.array_equal(c.idx27(l), o.idx27(l))
        assert np.array_equal(c.dof_grid(l), o.dof_grid(l))
        np.testing.assert_allclose(c.rhs(l), o.rhs(l), rtol=1e-12, atol=1e-14)
    assert c.cells_per_dim3(2) == ((8, 4, 4), (8, 4, 4))
    c.close()
    o.close()


@pytest.mark.parametrize("world", [2, 4])
def test_gloo_exchange_protocol(world):
    outs = launch("host", world, 3, 2)
    assert all("host ok" in o for o in outs)


@pytest.mark.parametrize("world", [2, 8])
def test_gloo_exchange_protocol_block_split_cube(world):
    """the strong-scaling layout (square mesh with n_subdiv = 2, block-split 2x1x1 / 2x2x2): assembled
    rhs and unique ownership on every level against the single-domain oracle"""
    outs = launch("host", world, 2, 2, extra=("strong",))
    assert all("host ok" in o for o in outs), outs


def test_gloo_exchange_protocol_shell_sector():
    """the poisson_shell slice block-split over two ranks: right-hand side of the mapped mesh with the variable
    coefficient's solution, summed over the rank interface, against the single-domain oracle"""
    outs = launch("host", 2, 3, 2, extra=("shell_sector",))
    assert all("host ok" in o for o in outs), outs


@pytest.mark.parametrize("world,n_coarse,p,nr", [(2, 6, 2, 2), (3, 6, 3, 1), (4, 12, 2, 1), (2, 12, 4, 1), (8, 12, 2, 1), (4, 6, 2, 1),
                                                  (8, 6, 2, 2), (8, 6, 3, 1), (8, 12, 2, 2), (5, 6, 2, 1)])
def test_gloo_exchange_protocol_hyper_shell(world, n_coarse, p, nr):
    """hyper_shell(6 | 12) over 2, 3, 4, 5 and 8 ranks: the coarse cells in equal shares where the rank count divides
    them, else the cells of level 1 (8 ranks: 6 of 48 / 12 of 96 each; 4 ranks on the six-cell shell: 12 of 48), else
    (5 ranks) uneven shares of coarse cells: equal cell counts wherever possible, local index tables address the same
    DoFs as the whole mesh's, the right-hand side summed over the rank interfaces equals the single-domain one,
    every DoF is owned exactly once"""
    outs = launch("", world, 0, 0, extra=("host", str(n_coarse), str(p), str(nr)), worker="shell_dist_worker.py")
    assert all("host ok" in o for o in outs), outs
