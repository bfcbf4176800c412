"""The 2x2x2 decomposition of an 8-GPU run (bench.py --gpus 8, strong scaling layout) on the one GPU of the test box:
eight ranks as threads (tests/thread_ranks.py), every rank with its own context, cube and solver, exchanging through
the callback transport; the checks are those of the multi-process tests (tests/dist_worker.py): operator, smoother
parameters, V-cycle, FMG and PCG against the single-domain oracle on the same global mesh."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dg_dist_worker  # noqa: E402
import dg_mg_dist_worker  # noqa: E402
import dist_worker  # noqa: E402
import shell_dist_worker  # noqa: E402
from thread_ranks import run_ranks  # noqa: E402


def eight(p, nr, flags):
    lines = []
    run_ranks(8, lambda dist, r: dist_worker.run("gpu", p, nr, flags, dist, r, 8, say=lambda *a, **k: lines.append(a[0])))
    assert len(lines) == 8 and all("gpu ok" in s for s in lines), lines
    return lines


@pytest.mark.parametrize("p,nr", [(2, 3), (4, 2), (8, 2)])   # (8, 2): BASELINE config 3 on 8 ranks
def test_eight_ranks_block_split_cube(p, nr):
    eight(p, nr, ("strong",))


def test_eight_ranks_fused_transfers_and_device_rhs(monkeypatch):
    """eight-colour schedule on every level, no agglomeration: the fused residual + restriction / prolongation forms with
    the interface rows of DoFs shared by up to eight ranks; right-hand sides assembled on the device"""
    monkeypatch.setenv("MGX_FREE_ONE_MAX", "0")
    monkeypatch.setenv("MGX_FREE_MAX_BRICKS", "0")
    monkeypatch.setenv("MGX_AGGLOMERATE", "0")
    monkeypatch.setenv("MGX_OVERLAP_MIN_BRICKS", "1")
    monkeypatch.setenv("MGX_TEST_DEVICE_RHS", "1")
    eight(4, 3, ("strong",))


def test_eight_ranks_mixed_precision(monkeypatch):
    eight(3, 3, ("strong", "f32"))


@pytest.mark.parametrize("p,basis,steps,number", [(3, 0, 9, "f64"), (4, 1, 9, "f32")])
def test_eight_ranks_dg_operator(p, basis, steps, number):
    """DG-SIP operator with ghost cells on 2x2x2 ranks (cells with ghost neighbours in all three directions): operator,
    merged Chebyshev update and merged CG sums against the single-domain face-based oracle"""
    lines = []
    run_ranks(8, lambda dist, r: dg_dist_worker.run(p, basis, steps, number, dist, r, 8, say=lambda *a, **k: lines.append(a[0])))
    assert len(lines) == 8 and all("dg ok" in s for s in lines), lines


def test_eight_ranks_dg_multigrid():
    """MultigridSolverDG on 2x2x2 ranks: the merged residual + restriction to the decomposed FE_Q hierarchy below"""
    lines = []
    run_ranks(8, lambda dist, r: dg_mg_dist_worker.run(2, 2, 0, "f64", dist, r, 8, say=lambda *a, **k: lines.append(a[0])))
    assert len(lines) == 8 and all("dg multigrid ok" in s for s in lines), lines


def benchmark_problem(cells_log2):
    """what `bench.py --gpus 8 --cells 2^k` solves in its verification: PCG iterations and global L2 error"""
    def rank_body(dist, r):
        ctx = mg.Context(0)
        comm = mg.Communicator(ctx, dist)
        cube = mg.Cube(4, n_refine=cells_log2 - 1, box=(2, 2, 2), procs=(2, 2, 2), rank=r, origin=-0.9, h0=0.95)
        solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64, comm=comm, device_rhs=True)
        its, _ = solver.solve_cg()
        l2 = solver.compute_l2_error()
        agglomerated = solver.coarse is not None
        solver.close()
        cube.close()
        ctx.close()
        return its, l2, agglomerated
    return run_ranks(8, rank_body)


def test_eight_ranks_benchmark_problem_64_cubed():
    """FE_Q(4) on 64^3 cells block-split over 2x2x2 ranks (2.1 M DoFs and 512 bricks per rank: one-launch schedule on the
    finest level, agglomerated coarse levels): 8 PCG iterations to the README's L2 error (README.md:135-159)"""
    res = benchmark_problem(6)
    assert all(r[0] == 8 and r[2] for r in res), res
    assert all(abs(r[1] / 1.327e-8 - 1) < 5e-3 for r in res), res


@pytest.mark.parametrize("n_coarse,p,nr", [(12, 3, 1), (6, 2, 2), (12, 2, 2), (6, 4, 1)])
def test_eight_ranks_hyper_shell(n_coarse, p, nr):
    """BASELINE config 4 on eight ranks in EQUAL shares (the run of poisson_shell on 8 GPUs): the cells of level 1 of
    hyper_shell(6 | 12) dealt out, 6 of 48 / 12 of 96 per rank (the reference partitions the refined mesh cell by cell,
    poisson_shell/program.cc:249,274); the coarse cells live on the undecomposed copy of the coarse levels, from which
    the full multigrid cycle starts.  Variable coefficient with the 1e6 contrast; operator, diagonal, smoother parameters,
    V-cycle, FMG and PCG against the single-domain oracle on the whole shell."""
    lines = []
    run_ranks(8, lambda dist, r: shell_dist_worker.run("gpu", n_coarse, p, nr, "shell", dist, r, 8,
                                                      say=lambda *a, **k: lines.append(a[0])))
    assert len(lines) == 8 and all("gpu ok" in s and "agglomerated" in s for s in lines), lines
