"""MultigridSolverDG on the GPU (include/mgx_dg.h: the DG level on top of the FE_Q hierarchy,
common/multigrid_solver_dg.h:55-747) against its restatement (oracle/dg_oracle.py DGMultigridOracle on
top of the C oracle's FE_Q multigrid): transfers DG <-> FE_Q, eigenvalue estimate of the block-Jacobi
Chebyshev smoother, one DG V-cycle, and the V-cycle-preconditioned CG of poisson_dg/program.cc."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")
from oracle import Oracle, dg_oracle as dg  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    c = mg.Context(0)
    yield c
    c.close()


class Pair:
    def __init__(self, ctx, p, nr, basis, number, degree_pre=3):
        self.ctx = ctx
        self.cube = mg.Cube(p, 1, nr)
        self.solver = mg.DGMultigridSolver(ctx, self.cube, basis, degree_pre, number)
        n1 = 2 ** nr
        self.dgo = dg.DGOracle(p, basis, (n1, n1, n1), np.eye(3) * self.cube.cell_size(nr))
        self.fe = Oracle(p, 1, nr, degree=degree_pre, vfloat=(number == mg.F32))
        self.ijk = self.solver.cell_ijk
        n = self.solver.m()
        # deal.II: (global DoF index mod 11) - mean; the global index of a DG DoF is taken from the
        # lexicographic position of its cell, which is the oracle's own layout
        start = (np.arange(n) % 11).astype(float)
        start -= start.mean()
        self.orc = dg.DGMultigridOracle(self.dgo, self.fe, degree_pre, start.reshape(self.dgo.shape))
        # FE_Q vectors: provider numbering <-> oracle numbering through the lexicographic grid
        l = self.cube.max_level
        gc, go = self.cube.dof_grid(l), self.fe.dof_grid(l)
        pos = np.empty(go.size, np.int64)
        pos[go] = np.arange(go.size)
        self.cg_to_oracle = pos[gc]   # oracle index of provider DoF i

    def to_oracle(self, v):
        out = np.empty(self.dgo.shape)
        i = self.ijk
        out[i[:, 2], i[:, 1], i[:, 0]] = np.asarray(v, dtype=float).reshape(len(i), -1)
        return out

    def to_product(self, a):
        i = self.ijk
        return a[i[:, 2], i[:, 1], i[:, 0]].ravel()

    def close(self):
        self.solver.close()
        self.cube.close()
        self.fe.close()


def rel(a, b):
    return abs(a - b).max() / abs(b).max()


@pytest.mark.parametrize("number,tol", [(mg.F64, 1e-11), (mg.F32, 2e-5)], ids=["f64", "f32"])
@pytest.mark.parametrize("p,nr,basis", [(2, 2, 0), (3, 2, 0), (4, 2, 0), (3, 2, 1), (3, 2, 2), (1, 3, 0), (5, 1, 0)])
def test_transfers_between_dg_and_fe_q(ctx, p, nr, basis, number, tol):
    P = Pair(ctx, p, nr, basis, number)
    rng = np.random.default_rng(p + nr)
    l = P.cube.max_level
    r = rng.standard_normal(P.dgo.shape)
    src = ctx.vector(P.solver.m(), number, P.to_product(r))
    cg = ctx.vector(P.cube.n_dofs(l), number, np.full(P.cube.n_dofs(l), 7.0))
    P.solver.restrict_to_cg(cg, src)
    ref = P.orc.restrict_to_cg(r)[P.cg_to_oracle]
    assert rel(cg.download().astype(float), ref) < tol
    c = rng.standard_normal(P.cube.n_dofs(l))
    c_or = np.empty_like(c)
    c_or[P.cg_to_oracle] = c
    d0 = rng.standard_normal(P.dgo.shape)
    dst = ctx.vector(P.solver.m(), number, P.to_product(d0))
    P.solver.prolongate_add_cg_to_dg(dst, ctx.vector(c.size, number, c))
    ref = d0 + P.orc.prolongate_cg_to_dg(c_or)
    assert rel(P.to_oracle(dst.download()), ref) < tol
    P.close()


@pytest.mark.parametrize("number,tol", [(mg.F64, 1e-10), (mg.F32, 1e-4)], ids=["f64", "f32"])
@pytest.mark.parametrize("p,nr,basis", [(2, 2, 0), (3, 2, 0), (4, 2, 0), (3, 2, 1), (3, 2, 2), (1, 3, 0), (5, 1, 0), (3, 3, 0)])
def test_merged_residual_and_restriction(ctx, p, nr, basis, number, tol):
    """vmult_residual_and_restrict_to_cg (laplace_operator_dg.h:852-861, action 1) against the two steps of the
    oracle, and bit for bit against itself (eight colours of cells, plain adds)"""
    P = Pair(ctx, p, nr, basis, number)
    rng = np.random.default_rng(3 * p + nr)
    l = P.cube.max_level
    rhs, lhs = rng.standard_normal(P.dgo.shape), rng.standard_normal(P.dgo.shape)
    b, x = (ctx.vector(P.solver.m(), number, P.to_product(a)) for a in (rhs, lhs))
    cg = ctx.vector(P.cube.n_dofs(l), number, np.full(P.cube.n_dofs(l), 7.0))
    P.solver.vmult_residual_and_restrict_to_cg(cg, b, x)
    ref = P.orc.restrict_to_cg(rhs - P.dgo.vmult(lhs))[P.cg_to_oracle]
    got = cg.download()
    assert rel(got.astype(float), ref) < tol
    # the unmerged pair of kernels gives the same vector up to rounding
    t = ctx.vector(P.solver.m(), number)
    P.solver.matrix_dg.vmult_residual(t, b, x)
    cg2 = ctx.vector(P.cube.n_dofs(l), number)
    P.solver.restrict_to_cg(cg2, t)
    assert rel(got.astype(float), cg2.download().astype(float)) < (1e-12 if number == mg.F64 else 1e-5)
    if P.cube.n_cells(l) >= 64:   # eight colours of cells; atomics below
        P.solver.vmult_residual_and_restrict_to_cg(cg, b, x)
        assert np.array_equal(cg.download(), got)
    P.close()


@pytest.mark.parametrize("p,nr,basis", [(2, 2, 0), (3, 2, 0), (4, 2, 0), (3, 3, 0), (3, 2, 2), (2, 2, 1)])
def test_dg_v_cycle_and_pcg_fp64(ctx, p, nr, basis):
    P = Pair(ctx, p, nr, basis, mg.F64)
    info = P.solver.smoother_info()
    assert info["cg_its"] == P.orc.cg_its and info["degree"] == 3
    assert info["lambda_max"] == pytest.approx(P.orc.lambda_max, rel=1e-8)
    assert info["theta"] == pytest.approx(P.orc.theta, rel=1e-8)
    rng = np.random.default_rng(7)
    x = rng.standard_normal(P.dgo.shape)
    src, dst = ctx.vector(P.solver.m(), data=P.to_product(x)), ctx.vector(P.solver.m())
    for _ in range(3):  # eager, coarse levels captured, replayed
        P.solver.vmult(dst, src)
        assert rel(P.to_oracle(dst.download()), P.orc.v_cycle(x)) < 1e-8
    rhs = rng.standard_normal(P.dgo.shape)
    b, sol = ctx.vector(P.solver.m(), data=P.to_product(rhs)), ctx.vector(P.solver.m())
    its, red = P.solver.solve_cg(b, sol, 1e-9)
    xo, oits, ored = P.orc.solve_cg(rhs, 1e-9)
    assert its == oits and red == pytest.approx(ored, rel=1e-5)
    assert rel(P.to_oracle(sol.download()), xo) < 1e-7
    # the solution solves the DG system
    res = ctx.vector(P.solver.m())
    P.solver.matrix_dg_dp.vmult_residual(res, b, sol)
    assert ctx.l2_norm(res) < 2e-9 * ctx.l2_norm(b)
    P.close()


@pytest.mark.parametrize("p,nr,basis", [(3, 2, 0), (4, 2, 0), (2, 3, 0), (3, 2, 2)])
def test_dg_v_cycle_and_pcg_mixed_precision(ctx, p, nr, basis):
    """the reference's default: fp32 V-cycle inside the fp64 CG (poisson_dg/program.cc:72-73)"""
    P = Pair(ctx, p, nr, basis, mg.F32)
    info = P.solver.smoother_info()
    assert info["lambda_max"] == pytest.approx(P.orc.lambda_max, rel=1e-4)
    rng = np.random.default_rng(9)
    x = rng.standard_normal(P.dgo.shape)
    src, dst = ctx.vector(P.solver.m(), data=P.to_product(x)), ctx.vector(P.solver.m())
    P.solver.vmult(dst, src)
    assert rel(P.to_oracle(dst.download()), P.orc.v_cycle(x)) < 5e-4
    rhs = rng.standard_normal(P.dgo.shape)
    b, sol = ctx.vector(P.solver.m(), data=P.to_product(rhs)), ctx.vector(P.solver.m())
    its, red = P.solver.solve_cg(b, sol, 1e-9)
    xo, oits, ored = P.orc.solve_cg(rhs, 1e-9)
    assert abs(its - oits) <= 1 and red == pytest.approx(ored, rel=0.05)
    assert rel(P.to_oracle(sol.download()), xo) < 1e-6
    P.close()


def test_poisson_dg_harness_runs():
    """tools/poisson_dg.py (poisson_dg/program.cc): converges in a mesh-independent number of iterations
    and prints the reference's table row"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rows = []
    for nr in (2, 3):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "poisson_dg.py"), "3", str(nr)], cwd=root,
                             capture_output=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        lines = out.stdout.decode().strip().splitlines()
        assert lines[-2].split() == "cells dofs mv_outer mv_inner cg_L2error cg_time cg_its cg_reduction".split()
        rows.append(lines[-1].split())
    assert int(rows[0][0]) == 64 and int(rows[1][0]) == 512 and int(rows[1][1]) == 512 * 64
    assert 5 <= int(rows[0][6]) <= 10 and 5 <= int(rows[1][6]) <= 10


def test_poisson_dg_discretisation_error_converges():
    """the solve itself, checked end to end: for a solution that vanishes on the boundary (the benchmark's
    own does not, see tools/poisson_dg.py) the L2 error of FE_DGQHermite(3) falls with h^4"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    err = []
    for nr in (3, 4):
        out = subprocess.run([sys.executable, os.path.join(root, "tools", "poisson_dg.py"), "3", str(nr), "--solution",
                              "vanishing", "--vcycle", "f64"], cwd=root, capture_output=True, timeout=600)
        assert out.returncode == 0, out.stderr[-2000:]
        err.append(float(out.stdout.decode().strip().splitlines()[-1].split()[4]))
    assert err[0] < 2e-3 and 11 < err[0] / err[1] < 20, err


@pytest.mark.parametrize("world,p,nr,basis,number", [(2, 3, 2, 0, "f64"), (4, 2, 2, 0, "f64"), (2, 3, 2, 2, "f32")])
def test_decomposed_dg_multigrid_matches_single_domain_oracle(world, p, nr, basis, number):
    """DG level with ghost cells on top of the decomposed FE_Q hierarchy: 2 and 4 ranks over gloo on
    the one GPU, smoother parameters, V-cycle and CG against the single-domain restatement"""
    from test_decomposition import launch
    outs = launch(None, world, None, None, extra=(str(p), str(nr), str(basis), number), worker="dg_mg_dist_worker.py")
    assert all("dg multigrid ok" in o for o in outs), outs
