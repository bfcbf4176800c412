"""DG path on the GPU (include/mgx_dg.h) against the face-based oracle (oracle/dg_oracle.py): the
check the reference makes of its cell-based operator (matvec_dg/program.cc:206-207), for the three
local bases, fp64 and fp32 (the number type of matvec_dg_cheby/program.cc:88), the block-Jacobi
preconditioner in the eigenvector basis and the merged Chebyshev update."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")
from oracle import dg_oracle as dg  # noqa: E402

TOL = {mg.F64: 2e-11, mg.F32: 3e-5}


class Case:
    """oracle and HIP operator on the same box of cells; vectors travel in the oracle's layout"""

    def __init__(self, ctx, p, kind, cells, jac, number, ordering="z"):
        self.ctx, self.number = ctx, number
        self.orc = dg.DGOracle(p, kind, cells, jac)
        nb, ijk = mg.dg_box_neighbours(cells, ordering)
        self.ijk = ijk
        self.op = mg.DGLaplaceOperator(ctx, p, kind, nb, jac, number)
        assert self.op.m() == int(np.prod(self.orc.shape))

    def up(self, x):
        """oracle array [nz, ny, nx, n^3] -> device vector in the operator's cell order"""
        i = self.ijk
        return self.op.initialize_dof_vector(x[i[:, 2], i[:, 1], i[:, 0]].ravel())

    def down(self, v):
        out = np.empty(self.orc.shape)
        i = self.ijk
        out[i[:, 2], i[:, 1], i[:, 0]] = v.download().astype(float).reshape(len(i), -1)
        return out

    def close(self):
        self.op.clear()


def rel(a, b):
    return abs(a - b).max() / abs(b).max()


@pytest.fixture(scope="module")
def ctx():
    c = mg.Context(0)
    yield c
    c.close()


MESHES = [((2, 3, 2), 4), ((4, 2, 2), 4), ((1, 1, 1), 0), ((3, 1, 2), 5)]


@pytest.mark.parametrize("number", [mg.F64, mg.F32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS_LOBATTO, dg.GAUSS], ids=["hermite", "gl", "gauss"])
@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 6])
def test_cell_based_operator_equals_face_based_oracle(ctx, p, kind, number):
    rng = np.random.default_rng(100 * p + kind)
    for cells, steps in MESHES[: 4 if p <= 4 else 2]:
        _, jac = dg.cheby_mesh(steps)
        case = Case(ctx, p, kind, cells, jac, number)
        x = rng.standard_normal(case.orc.shape)
        src, dst = case.up(x), case.op.initialize_dof_vector()
        case.op.vmult(dst, src)
        assert rel(case.down(dst), case.orc.vmult(x)) < TOL[number], (cells, p, kind)
        rhs = rng.standard_normal(case.orc.shape)
        case.op.vmult_residual(dst, case.up(rhs), src)
        assert rel(case.down(dst), rhs - case.orc.vmult(x)) < TOL[number]
        case.close()


@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS_LOBATTO, dg.GAUSS], ids=["hermite", "gl", "gauss"])
@pytest.mark.parametrize("p", [7, 8, 9])
def test_high_degrees(ctx, p, kind):
    _, jac = dg.cheby_mesh(3)
    case = Case(ctx, p, kind, (2, 2, 1), jac, mg.F64)
    x = np.random.default_rng(p).standard_normal(case.orc.shape)
    src, dst = case.up(x), case.op.initialize_dof_vector()
    case.op.vmult(dst, src)
    assert rel(case.down(dst), case.orc.vmult(x)) < 1e-9
    case.op.jacobi_vmult(dst, src)
    assert rel(case.down(dst), case.orc.jacobi_vmult(x)) < 1e-9
    case.close()


@pytest.mark.parametrize("ordering", ["z", "lexicographic"])
def test_harness_mesh_and_cell_orderings(ctx, ordering):
    """matvec_dg_cheby mesh after 7 steps (8 x 4 x 4 cells), FE_DGQHermite(4), both cell orders"""
    cells, jac = mg.dg_cheby_mesh(7)
    ocells, ojac = dg.cheby_mesh(7)
    assert cells == ocells and np.allclose(jac, ojac, rtol=1e-14)
    case = Case(ctx, 4, dg.HERMITE, cells, jac, mg.F64, ordering)
    x = np.random.default_rng(5).standard_normal(case.orc.shape)
    src, dst = case.up(x), case.op.initialize_dof_vector()
    case.op.vmult(dst, src)
    assert rel(case.down(dst), case.orc.vmult(x)) < TOL[mg.F64]
    case.close()


@pytest.mark.parametrize("number", [mg.F64, mg.F32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS_LOBATTO, dg.GAUSS], ids=["hermite", "gl", "gauss"])
@pytest.mark.parametrize("p", [2, 3, 4, 5])
def test_block_jacobi_and_merged_chebyshev_update(ctx, p, kind, number):
    cells, jac = dg.cheby_mesh(5)   # 4 x 4 x 2: cells with 0 .. 3 Dirichlet faces
    case = Case(ctx, p, kind, cells, jac, number)
    o = case.orc
    info = case.op.info()
    assert np.isclose(info["hermite_derivative_on_face"], dg.hermite_like_basis(p)[0].d(0.0)) or kind != dg.HERMITE
    assert np.allclose(info["eigenvalues_1d"], o.eigenvalues_1d, rtol=1e-9)
    inv = np.linalg.inv(jac)
    assert np.allclose(info["penalty"], (p + 1) ** 2 * np.linalg.norm(inv, axis=1))  # laplace_operator_dg.h:789-793
    rng = np.random.default_rng(7 + p)
    rhs, x, xo = (rng.standard_normal(o.shape) for _ in range(3))
    tol = TOL[number] * 5
    d = case.op.initialize_dof_vector()
    case.op.jacobi_vmult(d, case.up(rhs))
    assert rel(case.down(d), o.jacobi_vmult(rhs)) < tol
    for idx in (0, 1, 2):   # laplace_operator_dg.h:910-955
        sol, old, b = case.up(x), case.up(xo), case.up(rhs)
        case.op.vmult_with_chebyshev_update(b, idx, 0.6, 0.2, sol, old)
        new_ref, old_ref = o.vmult_with_chebyshev_update(rhs, idx, 0.6, 0.2, x, xo)
        assert rel(case.down(sol), new_ref) < tol, idx
        assert rel(case.down(old), old_ref) < tol, idx
    # the loop of the harness (matvec_dg_cheby/program.cc:116-121): update, then swap output / input
    out, inp, b = case.up(x), case.up(xo), case.up(rhs)
    ro, ri = x, xo
    for _ in range(3):
        case.op.vmult_with_chebyshev_update(b, 2, 0.6, 0.2, out, inp)
        out, inp = inp, out
        ro, ri = o.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, ro, ri)
        ro, ri = ri, ro
    assert rel(case.down(out), ro) < tol * 10 and rel(case.down(inp), ri) < tol * 10
    case.close()


@pytest.mark.parametrize("number", [mg.F64, mg.F32], ids=["f64", "f32"])
@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS_LOBATTO, dg.GAUSS], ids=["hermite", "gl", "gauss"])
@pytest.mark.parametrize("p", [2, 3, 4, 6])
def test_merged_cg_update(ctx, p, kind, number):
    """vmult_with_cg_update (laplace_operator_dg.h:863-908, action 2): the vector updates, q = A p and the four
    sums of the cell kernel's epilogue; then the CG loop built on it converges like the textbook one"""
    cells, jac = dg.cheby_mesh(5)
    case = Case(ctx, p, kind, cells, jac, number)
    o = case.orc
    rng = np.random.default_rng(11 + p)
    r, q, pv, x = (rng.standard_normal(o.shape) for _ in range(4))
    tol = TOL[number] * 5
    for alpha, beta in ((0.0, 0.0), (0.37, 0.81)):
        R, Q, Pv, X = case.up(r), case.up(q), case.up(pv), case.up(x)
        sums = case.op.vmult_with_cg_update(alpha, beta, R, Q, Pv, X)
        if alpha == 0.0:
            x_ref, p_ref = x, q
        else:
            x_ref, p_ref = x + alpha * pv, beta * pv + q
        q_ref = o.vmult(p_ref)
        assert rel(case.down(X), x_ref) < tol and rel(case.down(Pv), p_ref) < tol
        assert rel(case.down(Q), q_ref) < tol
        ref = np.array([(q_ref * p_ref).sum(), (r * r).sum(), (q_ref * r).sum(), (q_ref * q_ref).sum()])
        assert np.allclose(sums, ref, rtol=tol * 20, atol=tol * 20 * abs(ref).max())
        again = case.op.vmult_with_cg_update(alpha, beta, case.up(r), case.up(q), case.up(pv), case.up(x))
        assert np.array_equal(again, sums)   # block sums added in a fixed order
    case.close()


def test_operator_properties_at_benchmark_size(ctx):
    """FE_DGQHermite(4) on the harness mesh after 15 steps (32^3 cells, 4.1 M DoFs, fp32): symmetry,
    definiteness and reproducibility -- the oracle cannot run this size in seconds"""
    cells, jac = mg.dg_cheby_mesh(15)
    nb, _ = mg.dg_box_neighbours(cells)
    op = mg.DGLaplaceOperator(ctx, 4, mg.DG_HERMITE, nb, jac, mg.F32)
    n = op.m()
    assert n == 32 ** 3 * 125
    rng = np.random.default_rng(0)
    x, y = (op.initialize_dof_vector(rng.standard_normal(n)) for _ in range(2))
    ax, ay, t = (op.initialize_dof_vector() for _ in range(3))
    op.vmult(ax, x)
    op.vmult(ay, y)
    yax, xay, xax = ctx.dot(y, ax), ctx.dot(x, ay), ctx.dot(x, ax)
    assert yax == pytest.approx(xay, rel=2e-4, abs=2e-5 * xax)
    assert xax > 0
    op.vmult(t, x)
    assert np.array_equal(t.download(), ax.download())
    op.clear()


def test_errors_are_reported():
    c = mg.Context(0)
    nb, _ = mg.dg_box_neighbours((2, 2, 2))
    with pytest.raises(mg.MgxError):
        mg.DGLaplaceOperator(c, 12, 0, nb, np.eye(3))
    with pytest.raises(mg.MgxError):
        mg.DGLaplaceOperator(c, 3, 0, nb, np.zeros((3, 3)))
    bad = nb.copy()
    bad[0, 1] = 99
    with pytest.raises(mg.MgxError):
        mg.DGLaplaceOperator(c, 3, 0, bad, np.eye(3))
    op = mg.DGLaplaceOperator(c, 3, 0, nb, np.eye(3))
    v = op.initialize_dof_vector()
    with pytest.raises(mg.MgxError):
        op.vmult(v, v)
    op.clear()
    c.close()


@pytest.mark.parametrize("world,p,basis,steps,number", [(2, 4, 0, 7, "f64"), (4, 3, 0, 7, "f64"), (2, 2, 2, 6, "f64"),
                                                        (4, 4, 1, 8, "f32"), (2, 3, 0, 5, "f32")])
def test_decomposed_operator_matches_single_domain_oracle(world, p, basis, steps, number):
    """block decomposition with ghost cells (mgx_dg_exchange_desc; the reference's MPI face exchange,
    laplace_operator_dg.h:986-1057): 2 and 4 ranks over gloo sharing the GPU"""
    from test_decomposition import launch
    outs = launch(None, world, None, None, extra=(str(p), str(basis), str(steps), number), worker="dg_dist_worker.py")
    assert all("dg ok" in o for o in outs), outs

