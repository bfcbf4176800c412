"""CPU checks of the DG oracle (oracle/dg_oracle.py): the restated symmetric-interior-penalty form
is symmetric, positive definite and consistent (A u_h = (f, phi) for a polynomial solution that
vanishes on the boundary), in all three local bases of the reference
(common/laplace_operator_dg.h:343-348), on the sheared mesh of matvec_dg_cheby/program.cc:55-77.
The reference itself holds no golden vectors for this path; its check is cell-based against
face-based (matvec_dg/program.cc:206-207), which tests/test_gpu_dg.py repeats with the HIP kernel."""
import numpy as np
import pytest

from oracle import dg_oracle as dg


def test_hermite_like_basis_has_the_documented_properties():
    b = dg.hermite_like_basis(3)
    assert np.isclose(min(b[0].r), 2.0 / 7.0)                     # root of p_0 for degree 3
    for p in range(1, 10):
        b = dg.hermite_like_basis(p)
        xs = np.linspace(0, 1, 9)
        assert np.allclose(sum(f(xs) for f in b), 1.0, atol=1e-11)    # partition of unity
        vals0 = np.array([f(0.0) for f in b])
        ders0 = np.array([f.d(0.0) for f in b])
        assert np.allclose(vals0[1:], 0, atol=1e-12) and np.isclose(vals0[0], 1)
        if p >= 2:
            assert np.allclose(ders0[2:], 0, atol=1e-10)
        assert np.isclose(ders0[0], -ders0[1])                        # laplace_operator_dg.h:1190-1198
        if p >= 3:
            xq, wq = dg.gauss01(p + 2)
            assert abs(np.sum(wq * b[0](xq) * b[1](xq))) < 1e-14          # p_0 orthogonal to p_1
        for i, f in enumerate(b):                                     # mirror symmetry
            assert np.allclose(f(xs), b[p - i](1 - xs), atol=1e-11)


def test_cheby_mesh_follows_the_harness():
    cells, J = dg.cheby_mesh(4)
    assert cells == (4, 2, 2)
    assert np.allclose(J[:, 0], np.array([1.12, 0.24, 0.36]) * (0.95 + 0.95) / 4)
    cells, _ = dg.cheby_mesh(9)
    assert cells == (8, 8, 8)
    cells, _ = dg.cheby_mesh(11)
    assert cells == (16, 16, 8)


@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS_LOBATTO, dg.GAUSS])
@pytest.mark.parametrize("p", [2, 3, 4])
def test_sip_form_is_symmetric_definite_and_consistent(kind, p):
    _, J = dg.cheby_mesh(4)
    o = dg.DGOracle(p, kind, (2, 3, 2), J)
    A = o.dense_matrix()
    assert abs(A - A.T).max() < 1e-13 * abs(A).max()
    assert np.linalg.eigvalsh(0.5 * (A + A.T))[0] > 0
    N = np.array(o.cells, float)
    inv = np.linalg.inv(o.J)
    G = inv @ inv.T

    def u(x):
        xi = x @ inv.T
        return np.prod(xi * (N - xi), axis=1)

    def f(x):  # -Laplace u
        xi = x @ inv.T
        w, dw = xi * (N - xi), N - 2 * xi
        lap = 0
        for a in range(3):
            for c in range(3):
                if a == c:
                    t = -2.0 * np.prod(np.delete(w, a, axis=1), axis=1)
                else:
                    t = dw[:, a] * dw[:, c] * w[:, 3 - a - c]
                lap = lap + G[a, c] * t
        return -lap

    rhs = o.load_vector(f)
    assert abs(o.vmult(o.interpolate(u)) - rhs).max() < 1e-11 * abs(rhs).max()


@pytest.mark.parametrize("kind", [dg.HERMITE, dg.GAUSS])
def test_operator_is_the_same_in_every_basis(kind):
    """A_basis = B^T A_gauss B with B the change of basis (values in the Gauss points)"""
    p = 3
    _, J = dg.cheby_mesh(3)
    o, g = dg.DGOracle(p, kind, (2, 2, 2), J), dg.DGOracle(p, dg.GAUSS, (2, 2, 2), J)
    x = np.random.default_rng(1).standard_normal(o.shape)
    B = dg.kron3(o.S, o.S, o.S)
    assert np.allclose(o.vmult(x), g.vmult(x @ B.T) @ B, rtol=1e-11, atol=1e-11)


def test_block_jacobi_is_the_inverse_of_the_transformed_diagonal():
    p = 2
    _, J = dg.cheby_mesh(3)
    o = dg.DGOracle(p, dg.HERMITE, (2, 2, 2), J)
    # eigenvectors are mass-orthonormal
    mass = o.S.T @ (o.wq[:, None] * o.S)
    assert np.allclose(o.T.T @ mass @ o.T, np.eye(p + 1), atol=1e-12)
    A = o.dense_matrix()
    n3 = (p + 1) ** 3
    blk = A[:n3, :n3]  # cell (0,0,0): lower faces Dirichlet, upper ones interior
    assert np.allclose(blk, o.own_block([True] * 3, [False] * 3), atol=1e-12)
    r = np.zeros(o.shape)
    r[0, 0, 0] = np.random.default_rng(2).standard_normal(n3)
    z = o.jacobi_vmult(r)[0, 0, 0]
    d = np.diag(o.T3.T @ blk @ o.T3)
    assert np.allclose(z, o.T3 @ ((o.T3.T @ r[0, 0, 0]) / d), rtol=1e-12)
    # the update formulas of laplace_operator_dg.h:1839-1860
    rng = np.random.default_rng(3)
    rhs, x, xo = (rng.standard_normal(o.shape) for _ in range(3))
    new, old = o.vmult_with_chebyshev_update(rhs, 2, 0.6, 0.2, x, xo)
    assert old is not None and np.allclose(old, x)
    assert np.allclose(new, 0.2 * o.jacobi_vmult(rhs - o.vmult(x)) + 1.6 * x - 0.6 * xo)
    new1, _ = o.vmult_with_chebyshev_update(rhs, 1, 0.6, 0.2, x, xo)
    assert np.allclose(new1, 0.2 * o.jacobi_vmult(rhs - o.vmult(x)) + 1.6 * x)
    new0, _ = o.vmult_with_chebyshev_update(rhs, 0, 0.6, 0.2, x, xo)
    assert np.allclose(new0, 0.2 * o.jacobi_vmult(rhs))


def test_partition_of_a_box_is_consistent():
    multigrid_amd = pytest.importorskip("multigrid_amd")
    cells, procs = (8, 4, 4), (2, 2, 2)
    parts = [multigrid_amd.dg_box_partition(cells, procs, r) for r in range(8)]
    assert sum(len(q["ijk"]) for q in parts) == 128
    for r, q in enumerate(parts):
        assert q["neighbours"].max() < len(q["ijk"]) + q["n_ghost"]
        for (rk, send, first, cnt) in q["exchange"]:
            back = [e for e in parts[rk]["exchange"] if e[0] == r]
            assert len(back) == 1 and back[0][3] == cnt and len(send) == cnt
