"""Independent (non-reference) checks of the CPU oracle: dense element matrix,
symmetry, definiteness, null space, R = P^T, polynomial reproduction, Chebyshev polynomial."""
import numpy as np
import pytest

from oracle import Oracle


def _lex(o, l, v):
    """vector in oracle numbering -> lexicographic grid"""
    g = o.dof_grid(l)
    out = np.zeros(g.size)
    out[g] = v
    return out


@pytest.mark.parametrize("p,ns,nr", [(1, 2, 0), (2, 1, 1), (3, 1, 1), (4, 1, 1), (4, 3, 0), (5, 1, 1),
                                      (8, 1, 1)])
def test_vmult_matches_dense_element_matrix(p, ns, nr):
    o = Oracle(p, ns, nr)
    l = o.max_level
    rng = np.random.default_rng(p * 100 + ns)
    x = rng.uniform(-1, 1, o.n_dofs(l))
    y = o.vmult(l, x)
    g = o.dof_grid(l)
    assert np.array_equal(np.sort(g), np.arange(g.size))  # numbering is a permutation of the grid
    ylex = o.vmult_dense_lex(l, _lex(o, l, x))
    np.testing.assert_allclose(y, ylex[g], rtol=0, atol=2e-12 * np.abs(ylex).max())
    o.close()


def test_numbering_contract():
    """entity-contiguous numbering with constrained DoFs last (laplace_operator.h:272-340,
    multigrid_solver.h:525,570-577)."""
    p = 4
    o = Oracle(p, 1, 2)
    l = 2
    idx = o.idx27(l)
    plain = o.idx27_plain(l)
    n_free = o.n_dofs(l) - o.n_constrained(l)
    assert (plain[idx == 0xFFFFFFFF] >= n_free).all()
    assert (idx[idx != 0xFFFFFFFF] < n_free).all()
    assert np.array_equal(o.constrained(l), np.arange(n_free, o.n_dofs(l)))
    # DoFs of the hex interior entity (e=13) are contiguous and lexicographic on the grid
    g = o.dof_grid(l)
    G = o.cells_per_dim(l) * p + 1
    cc = o.cell_coords(l)
    for c in (0, 7, 33):
        base = plain[c, 13]
        ids = g[base:base + (p - 1) ** 3].reshape(p - 1, p - 1, p - 1)
        X, Y, Z = cc[c]
        for k in range(p - 1):
            for j in range(p - 1):
                for i in range(p - 1):
                    assert ids[k, j, i] == ((Z * p + 1 + k) * G + Y * p + 1 + j) * G + X * p + 1 + i


@pytest.mark.parametrize("p", [2, 4, 7])
def test_symmetric_positive_definite(p):
    o = Oracle(p, 1, 1)
    l = 1
    rng = np.random.default_rng(1)
    x = rng.uniform(-1, 1, o.n_dofs(l))
    y = rng.uniform(-1, 1, o.n_dofs(l))
    assert np.dot(y, o.vmult(l, x)) == pytest.approx(np.dot(x, o.vmult(l, y)), rel=1e-12)
    assert np.dot(x, o.vmult(l, x)) > 0
    o.close()


def test_constants_in_null_space_before_bc():
    """With all entities unconstrained inside the domain, A * 1 vanishes at DoFs whose cell patch
    does not touch the Dirichlet boundary (the constrained columns are dropped)."""
    p = 4
    o = Oracle(p, 1, 2)
    l = 2
    y = _lex(o, l, o.vmult(l, np.ones(o.n_dofs(l))))
    G = o.cells_per_dim(l) * p + 1
    y = y.reshape(G, G, G)
    inner = y[p + 1:G - p - 1, p + 1:G - p - 1, p + 1:G - p - 1]
    assert np.abs(inner).max() < 1e-12


def test_restriction_is_transposed_prolongation():
    o = Oracle(4, 1, 2)
    rng = np.random.default_rng(5)
    for with_bc in (False, True):
        xc = rng.uniform(-1, 1, o.n_dofs(1))
        yf = rng.uniform(-1, 1, o.n_dofs(2))
        Pxc = o.prolongate(2, xc, with_bc=with_bc)
        Rty = o.restrict_and_add(2, np.zeros(o.n_dofs(1)), yf, with_bc=with_bc)
        assert np.dot(Pxc, yf) == pytest.approx(np.dot(xc, Rty), rel=1e-12)
    o.close()


def test_prolongation_reproduces_polynomials():
    p = 4
    o = Oracle(p, 1, 2)

    def poly(l):
        G = o.cells_per_dim(l) * p + 1
        h = o.cell_size(l)
        gl = o.gll()
        x1 = np.array([-0.9 + h * (min(i // p, o.cells_per_dim(l) - 1)
                                   + gl[i - p * min(i // p, o.cells_per_dim(l) - 1)]) for i in range(G)])
        g = o.dof_grid(l)
        gx, gy, gz = g % G, (g // G) % G, g // (G * G)
        x, y, z = x1[gx], x1[gy], x1[gz]
        return x ** 4 - 2 * x * y ** 3 + z ** 4 * y + 0.3 * x * y * z

    fine = o.prolongate(2, poly(1))
    np.testing.assert_allclose(fine, poly(2), atol=1e-13)
    # prolongate_and_add accumulates
    fine2 = o.prolongate(2, poly(1), fine=np.ones(o.n_dofs(2)))
    np.testing.assert_allclose(fine2, poly(2) + 1, atol=1e-13)
    o.close()


def test_convergence_order():
    """observed L2 rate ~ p+1 (README.md:144-159 columns after the errors)."""
    p = 3
    errs = []
    for nr in (2, 3):
        o = Oracle(p, 1, nr, degree=3, n_cycles=2)
        o.solve_cg()
        errs.append(o.l2_error())
        o.close()
    rate = np.log2(errs[0] / errs[1])
    assert 3.5 < rate < 4.6


def test_chebyshev_is_the_chebyshev_polynomial():
    """On the level operator, x = p(D^-1 A) D^-1 b; the error propagator 1 - lambda p(lambda)
    equals T_n((theta - lambda)/delta) / T_n(theta/delta) (first-kind polynomial).
    Check through the action on b: A-eigenvector content is not accessible matrix-free, so use
    linearity + the recurrence evaluated in numpy."""
    o = Oracle(3, 1, 1, degree=4)
    l = 1
    info = o.cheb_info(l)
    assert info["degree"] == 4
    rng = np.random.default_rng(0)
    b = rng.uniform(-1, 1, o.n_dofs(l))
    dinv = o.inv_diag(l)
    theta, delta = info["theta"], info["delta"]
    # numpy restatement of the recurrence
    x_old = np.zeros_like(b)
    x = dinv * b / theta
    rho, sigma = delta / theta, theta / delta
    for _ in range(info["degree"] - 1):
        rho_new = 1.0 / (2 * sigma - rho)
        f1, f2 = rho_new * rho, 2 * rho_new / delta
        rho = rho_new
        x, x_old = x + f1 * (x - x_old) + f2 * dinv * (b - o.vmult(l, x)), x
    np.testing.assert_allclose(o.cheb_vmult(l, b), x, rtol=1e-13, atol=1e-15)
    # step() from a nonzero start reduces the residual
    x0 = rng.uniform(-1, 1, o.n_dofs(l))
    x1 = o.cheb_step(l, x0, b)
    assert np.linalg.norm(b - o.vmult(l, x1)) < np.linalg.norm(b - o.vmult(l, x0))
    o.close()


def test_inverse_diagonal():
    p = 2
    o = Oracle(p, 1, 1)
    l = 1
    n = o.n_dofs(l)
    d = np.array([o.vmult(l, np.eye(1, n, i).ravel())[i] for i in range(n)])
    np.testing.assert_allclose(o.inv_diag(l), 1.0 / d, rtol=1e-12)
    o.close()


@pytest.mark.parametrize("degree", [2, 3, 5])
def test_fourth_kind_chebyshev_matches_the_closed_form(degree):
    """PolynomialType::fourth_kind (multigrid_solver.h:951-952; deal.II's recurrence, recalled) pinned
    against the closed form: after n steps from a zero guess the error is
    W_n(1 - 2 t / lambda_max) / (2n + 1) times the initial one on every eigenvector of D^-1 A with
    eigenvalue t, W_n the Chebyshev polynomial of the fourth kind, W_n(cos th) = sin((n+1/2) th) / sin(th/2)
    (Lottes, Optimal polynomial smoothers for multigrid V-cycles, 2023)."""
    import scipy.linalg
    o = Oracle(2, 1, 1, degree=degree, polynomial="fourth_kind")
    l, n = 1, o.n_dofs(1)
    A = np.empty((n, n))
    e = np.zeros(n)
    for i in range(n):
        e[i] = 1.0
        A[:, i] = o.vmult(l, e)
        e[i] = 0.0
    D = 1.0 / o.inv_diag(l)
    lam, V = scipy.linalg.eigh(A, np.diag(D))
    info = o.cheb_info(l)
    assert info["delta"] == info["lambda_max"] and info["degree"] == degree
    for k in (0, n // 3, n - 1):
        v = V[:, k]
        x = o.cheb_vmult(l, A @ v)
        th = np.arccos(1.0 - 2.0 * lam[k] / info["lambda_max"])
        expected = np.sin((degree + 0.5) * th) / np.sin(0.5 * th) / (2 * degree + 1)
        assert np.allclose(v - x, expected * v, atol=1e-11 * np.abs(v).max()), (k, lam[k])
    # step() from a non-zero guess reduces the error by the same polynomial
    v = V[:, n // 2]
    x0 = 0.3 * V[:, 1]
    x = o.cheb_step(l, x0, A @ v)
    c = np.linalg.solve(V, v - x)
    c0 = np.linalg.solve(V, v - x0)
    th = np.arccos(1.0 - 2.0 * lam / info["lambda_max"])
    assert np.allclose(c, np.sin((degree + 0.5) * th) / np.sin(0.5 * th) / (2 * degree + 1) * c0, atol=1e-10)
    o.close()
