"""CPU restatement (numpy, dense) of the reference's symmetric-interior-penalty DG Laplace operator,
its block-Jacobi preconditioner in the eigenvector basis and the merged Chebyshev update.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the cpu_baseline
leg of bench tools -- never from the product package (multigrid_amd).

What is restated, and from where
  * the bilinear form, written FACE BY FACE exactly as the reference's verification operator
    does (common/laplace_operator_dg_face.h:66-84 cell term, :86-128 interior faces with
    sigma_F = (p+1)^2 * 1/2 (|n.J^-1|_- + |n.J^-1|_+), :131-160 Dirichlet faces with
    sigma_F = 2 (p+1)^2 |n.J^-1|).  The reference checks its fast cell-based operator
    (common/laplace_operator_dg.h:1110-1861) against that face-based one
    (matvec_dg/program.cc:206-207); tests/ do the same with the HIP kernel.
  * JacobiTransformed (laplace_operator_dg.h:2028-2256): per cell, T D^-1 T^T with T the tensor
    product of the 1D generalised eigenvectors of (Laplace + penalty, mass) on the unit cell
    (:179-215, LAPACK sygv = scipy.linalg.eigh here) and D the diagonal of T^T A_KK T, A_KK being
    the cell's own block: cell term + its 6 faces with the neighbour set to zero on interior faces
    (factor 1/2) and mirrored on Dirichlet faces (:1895-1968).
  * the merged Chebyshev step (:910-955, epilogue :1839-1860) and the matvec_dg_cheby mesh
    (matvec_dg_cheby/program.cc:55-77).

Parity status.  The reference holds no golden vectors for this path and deal.II cannot be built
here, so the operator is pinned the way the reference pins it (cell-based against face-based) plus
basis-independent properties (symmetry, definiteness, consistency with -Laplace u on polynomials).
The three local bases are deal.II's (external): FE_DGQArbitraryNodes(QGauss) and FE_DGQ are
Lagrange bases in the Gauss / Gauss-Lobatto points; FE_DGQHermite is restated from the documented
construction of Polynomials::HermiteLikeInterpolation (recalled, see hermite_like_basis) -- the
coefficient VALUES in that basis are "parity unpinned"; the operator it represents is not.
"""
import numpy as np
import scipy.linalg
import scipy.special

HERMITE, GAUSS_LOBATTO, GAUSS = 0, 1, 2
PENALTY_FACTOR = 1.0  # laplace_operator_dg.h:47


def gauss01(n):
    x, w = np.polynomial.legendre.leggauss(n)
    return 0.5 * (x + 1.0), 0.5 * w


def gauss_lobatto01(n):
    if n == 1:
        return np.array([0.5])
    if n == 2:
        return np.array([0.0, 1.0])
    # interior nodes: roots of P'_{n-1}, i.e. of the Jacobi polynomial P^{(1,1)}_{n-2}
    xi, _ = scipy.special.roots_jacobi(n - 2, 1.0, 1.0)
    return np.concatenate(([0.0], 0.5 * (xi + 1.0), [1.0]))


class RootPoly:
    """c * prod (x - r_k): evaluated in product form (monomial coefficients of the degree-9 bases
    reach 1e7 and would cost six digits)"""

    def __init__(self, roots, c=1.0):
        self.r = np.asarray(roots, dtype=float).reshape(-1)
        self.c = float(c)

    def __call__(self, x):
        x = np.asarray(x, dtype=float)
        return self.c * np.prod(x[..., None] - self.r, axis=-1) if self.r.size else self.c * np.ones_like(x)

    def d(self, x):
        """first derivative"""
        x = np.asarray(x, dtype=float)
        out = np.zeros_like(x)
        for m in range(self.r.size):
            out = out + np.prod(x[..., None] - np.delete(self.r, m), axis=-1)
        return self.c * out

    def scaled_to(self, x, value):
        return RootPoly(self.r, self.c * value / self(x))

    def mirrored(self):
        """f(1 - x)"""
        return RootPoly(1.0 - self.r, self.c * (-1.0) ** self.r.size)


def lagrange_basis(nodes):
    nodes = np.asarray(nodes, dtype=float)
    return [RootPoly(np.delete(nodes, i)).scaled_to(xi, 1.0) for i, xi in enumerate(nodes)]


def hermite_like_basis(p):
    """FE_DGQHermite's 1D polynomials (deal.II Polynomials::HermiteLikeInterpolation, external,
    restated from its documentation): p_0 is the only function with a value at x = 0, p_0 and p_1
    the only ones with a derivative there (mirror image at x = 1); p_0 is L2-orthogonal to p_1
    (root at 2/7 for degree 3); the inner functions are Lagrange polynomials in the roots of the
    Jacobi polynomial P^{(4,4)}_{p-3} times x^2 (1-x)^2; the functions sum to one, whence
    p_1'(0) = -p_0'(0) -- the property the reference uses at laplace_operator_dg.h:1190-1198.
    Degree 1: hat functions, degree 2: Bernstein polynomials."""
    if p == 0:
        return [RootPoly([])]
    if p == 1:
        return [RootPoly([1.0], -1.0), RootPoly([0.0])]
    if p == 2:
        return [RootPoly([1.0, 1.0]), RootPoly([0.0, 1.0], -2.0), RootPoly([0.0, 0.0])]
    if p > 3:
        xi, _ = scipy.special.roots_jacobi(p - 3, 4.0, 4.0)
        inner = 0.5 * (xi + 1.0)
    else:
        inner = np.zeros(0)
    q0 = RootPoly(np.concatenate(([1.0, 1.0], inner)))        # p_0 = c (x - r) q0
    q1 = RootPoly(np.concatenate(([0.0, 1.0, 1.0], inner)))   # shape of p_1
    xq, wq = gauss01(p + 2)
    r = np.sum(wq * xq * q0(xq) * q1(xq)) / np.sum(wq * q0(xq) * q1(xq))  # int (x - r) q0 q1 = 0
    p0 = RootPoly(np.concatenate((q0.r, [r]))).scaled_to(0.0, 1.0)
    p1 = RootPoly(q1.r, -p0.d(0.0) / q1.d(0.0))
    basis = [p0, p1]
    for j, xj in enumerate(inner):
        basis.append(RootPoly(np.concatenate(([0.0, 0.0, 1.0, 1.0], np.delete(inner, j)))).scaled_to(xj, 1.0))
    basis += [p1.mirrored(), p0.mirrored()]
    return basis


def basis_1d(p, kind):
    if kind == HERMITE:
        return hermite_like_basis(p)
    if kind == GAUSS_LOBATTO:
        return lagrange_basis(gauss_lobatto01(p + 1))
    if kind == GAUSS:
        return lagrange_basis(gauss01(p + 1)[0])
    raise ValueError(kind)


def cheby_mesh(n_cell_steps):
    """cells per direction and the one cell Jacobian of matvec_dg_cheby/program.cc:55-77"""
    dim = 3
    left = np.array([-1.0 + 0.05 * (d + 1) for d in range(dim)])
    right = np.array([0.95 - 0.06 * d for d in range(dim)])
    sub = np.array([2 if c < n_cell_steps % dim else 1 for c in range(dim)])
    cells = sub * 2 ** (n_cell_steps // dim)
    trafo = np.eye(dim) + np.array([[0.12 * (d + 1) * (e + 1) for e in range(dim)] for d in range(dim)])
    jac = trafo @ np.diag((right - left) / cells)
    return tuple(int(c) for c in cells), jac


def kron3(mx, my, mz):
    """local index (k, j, i), i fastest along x"""
    return np.kron(mz, np.kron(my, mx))


class DGOracle:
    """vectors are arrays [nz, ny, nx, (p+1)^3] over cells in lexicographic order, x fastest, all
    outer faces Dirichlet (homogeneous), one Jacobian for all cells (laplace_operator_dg.h:749)"""

    def __init__(self, degree, kind, cells, jacobian):
        self.p, self.kind = degree, kind
        self.n = n = degree + 1
        self.cells = tuple(cells)
        nx, ny, nz = self.cells
        self.shape = (nz, ny, nx, n ** 3)
        self.J = np.asarray(jacobian, dtype=float).reshape(3, 3)
        polys = basis_1d(degree, kind)
        xq, wq = gauss01(n)
        S = np.array([f(xq) for f in polys]).T             # values at Gauss points [q, i]
        SD = np.array([f.d(xq) for f in polys]).T           # derivatives there
        b = [np.array([[float(f(s)) for f in polys]]) for s in (0.0, 1.0)]
        g = [np.array([[float(f.d(s)) for f in polys]]) for s in (0.0, 1.0)]
        self.S, self.SD, self.xq, self.wq = S, SD, xq, wq
        self.hermite_derivative_on_face = float(polys[0].d(0.0))
        inv = np.linalg.inv(self.J)             # row a = grad xi_a
        det = abs(np.linalg.det(self.J))
        K = det * inv @ inv.T
        W3 = np.kron(wq, np.kron(wq, wq))
        grads = [kron3(SD, S, S), kron3(S, SD, S), kron3(S, S, SD)]
        A_vol = sum(K[a, c] * grads[a].T @ (W3[:, None] * grads[c]) for a in range(3) for c in range(3))
        W2 = np.kron(wq, wq)

        def trace_ops(d, s):
            def op(kind_d, deriv_dir):
                m = []
                for a in range(3):
                    if a == d:
                        m.append(g[s] if kind_d else b[s])
                    else:
                        m.append(SD if a == deriv_dir else S)
                return kron3(*m)
            V = op(False, -1)
            G = [op(a == d, a) for a in range(3)]
            return V, G

        self.blocks = {}
        for d in range(3):
            nrm = np.linalg.norm(inv[d])
            c = (inv @ inv.T)[d] / nrm               # n . grad xi_a, n pointing towards +xi_d
            Wf = W2 * det * nrm
            sigma = PENALTY_FACTOR * n * n * abs(c[d])
            V0, G0 = trace_ops(d, 0)
            V1, G1 = trace_ops(d, 1)
            N0 = sum(c[a] * G0[a] for a in range(3))
            N1 = sum(c[a] * G1[a] for a in range(3))
            W = Wf[:, None]
            # interior face between - (its upper face) and + (its lower face), normal from - to +
            mm = V1.T @ (W * (sigma * V1 - 0.5 * N1)) - 0.5 * N1.T @ (W * V1)
            mp = V1.T @ (W * (-sigma * V0 - 0.5 * N0)) + 0.5 * N1.T @ (W * V0)
            pm = -V0.T @ (W * (sigma * V1 - 0.5 * N1)) - 0.5 * N0.T @ (W * V1)
            pp = V0.T @ (W * (sigma * V0 + 0.5 * N0)) + 0.5 * N0.T @ (W * V0)
            # Dirichlet faces: sigma_F = 2 sigma, outward normal +n on the upper and -n on the lower face
            bu = V1.T @ (W * (2 * sigma * V1 - N1)) - N1.T @ (W * V1)
            bl = V0.T @ (W * (2 * sigma * V0 + N0)) + N0.T @ (W * V0)
            self.blocks[d] = dict(mm=mm, mp=mp, pm=pm, pp=pp, bu=bu, bl=bl)
        self.A_vol = A_vol

        # JacobiTransformed: 1D generalised eigenvectors (laplace_operator_dg.h:179-215)
        mass = S.T @ (wq[:, None] * S)
        lapl = SD.T @ (wq[:, None] * SD)
        pen = n * n * PENALTY_FACTOR
        lapl = lapl + pen * b[0].T @ b[0] + 0.5 * (g[0].T @ b[0] + b[0].T @ g[0])
        lapl = lapl + pen * b[1].T @ b[1] - 0.5 * (g[1].T @ b[1] + b[1].T @ g[1])
        self.eigenvalues_1d, T = scipy.linalg.eigh(lapl, mass)
        self.T = T
        self.T3 = kron3(T, T, T)
        self._diag_cache = {}

    # ---------------------------------------------------------------- operator
    def own_block(self, lower, upper):
        """A_KK of a cell whose lower/upper faces in direction d are Dirichlet where flagged"""
        A = self.A_vol.copy()
        for d in range(3):
            B = self.blocks[d]
            A += B["bl"] if lower[d] else B["pp"]
            A += B["bu"] if upper[d] else B["mm"]
        return A

    def _slices(self, d, lo, hi):
        # arrays are [z, y, x, dof]: axis of direction d is 2 - d
        s = [slice(None)] * 4
        s[2 - d] = slice(lo, hi)
        return tuple(s)

    def vmult(self, x):
        x = np.asarray(x, dtype=float).reshape(self.shape)
        y = x @ self.A_vol.T
        for d in range(3):
            B = self.blocks[d]
            N = self.cells[d]
            first, last = self._slices(d, 0, 1), self._slices(d, N - 1, N)
            lower, upper = self._slices(d, 0, N - 1), self._slices(d, 1, N)
            y[first] += x[first] @ B["bl"].T
            y[last] += x[last] @ B["bu"].T
            if N > 1:
                y[lower] += x[lower] @ B["mm"].T + x[upper] @ B["mp"].T
                y[upper] += x[upper] @ B["pp"].T + x[lower] @ B["pm"].T
        return y

    def dense_matrix(self):
        n = int(np.prod(self.shape))
        A = np.empty((n, n))
        e = np.zeros(n)
        for i in range(n):
            e[i] = 1.0
            A[:, i] = self.vmult(e).ravel()
            e[i] = 0.0
        return A

    # ---------------------------------------------------------------- block Jacobi
    def inverse_diagonal(self, lower, upper):
        key = (tuple(lower), tuple(upper))
        if key not in self._diag_cache:
            A = self.own_block(lower, upper)
            self._diag_cache[key] = 1.0 / np.einsum("ij,ij->j", self.T3, A @ self.T3)
        return self._diag_cache[key]

    def jacobi_vmult(self, r):
        r = np.asarray(r, dtype=float).reshape(self.shape)
        t = r @ self.T3                      # T^T r
        nx, ny, nz = self.cells
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    idx = (i, j, k)
                    lower = [idx[d] == 0 for d in range(3)]
                    upper = [idx[d] == self.cells[d] - 1 for d in range(3)]
                    t[k, j, i] *= self.inverse_diagonal(lower, upper)
        return t @ self.T3.T

    # ---------------------------------------------------------------- merged Chebyshev step
    def vmult_with_chebyshev_update(self, rhs, iteration_index, factor1, factor2, solution, solution_old):
        """returns (solution, solution_old) after the call, including the swap of :931"""
        rhs = np.asarray(rhs, dtype=float).reshape(self.shape)
        solution = np.asarray(solution, dtype=float).reshape(self.shape)
        solution_old = np.asarray(solution_old, dtype=float).reshape(self.shape)
        if iteration_index == 0:
            return factor2 * self.jacobi_vmult(rhs), solution_old
        z = self.jacobi_vmult(rhs - self.vmult(solution))
        new = factor2 * z + (1.0 + factor1) * solution
        if iteration_index > 1:
            new = new - factor1 * solution_old
        return new, solution

    # ---------------------------------------------------------------- helpers for the tests
    def interpolate(self, fn):
        """coefficients of the L2 projection of fn (exact for polynomials of degree <= p)"""
        n = self.n
        nx, ny, nz = self.cells
        mass = self.S.T @ (self.wq[:, None] * self.S)
        P = np.linalg.solve(mass, self.S.T * self.wq)   # Gauss values -> coefficients, 1D
        P3 = kron3(P, P, P)
        out = np.empty(self.shape)
        q = np.array([[self.xq[i], self.xq[j], self.xq[k]] for k in range(n) for j in range(n) for i in range(n)])
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    ref = q + np.array([i, j, k])
                    out[k, j, i] = P3 @ fn(ref @ self.J.T)
        return out

    def load_vector(self, fn):
        """(f, phi_i) with Gauss quadrature"""
        n = self.n
        nx, ny, nz = self.cells
        S3 = kron3(self.S, self.S, self.S)
        W3 = np.kron(self.wq, np.kron(self.wq, self.wq)) * abs(np.linalg.det(self.J))
        q = np.array([[self.xq[i], self.xq[j], self.xq[k]] for k in range(n) for j in range(n) for i in range(n)])
        out = np.empty(self.shape)
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    ref = q + np.array([i, j, k])
                    out[k, j, i] = S3.T @ (W3 * fn(ref @ self.J.T))
        return out


class DGMultigridOracle:
    """multigrid::MultigridSolverDG (common/multigrid_solver_dg.h:55-747) restated: the DG level --
    Chebyshev smoother with the JacobiTransformed preconditioner, residual restricted to FE_Q(p)
    (laplace_operator_dg.h:1798-1819), V-cycle of the FE_Q hierarchy, correction embedded back
    (:1863-1894) -- on top of the C oracle's FE_Q multigrid (oracle/mg_oracle.c) on the same Cartesian
    mesh.  `dg`: DGOracle, `fe`: oracle.Oracle of the same mesh and degree (its smoothers are
    re-configured as multigrid_solver_dg.h:271-291 does), `start`: the start vector of the
    eigenvalue estimate in the DG oracle's layout (deal.II: (global DoF index mod 11) - mean)."""

    def __init__(self, dg, fe, degree_pre, start):
        self.dg, self.fe, self.degree = dg, fe, degree_pre
        p, n = dg.p, dg.n
        lmax = fe.max_level
        self.lmax = lmax
        if lmax > 0:
            fe.reset_smoother(lmax, 20.0, max(1, degree_pre - 1), 15)
        fe.reset_smoother(0, 2e-3, -1, max(3, fe.n_dofs(0)))
        polys = basis_1d(p, dg.kind)
        g = gauss_lobatto01(n)
        B = np.array([f(g) for f in polys]).T            # B[q, i] = phi_i(g_q)
        self.P1 = np.linalg.inv(B)                        # GLL values -> DG coefficients
        self.P3 = kron3(self.P1, self.P1, self.P1)
        nx, ny, nz = dg.cells
        self.G = (nx * p + 1, ny * p + 1, nz * p + 1)
        self.fe_grid = fe.dof_grid(lmax).astype(np.int64)
        # eigenvalue estimate: CG preconditioned with JacobiTransformed (multigrid_solver_dg.h:293-303)
        r = np.array(start, dtype=float).reshape(dg.shape)
        d = None
        diag, off = [], []
        res, rz, alpha, it = np.linalg.norm(r), 0.0, 0.0, 0
        while it < 15 and res > 1e-10:
            it += 1
            rz_old = rz
            z = dg.jacobi_vmult(r)
            rz = float(np.vdot(r, z))
            if it > 1:
                beta = rz / rz_old
                d = z + beta * d
            else:
                beta = 0.0
                d = z
            alpha_old = alpha
            h = dg.vmult(d)
            alpha = rz / float(np.vdot(d, h))
            r = r - alpha * h
            res = np.linalg.norm(r)
            if it == 1:
                diag.append(1.0 / alpha)
            else:
                off.append(np.sqrt(beta) / alpha_old)
                diag.append(1.0 / alpha + beta / alpha_old)
        T = np.diag(diag) + np.diag(off, 1) + np.diag(off, -1)
        ev = np.linalg.eigvalsh(T)
        self.lambda_max = 1.2 * ev[-1]
        a = self.lambda_max / 20.0
        self.delta, self.theta = 0.5 * (self.lambda_max - a), 0.5 * (self.lambda_max + a)
        self.cg_its = it

    # ---- transfers ----
    def restrict_to_cg(self, r_dg):
        """P^T r, as a vector of the FE_Q oracle's finest level (constrained rows zero)"""
        p, n = self.dg.p, self.dg.n
        nx, ny, nz = self.dg.cells
        Gx, Gy, Gz = self.G
        grid = np.zeros((Gz, Gy, Gx))
        loc = np.asarray(r_dg).reshape(self.dg.shape) @ self.P3      # (P3^T r)_q = sum_i P3[i, q] r_i
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    grid[k * p:k * p + n, j * p:j * p + n, i * p:i * p + n] += loc[k, j, i].reshape(n, n, n)
        grid[0], grid[-1], grid[:, 0], grid[:, -1], grid[:, :, 0], grid[:, :, -1] = 0, 0, 0, 0, 0, 0
        return grid.ravel()[self.fe_grid]

    def prolongate_cg_to_dg(self, u_cg):
        p, n = self.dg.p, self.dg.n
        nx, ny, nz = self.dg.cells
        Gx, Gy, Gz = self.G
        grid = np.zeros(Gx * Gy * Gz)
        grid[self.fe_grid] = u_cg
        grid = grid.reshape(Gz, Gy, Gx)
        grid[0], grid[-1], grid[:, 0], grid[:, -1], grid[:, :, 0], grid[:, :, -1] = 0, 0, 0, 0, 0, 0
        out = np.empty(self.dg.shape)
        for k in range(nz):
            for j in range(ny):
                for i in range(nx):
                    c = grid[k * p:k * p + n, j * p:j * p + n, i * p:i * p + n].ravel()
                    out[k, j, i] = self.P3 @ c
        return out

    # ---- smoother: PreconditionChebyshev with the merged operation (laplace_operator_dg.h:910-955) ----
    def smooth(self, x, b, is_step):
        dg = self.dg
        if not is_step:
            x, old = dg.vmult_with_chebyshev_update(b, 0, 0.0, 1.0 / self.theta, None if x is None else x, np.zeros(dg.shape))
            index = 1
        else:
            x, old = dg.vmult_with_chebyshev_update(b, 1, 0.0, 1.0 / self.theta, x, np.zeros(dg.shape))
            index = 2
        rhok, sigma = self.delta / self.theta, self.theta / self.delta
        for _ in range(self.degree - 1):
            rhokp = 1.0 / (2.0 * sigma - rhok)
            f1, f2 = rhokp * rhok, 2.0 * rhokp / self.delta
            rhok = rhokp
            x, old = dg.vmult_with_chebyshev_update(b, index, f1, f2, x, old)
            index += 1
        return x

    def v_cycle(self, defect):
        """dg_v_cycle(1), multigrid_solver_dg.h:605-633"""
        defect = np.asarray(defect, dtype=float).reshape(self.dg.shape)
        x = self.smooth(np.zeros(self.dg.shape), defect, False)
        t = defect - self.dg.vmult(x)
        x = x + self.prolongate_cg_to_dg(self.fe.vcycle(self.restrict_to_cg(t)))
        return self.smooth(x, defect, True)

    def solve_cg(self, rhs, tolerance=1e-9):
        """(solution, iterations, reduction rate), multigrid_solver_dg.h:410-424"""
        rhs = np.asarray(rhs, dtype=float).reshape(self.dg.shape)
        x = np.zeros(self.dg.shape)
        r = rhs.copy()
        res0 = res = np.linalg.norm(r)
        it, rz, d = 0, 0.0, None
        while res > max(1e-16, tolerance * res0) and it < 100:
            it += 1
            z = self.v_cycle(r)
            rz_old, rz = rz, float(np.vdot(r, z))
            d = z if it == 1 else z + (rz / rz_old) * d
            h = self.dg.vmult(d)
            alpha = rz / float(np.vdot(d, h))
            x = x + alpha * d
            r = r - alpha * h
            res = np.linalg.norm(r)
        return x, it, (res / res0) ** (1.0 / max(it, 1))
