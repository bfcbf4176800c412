"""The CPU oracle seen through the DoF numbering of a provider cube.

The oracle numbers the mesh entities cell by cell; the provider (mgx_cube) by default groups them
per brick for the device cell loop.  Both expose dof -> global lexicographic grid id, so a level
vector is carried from one numbering to the other by a permutation.  `oracle_for(cube, ...)`
returns an object with the Oracle's methods whose level vectors (inputs and results) are in the
CUBE's numbering, so that a parity test reads `assert rel(gpu_result, orc.vmult(l, x)) < tol`."""
import numpy as np

from oracle import Oracle


class OracleView:
    def __init__(self, orc, cube):
        self.orc, self.cube = orc, cube
        self._maps = {}

    def __getattr__(self, name):  # everything that carries no level vector
        return getattr(self.orc, name)

    def maps(self, l):
        """(ofc, cfo): v_oracle = v_cube[ofc], v_cube = v_oracle[cfo]; None, None if identical."""
        if l not in self._maps:
            gc, go = self.cube.dof_grid(l).astype(np.int64), self.orc.dof_grid(l).astype(np.int64)
            assert gc.size == go.size
            if np.array_equal(gc, go):
                self._maps[l] = (None, None)
            else:
                pos_c = np.empty(gc.size, np.int64)
                pos_c[gc] = np.arange(gc.size)
                pos_o = np.empty(go.size, np.int64)
                pos_o[go] = np.arange(go.size)
                self._maps[l] = (pos_c[go], pos_o[gc])
        return self._maps[l]

    def to_o(self, l, v):
        ofc = self.maps(l)[0]
        return np.ascontiguousarray(v if ofc is None else np.asarray(v)[ofc])

    def to_c(self, l, v):
        cfo = self.maps(l)[1]
        return np.ascontiguousarray(v if cfo is None else np.asarray(v)[cfo])

    @property
    def lmax(self):
        return self.orc.n_levels - 1

    # per-DoF data
    def rhs(self, l):
        return self.to_c(l, self.orc.rhs(l))

    def inv_diag(self, l):
        return self.to_c(l, self.orc.inv_diag(l))

    def solution(self, l):
        return self.to_c(l, self.orc.solution(l))

    def bc(self, l):
        idx, val = self.orc.bc(l)
        ofc = self.maps(l)[0]
        if ofc is not None:
            idx = ofc[idx]
            order = np.argsort(idx)
            idx, val = idx[order].astype(np.uint32), val[order]
        return idx, val

    # operators
    def vmult(self, l, src):
        return self.to_c(l, self.orc.vmult(l, self.to_o(l, src)))

    def vmult_residual(self, l, rhs, lhs):
        return self.to_c(l, self.orc.vmult_residual(l, self.to_o(l, rhs), self.to_o(l, lhs)))

    def cheb_vmult(self, l, b):
        return self.to_c(l, self.orc.cheb_vmult(l, self.to_o(l, b)))

    def cheb_step(self, l, x, b):
        return self.to_c(l, self.orc.cheb_step(l, self.to_o(l, x), self.to_o(l, b)))

    def prolongate(self, l, coarse, fine=None, with_bc=False):
        f = None if fine is None else self.to_o(l, fine)
        return self.to_c(l, self.orc.prolongate(l, self.to_o(l - 1, coarse), fine=f, with_bc=with_bc))

    def restrict_and_add(self, l, coarse, fine, with_bc=False):
        return self.to_c(l - 1, self.orc.restrict_and_add(l, self.to_o(l - 1, coarse), self.to_o(l, fine),
                                                          with_bc=with_bc))

    def vmult_with_cg_update(self, l, alpha, beta, r, q, p, x):
        """returns (sums, q, p, x) in the cube's numbering; the inputs are not modified"""
        ro, qo, po, xo = (self.to_o(l, a).copy() for a in (r, q, p, x))
        sums = self.orc.vmult_with_cg_update(l, alpha, beta, ro, qo, po, xo)
        return sums, self.to_c(l, qo), self.to_c(l, po), self.to_c(l, xo)

    def vmult_with_residual_update(self, residual, update, factor):
        l = self.lmax
        ro, uo = self.to_o(l, residual).copy(), self.to_o(l, update).copy()
        out = self.orc.vmult_with_residual_update(ro, uo, factor)
        return out, self.to_c(l, ro), self.to_c(l, uo)

    def vcycle(self, src):
        l = self.lmax
        return self.to_c(l, self.orc.vcycle(self.to_o(l, src)))


def oracle_for(cube, *args, **kwargs):
    return OracleView(Oracle(*args, **kwargs), cube)


def assert_same_cg(solver, orc, rtol=1e-6):
    """solver.solve_cg() against orc.solve_cg() (multigrid_solver.h:483-493) through the residual history
    SolverCG hands to its ReductionControl(1000, 1e-16, 1e-9).

    * While the residual is above 1e-4 of its start value, every entry agrees to `rtol`, relative.
    * Below that, preconditioned CG on an ill-conditioned operator (the 1e6 coefficient contrast of
      poisson_shell, 25 - 45 iterations) amplifies the last-bit differences between two correct
      implementations until the two paths decorrelate (measured: 1e-8 up to iteration 20, then
      0.5 within four iterations, while iteration count and final L2 error stay identical): there the
      entries must stay within a factor of 3 of each other.  Well-conditioned problems never get there.
    * The iteration counts are equal; they may differ by one if the oracle's residual at the deciding
      iteration lies within 1 % of the stopping threshold, and by two if the paths have decorrelated.
    Returns (iterations, reduction rate) of the solver."""
    import numpy as np

    its, red = solver.solve_cg()
    oits, _ = orc.solve_cg()
    h, oh = np.asarray(solver.cg_history()), np.asarray(orc.cg_history())
    assert len(h) == its + 1 and len(oh) == oits + 1
    n = min(len(h), len(oh))
    err = np.abs(h[:n] - oh[:n]) / oh[:n]
    early = oh[:n] >= 1e-4 * oh[0]
    assert err[early].max() <= rtol, "PCG residual histories differ: %g at iteration %d" % (err[early].max(), int(err[early].argmax()))
    ratio = h[:n] / oh[:n]
    assert ratio.min() > 1. / 3. and ratio.max() < 3., "PCG residual histories differ by a factor of %g" % max(ratio.max(), 1. / ratio.min())
    if its != oits:
        k = min(its, oits)  # the iteration after which one of the two stopped
        threshold = max(1e-9 * oh[0], 1e-16)
        near = abs(its - oits) == 1 and abs(oh[k] - threshold) <= 0.01 * threshold
        decorrelated = abs(its - oits) <= 2 and err.max() > 1e-3
        assert near or decorrelated, (
            "PCG iteration counts %d (device) vs %d (oracle); oracle residual %g at iteration %d, threshold %g"
            % (its, oits, oh[k], k, threshold))
    return its, red
