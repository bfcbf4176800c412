"""poisson_shell slice (BASELINE config 4) on the CPU: the provider's mapped meshes -- per-point merged
coefficient (evaluate_coefficient, laplace_operator.h:388-430), boundary values, right-hand side --
against the oracle's independent restatement, and the oracle's general branch against its affine
one where both apply."""
import numpy as np
import pytest

import multigrid_amd as mg
from oracle import Oracle
from oracle_view import oracle_for

CASES = [("sheared", "cube"), ("shell_sector", "shell"), ("sheared", "shell")]


def test_general_branch_equals_affine_branch_on_the_cube():
    a, b = Oracle(3, 2, 1), Oracle(3, 2, 1, geometry="cartesian", problem="cube")
    l = a.max_level
    x = np.random.default_rng(0).uniform(-1, 1, a.n_dofs(l))
    assert np.abs(a.vmult(l, x) - b.vmult(l, x)).max() < 1e-13 * np.abs(a.vmult(l, x)).max()
    np.testing.assert_allclose(a.rhs(l), b.rhs(l), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(a.inv_diag(l), b.inv_diag(l), rtol=1e-12)
    a.close()
    b.close()


@pytest.mark.parametrize("geometry,problem", CASES)
@pytest.mark.parametrize("p", [2, 4])
def test_provider_matches_oracle_on_mapped_meshes(geometry, problem, p):
    cube = mg.Cube(p, n_refine=2, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry=geometry, problem=problem)
    orc = oracle_for(cube, p, 1, 2, geometry=geometry, problem=problem, origin=-0.9, h0=1.9)
    n3 = (p + 1) ** 3
    for l in range(cube.n_levels):
        cq = cube.coef_q(l)                                  # [cell][6][n^3]
        oq = orc.coef_q(l)                                   # [cell][n^3][6]
        np.testing.assert_allclose(cq, np.transpose(oq, (0, 2, 1)), rtol=1e-10, atol=1e-10 * np.abs(oq).max())
        np.testing.assert_allclose(cube.rhs(l), orc.rhs(l), rtol=1e-9, atol=1e-9 * np.abs(orc.rhs(l)).max())
        # boundary values are stored where they are nonzero (multigrid_solver.h:250): compare as
        # full vectors (a value of 1e-17 on one side is a zero on the other)
        full_c, full_o = np.zeros(cube.n_dofs(l)), np.zeros(cube.n_dofs(l))
        ci, cv = cube.bc(l)
        oi, ov = orc.bc(l)
        full_c[ci], full_o[oi] = cv, ov
        np.testing.assert_allclose(full_c, full_o, rtol=1e-13, atol=1e-13)
    assert cq.shape == (cube.n_cells(cube.max_level), 6, n3)
    cube.close()
    orc.close()


def test_shell_problem_converges_with_the_expected_order():
    """manufactured solution on the curved sector with the 1e6 coefficient contrast: the PCG solution
    converges with order p + 1 (independent of any implementation detail of the operator)"""
    errs = []
    for nr in (2, 3):
        o = Oracle(3, 1, nr, degree=3, n_cycles=1, geometry="shell_sector", problem="shell", origin=-0.9, h0=1.9)
        o.solve_cg()
        errs.append(o.l2_error())
        o.close()
    assert np.log2(errs[0] / errs[1]) > 3.5
