"""Right-hand side assembly on the GPU (mgx_compute_residual / mgx_solver_compute_rhs = LaplaceOperator::compute_residual,
laplace_operator.h:804-845, as MultigridSolver's constructor calls it, multigrid_solver.h:225-261) against the host
assembly of the provider, which the CPU suite pins to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")


@pytest.fixture(scope="module")
def ctx():
    c = mg.Context(0)
    yield c
    c.close()


CASES = {
    "cube_p2": lambda: mg.Cube(2, 1, 4),
    "cube_p4": lambda: mg.Cube(4, 1, 3),
    "cube_p5": lambda: mg.Cube(5, 1, 2),
    "cube_p8": lambda: mg.Cube(8, 1, 2),
    "box_p3": lambda: mg.Cube(3, n_refine=2, box=(2, 1, 1), numbering="cell"),
    "sheared_p2": lambda: mg.Cube(2, n_refine=3, box=(1, 1, 1), geometry="sheared"),
    "shell_sector_p4": lambda: mg.Cube(4, n_refine=2, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry="shell_sector", problem="shell"),
    "hyper_shell6_p3": lambda: mg.Cube(3, n_refine=2, shell=6, problem="shell"),
    "hyper_shell12_p2": lambda: mg.Cube(2, n_refine=1, shell=12, problem="shell"),
}


@pytest.mark.parametrize("case", sorted(CASES))
def test_compute_residual_equals_host_assembly(ctx, case):
    cube = CASES[case]()
    for l in range(cube.n_levels):
        n = cube.n_dofs(l)
        op = mg.LaplaceOperator.from_cube(ctx, cube, l)
        u = np.zeros(n)
        bi, bv = cube.bc(l)
        u[bi] = bv
        fq = cube.rhs_quadrature(l)
        dst, src, f = ctx.vector(n, data=np.full(n, 3.0)), ctx.vector(n, data=u), ctx.vector(fq.size, data=fq.ravel())
        op.compute_residual(dst, src, f)
        got, ref = dst.download(), cube.rhs(l)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max(), (case, l, np.abs(got - ref).max(), np.abs(ref).max())
        assert (got[cube.constrained(l)] == 0).all()
        op.compute_residual(dst, src, f)
        assert np.array_equal(dst.download(), got)   # fixed order of the additions
        # the two terms on their own: f = 0 and homogeneous boundary values
        a, b = ctx.vector(n), ctx.vector(n)
        op.compute_residual(a, src, None)
        op.compute_residual(b, None, f)
        assert np.abs(a.download() + b.download() - got).max() <= 1e-13 * np.abs(ref).max()
    cube.close()


@pytest.mark.parametrize("p,nr,number", [(4, 3, mg.F64), (2, 4, mg.F32), (5, 2, mg.F64)])
def test_solver_with_device_rhs_solves_like_the_host_one(ctx, p, nr, number):
    cube = mg.Cube(p, 1, nr)
    host = mg.MultigridSolver(ctx, cube, 3, 3, 1, number)
    dev = mg.MultigridSolver(ctx, cube, 3, 3, 1, number, device_rhs=True)
    for l in range(cube.n_levels):
        a, b = host.get_vector(l, "rhs").download(), dev.get_vector(l, "rhs").download()
        assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max(), l
    host.solve()
    dev.solve()
    eh, ed = host.compute_l2_error(), dev.compute_l2_error()
    assert abs(eh - ed) <= 1e-6 * eh
    ih, id_ = host.solve_cg()[0], dev.solve_cg()[0]
    assert abs(ih - id_) <= 1
    host.close()
    dev.close()
    cube.close()


def test_device_rhs_at_benchmark_size(ctx):
    """FE_Q(4), 64^3 cells (17 M DoFs, brick schedule: 64 launches of cells that share no DoF): the device rhs against
    the host one, and the README's discretisation error from the solver built on it"""
    cube = mg.Cube(4, 1, 6)
    dev = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64, device_rhs=True)
    l = cube.max_level
    a, b = cube.rhs(l), dev.get_vector(l, "rhs").download()
    assert np.abs(a - b).max() <= 1e-12 * np.abs(a).max()
    its, _ = dev.solve_cg()
    assert its == 8 and abs(dev.compute_l2_error() / 1.327e-8 - 1) < 5e-3   # README.md:135-159
    dev.close()
    cube.close()
