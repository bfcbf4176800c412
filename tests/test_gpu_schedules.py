"""The three schedules of the macro-element brick loop -- eight colour launches, two classes with the
brick edges and corners in private blocks, one launch with every brick surface in private blocks
(mgx_macro.hip, FREE) -- against the oracle and against each other on the same meshes: operator,
residual, Chebyshev smoother from a zero and a non-zero start, V-cycle, PCG; results bitwise
reproducible.  Production picks the schedule per level from the number of bricks
(Tunables::free_one_max / free_max_bricks); the overrides below force each one on every level."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")
from oracle_view import assert_same_cg, oracle_for  # noqa: E402

SCHEDULES = {"eight": {"MGX_FREE_MAX_BRICKS": "0"},
             "two": {"MGX_FREE_MAX_BRICKS": "4000000000", "MGX_FREE_ONE_MAX": "0"},
             "one": {"MGX_FREE_MAX_BRICKS": "4000000000", "MGX_FREE_ONE_MAX": "4000000000"}}


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("p,ns,nr,number", [(4, 1, 3, "f64"), (4, 3, 2, "f64"), (2, 1, 4, "f64"), (1, 1, 4, "f64"), (3, 3, 2, "f64"),
                                            (5, 1, 3, "f64"), (8, 1, 2, "f64"), (4, 1, 3, "f32"), (7, 3, 1, "f64")])
def test_schedules_against_oracle_and_each_other(monkeypatch, p, ns, nr, number):
    vnum = mg.F32 if number == "f32" else mg.F64
    tol_op, tol_sm, tol_v = (2e-5, 5e-5, 2e-4) if number == "f32" else (1e-12, 1e-10, 1e-9)
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=3, n_cycles=1, vfloat=number == "f32")
    l = cube.max_level
    x, b = cube.seeded_vector(l, 21), cube.seeded_vector(l, 22)
    if number == "f32":
        x, b = x.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    results = {}
    for name, env in SCHEDULES.items():
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        c = mg.Context(0)
        solver = mg.MultigridSolver(c, cube, 3, 3, 1, vnum)
        A = solver.matrix(l)
        src, rhs, dst, again = (c.vector(x.size, vnum, x), c.vector(x.size, vnum, b), c.vector(x.size, vnum), c.vector(x.size, vnum))
        out = []
        A.vmult(dst, src)
        A.vmult(again, src)
        assert np.array_equal(dst.download(), again.download())
        out.append(dst.download().astype(np.float64))
        assert rel(out[-1], orc.vmult(l, x)) < tol_op
        A.vmult_residual(rhs, src, dst)
        out.append(dst.download().astype(np.float64))
        assert rel(out[-1], orc.vmult_residual(l, b, x)) < tol_op
        if number == "f64":
            sm = solver.smoother(l)
            sm.vmult(dst, rhs)
            x_ref = orc.cheb_vmult(l, b)
            out.append(dst.download())
            assert rel(out[-1], x_ref) < tol_sm
            sm.step(dst, rhs)
            out.append(dst.download())
            assert rel(out[-1], orc.cheb_step(l, x_ref, b)) < tol_sm
        xd, yd = c.vector(x.size, data=x), c.vector(x.size)
        for _ in range(2):
            solver.vmult(yd, xd)
            assert rel(yd.download(), orc.vcycle(x)) < tol_v
        out.append(yd.download())
        if number == "f64":
            assert_same_cg(solver, orc)
            out.append(solver.get_solution().download())
        results[name] = out
        solver.close()
        c.close()
    for name in ("two", "one"):
        for a, e in zip(results[name], results["eight"]):
            assert rel(a, e) < 10 * tol_op
    cube.close()
    orc.close()


@pytest.mark.parametrize("p,ns,nr,number", [(1, 1, 4, "f64"), (2, 1, 4, "f64"), (3, 3, 2, "f64"), (4, 1, 3, "f64"), (4, 3, 2, "f64"),
                                            (5, 1, 3, "f64"), (6, 1, 2, "f64"), (7, 3, 1, "f64"), (8, 1, 2, "f64"), (9, 1, 2, "f64"),
                                            (4, 1, 3, "f32"), (8, 1, 2, "f32")])
def test_second_pipeline_of_the_plain_and_residual_forms(p, ns, nr, number):
    """mgx_macro2.hip (interior points first, partial sums requested before the sweeps) runs the plain and the residual
    form of the eight-colour schedule: against the oracle and BITWISE against the first pipeline (mgx_macro.hip, option
    no_macro_v2) -- same sweeps, same order of every sum."""
    vnum = mg.F32 if number == "f32" else mg.F64
    tol = 2e-5 if number == "f32" else 1e-12
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr)
    l = cube.max_level
    x, b = cube.seeded_vector(l, 31), cube.seeded_vector(l, 32)
    if number == "f32":
        x, b = x.astype(np.float32).astype(np.float64), b.astype(np.float32).astype(np.float64)
    results = []
    for opts in ({"free_max_bricks": 0}, {"free_max_bricks": 0, "no_macro_v2": 1}):
        c = mg.Context(0, options=opts)
        op = mg.LaplaceOperator.from_cube(c, cube, l, number=vnum)
        src, rhs, dst = c.vector(x.size, vnum, x), c.vector(x.size, vnum, b), c.vector(x.size, vnum)
        op.vmult(dst, src)
        y = dst.download()
        assert rel(y.astype(np.float64), orc.vmult(l, x)) < tol
        op.vmult_residual(rhs, src, dst)
        r = dst.download()
        assert rel(r.astype(np.float64), orc.vmult_residual(l, b, x)) < tol
        results.append((y, r))
        op.clear()
        c.close()
    assert np.array_equal(results[0][0], results[1][0]) and np.array_equal(results[0][1], results[1][1])
    cube.close()
    orc.close()
