"""poisson_shell slice on the GPU (BASELINE config 4): the quadrature-point operation with the full
symmetric tensor -- one per mesh (affine cells, laplace_operator.h:473-486) or one per cell and
quadrature point (variable coefficient, curved cells, :493-522) -- and the multigrid solver on it,
against the oracle.  Tolerances as in test_gpu_parity.py; the 1e6 coefficient contrast of the shell
problem enters the operator's scale, not the relative accuracy."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")
from oracle import Oracle  # noqa: E402
from oracle_view import assert_same_cg, oracle_for  # noqa: E402


@pytest.fixture(scope="module")
def ctx():
    c = mg.Context(0)
    yield c
    c.close()


def rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.mark.parametrize("geometry,problem", [("sheared", "cube"), ("shell_sector", "shell"), ("sheared", "shell")])
@pytest.mark.parametrize("p,nr", [(2, 3), (4, 2), (5, 2), (8, 1)])
def test_mapped_operator(ctx, geometry, problem, p, nr):
    cube = mg.Cube(p, n_refine=nr, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry=geometry, problem=problem)
    orc = oracle_for(cube, p, 1, nr, geometry=geometry, problem=problem, origin=-0.9, h0=1.9)
    for l in range(cube.n_levels):
        for number, tol in ((mg.F64, 1e-12), (mg.F32, 2e-5)):
            op = mg.LaplaceOperator.from_cube(ctx, cube, l, number)
            x, b = cube.seeded_vector(l, 1), cube.seeded_vector(l, 2)
            src, rhs, dst = ctx.vector(x.size, number, x), ctx.vector(x.size, number, b), ctx.vector(x.size, number)
            x_in = x.astype(np.float32).astype(np.float64) if number == mg.F32 else x
            b_in = b.astype(np.float32).astype(np.float64) if number == mg.F32 else b
            op.vmult(dst, src)
            assert rel(dst.download().astype(np.float64), orc.vmult(l, x_in)) < tol
            op.vmult_residual(rhs, src, dst)
            assert rel(dst.download().astype(np.float64), orc.vmult_residual(l, b_in, x_in)) < tol
            if number == mg.F64:
                op.compute_diagonal()
                assert rel(op.get_matrix_diagonal_inverse().download(), orc.inv_diag(l)) < 1e-12
            op.clear()
    cube.close()
    orc.close()


@pytest.mark.parametrize("geometry,problem", [("sheared", "cube"), ("shell_sector", "shell")])
def test_mapped_multigrid_solver(ctx, geometry, problem):
    p, nr = 4, 2
    cube = mg.Cube(p, n_refine=nr, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry=geometry, problem=problem)
    orc = oracle_for(cube, p, 1, nr, degree=3, n_cycles=1, geometry=geometry, problem=problem, origin=-0.9, h0=1.9)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    l = cube.max_level
    for lev in range(cube.n_levels):
        gi, oi = solver.smoother(lev).info(), orc.cheb_info(lev)
        assert gi["degree"] == oi["degree"] and abs(gi["cg_its"] - oi["cg_its"]) <= (2 if lev == 0 else 0)
        assert gi["lambda_max"] == pytest.approx(oi["lambda_max"], rel=1e-8)
    x = cube.seeded_vector(l, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    solver.vmult(dst, src)
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    rate, trace = solver.solve(True)
    orate, otrace = orc.solve(True)
    assert rate == pytest.approx(orate, rel=1e-6)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-7)
    assert_same_cg(solver, orc)  # 1e6 coefficient contrast: residual history, not a bare iteration count
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-7)
    solver.close()
    cube.close()
    orc.close()


@pytest.mark.parametrize("p", [2, 4, 7])
def test_affine_full_tensor(ctx, p):
    """affine cells with off-diagonal coefficient entries (laplace_operator.h:473-486): the tensor is
    the operator descriptor's coef[6], the weight is applied per point"""
    nr = 2
    orc = Oracle(p, 1, nr)
    l = nr
    coef = np.array([1.3, 0.9, 1.1, 0.25, -0.15, 0.2]) * orc.cell_size(l)
    orc.set_affine_coef(l, coef)
    idx = np.ascontiguousarray(orc.idx27(l)).ravel()
    plain = np.ascontiguousarray(orc.idx27_plain(l)).ravel()
    cons = orc.constrained(l)
    S, D, w = orc.shape_values().ravel(), orc.colloc_grad().ravel(), orc.qweights()
    d = mg._lib.OperatorDesc()
    d.degree, d.number, d.n_cells, d.n_dofs = p, mg.F64, orc.n_cells(l), orc.n_dofs(l)
    d.idx27 = idx.ctypes.data_as(mg._lib.u32p)
    d.idx27_plain = plain.ctypes.data_as(mg._lib.u32p)
    d.constrained = cons.ctypes.data_as(mg._lib.u32p)
    d.n_constrained = cons.size
    for i, v in enumerate(coef):
        d.coef[i] = v
    d.shape_values = S.ctypes.data_as(mg._lib.f64p)
    d.colloc_grad = D.ctypes.data_as(mg._lib.f64p)
    d.qweights = w.ctypes.data_as(mg._lib.f64p)
    op = mg.LaplaceOperator(ctx, d)
    x = np.random.default_rng(3).uniform(-1, 1, orc.n_dofs(l))
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    op.vmult(dst, src)
    assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
    # diagonal: (A e_i)_i of the oracle for a sample of DoFs (its stored diagonal predates the tensor)
    op.compute_diagonal()
    inv = op.get_matrix_diagonal_inverse().download()
    free = np.setdiff1d(np.arange(orc.n_dofs(l)), cons)
    for i in np.random.default_rng(5).choice(free, 40, replace=False):
        e = np.zeros(orc.n_dofs(l))
        e[i] = 1.0
        assert inv[i] == pytest.approx(1.0 / orc.vmult(l, e)[i], rel=1e-12)
    assert np.array_equal(inv[cons], np.ones(cons.size))
    op.clear()
    orc.close()


@pytest.mark.parametrize("geometry,problem,p,nr", [("shell_sector", "shell", 4, 2), ("sheared", "cube", 2, 3), ("sheared", "shell", 3, 2),
                                                   ("shell_sector", "shell", 1, 3)])
def test_colour_by_colour_launches_of_the_general_branch(monkeypatch, geometry, problem, p, nr):
    """levels with many cells run the general branch colour by colour without atomics (greedy cell
    colouring from the shared entities; production threshold 4096 cells): forced on every level here,
    against the oracle, bitwise reproducible, and equal to the atomic form up to the summation order"""
    monkeypatch.setenv("MGX_CELL_COLOUR_MIN", "1")
    c = mg.Context(0)
    monkeypatch.setenv("MGX_CELL_COLOUR_MIN", "4000000000")
    c_atomic = mg.Context(0)
    cube = mg.Cube(p, n_refine=nr, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry=geometry, problem=problem)
    orc = oracle_for(cube, p, 1, nr, geometry=geometry, problem=problem, origin=-0.9, h0=1.9)
    for l in range(cube.n_levels):
        op, opa = mg.LaplaceOperator.from_cube(c, cube, l), mg.LaplaceOperator.from_cube(c_atomic, cube, l)
        x = cube.seeded_vector(l, 3)
        src, dst, again = c.vector(x.size, data=x), c.vector(x.size), c.vector(x.size)
        op.vmult(dst, src)
        op.vmult(again, src)
        assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
        assert np.array_equal(dst.download(), again.download())
        sa, da = c_atomic.vector(x.size, data=x), c_atomic.vector(x.size)
        opa.vmult(da, sa)
        assert rel(da.download(), dst.download()) < 1e-13
        op.clear()
        opa.clear()
    cube.close()
    orc.close()
    c.close()
    c_atomic.close()


@pytest.mark.parametrize("mesh,nr,problem,number", [(("shell", 6), 3, "shell", "f64"), (("shell", 12), 2, "shell", "f64"),
                                                    (("shell", 6), 3, "cube", "f32"), (("box", "sheared"), 3, "cube", "f64"),
                                                    (("box", "shell_sector"), 3, "shell", "f64")])
def test_brick_form_of_the_general_operator(mesh, nr, problem, number):
    """vmult of a general-tensor operator at p = 4 on its brick schedule (brick_general_kernel: the cells of a brick added
    up in LDS, brick surfaces through private blocks and the finish kernel; production from 2048 bricks on, forced
    here): against the oracle on the same mesh, against the per-cell kernel with ordered assembly, bitwise reproducible"""
    cb = mg.Context(0, options={"general_brick_min": 1})
    cc = mg.Context(0, options={"no_general_bricks": 1})
    if mesh[0] == "shell":
        cube = mg.Cube(4, n_refine=nr, shell=mesh[1], problem=problem)
        orc = Oracle(4, degree=3, n_cycles=1, mesh=cube, problem=problem)
    else:
        cube = mg.Cube(4, n_refine=nr, box=(1, 1, 1), origin=-0.9, h0=1.9, geometry=mesh[1], problem=problem)
        orc = oracle_for(cube, 4, 1, nr, geometry=mesh[1], problem=problem, origin=-0.9, h0=1.9)
    num, dt, tol = (mg.F32, np.float32, 2e-6) if number == "f32" else (mg.F64, np.float64, 1e-12)
    l = cube.max_level
    x = cube.seeded_vector(l, 5)
    res = []
    for c in (cb, cc):
        op = mg.LaplaceOperator.from_cube(c, cube, l, num)
        src, dst, again = c.vector(x.size, num, x.astype(dt)), c.vector(x.size, num), c.vector(x.size, num)
        op.vmult(dst, src)
        op.vmult(again, src)
        assert np.array_equal(dst.download(), again.download())
        res.append(dst.download().astype(np.float64))
        op.clear()
    assert rel(res[0], res[1]) < tol
    if orc is not None:
        assert rel(res[0], orc.vmult(l, x)) < (1e-5 if number == "f32" else 1e-12)
        orc.close()
    cube.close()
    cb.close()
    cc.close()


@pytest.mark.parametrize("n_coarse,p,nr,problem", [(6, 4, 2, "shell"), (12, 4, 1, "shell"), (6, 2, 3, "shell"), (12, 3, 2, "cube"),
                                                   (6, 5, 1, "shell"), (6, 1, 3, "cube")])
def test_hyper_shell_solver_against_oracle(ctx, n_coarse, p, nr, problem):
    """The mesh of poisson_shell itself -- GridGenerator::hyper_shell(0, 0.5, 1.0, 6 | 12) + refine_global
    (poisson_shell/program.cc:425-431): operator, diagonal, smoother parameters, transfers, V-cycle, FMG and
    PCG on the device against the oracle assembled on the same mesh tables (tests/test_hyper_shell.py pins
    the tables).  Three blocks meet at the radial edges through the polyhedron's vertices: the transfer
    runs with owner weights there, the oracle with 1/multiplicity -- the same operator."""
    cube = mg.Cube(p, n_refine=nr, shell=n_coarse, problem=problem)
    orc = Oracle(p, degree=3, n_cycles=1, mesh=cube, problem=problem)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    for l in range(cube.n_levels):
        A = solver.matrix_dp(l)
        x, b = cube.seeded_vector(l, 1), cube.seeded_vector(l, 2)
        src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
        A.vmult(dst, src)
        assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
        A.vmult_residual(rhs, src, dst)
        assert rel(dst.download(), orc.vmult_residual(l, b, x)) < 1e-12
        assert rel(A.get_matrix_diagonal_inverse().download(), orc.inv_diag(l)) < 1e-12
        gi, oi = solver.smoother(l).info(), orc.cheb_info(l)
        # (level 0: the eigenvalue CG runs to a 1e-10 residual -- 75 iterations on the 12-cell mesh with the 1e6
        # contrast -- and stops an iteration earlier or later with the last bits of either implementation; the
        # levels above run a fixed 15 iterations)
        assert gi["degree"] == oi["degree"] and abs(gi["cg_its"] - oi["cg_its"]) <= (2 if l == 0 else 0)
        assert gi["lambda_max"] == pytest.approx(oi["lambda_max"], rel=1e-8)
        if l > 0:
            T = mg.Transfer(solver.matrix_dp(l - 1), A, cube.children(l), cube.prolong_1d())
            xc = cube.seeded_vector(l - 1, 3)
            cv, fv = ctx.vector(xc.size, data=xc), ctx.vector(x.size, data=x)
            T.prolongate_and_add(fv, cv)
            assert rel(fv.download(), orc.prolongate(l, xc, fine=x, with_bc=True)) < 1e-13
            cv2 = ctx.vector(xc.size, data=xc)
            T.restrict_and_add(cv2, src)
            assert rel(cv2.download(), orc.restrict_and_add(l, xc, x, with_bc=True)) < 1e-12
            T.clear()
    lmax = cube.max_level
    x = cube.seeded_vector(lmax, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    for _ in range(2):
        solver.vmult(dst, src)
        assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    rate, trace = solver.solve(True)
    orate, otrace = orc.solve(True)
    assert rate == pytest.approx(orate, rel=1e-6)
    # (one V-cycle per level reduces the residual of the 1e6-contrast problem by 0.15 only: the FMG iterate
    # carries the 1e-9 differences of the cycles amplified by the conditioning of the operator)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=2e-6)
    assert_same_cg(solver, orc)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=2e-6)
    solver.close()
    cube.close()
    orc.close()


def test_poisson_shell_harness_runs():
    """tools/poisson_shell.py (poisson_shell/program.cc): command line, cycle protocol (6- and 12-cell shells in
    turn), the program's output lines and table"""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "poisson_shell.py"), "2", "4000", "--cycles", "0:6"],
                         cwd=root, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    text = out.stdout.decode()
    assert "Testing FE_Q<3>(2)" in text and "Max size reached, terminating." in text
    for n_dofs in (78, 150, 490, 970, 3474):  # cycles 0 .. 4: (6 | 12) N^2 + 2 points per sphere, N + 1 spheres
        assert "Best timings for ndof = %d " % n_dofs in text and "L2 error with ndof = %d " % n_dofs in text
    assert "Number of degrees of freedom: 6930" in text  # cycle 5 exceeds the maximum size
    rows = [l.split() for l in text.splitlines() if l[:1].isdigit() and len(l.split()) == 11]
    assert [int(r[0]) for r in rows] == [6, 12, 48, 96, 384]
    assert all(int(r[9]) < 60 for r in rows)  # PCG iterations
