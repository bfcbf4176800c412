"""The C++ boundary (include/multigrid_shim.hpp): tests/shim_check.cpp calls every member of the mirrored classes once
(LaplaceOperator, MultigridSolver, LaplaceOperatorCompactCombine, JacobiTransformed, MultigridSolverDG); this test
compiles it against libmgx.so, runs it on the GPU and compares every printed value with the same call made through
ctypes (which the other GPU tests compare with the oracle)."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

mg = pytest.importorskip("multigrid_amd")

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_shim_check(tmp_path):
    exe = str(tmp_path / "shim_check")
    libdir = os.path.dirname(mg._lib.LIB_PATH)
    subprocess.run(["g++", "-O1", "-std=c++17", "-I" + os.path.join(ROOT, "include"), "-o", exe,
                    os.path.join(ROOT, "tests", "shim_check.cpp"), "-L" + libdir, "-l:" + os.path.basename(mg._lib.LIB_PATH),
                    "-Wl,-rpath," + libdir], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, timeout=600).stdout.decode()
    vals = {}
    for line in out.splitlines():
        k, v = line.split(" ", 1)
        assert k != "error", v
        vals[k] = float(v)
    assert vals.get("done") == 1
    return vals


def close(a, b, tol=1e-11):
    return abs(a - b) <= tol * max(abs(a), abs(b), 1e-300)


def test_every_shim_member_against_ctypes(tmp_path):
    v = run_shim_check(tmp_path)
    ctx = mg.Context(0)
    norm = lambda vec: float(np.linalg.norm(vec.download().astype(np.float64)))  # noqa: E731
    # ---- LaplaceOperator ----
    cube = mg.Cube(4, 1, 3)
    l = cube.max_level
    op = mg.LaplaceOperator.from_cube(ctx, cube, l)
    n = cube.n_dofs(l)
    assert v["op_m"] == n
    seeded = lambda s: ctx.vector(n, data=cube.seeded_vector(l, s))  # noqa: E731
    x, b, y, r = seeded(1), seeded(2), ctx.vector(n), ctx.vector(n)
    op.vmult(y, x)
    assert close(v["vmult_l2"], norm(y))
    op.vmult_residual(b, x, r)
    assert close(v["vmult_residual_l2"], norm(r))
    op.compute_diagonal()
    assert close(v["diag_inverse_l2"], norm(op.get_matrix_diagonal_inverse()))
    q, p, xx = seeded(3), seeded(4), seeded(5)
    sums = op.vmult_with_cg_update(0.3, 0.7, b, q, p, xx)
    for name, s in zip(("qp", "rr", "qr", "qq"), sums):
        assert close(v["cg_update_" + name], s)
    assert close(v["cg_update_x_l2"], norm(xx))
    rhs, u0 = ctx.vector(n), ctx.vector(n)
    rq = cube.rhs_quadrature(l)
    rhs_q = ctx.vector(rq.size, data=rq)
    op.compute_residual(rhs, u0, rhs_q)
    assert close(v["compute_residual_l2"], norm(rhs))
    assert v["evaluate_coefficient_defect"] < 1e-12  # twice the coefficient through the per-point branch: twice the product
    op.clear()
    # ---- MultigridSolver<3,4,float,double> (iterative results of an fp32 V-cycle: 1e-6, as the oracle comparisons) ----
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F32)
    rate = solver.solve(False)[0]
    assert close(v["fmg_reduction"], rate, 1e-6)
    assert close(v["fmg_l2_error"], solver.compute_l2_error(), 1e-6)
    its, red = solver.solve_cg()
    assert v["cg_its"] == its and close(v["cg_reduction"], red, 1e-6)
    assert close(v["cg_l2_error"], solver.compute_l2_error(), 1e-6)
    assert close(v["solution_l2"], norm(solver.get_solution()), 1e-8)
    src, dst = seeded(7), ctx.vector(n)
    solver.vmult(dst, src)
    assert close(v["vcycle_l2"], norm(dst), 1e-6)
    res, upd = seeded(8), seeded(9)
    dots = solver.vmult_with_residual_update(res, upd, 0.25)
    assert close(v["residual_update_zr"], dots[0], 1e-6) and close(v["residual_update_zu"], dots[1], 1e-6)
    solver.matrix_dp(l).vmult(dst, src)
    assert close(v["solver_operator_vmult_l2"], norm(dst))
    solver.close()
    cube.close()
    # ---- DG ----
    cube3 = mg.Cube(3, 1, 3)
    dgs = mg.DGMultigridSolver(ctx, cube3, mg.DG_HERMITE, 3, mg.F32)
    A = dgs.matrix_dg_dp
    m = A.m()
    assert v["dg_m"] == m
    assert close(v["dg_penalty"], A.info()["penalty"][0])
    i = np.arange(m, dtype=np.float64)
    hx, hb, ho = np.sin(0.37 * i), np.cos(0.11 * i), np.sin(0.05 * i + 1.)
    xd, bd, od, yd = (A.initialize_dof_vector(a) for a in (hx, hb, ho, np.zeros(m)))
    A.vmult(yd, xd)
    assert close(v["dg_vmult_l2"], norm(yd))
    A.vmult_residual(yd, bd, xd)
    assert close(v["dg_vmult_residual_l2"], norm(yd))
    A.jacobi_vmult(yd, xd)
    assert close(v["dg_jacobi_l2"], norm(yd))
    A.vmult_with_chebyshev_update(bd, 2, 0.6, 0.2, xd, od)
    assert close(v["dg_chebyshev_l2"], norm(xd))
    qd, pd, xxd = (A.initialize_dof_vector(a) for a in (hb, ho, hx))
    sums = A.vmult_with_cg_update(0.3, 0.7, bd, qd, pd, xxd)
    assert close(v["dg_cg_update_qp"], sums[0]) and close(v["dg_cg_update_qq"], sums[3])
    rhsd, sold = dgs.initialize_dof_vector(hb), dgs.initialize_dof_vector()
    dgs.vmult(sold, rhsd)
    assert close(v["dg_vcycle_l2"], norm(sold), 1e-6)
    its, _ = dgs.solve_cg(rhsd, sold, 1e-9)
    assert v["dg_cg_its"] == its
    assert close(v["dg_solution_l2"], norm(sold), 1e-7)
    assert v["dg_smoother_degree"] == dgs.smoother_info()["degree"]
    dgs.close()
    cube3.close()
    ctx.close()
