"""GPU parity tests: the HIP path (through the C ABI of include/mgx.h) against the CPU oracle on the
same inputs.  Tolerances (fp64): the arithmetic is the same sum-factorisation in a different
summation order (thread-tile sweeps, atomic scatter-add order), so agreement is at round-off
level: 1e-12 relative to the max-norm for one operator application, 1e-9 for composite cycles
(SURVEY.md 8a/BASELINE.md 3).  fp32 V-cycle: 2e-4."""
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


# levels with >= 64 cells in Morton bricks and p <= 4 run the brick cell loop (mgx_brick.hip), the
# others the per-cell kernel (mgx_kernels.hip): both are covered
CASES = [(1, 2, 1), (1, 1, 3), (2, 1, 2), (2, 1, 3), (3, 1, 2), (3, 3, 2), (4, 1, 2), (4, 1, 3), (4, 3, 1),
         (4, 3, 2), (5, 1, 1), (5, 1, 2), (6, 1, 1), (7, 1, 1), (8, 1, 1), (8, 1, 2), (9, 1, 1)]


@pytest.mark.parametrize("p,ns,nr", CASES)
def test_vmult_and_residual(ctx, p, ns, nr):
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr)
    for l in range(cube.n_levels):
        op = mg.LaplaceOperator.from_cube(ctx, cube, l)
        x = cube.seeded_vector(l, 1)
        b = cube.seeded_vector(l, 2)
        src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
        op.vmult(dst, src)
        assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
        op.vmult_residual(rhs, src, dst)
        assert rel(dst.download(), orc.vmult_residual(l, b, x)) < 1e-12
        op.compute_diagonal()
        assert rel(op.get_matrix_diagonal_inverse().download(), orc.inv_diag(l)) < 1e-13
        op.clear()
    cube.close()
    orc.close()


@pytest.mark.parametrize("p,ns,nr", [(4, 1, 3), (2, 1, 3), (3, 3, 2)])
@pytest.mark.parametrize("wide_max", ["0", "100000"])
def test_cell_by_cell_brick_kernel(monkeypatch, p, ns, nr, wide_max):
    """The cell-by-cell form of the brick loop (round-1 kernels, built only into the cross-check library:
    make -C multigrid_amd/csrc crosscheck, MGX_LIB_PATH=multigrid_amd/libmgx_crosscheck.so; production: the
    macro-element form, which every other test runs), in its 256-thread (brick_wide_max = 0) and its
    512-thread form."""
    if not mg._lib.load().mgx_has_cells_form():
        pytest.skip("library built without the cell-by-cell cross-check kernels")
    ctx = mg.Context(0, options={"cells_form": 1, "brick_wide_max": float(wide_max)})
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=3)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    l = cube.max_level
    x, b = cube.seeded_vector(l, 1), cube.seeded_vector(l, 2)
    src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
    A = solver.matrix_dp(l)
    A.vmult(dst, src)
    assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
    A.vmult_residual(rhs, src, dst)
    assert rel(dst.download(), orc.vmult_residual(l, b, x)) < 1e-12
    sm = solver.smoother(l)
    sm.vmult(dst, rhs)
    x_ref = orc.cheb_vmult(l, b)
    assert rel(dst.download(), x_ref) < 1e-10
    sm.step(dst, rhs)
    assert rel(dst.download(), orc.cheb_step(l, x_ref, b)) < 1e-10
    solver.vmult(dst, src)
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    solver.close()
    cube.close()
    orc.close()
    ctx.close()


def test_cell_by_cell_brick_kernel_in_the_crosscheck_library():
    """The six cases above need the cell-by-cell kernels, which the production library does not carry: they run here
    against multigrid_amd/libmgx_crosscheck.so (built by __graft_entry__.build(): make crosscheck) in ONE child process
    (a process loads one build of the library)."""
    import os
    import subprocess
    import sys
    if mg._lib.load().mgx_has_cells_form():
        pytest.skip("this process already runs the cross-check library")
    lib = os.path.join(os.path.dirname(mg._lib.LIB_PATH), "libmgx_crosscheck.so")
    if not os.path.exists(lib):
        pytest.skip("libmgx_crosscheck.so not built (make -C multigrid_amd/csrc crosscheck)")
    env = dict(os.environ, MGX_LIB_PATH=lib)
    out = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-k",
                          "test_cell_by_cell_brick_kernel and not crosscheck", "-p", "no:cacheprovider"],
                         env=env, capture_output=True, timeout=900)
    tail = out.stdout.decode()[-600:]
    assert out.returncode == 0 and "6 passed" in tail, tail


@pytest.mark.parametrize("p,nr", [(4, 2), (4, 3), (2, 4)])
def test_operator_from_foreign_tables(ctx, p, nr):
    """Drop-in scenario: tables come from the caller (here: the oracle's own arrays), not from
    mgx_cube, and without the brick colour hint (the library colours the bricks itself)."""
    orc = Oracle(p, 1, nr)
    l = nr
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
    h = orc.cell_size(l)
    for i, v in enumerate([h, h, h, 0, 0, 0]):
        d.coef[i] = v
    d.shape_values = S.ctypes.data_as(mg._lib.f64p)
    d.colloc_grad = D.ctypes.data_as(mg._lib.f64p)
    d.qweights = w.ctypes.data_as(mg._lib.f64p)
    d.brick_colour = None
    op = mg.LaplaceOperator(ctx, d)
    x = np.random.default_rng(3).uniform(-1, 1, orc.n_dofs(l))
    b = np.random.default_rng(4).uniform(-1, 1, orc.n_dofs(l))
    src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
    dst.upload(np.full(x.size, np.nan))  # the loop must overwrite, never read, the old dst
    op.vmult(dst, src)
    assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
    first = dst.download()
    op.vmult(dst, src)
    assert np.array_equal(first, dst.download())  # atomic-free => bitwise reproducible
    dst.upload(np.full(x.size, np.nan))
    op.vmult_residual(rhs, src, dst)
    assert rel(dst.download(), orc.vmult_residual(l, b, x)) < 1e-12
    op.clear()


def test_error_paths(ctx):
    cube = mg.Cube(3, 1, 1)
    d = cube.operator_desc(1)
    bad = cube.idx27(1).copy().ravel()
    bad[5] = cube.n_dofs(1) + 7  # out of range compressed index must be refused on the host
    d.idx27 = bad.ctypes.data_as(mg._lib.u32p)
    with pytest.raises(mg.MgxError):
        mg.LaplaceOperator(ctx, d)
    op = mg.LaplaceOperator.from_cube(ctx, cube, 1)
    v = op.initialize_dof_vector()
    with pytest.raises(mg.MgxError):
        op.vmult(v, v)  # aliasing refused (laplace_operator.h:573-601 needs distinct vectors)
    d2 = cube.operator_desc(1)
    d2.coef[3] = 10.0 * d2.coef[0]  # an indefinite coefficient tensor is refused (a definite full one is
    with pytest.raises(mg.MgxError):  # the sheared-mesh case of tests/test_gpu_shell.py)
        mg.LaplaceOperator(ctx, d2)
    op.clear()
    cube.close()


@pytest.mark.parametrize("p,ns,nr", [(2, 1, 2), (4, 1, 2), (4, 1, 3), (3, 3, 2), (4, 3, 1), (7, 1, 1)])
def test_chebyshev(ctx, p, ns, nr):
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=3)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    for l in range(cube.n_levels):
        sm = solver.smoother(l)
        gi, oi = sm.info(), orc.cheb_info(l)
        assert gi["degree"] == oi["degree"] and gi["cg_its"] == oi["cg_its"]
        for k in ("lambda_max", "theta", "delta"):
            assert gi[k] == pytest.approx(oi[k], rel=1e-8)
        b = cube.seeded_vector(l, 7)
        bd, xd = ctx.vector(b.size, data=b), ctx.vector(b.size)
        sm.vmult(xd, bd)
        x_ref = orc.cheb_vmult(l, b)
        assert rel(xd.download(), x_ref) < 1e-10
        sm.step(xd, bd)
        assert rel(xd.download(), orc.cheb_step(l, x_ref, b)) < 1e-10
    solver.close()
    cube.close()
    orc.close()


@pytest.mark.parametrize("p,ns,nr", [(1, 2, 1), (2, 1, 2), (4, 1, 2), (4, 3, 1), (8, 1, 1)])
def test_transfers(ctx, p, ns, nr):
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr)
    ops = [mg.LaplaceOperator.from_cube(ctx, cube, l) for l in range(cube.n_levels)]
    for l in range(1, cube.n_levels):
        tr = mg.Transfer(ops[l - 1], ops[l], cube.children(l), cube.prolong_1d())
        xc = cube.seeded_vector(l - 1, 11)
        xf = cube.seeded_vector(l, 12)
        dc, df = ctx.vector(xc.size, data=xc), ctx.vector(xf.size, data=xf)
        out = ctx.vector(xf.size, data=np.full(xf.size, 7.0))
        tr.prolongate(out, dc, with_constraints=False)  # overwrite
        assert rel(out.download(), orc.prolongate(l, xc, with_bc=False)) < 1e-13
        out.upload(xf)
        tr.prolongate_and_add(out, dc, with_constraints=True)
        assert rel(out.download(), orc.prolongate(l, xc, fine=xf, with_bc=True)) < 1e-13
        outc = ctx.vector(xc.size, data=xc)
        tr.restrict_and_add(outc, df, with_constraints=True)
        assert rel(outc.download(), orc.restrict_and_add(l, xc, xf, with_bc=True)) < 1e-13
        outc.upload(xc)
        tr.restrict_and_add(outc, df, with_constraints=False)
        assert rel(outc.download(), orc.restrict_and_add(l, xc, xf, with_bc=False)) < 1e-13
        tr.clear()
    for o in ops:
        o.clear()
    cube.close()
    orc.close()


@pytest.mark.parametrize("p,ns,nr,degree,ncyc", [(4, 1, 3, 3, 1), (4, 1, 3, 3, 2), (4, 3, 1, 2, 1), (2, 1, 3, 3, 1),
                                                  (8, 1, 2, 3, 1), (4, 1, 0, 3, 1), (4, 3, 2, 4, 1),
                                                  (3, 1, 3, 2, 2), (1, 1, 4, 3, 1)])
def test_vcycle_fmg_pcg(ctx, p, ns, nr, degree, ncyc):
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=degree, n_cycles=ncyc)
    solver = mg.MultigridSolver(ctx, cube, degree, degree, ncyc, mg.F64)
    lmax = cube.max_level
    x = cube.seeded_vector(lmax, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=np.full(x.size, np.nan))
    solver.vmult(dst, src)  # MultigridSolver::vmult = one V-cycle; every entry of dst is written
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    assert np.array_equal(src.download(), x)  # the source is left alone
    rate, trace = solver.solve(True)
    orate, otrace = orc.solve(True)
    if nr > 0:
        assert rate == pytest.approx(orate, rel=1e-6)
        np.testing.assert_allclose(trace[1:, 0], otrace[1:, 1], rtol=1e-9)  # residual start
        np.testing.assert_allclose(trace[1:, 1], otrace[1:, 2], rtol=1e-6)  # residual end
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-8)
    assert rel(solver.get_solution().download(), orc.solution(lmax)) < 1e-9
    its, red = assert_same_cg(solver, orc)
    ored = orc.solve_cg()[1]
    assert red == pytest.approx(ored, rel=1e-5)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-8)
    solver.close()
    cube.close()
    orc.close()


def test_default_brick_threshold(monkeypatch):
    """With the production threshold (bricks from 512 per level on, 1024 for p <= 2) small levels use the per-cell
    kernel and the finest level of a 64^3 mesh (4096 bricks) the brick loop; same results."""
    monkeypatch.delenv("MGX_BRICK_MIN", raising=False)
    monkeypatch.delenv("MGX_RESTRICT_COLOUR_MIN", raising=False)
    ctx = mg.Context(0)  # thresholds are read when the context is created
    p, nr = 2, 6
    cube = mg.Cube(p, 1, nr)
    orc = oracle_for(cube, p, 1, nr, degree=3, n_cycles=1)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    x = cube.seeded_vector(nr, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    solver.matrix_dp(nr).vmult(dst, src)
    assert rel(dst.download(), orc.vmult(nr, x)) < 1e-12
    solver.vmult(dst, src)
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    solver.close()
    cube.close()
    orc.close()
    ctx.close()


def test_production_thresholds_p4_against_oracle(monkeypatch):
    """FE_Q(4) on 64^3 cells (16 974 593 DoFs, 4096 bricks on the finest level): the production
    configuration -- default brick / colour thresholds, macro-element brick loop on the fine levels,
    per-cell kernel and graph replay on the coarse ones -- against the oracle, not against another
    path of the same library."""
    monkeypatch.delenv("MGX_BRICK_MIN", raising=False)
    monkeypatch.delenv("MGX_RESTRICT_COLOUR_MIN", raising=False)
    ctx = mg.Context(0)  # thresholds are read when the context is created
    p, nr = 4, 6
    cube = mg.Cube(p, 1, nr)
    orc = oracle_for(cube, p, 1, nr, degree=3, n_cycles=1)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F64)
    x, b = cube.seeded_vector(nr, 5), cube.seeded_vector(nr, 6)
    src, rhs, dst = ctx.vector(x.size, data=x), ctx.vector(x.size, data=b), ctx.vector(x.size)
    A = solver.matrix_dp(nr)
    A.vmult(dst, src)
    assert rel(dst.download(), orc.vmult(nr, x)) < 1e-12
    A.vmult_residual(rhs, src, dst)
    assert rel(dst.download(), orc.vmult_residual(nr, b, x)) < 1e-12
    sm = solver.smoother(nr)
    sm.vmult(dst, rhs)
    x_ref = orc.cheb_vmult(nr, b)
    assert rel(dst.download(), x_ref) < 1e-10
    sm.step(dst, rhs)
    assert rel(dst.download(), orc.cheb_step(nr, x_ref, b)) < 1e-10
    solver.vmult(dst, src)
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    solver.vmult(dst, src)  # coarse levels replayed from the HIP graph
    assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    for v in (src, rhs, dst):
        v.free()
    solver.close()
    cube.close()
    orc.close()
    ctx.close()


@pytest.mark.parametrize("p,nr,brick_min", [(4, 4, None), (3, 3, "4000000000"), (2, 4, "4000000000"), (8, 2, None)])
def test_per_cell_levels_are_reproducible_and_match_the_oracle(monkeypatch, p, nr, brick_min):
    """Levels below the brick threshold (production: < 512 bricks; "4000000000": every level) run the per-cell
    kernel with the ordered assembly instead of atomic adds, their restriction and diagonal likewise:
    operator, diagonal, smoother parameters, V-cycle and PCG against the oracle, and every result bitwise
    the same when computed twice on two solvers (nothing on the path depends on the order in which
    workgroups run)."""
    if brick_min is None:
        monkeypatch.delenv("MGX_BRICK_MIN", raising=False)
    else:
        monkeypatch.setenv("MGX_BRICK_MIN", brick_min)
    monkeypatch.delenv("MGX_RESTRICT_COLOUR_MIN", raising=False)
    c = mg.Context(0)
    cube = mg.Cube(p, 1, nr)
    orc = oracle_for(cube, p, 1, nr, degree=3, n_cycles=1)
    results = []
    for attempt in range(2):
        solver = mg.MultigridSolver(c, cube, 3, 3, 1, mg.F64)
        out = []
        for l in range(cube.n_levels):
            A = solver.matrix_dp(l)
            x, b = cube.seeded_vector(l, 11), cube.seeded_vector(l, 12)
            src, rhs, dst = c.vector(x.size, data=x), c.vector(x.size, data=b), c.vector(x.size)
            dst.upload(np.full(x.size, np.nan))
            A.vmult(dst, src)
            assert rel(dst.download(), orc.vmult(l, x)) < 1e-12
            out.append(dst.download())
            A.vmult_residual(rhs, src, dst)
            assert rel(dst.download(), orc.vmult_residual(l, b, x)) < 1e-12
            out.append(dst.download())
            out.append(A.get_matrix_diagonal_inverse().download())
            assert rel(out[-1], orc.inv_diag(l)) < 1e-12
            gi, oi = solver.smoother(l).info(), orc.cheb_info(l)
            assert gi["degree"] == oi["degree"] and gi["cg_its"] == oi["cg_its"]
            assert gi["lambda_max"] == pytest.approx(oi["lambda_max"], rel=1e-8)
            out.append(np.array([gi["lambda_max"], gi["lambda_min"]]))
            for v in (src, rhs, dst):
                v.free()
        lmax = cube.max_level
        x = cube.seeded_vector(lmax, 5)
        src, dst = c.vector(x.size, data=x), c.vector(x.size)
        for _ in range(3):  # eager, captured, replayed
            solver.vmult(dst, src)
            assert rel(dst.download(), orc.vcycle(x)) < 1e-9
            out.append(dst.download())
        assert_same_cg(solver, orc)
        out.append(solver.cg_history())
        out.append(solver.get_solution().download())
        results.append(out)
        solver.close()
    for a, b in zip(*results):
        assert np.array_equal(a, b)
    cube.close()
    orc.close()
    c.close()


def test_level_errors_of_the_analysed_solve(ctx):
    """multigrid_solver.h:420-424, 468-472: L2 error of every level before and after its cycles"""
    cube = mg.Cube(4, 1, 3)
    orc = oracle_for(cube, 4, 1, 3, degree=3, n_cycles=2)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 2, mg.F64)
    rate, trace, errors = solver.solve(True, level_errors=True)
    orate, otrace = orc.solve(True)
    assert rate == pytest.approx(orate, rel=1e-6)
    np.testing.assert_allclose(errors[1:, 0], otrace[1:, 0], rtol=1e-8)  # error start
    np.testing.assert_allclose(trace[1:, 0], otrace[1:, 1], rtol=1e-9)   # residual start
    np.testing.assert_allclose(trace[1:, 1], otrace[1:, 2], rtol=1e-6)   # residual end
    np.testing.assert_allclose(errors[1:, 1], otrace[1:, 3], rtol=1e-8)  # error end
    solver.close()
    cube.close()
    orc.close()


def test_poisson_cube_driver_reproduces_the_readme_row():
    """the C++ driver (reference CLI, protocol and table on the shim classes) at README.md:143:
    ./poisson_cube 4 30000 40000 2 3 3 s  ->  512 cells, 35937 DoFs, mixed precision"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "multigrid_amd", "poisson_cube")
    out = subprocess.run([exe, "4", "30000", "40000", "2", "3", "3", "s", "f32"], cwd=root, capture_output=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    text = out.stdout.decode()
    for l in range(1, 4):  # the four analysis lines per level of solve(true)
        for key in ("error start         level %d:", "residual norm start level %d:", "residual norm end   level %d:",
                    "error end           level %d:"):
            assert key % l in text, key % l
    rows = [ln.split() for ln in text.splitlines() if ln.split()[:2] == ["512", "35937"]]
    assert rows, text[-2000:]
    r = rows[-1]
    # cells dofs mv_outer mv_inner reduction fmg_L2error rate fmg_time cg_L2error rate cg_time cg_its cg_reduction
    assert float(r[4]) == pytest.approx(1.319e-01, rel=0.10)
    assert float(r[5]) == pytest.approx(4.037e-04, rel=0.05)
    assert float(r[8]) == pytest.approx(3.822e-04, rel=0.01)
    assert int(r[11]) == 8
    assert float(r[12]) == pytest.approx(6.689e-02, rel=0.05)


@pytest.mark.parametrize("p,ns,nr", [(4, 1, 3), (2, 1, 3), (3, 3, 2), (8, 1, 2), (4, 1, 1)])
def test_vmult_with_cg_update(ctx, p, ns, nr):
    """laplace_operator.h:638-719: x += alpha p; p = beta p + q; q = A p and the four sums, fused
    into the brick loop (levels with bricks) or run one after the other (coarse levels)"""
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr)
    for l in range(cube.n_levels):
        A = mg.LaplaceOperator.from_cube(ctx, cube, l)
        vecs = [cube.seeded_vector(l, s) for s in (11, 12, 13, 14)]
        for alpha, beta in ((0., 0.), (0.37, 0.81)):
            r, q, pp, x = (ctx.vector(v.size, data=v) for v in vecs)
            sums = A.vmult_with_cg_update(alpha, beta, r, q, pp, x)
            osums, oq, op_, ox = orc.vmult_with_cg_update(l, alpha, beta, *vecs)
            np.testing.assert_allclose(sums, osums, rtol=1e-11, atol=1e-11 * abs(osums).max())
            assert rel(q.download(), oq) < 1e-12
            assert rel(pp.download(), op_) < 1e-14
            assert rel(x.download(), ox) < 1e-14
        A.clear()
    cube.close()
    orc.close()


@pytest.mark.parametrize("number", ["f64", "f32"])
def test_fused_pcg(ctx, number):
    """multigrid_solver.h:516-619 and the PCG built on the two merged operations: same iteration
    count and error as the plain PCG and as the oracle's"""
    vf = number == "f32"
    cube = mg.Cube(4, 1, 3)
    orc = oracle_for(cube, 4, 1, 3, degree=3, n_cycles=1, vfloat=vf)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F32 if vf else mg.F64)
    l = cube.max_level
    res, upd = cube.seeded_vector(l, 21), cube.seeded_vector(l, 22)
    for factor in (0., -0.43):
        r, u = ctx.vector(res.size, data=res), ctx.vector(res.size, data=upd)
        out = solver.vmult_with_residual_update(r, u, factor)
        oout, ores, oupd = orc.vmult_with_residual_update(res, upd, factor)
        tol = 2e-4 if vf else 1e-9
        np.testing.assert_allclose(out, oout[:2], rtol=tol)
        assert rel(r.download(), ores) < 1e-14
        assert rel(u.download(), oupd) < tol
    its, red = solver.solve_cg()
    err = solver.compute_l2_error()
    fits, fred = solver.solve_cg_fused()
    ferr = solver.compute_l2_error()
    oits, ored = orc.solve_cg()
    assert fits == its == oits
    assert fred == pytest.approx(red, rel=1e-3 if vf else 1e-6)
    assert ferr == pytest.approx(err, rel=1e-6)
    solver.close()
    cube.close()
    orc.close()


def test_readme_known_answers_on_gpu(ctx):
    """README.md:143 (512 cells): the GPU path itself reproduces the reference's printed numbers."""
    cube = mg.Cube(4, 1, 3)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 2, mg.F32)  # README run: fp32 V-cycle, 2 cycles
    rate, _ = solver.solve(True)
    assert rate == pytest.approx(1.319e-01, rel=0.10)
    assert solver.compute_l2_error() == pytest.approx(4.037e-04, rel=0.05)
    its, red = solver.solve_cg()
    assert its == 8
    assert red == pytest.approx(6.689e-02, rel=0.05)
    assert solver.compute_l2_error() == pytest.approx(3.822e-04, rel=0.01)
    solver.close()
    cube.close()


@pytest.mark.parametrize("p,ns,nr", [(4, 1, 3), (2, 1, 3), (5, 1, 2), (8, 1, 2), (3, 3, 2)])
def test_mixed_precision_vcycle(ctx, p, ns, nr):
    """the reference's default: fp32 V-cycle (fp32 level operators, smoothers, transfers) inside the
    fp64 outer iteration, against the oracle run the same way"""
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=3, n_cycles=1, vfloat=True)
    solver = mg.MultigridSolver(ctx, cube, 3, 3, 1, mg.F32)
    lmax = cube.max_level
    x = cube.seeded_vector(lmax, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    solver.vmult(dst, src)
    assert rel(dst.download(), orc.vcycle(x)) < 2e-4
    # the fp32 level operator itself
    A = solver.matrix(lmax)
    s32, d32 = ctx.vector(x.size, mg.F32, x), ctx.vector(x.size, mg.F32)
    A.vmult(d32, s32)
    x32 = x.astype(np.float32).astype(np.float64)
    assert rel(d32.download().astype(np.float64), orc.vmult(lmax, x32)) < 3e-6
    solver.solve(False)
    orc.solve(False)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-4)
    solver.close()
    cube.close()
    orc.close()


def test_vector_kernels(ctx):
    rng = np.random.default_rng(9)
    n = 100003
    a, b = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
    da, db = ctx.vector(n, data=a), ctx.vector(n, data=b)
    assert ctx.dot(da, db) == pytest.approx(np.dot(a, b), rel=1e-12)
    assert ctx.l2_norm(da) == pytest.approx(np.linalg.norm(a), rel=1e-13)
    lib = ctx.lib
    mg.check(lib.mgx_sadd(ctx.h, mg.F64, da.ptr, -1.0, 1.0, db.ptr, n))
    np.testing.assert_array_equal(da.download(), -a + b)
    f = ctx.vector(n, mg.F32)
    mg.check(lib.mgx_copy_cast(ctx.h, f.ptr, mg.F32, db.ptr, mg.F64, n))
    np.testing.assert_array_equal(f.download(), b.astype(np.float32))
    mg.check(lib.mgx_add_cast(ctx.h, db.ptr, mg.F64, f.ptr, mg.F32, n))
    np.testing.assert_array_equal(db.download(), b + b.astype(np.float32).astype(np.float64))
    z = ctx.vector(0)
    assert ctx.dot(z, z) == 0.0  # empty input


@pytest.mark.parametrize("p,ns,nr,degree", [(4, 1, 3, 3), (2, 1, 3, 2), (4, 3, 1, 4), (8, 1, 2, 3), (3, 1, 3, 1)])
def test_fourth_kind_chebyshev_smoother(ctx, p, ns, nr, degree):
    """PolynomialType::fourth_kind, the choice of the reference's MultigridSolver<dim,p,Number,Number>
    specialisation (multigrid_solver.h:951-952): smoother, V-cycle, FMG and PCG against the oracle
    (whose recurrence tests/test_oracle_properties.py pins to the closed form), then back to the first kind
    on the same solver (the replayed graph of the coarse levels has to be dropped)"""
    cube = mg.Cube(p, ns, nr)
    orc = oracle_for(cube, p, ns, nr, degree=degree, polynomial="fourth_kind")
    first = oracle_for(cube, p, ns, nr, degree=degree)
    solver = mg.MultigridSolver(ctx, cube, degree, degree, 1, mg.F64, polynomial="fourth_kind")
    for l in range(cube.n_levels):
        sm = solver.smoother(l)
        gi, oi = sm.info(), orc.cheb_info(l)
        assert gi["degree"] == oi["degree"]
        assert gi["delta"] == pytest.approx(oi["delta"], rel=1e-8)
        if l > 0:
            assert gi["delta"] == gi["lambda_max"]
        b = cube.seeded_vector(l, 7)
        bd, xd = ctx.vector(b.size, data=b), ctx.vector(b.size)
        sm.vmult(xd, bd)
        x_ref = orc.cheb_vmult(l, b)
        assert rel(xd.download(), x_ref) < 1e-10
        sm.step(xd, bd)
        assert rel(xd.download(), orc.cheb_step(l, x_ref, b)) < 1e-10
    lmax = cube.max_level
    x = cube.seeded_vector(lmax, 5)
    src, dst = ctx.vector(x.size, data=x), ctx.vector(x.size)
    for _ in range(3):  # eager, captured, replayed
        solver.vmult(dst, src)
        assert rel(dst.download(), orc.vcycle(x)) < 1e-9
    rate, trace = solver.solve(True)
    orate, otrace = orc.solve(True)
    assert rate == pytest.approx(orate, rel=1e-6)
    np.testing.assert_allclose(trace[1:, 0], otrace[1:, 1], rtol=1e-9)
    assert solver.compute_l2_error() == pytest.approx(orc.l2_error(), rel=1e-8)
    assert_same_cg(solver, orc)
    mg.check(ctx.lib.mgx_solver_set_polynomial_type(solver.h, 0))
    solver.vmult(dst, src)
    assert rel(dst.download(), first.vcycle(x)) < 1e-9
    solver.close()
    cube.close()
    orc.close()
    first.close()
