"""CPU-side checks of the product's host logic and of the C-ABI library (no compute calls, no GPU):
libmgx.so loads, exports every symbol include/*.h declares, fails loudly without a device, and the
host-side cube provider (mgx_cube) agrees with the oracle's independently written tables."""
import os
import re

import numpy as np
import pytest

from oracle import Oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as g
    if not os.path.exists(os.path.join(ROOT, "multigrid_amd", "libmgx.so")):
        g.build()
    from multigrid_amd import _lib
    return _lib.load()


def test_exports_every_declared_symbol(lib):
    from multigrid_amd import _lib
    declared = set()
    for hdr in ("mgx.h", "mgx_cube.h", "mgx_dg.h"):
        text = open(os.path.join(ROOT, "include", hdr)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        declared |= set(re.findall(r"\b(mgx_[a-z0-9_]+)\s*\(", text))
    assert len(declared) > 60
    for name in sorted(declared):
        assert hasattr(lib, name), "libmgx.so does not export %s" % name
        assert name in _lib.SIGNATURES, "python binding lacks %s" % name


def test_no_cpu_fallback(lib):
    """Without a HIP device the product refuses to run (this container has none); on the GPU box
    the context is created."""
    import multigrid_amd as mg
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(mg.MgxError) as e:
        mg.Context(0)
    assert e.value.status == -2  # MGX_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_touch_the_oracle():
    """oracle/ is test infrastructure: nothing in the product tree may reference it."""
    for base, _, files in os.walk(os.path.join(ROOT, "multigrid_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".hpp", ".h", "Makefile")):
                text = open(os.path.join(base, f), errors="ignore").read()
                assert "oracle" not in text.lower(), os.path.join(base, f)
    for f in os.listdir(os.path.join(ROOT, "include")):
        assert "oracle" not in open(os.path.join(ROOT, "include", f)).read().lower()


@pytest.mark.parametrize("p,ns,nr", [(1, 2, 2), (2, 1, 2), (4, 1, 3), (4, 3, 1), (5, 3, 1), (8, 1, 1), (9, 1, 1)])
def test_cube_provider_matches_oracle(lib, p, ns, nr):
    import multigrid_amd as mg
    c = mg.Cube(p, ns, nr, numbering="cell")  # the oracle's entity order: tables must be identical
    o = Oracle(p, ns, nr)
    assert c.n_levels == o.n_levels
    for l in range(c.n_levels):
        assert c.n_dofs(l) == (c.cells_per_dim(l) * p + 1) ** 3
        assert np.array_equal(c.idx27(l), o.idx27(l))
        assert np.array_equal(c.idx27_plain(l), o.idx27_plain(l))
        assert np.array_equal(c.dof_grid(l), o.dof_grid(l))
        assert np.array_equal(c.constrained(l), o.constrained(l))
        assert np.array_equal(c.cell_coords(l), o.cell_coords(l))
        np.testing.assert_allclose(c.rhs(l), o.rhs(l), rtol=1e-12, atol=1e-14)
        bi, bv = c.bc(l)
        oi, ov = o.bc(l)
        order = np.argsort(oi)
        assert np.array_equal(bi, oi[order])
        np.testing.assert_allclose(bv, ov[order], rtol=1e-14)
        if l > 0:
            ch = c.children(l)
            cc, cf = c.cell_coords(l - 1), c.cell_coords(l)
            for k in range(8):  # child k = x + 2y + 4z of its parent
                off = np.array([k & 1, (k >> 1) & 1, k >> 2])
                assert np.array_equal(cf[ch[:, k]], 2 * cc + off)
    np.testing.assert_allclose(c.shape_values(), o.shape_values(), atol=1e-15)
    np.testing.assert_allclose(c.colloc_grad(), o.colloc_grad(), atol=1e-13)
    np.testing.assert_allclose(c.qweights(), o.qweights(), atol=1e-16)
    np.testing.assert_allclose(c.prolong_1d(), o.prolong_1d(), atol=1e-15)
    # L2 error functional agrees on an arbitrary vector
    l = c.max_level
    v = c.seeded_vector(l, 3)
    assert c.l2_error(l, v) > 0
    c.close()
    o.close()


@pytest.mark.parametrize("p,ns,nr", [(1, 1, 3), (2, 3, 2), (4, 1, 3), (4, 3, 2), (5, 1, 2), (8, 1, 1)])
def test_brick_numbering_contract(lib, p, ns, nr):
    """The default (brick-grouped) numbering: same mesh, same data as the cell numbering up to a
    permutation; entity-contiguous with Dirichlet DoFs last (laplace_operator.h:272-340); and the
    DoFs numbered by one brick are one contiguous range with the brick interior first."""
    import multigrid_amd as mg
    c, r = mg.Cube(p, ns, nr), mg.Cube(p, ns, nr, numbering="cell")
    nb = 4 if p <= 4 else 2
    for l in range(c.n_levels):
        n = c.n_dofs(l)
        gc, gr = c.dof_grid(l).astype(np.int64), r.dof_grid(l).astype(np.int64)
        assert np.array_equal(np.sort(gc), np.arange(n))
        assert c.n_constrained(l) == r.n_constrained(l)
        assert np.array_equal(c.constrained(l), np.arange(n - c.n_constrained(l), n))
        assert np.array_equal(np.sort(gc[c.constrained(l)]), np.sort(gr[r.constrained(l)]))
        for name in ("rhs",):
            a, b = np.empty(n), np.empty(n)
            a[gc], b[gr] = getattr(c, name)(l), getattr(r, name)(l)
            np.testing.assert_allclose(a, b, rtol=1e-12, atol=1e-14)  # cell contributions add in another order
        # entity contiguity: the DoFs of entity e of cell c are base .. base+size-1 and lie on the
        # grid points the cell-numbered provider assigns to the same entity
        idx, ridx = c.idx27_plain(l).reshape(-1, 27), r.idx27_plain(l).reshape(-1, 27)
        for e in (0, 1, 4, 13, 14, 22, 26):
            cx, cy, cz = e % 3, (e // 3) % 3, e // 9
            size = (p - 1 if cx == 1 else 1) * (p - 1 if cy == 1 else 1) * (p - 1 if cz == 1 else 1)
            for k in range(size):
                assert np.array_equal(gc[idx[:, e] + k], gr[ridx[:, e] + k])
        if c.n_cells(l) < nb ** 3 or l < (2 if p <= 4 else 1):
            assert np.array_equal(gc, gr)  # no bricks on this level: plain first-touch order
            continue
        # unconstrained DoFs first numbered by brick b form one range; brick interiors lead
        n_free = n - c.n_constrained(l)
        cells = idx.shape[0]
        first = np.full(n, -1, np.int64)
        sizes = np.array([(p - 1 if e % 3 == 1 else 1) * (p - 1 if (e // 3) % 3 == 1 else 1) *
                          (p - 1 if e // 9 == 1 else 1) for e in range(27)])
        for cell in range(cells - 1, -1, -1):  # reverse order: the first cell wins
            for e in range(27):
                if sizes[e]:
                    first[idx[cell, e]:idx[cell, e] + sizes[e]] = cell // nb ** 3
        owner = first[:n_free]
        assert (np.diff(owner) >= 0).all()  # ranges of consecutive bricks, in brick order
    c.close()
    r.close()


def test_seeded_vector_is_numbering_independent(lib):
    import multigrid_amd as mg
    c = mg.Cube(3, 1, 2, numbering="cell")
    v = c.seeded_vector(2, 42)
    g = c.dof_grid(2)
    assert v.min() >= -1 and v.max() < 1 and abs(v.mean()) < 0.05
    lex = np.empty_like(v)
    lex[g] = v
    c2 = mg.Cube(3, 1, 2)
    lex2 = np.empty_like(v)
    lex2[c2.dof_grid(2)] = c2.seeded_vector(2, 42)
    assert np.array_equal(lex, lex2)
    c.close()
    c2.close()


def test_invalid_arguments(lib):
    import ctypes as C
    h = C.c_void_p()
    assert lib.mgx_cube_create(0, 1, 1, C.byref(h)) == -1
    assert lib.mgx_cube_create(4, 1, 12, C.byref(h)) != 0
    assert lib.mgx_cube_create(9, 3, 8, C.byref(h)) == -4  # 32-bit DoF index overflow refused


def test_shim_declares_the_reference_interface_and_compiles(tmp_path):
    """include/multigrid_shim.hpp mirrors the public members of the reference classes on the hot path:
    every one of them is declared, and a translation unit that calls all of them (tests/shim_check.cpp) compiles."""
    import re
    import subprocess
    src = open(os.path.join(ROOT, "include", "multigrid_shim.hpp")).read()

    def members(cls):
        body = src[src.index("  class " + cls):]
        body = body[:body.index("\n  };")]
        return set(re.findall(r"\b([a-z_0-9]+)\(", body))
    need = {
        # common/laplace_operator.h:60-124
        "LaplaceOperator": {"initialize", "clear", "vmult", "vmult_residual", "vmult_with_cg_update", "compute_residual",
                            "evaluate_coefficient", "compute_diagonal", "get_matrix_diagonal_inverse", "m",
                            "initialize_dof_vector"},
        # common/multigrid_solver.h:100-637
        "MultigridSolver": {"solve", "solve_cg", "vmult", "vmult_with_residual_update", "do_matvec", "do_matvec_smoother",
                            "compute_l2_error", "get_solution", "print_wall_times"},
        # common/laplace_operator_dg.h:350-2025, 2028-2256; common/multigrid_solver_dg.h:55-747
        "LaplaceOperatorCompactCombine": {"reinit", "m", "initialize_dof_vector", "get_penalty", "vmult", "vmult_residual",
                                          "vmult_with_cg_update", "vmult_with_chebyshev_update"},
        "JacobiTransformed": {"m", "vmult"},
        "MultigridSolverDG": {"vmult", "solve_cg"},
    }
    for cls, names in need.items():
        assert names <= members(cls), (cls, names - members(cls))
    subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"),
                    os.path.join(ROOT, "tests", "shim_check.cpp")], check=True)


@pytest.mark.parametrize("p", range(1, 10))
def test_embedding_is_symmetric_and_its_even_odd_form_reproduces_it(lib, p):
    """what the fused transfer forms rely on (Basis1D::P1eo, mgx_brick_device.hpp restrict_half / prolong_line): the 1D
    embedding of a parent into its two children is symmetric under reversal of both indices, and the half-size
    products with the sums and differences at mirrored positions give the dense restriction and prolongation"""
    import multigrid_amd as mg
    c = mg.Cube(p, 1, 1)
    P1 = c.prolong_1d()
    c.close()
    n, nh = p + 1, (p + 1) // 2
    assert P1.shape == (2 * p + 1, n)
    np.testing.assert_allclose(P1[::-1, ::-1], P1, atol=1e-14)
    he = 0.5 * (P1[:n, :nh] + P1[:n, ::-1][:, :nh])
    ho = 0.5 * (P1[:p, :nh] - P1[:p, ::-1][:, :nh])
    pc = P1[:n, p // 2]
    rng = np.random.default_rng(p)
    r = rng.standard_normal(2 * p + 1)
    re, ro = np.append(r[:p] + r[::-1][:p], r[p]), r[:p] - r[::-1][:p]
    o = np.zeros(n)
    se, so = he.T @ re, ho.T @ ro
    o[:nh], o[::-1][:nh] = se + so, se - so
    if p % 2 == 0:
        o[p // 2] = pc @ re
    np.testing.assert_allclose(o, P1.T @ r, atol=1e-13)
    cv = rng.standard_normal(n)
    ce, co = cv[:nh] + cv[::-1][:nh], cv[:nh] - cv[::-1][:nh]
    fe = he @ ce + (pc * cv[p // 2] if p % 2 == 0 else 0.)
    fo = ho @ co
    f = np.zeros(2 * p + 1)
    f[:p], f[::-1][:p], f[p] = fe[:p] + fo, fe[:p] - fo, fe[p]
    np.testing.assert_allclose(f, P1 @ cv, atol=1e-13)
