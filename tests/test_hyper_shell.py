"""The mesh of poisson_shell (BASELINE config 4): GridGenerator::hyper_shell(0, 0.5, 1.0, 6 | 12) +
refine_global (poisson_shell/program.cc:425-431) as the provider builds it (mgx_cube_create_shell), on
the CPU.  The reference holds no numbers for this program (parity unpinned); what pins the mesh
here are properties no implementation detail can fake -- the DoF count of the shell in closed form, the
volume of the shell, the entity-orientation contract of the compressed index table checked DoF by DoF
at creation, symmetry / definiteness / the constant null space of the operator the oracle assembles
on it, agreement of the provider's merged coefficient and right-hand side with the oracle's own
computation from the node coordinates, and the convergence order of the solution."""
import numpy as np
import pytest

import multigrid_amd as mg
from oracle import Oracle


@pytest.mark.parametrize("n_coarse", [6, 12])
@pytest.mark.parametrize("p", [1, 2, 3, 4, 5])
def test_dof_counts_multiplicities_and_volume(n_coarse, p):
    nr = 2 if p < 4 else 1
    c = mg.Cube(p, n_refine=nr, shell=n_coarse)
    vol = []
    for l in range(c.n_levels):
        N = p * 2 ** l  # lattice intervals per block direction
        on_sphere = n_coarse * N * N + 2  # Euler: points of a quadrangulated sphere with n_coarse N^2 quads
        assert c.n_cells(l) == n_coarse * 8 ** l
        assert c.n_dofs(l) == on_sphere * (N + 1)
        assert c.n_constrained(l) == 2 * on_sphere  # Dirichlet values on both spheres
        # three cube faces / three or four rhombi meet at a vertex of the polyhedron: 3 (and 4) cells around a
        # radial edge, up to 8 (6 at those edges) around a vertex
        m = c.entity_multiplicity(l)
        assert m[:, 13].min() == 1 and m[:, 13].max() == 1 and m.max() <= 8 and 3 in m
        idx = c.idx27(l)
        cons = c.constrained(l)
        assert np.array_equal(cons, np.arange(c.n_dofs(l) - cons.size, c.n_dofs(l)))  # numbered last
        assert (idx[idx != 0xFFFFFFFF] < c.n_dofs(l) - cons.size).all()
        assert np.unique(c.dof_grid(l)).size == c.n_dofs(l)
        if l > 0:
            assert np.array_equal(c.children(l).ravel(), np.arange(c.n_cells(l)))  # forest order
        vol.append(c.l2_error_parts(l, np.zeros(c.n_dofs(l)))[1])
    exact = 4. / 3. * np.pi * (1 - 0.5 ** 3)
    assert abs(vol[-1] - exact) < abs(vol[0] - exact) or abs(vol[0] - exact) < 1e-12
    if p >= 3:
        assert abs(vol[-1] - exact) < 2e-4 * exact
    c.close()


@pytest.mark.parametrize("n_coarse,p,nr", [(6, 2, 2), (12, 2, 1), (6, 4, 1), (12, 3, 1), (6, 1, 2)])
def test_oracle_on_the_shell_and_provider_tables(n_coarse, p, nr):
    c = mg.Cube(p, n_refine=nr, shell=n_coarse, problem="shell")
    o = Oracle(p, degree=3, n_cycles=1, mesh=c, problem="shell")
    rng = np.random.default_rng(n_coarse + p)
    for l in range(c.n_levels):
        n = c.n_dofs(l)
        x, y = rng.uniform(-1, 1, n), rng.uniform(-1, 1, n)
        free = np.ones(n, bool)
        free[c.constrained(l)] = False
        x[~free] = y[~free] = 0
        Ax, Ay = o.vmult(l, x), o.vmult(l, y)
        if free.any():  # (p = 1: every DoF of the coarse mesh lies on one of the two spheres)
            assert abs(x @ Ay - y @ Ax) < 1e-12 * abs(x @ Ay)
            assert x @ Ax > 0
        # the provider's merged coefficient, right-hand side and boundary values against the oracle's,
        # which starts from the node coordinates alone
        cq, oq = c.coef_q(l), o.coef_q(l)
        np.testing.assert_allclose(cq, np.transpose(oq, (0, 2, 1)), rtol=1e-10, atol=1e-12 * np.abs(oq).max())
        np.testing.assert_allclose(c.rhs(l), o.rhs(l), rtol=1e-9, atol=1e-10 * np.abs(o.rhs(l)).max())
        full_c, full_o = np.zeros(n), np.zeros(n)
        (ci, cv), (oi, ov) = c.bc(l), o.bc(l)
        full_c[ci], full_o[oi] = cv, ov
        np.testing.assert_allclose(full_c, full_o, rtol=1e-13, atol=1e-13)
    # transfers: R = P^T (with the 1/multiplicity weights of three blocks around an edge), and the
    # prolongation reproduces the coarse function: the embedding of a coarse field keeps constants
    if c.n_levels > 1:
        l = c.max_level
        xc, xf = rng.uniform(-1, 1, c.n_dofs(l - 1)), rng.uniform(-1, 1, c.n_dofs(l))
        Pxc = o.prolongate(l, xc, with_bc=False)
        Rxf = o.restrict_and_add(l, np.zeros(c.n_dofs(l - 1)), xf, with_bc=False)
        assert abs(xf @ Pxc - Rxf @ xc) < 1e-12 * abs(xf @ Pxc)
        np.testing.assert_allclose(o.prolongate(l, np.ones(c.n_dofs(l - 1)), with_bc=False), 1.0, rtol=1e-13)
    o.close()
    c.close()


@pytest.mark.parametrize("n_coarse", [6, 12])
def test_convergence_order_on_the_shell(n_coarse):
    """constant coefficient, u = prod sin(3 pi x_d): the error of the PCG solution falls with order p + 1"""
    p, errs = 2, []
    for nr in (2, 3, 4):
        c = mg.Cube(p, n_refine=nr, shell=n_coarse, problem="cube")
        o = Oracle(p, degree=3, n_cycles=1, mesh=c, problem="cube")
        its, _ = o.solve_cg()
        assert its <= 12
        errs.append(o.l2_error())
        o.close()
        c.close()
    assert np.log2(errs[1] / errs[2]) > p + 0.7  # 2.86 / 2.90 measured, -> 3


def _local_dofs(c, l, tables):
    """[n_cells, (p+1)^3]: DoF of every Gauss-Lobatto node of every cell through the compressed index table (27 entity
    starts per cell, lexicographic inside an entity: vector_access_reduced.h:11-505)"""
    p, n = c.degree, c.degree + 1
    out = np.empty((tables.shape[0], n ** 3), dtype=np.int64)
    for e in range(27):
        cx, cy, cz = e % 3, (e // 3) % 3, e // 9
        rng = [([0], [1 + o for o in range(p - 1)], [p])[cc] for cc in (cx, cy, cz)]
        k = 0
        for iz in rng[2]:
            for iy in rng[1]:
                for ix in rng[0]:
                    out[:, (iz * n + iy) * n + ix] = tables[:, e].astype(np.int64) + k
                    k += 1
    return out


@pytest.mark.parametrize("n_coarse,p,nr", [(6, 2, 1), (12, 2, 1), (6, 3, 2), (12, 4, 1), (6, 1, 2)])
def test_index_tables_glue_exactly_the_coincident_points(n_coarse, p, nr):
    """Topology from geometry alone, independent of how the provider identified the entities of neighbouring blocks: the
    physical Gauss-Lobatto points of all cells are enumerated on their own (points that coincide in space are one point:
    the union of the closed cells IS the shell), and the gluing the compressed index tables imply must be exactly that --
    a DoF has one position whatever cell looks at it, different DoFs sit at different positions, and there are as many
    DoFs as distinct points.  With the sphere radii and the volume (above) this pins the mesh the oracle and the GPU path
    run on without any table the two share."""
    c = mg.Cube(p, n_refine=nr, shell=n_coarse)
    for l in range(c.n_levels):
        X = np.transpose(c.cell_nodes(l), (0, 2, 1)).reshape(-1, 3)       # [cell * node, xyz]
        dof = _local_dofs(c, l, c.idx27_plain(l)).ravel()
        # independent enumeration: distinct positions up to a tolerance far below the node spacing (h / p^2 > 1e-3)
        key = np.round(X / 1e-9).astype(np.int64)
        _, point = np.unique(key, axis=0, return_inverse=True)
        point = point.ravel()
        n_points = point.max() + 1
        assert n_points == c.n_dofs(l)                                      # as many DoFs as points of the shell
        # one position per DoF, one DoF per position
        first = np.full(c.n_dofs(l), -1, dtype=np.int64)
        first[dof] = point
        assert (first >= 0).all() and np.array_equal(first[dof], point)
        back = np.full(n_points, -1, dtype=np.int64)
        back[point] = dof
        assert np.array_equal(back[point], dof)
        # the Dirichlet DoFs are the points on the two spheres (|x| = 0.5 and 1), and only those
        r = np.linalg.norm(X, axis=1)
        on_sphere = np.zeros(c.n_dofs(l), bool)
        on_sphere[dof[(np.abs(r - 0.5) < 1e-12) | (np.abs(r - 1.0) < 1e-12)]] = True
        cons = np.zeros(c.n_dofs(l), bool)
        cons[c.constrained(l)] = True
        assert np.array_equal(on_sphere, cons)
        assert r.min() > 0.5 - 1e-12 and r.max() < 1 + 1e-12
    c.close()
