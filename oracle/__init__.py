"""ctypes binding of the CPU oracle (oracle/libmg_oracle.so).

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py -- never from the product package (multigrid_amd).
See oracle/mg_oracle.h for the parity status and the reference citations.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libmg_oracle.so")


def build(force=False):
    """Compile oracle/libmg_oracle.so with gcc (no GPU, no reference sources involved)."""
    srcs = [os.path.join(_HERE, f) for f in ("mg_oracle.c", "mg_oracle_num.inc", "mg_oracle.h")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmg_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _LIB_PATH


_lib = None

_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


def lib():
    global _lib
    if _lib is not None:
        return _lib
    # many small parallel regions: sleeping waiters behave far better than spinning ones when the
    # host grants fewer CPUs than it shows (GPU boxes hand out a cgroup share)
    os.environ.setdefault("OMP_WAIT_POLICY", "passive")
    build()
    L = C.CDLL(_LIB_PATH)
    vp = C.c_void_p
    L.orc_create.restype = vp
    L.orc_create.argtypes = [C.c_int] * 6
    L.orc_create_box.restype = vp
    L.orc_create_box.argtypes = [C.c_int] * 8
    L.orc_create_mapped.restype = vp
    L.orc_create_mapped.argtypes = [C.c_int] * 4 + [C.c_double, C.c_double] + [C.c_int] * 6
    L.orc_set_affine_coef.argtypes = [vp, C.c_int, _f64p]
    L.orc_coef_q.restype = C.POINTER(C.c_double)
    L.orc_coef_q.argtypes = [vp, C.c_int]
    L.orc_cells_per_dim3.argtypes = [vp, C.c_int, C.POINTER(C.c_int * 3)]
    L.orc_destroy.argtypes = [vp]
    for name in ("orc_n_levels", "orc_degree"):
        getattr(L, name).restype = C.c_int
        getattr(L, name).argtypes = [vp]
    for name in ("orc_n_cells", "orc_n_dofs", "orc_n_constrained"):
        getattr(L, name).restype = C.c_uint32
        getattr(L, name).argtypes = [vp, C.c_int]
    L.orc_cells_per_dim.restype = C.c_int
    L.orc_cells_per_dim.argtypes = [vp, C.c_int]
    for name in ("orc_idx27", "orc_idx27_plain", "orc_constrained", "orc_cell_coords", "orc_dof_grid"):
        getattr(L, name).restype = _u32p
        getattr(L, name).argtypes = [vp, C.c_int]
    for name in ("orc_shape_values", "orc_colloc_grad", "orc_qweights", "orc_qpoints", "orc_gll",
                 "orc_prolong_1d"):
        getattr(L, name).restype = _f64p
        getattr(L, name).argtypes = [vp]
    for name in ("orc_rhs", "orc_inv_diag", "orc_solution"):
        getattr(L, name).restype = _f64p
        getattr(L, name).argtypes = [vp, C.c_int]
    L.orc_h.restype = C.c_double
    L.orc_h.argtypes = [vp, C.c_int]
    L.orc_set_polynomial_type.argtypes = [vp, C.c_int]
    L.orc_reset_smoother.argtypes = [vp, C.c_int, C.c_double, C.c_int, C.c_int]
    L.orc_cheb_info.argtypes = [vp, C.c_int] + [_f64p] * 4 + [C.POINTER(C.c_int)] * 2
    L.orc_bc.restype = C.c_uint32
    L.orc_bc.argtypes = [vp, C.c_int, _u32p, _f64p]
    L.orc_vmult.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.orc_vmult_residual.argtypes = [vp, C.c_int, _f64p, _f64p, _f64p]
    L.orc_vmult_with_cg_update.argtypes = [vp, C.c_int, C.c_double, C.c_double, _f64p, _f64p, _f64p, _f64p, _f64p]
    L.orc_vmult_with_residual_update.argtypes = [vp, _f64p, _f64p, C.c_double, _f64p]
    L.orc_vmult_dense_lex.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.orc_cheb_vmult.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.orc_cheb_step.argtypes = [vp, C.c_int, _f64p, _f64p]
    L.orc_prolongate.argtypes = [vp, C.c_int, _f64p, _f64p, C.c_int, C.c_int]
    L.orc_restrict_and_add.argtypes = [vp, C.c_int, _f64p, _f64p, C.c_int]
    L.orc_vcycle_apply.argtypes = [vp, _f64p, _f64p]
    L.orc_solve.restype = C.c_double
    L.orc_solve.argtypes = [vp, C.c_int, _f64p]
    L.orc_solve_cg.restype = C.c_int
    L.orc_solve_cg.argtypes = [vp, _f64p]
    L.orc_cg_history.restype = C.c_int
    L.orc_cg_history.argtypes = [vp, _f64p, C.c_int]
    L.orc_l2_error.restype = C.c_double
    L.orc_l2_error.argtypes = [vp, C.c_int]
    L.orc_time_vmult.restype = C.c_double
    L.orc_time_vmult.argtypes = [vp, C.c_int, C.c_int]
    L.orc_time_vcycle.restype = C.c_double
    L.orc_time_vcycle.argtypes = [vp, C.c_int]
    L.orc_num_threads.restype = C.c_int
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = None
    L.orc_create_from_mesh.restype = vp
    L.orc_create_from_mesh.argtypes = [C.c_int, C.c_int, C.POINTER(MeshLevel), C.c_int, C.c_int, C.c_int, C.c_int]
    _lib = L
    return L


class MeshLevel(C.Structure):
    """orc_mesh_level (mg_oracle.h)"""
    _fields_ = [("n_cells", C.c_uint32), ("n_dofs", C.c_uint32), ("n_constrained", C.c_uint32),
                ("idx27", C.POINTER(C.c_uint32)), ("idx27_plain", C.POINTER(C.c_uint32)),
                ("constrained", C.POINTER(C.c_uint32)), ("children", C.POINTER(C.c_uint32)),
                ("dof_gid", C.POINTER(C.c_uint32)), ("ent_mult", C.POINTER(C.c_uint8)),
                ("cell_nodes", C.POINTER(C.c_double))]


def _p(a):
    assert a.dtype == np.float64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(_f64p)


class Oracle:
    """MultigridSolver<3,p,Number,double> of the reference, restated on the CPU."""

    GEOMETRY = {"cartesian": 0, "sheared": 1, "shell_sector": 2}
    PROBLEM = {"cube": 0, "shell": 1}

    def __init__(self, p, n_subdiv=1, n_refine=3, degree=3, n_cycles=1, vfloat=False, box=None, geometry=None,
                 problem="cube", origin=-0.9, h0=None, polynomial="first_kind", mesh=None):
        """box=(sx,sy,sz): the doubling-mesh family (coarse cubes of size 1.9 from (-1,-1,-1));
        otherwise the square mesh [-0.9,1]^3 with n_subdiv coarse cells per direction"""
        self.L = lib()
        if mesh is not None:
            # a mesh described by tables (multi-block shell): `mesh` provides the per-level arrays of orc_mesh_level
            u32, u8, f64 = C.POINTER(C.c_uint32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
            keep, levels = [], (MeshLevel * mesh.n_levels)()
            for l in range(mesh.n_levels):
                arrs = dict(idx27=np.ascontiguousarray(mesh.idx27(l), np.uint32), idx27_plain=np.ascontiguousarray(mesh.idx27_plain(l), np.uint32),
                            constrained=np.ascontiguousarray(mesh.constrained(l), np.uint32),
                            children=np.ascontiguousarray(mesh.children(l), np.uint32) if l > 0 else np.zeros(1, np.uint32),
                            dof_gid=np.ascontiguousarray(mesh.dof_grid(l), np.uint32),
                            ent_mult=np.ascontiguousarray(mesh.entity_multiplicity(l), np.uint8),
                            cell_nodes=np.ascontiguousarray(mesh.cell_nodes(l), np.float64))
                keep.append(arrs)
                levels[l].n_cells, levels[l].n_dofs, levels[l].n_constrained = mesh.n_cells(l), mesh.n_dofs(l), mesh.n_constrained(l)
                for k, t in (("idx27", u32), ("idx27_plain", u32), ("constrained", u32), ("children", u32), ("dof_gid", u32),
                             ("ent_mult", u8), ("cell_nodes", f64)):
                    setattr(levels[l], k, arrs[k].ctypes.data_as(t))
            self.h = self.L.orc_create_from_mesh(p, mesh.n_levels, levels, degree, n_cycles, int(vfloat), self.PROBLEM[problem])
        elif geometry is not None:
            # mapped mesh / variable coefficient (poisson_shell): box of `box` (default n_subdiv^3) coarse cells
            b = box if box is not None else (n_subdiv,) * 3
            self.h = self.L.orc_create_mapped(p, b[0], b[1], b[2], origin, 1.9 / b[0] if h0 is None else h0, n_refine,
                                              degree, n_cycles, int(vfloat), self.GEOMETRY[geometry], self.PROBLEM[problem])
        elif box is not None:
            self.h = self.L.orc_create_box(p, box[0], box[1], box[2], n_refine, degree, n_cycles, int(vfloat))
        else:
            self.h = self.L.orc_create(p, n_subdiv, n_refine, degree, n_cycles, int(vfloat))
        if not self.h:
            raise ValueError("orc_create failed")
        self.p = p
        self.n_levels = self.L.orc_n_levels(self.h)
        self.max_level = self.n_levels - 1
        if polynomial != "first_kind":  # multigrid_solver.h:951-952 (Number == Number2 specialisation)
            assert polynomial == "fourth_kind"
            self.L.orc_set_polynomial_type(self.h, 1)

    def close(self):
        if self.h:
            self.L.orc_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # sizes / tables -------------------------------------------------------
    def n_dofs(self, l):
        return self.L.orc_n_dofs(self.h, l)

    def n_cells(self, l):
        return self.L.orc_n_cells(self.h, l)

    def n_constrained(self, l):
        return self.L.orc_n_constrained(self.h, l)

    def cells_per_dim(self, l):
        return self.L.orc_cells_per_dim(self.h, l)

    def cells_per_dim3(self, l):
        out = (C.c_int * 3)()
        self.L.orc_cells_per_dim3(self.h, l, C.byref(out))
        return tuple(out)

    def _u32(self, fn, l, n):
        return np.ctypeslib.as_array(getattr(self.L, fn)(self.h, l), shape=(n,)).copy()

    def idx27(self, l):
        return self._u32("orc_idx27", l, 27 * self.n_cells(l)).reshape(-1, 27)

    def idx27_plain(self, l):
        return self._u32("orc_idx27_plain", l, 27 * self.n_cells(l)).reshape(-1, 27)

    def constrained(self, l):
        return self._u32("orc_constrained", l, self.n_constrained(l))

    def cell_coords(self, l):
        return self._u32("orc_cell_coords", l, 3 * self.n_cells(l)).reshape(-1, 3)

    def dof_grid(self, l):
        return self._u32("orc_dof_grid", l, self.n_dofs(l))

    def _f64(self, fn, n, *args):
        return np.ctypeslib.as_array(getattr(self.L, fn)(self.h, *args), shape=(n,)).copy()

    def shape_values(self):
        n = self.p + 1
        return self._f64("orc_shape_values", n * n).reshape(n, n)

    def colloc_grad(self):
        n = self.p + 1
        return self._f64("orc_colloc_grad", n * n).reshape(n, n)

    def qweights(self):
        return self._f64("orc_qweights", self.p + 1)

    def qpoints(self):
        return self._f64("orc_qpoints", self.p + 1)

    def gll(self):
        return self._f64("orc_gll", self.p + 1)

    def prolong_1d(self):
        n = self.p + 1
        return self._f64("orc_prolong_1d", (2 * n - 1) * n).reshape(2 * n - 1, n)

    def rhs(self, l):
        return self._f64("orc_rhs", self.n_dofs(l), l)

    def inv_diag(self, l):
        return self._f64("orc_inv_diag", self.n_dofs(l), l)

    def solution(self, l):
        return self._f64("orc_solution", self.n_dofs(l), l)

    def cell_size(self, l):
        return self.L.orc_h(self.h, l)

    def reset_smoother(self, l, smoothing_range, degree, eig_cg_n_iterations):
        """multigrid_solver_dg.h:271-291 configures the FE_Q hierarchy under a DG level differently"""
        self.L.orc_reset_smoother(self.h, l, smoothing_range, degree, eig_cg_n_iterations)

    def cheb_info(self, l):
        v = [C.c_double() for _ in range(4)]
        i = [C.c_int() for _ in range(2)]
        self.L.orc_cheb_info(self.h, l, *[C.byref(x) for x in v], *[C.byref(x) for x in i])
        return dict(lambda_min=v[0].value, lambda_max=v[1].value, theta=v[2].value,
                    delta=v[3].value, degree=i[0].value, cg_its=i[1].value)

    def bc(self, l):
        n = self.L.orc_bc(self.h, l, None, None)
        idx = np.zeros(n, np.uint32)
        val = np.zeros(n, np.float64)
        self.L.orc_bc(self.h, l, idx.ctypes.data_as(_u32p), _p(val))
        return idx, val

    # operator ---------------------------------------------------------------
    def vmult(self, l, src):
        dst = np.empty_like(src)
        self.L.orc_vmult(self.h, l, _p(dst), _p(src))
        return dst

    def vmult_residual(self, l, rhs, lhs):
        res = np.empty_like(rhs)
        self.L.orc_vmult_residual(self.h, l, _p(rhs), _p(lhs), _p(res))
        return res

    def vmult_dense_lex(self, l, src_lex):
        dst = np.empty_like(src_lex)
        self.L.orc_vmult_dense_lex(self.h, l, _p(dst), _p(src_lex))
        return dst

    def cheb_vmult(self, l, b):
        x = np.zeros_like(b)
        self.L.orc_cheb_vmult(self.h, l, _p(x), _p(b))
        return x

    def cheb_step(self, l, x, b):
        x = x.copy()
        self.L.orc_cheb_step(self.h, l, _p(x), _p(b))
        return x

    def prolongate(self, l, coarse, fine=None, with_bc=False):
        add = fine is not None
        out = fine.copy() if add else np.zeros(self.n_dofs(l))
        self.L.orc_prolongate(self.h, l, _p(out), _p(coarse), int(add), int(with_bc))
        return out

    def restrict_and_add(self, l, coarse, fine, with_bc=False):
        out = coarse.copy()
        self.L.orc_restrict_and_add(self.h, l, _p(out), _p(fine), int(with_bc))
        return out

    def vcycle(self, src):
        dst = np.empty_like(src)
        self.L.orc_vcycle_apply(self.h, _p(dst), _p(src))
        return dst

    def vmult_with_cg_update(self, l, alpha, beta, r, q, p, x):
        """in place on q, p, x (contiguous float64 arrays); returns the four sums"""
        sums = np.zeros(4)
        self.L.orc_vmult_with_cg_update(self.h, l, alpha, beta, _p(r), _p(q), _p(p), _p(x), _p(sums))
        return sums

    def vmult_with_residual_update(self, residual, update, factor):
        """in place on residual, update; returns {z.res, z.(factor upd), res.res}"""
        out = np.zeros(3)
        self.L.orc_vmult_with_residual_update(self.h, _p(residual), _p(update), factor, _p(out))
        return out

    def set_affine_coef(self, l, coef6):
        self.L.orc_set_affine_coef(self.h, l, _p(np.ascontiguousarray(coef6, dtype=np.float64)))

    def coef_q(self, l):
        """[n_cells, (p+1)^3, 6] merged coefficient of the general branch (None on the affine branch)"""
        ptr = self.L.orc_coef_q(self.h, l)
        if not ptr:
            return None
        n3 = (self.p + 1) ** 3
        return np.ctypeslib.as_array(ptr, shape=(self.n_cells(l), n3, 6)).copy()

    def solve(self, analyze=False):
        trace = np.zeros(4 * self.n_levels)
        rate = self.L.orc_solve(self.h, int(analyze), _p(trace))
        return rate, trace.reshape(-1, 4)

    def solve_cg(self):
        red = C.c_double()
        its = self.L.orc_solve_cg(self.h, C.byref(red))
        return its, red.value

    def cg_history(self):
        """residual norms of the last solve_cg: [0] at the start, [k] after iteration k"""
        out = np.zeros(1001)
        n = self.L.orc_cg_history(self.h, _p(out), out.size)
        return out[:n].copy()

    def l2_error(self, l=None):
        return self.L.orc_l2_error(self.h, self.max_level if l is None else l)

    def time_vmult(self, l, n):
        return self.L.orc_time_vmult(self.h, l, n)

    def time_vcycle(self, n):
        return self.L.orc_time_vcycle(self.h, n)

    def num_threads(self):
        return self.L.orc_num_threads()

    def set_num_threads(self, n):
        """OpenMP threads of the following calls (n <= 0: the default)"""
        self.L.orc_set_num_threads(int(n))
