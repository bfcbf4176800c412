"""ctypes declarations for libmgx.so (include/mgx.h, include/mgx_cube.h, include/mgx_dg.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C multigrid_amd/csrc`.
There is no CPU fallback: a missing library raises, and so does a missing HIP device at
context creation (MGX_ERR_NO_DEVICE).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# MGX_LIB_PATH: A/B timing of two builds of the same library (tools/); never a different backend
LIB_PATH = os.environ.get("MGX_LIB_PATH") or os.path.join(_HERE, "libmgx.so")

F32, F64 = 0, 1
INVALID_INDEX = 0xFFFFFFFF

vp = C.c_void_p
u32p = C.POINTER(C.c_uint32)
f64p = C.POINTER(C.c_double)


class ExchangeDesc(C.Structure):
    _fields_ = [("plan_id", C.c_int), ("n_neighbors", C.c_int), ("neighbor_rank", C.POINTER(C.c_int)),
                ("count", u32p), ("index", C.POINTER(u32p)), ("shared", u32p), ("n_shared", C.c_uint32),
                ("not_owned", u32p), ("n_not_owned", C.c_uint32), ("send_buf", C.POINTER(vp)),
                ("recv_buf", C.POINTER(vp))]


EXCHANGE_FN = C.CFUNCTYPE(C.c_int, vp, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), u32p, C.POINTER(vp), C.POINTER(vp))
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, vp, f64p, C.c_int)
ALLOC_FN = C.CFUNCTYPE(vp, vp, C.c_size_t)
LEVEL_HOOK_FN = C.CFUNCTYPE(None, vp, C.c_int, C.c_int)


class CommDesc(C.Structure):
    _fields_ = [("rank", C.c_int), ("size", C.c_int), ("user", vp), ("exchange", EXCHANGE_FN),
                ("allreduce_sum", ALLREDUCE_FN), ("alloc_device", ALLOC_FN)]


class CubeBoxDesc(C.Structure):
    _fields_ = [("degree", C.c_int), ("n_refine", C.c_int), ("roots", C.c_int * 3), ("origin", C.c_double),
                ("h0", C.c_double), ("procs", C.c_int * 3), ("rank", C.c_int), ("numbering", C.c_int),
                ("geometry", C.c_int), ("problem", C.c_int)]


class OperatorDesc(C.Structure):
    _fields_ = [("degree", C.c_int), ("number", C.c_int), ("n_cells", C.c_uint32), ("n_dofs", C.c_uint32),
                ("idx27", u32p), ("idx27_plain", u32p), ("constrained", u32p), ("n_constrained", C.c_uint32),
                ("coef", C.c_double * 6), ("shape_values", f64p), ("colloc_grad", f64p), ("qweights", f64p),
                ("brick_colour", C.POINTER(C.c_uint8)), ("global_index", u32p),
                ("exchange", C.POINTER(ExchangeDesc)), ("coef_q", f64p)]


class DGExchangeDesc(C.Structure):
    _fields_ = [("plan_id", C.c_int), ("n_neighbors", C.c_int), ("neighbor_rank", C.POINTER(C.c_int)),
                ("count", u32p), ("send_cells", C.POINTER(u32p)), ("recv_first", u32p)]


class DGOperatorDesc(C.Structure):
    _fields_ = [("degree", C.c_int), ("basis", C.c_int), ("number", C.c_int), ("n_cells", C.c_uint32),
                ("neighbours", C.POINTER(C.c_int32)), ("jacobian", C.c_double * 9),
                ("n_ghost_cells", C.c_uint32), ("exchange", C.POINTER(DGExchangeDesc))]


class DGSolverDesc(C.Structure):
    _fields_ = [("matrix_dg", vp), ("matrix_dg_dp", vp), ("cfe", vp), ("degree_pre", C.c_int),
                ("cell_global_id", u32p)]


class SmootherInfo(C.Structure):
    _fields_ = [("lambda_min", C.c_double), ("lambda_max", C.c_double), ("theta", C.c_double),
                ("delta", C.c_double), ("degree", C.c_int), ("cg_iterations", C.c_int)]


class TransferDesc(C.Structure):
    _fields_ = [("children", u32p), ("prolong_1d", f64p), ("weight_shift", C.POINTER(C.c_uint8))]


class SolverDesc(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("degree_pre", C.c_int), ("n_cycles", C.c_int),
                ("matrix", C.POINTER(vp)), ("matrix_dp", C.POINTER(vp)),
                ("transfer", C.POINTER(vp)), ("transfer_dp", C.POINTER(vp)),
                ("rhs", C.POINTER(f64p)), ("bc_index", C.POINTER(u32p)), ("bc_value", C.POINTER(f64p)),
                ("bc_count", u32p)]


class CubeSolver(C.Structure):
    _fields_ = [("n_levels", C.c_int), ("matrix", C.POINTER(vp)), ("matrix_dp", C.POINTER(vp)),
                ("transfer", C.POINTER(vp)), ("transfer_dp", C.POINTER(vp)), ("solver", vp)]


# every symbol include/mgx.h and include/mgx_cube.h declare: name -> (restype, argtypes)
SIGNATURES = {
    "mgx_last_error": (C.c_char_p, []),
    "mgx_version": (C.c_char_p, []),
    "mgx_context_create": (C.c_int, [C.POINTER(vp), C.c_int]),
    "mgx_context_set_option": (C.c_int, [vp, C.c_char_p, C.c_double]),
    "mgx_has_cells_form": (C.c_int, []),
    "mgx_context_destroy": (C.c_int, [vp]),
    "mgx_sync": (C.c_int, [vp]),
    "mgx_device_memory_info": (C.c_int, [vp, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t)]),
    "mgx_context_stream": (vp, [vp]),
    "mgx_context_set_comm": (C.c_int, [vp, C.POINTER(CommDesc)]),
    "mgx_rccl_unique_id": (C.c_int, [vp]),
    "mgx_context_set_rccl": (C.c_int, [vp, C.c_int, C.c_int, vp]),
    "mgx_context_use_rccl": (C.c_int, [vp, C.c_int]),
    "mgx_copy_device": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "mgx_operator_exchange_buffers": (C.c_int, [vp, C.c_int, C.POINTER(vp), C.POINTER(vp), u32p, C.POINTER(C.c_int)]),
    "mgx_exchange_add": (C.c_int, [vp, vp]),
    "mgx_profile_enable": (C.c_int, [vp, C.c_int]),
    "mgx_range_push": (C.c_int, [vp, C.c_char_p]),
    "mgx_range_pop": (C.c_int, [vp]),
    "mgx_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_uint64), f64p]),
    "mgx_operator_set_profiled": (C.c_int, [vp, C.c_int]),
    "mgx_malloc": (C.c_int, [vp, C.POINTER(vp), C.c_size_t]),
    "mgx_free": (C.c_int, [vp, vp]),
    "mgx_upload": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "mgx_download": (C.c_int, [vp, vp, vp, C.c_size_t]),
    "mgx_memset_zero": (C.c_int, [vp, vp, C.c_size_t]),
    "mgx_copy_cast": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_size_t]),
    "mgx_add_cast": (C.c_int, [vp, vp, C.c_int, vp, C.c_int, C.c_size_t]),
    "mgx_sadd": (C.c_int, [vp, C.c_int, vp, C.c_double, C.c_double, vp, C.c_size_t]),
    "mgx_dot": (C.c_int, [vp, C.c_int, vp, vp, C.c_size_t, f64p]),
    "mgx_l2_norm": (C.c_int, [vp, C.c_int, vp, C.c_size_t, f64p]),
    "mgx_operator_dot": (C.c_int, [vp, vp, vp, f64p]),
    "mgx_operator_l2_norm": (C.c_int, [vp, vp, f64p]),
    "mgx_set_entries": (C.c_int, [vp, C.c_int, vp, u32p, f64p, C.c_uint32]),
    "mgx_operator_create": (C.c_int, [vp, C.POINTER(OperatorDesc), C.POINTER(vp)]),
    "mgx_operator_destroy": (C.c_int, [vp]),
    "mgx_operator_n_dofs": (C.c_uint32, [vp]),
    "mgx_operator_number": (C.c_int, [vp]),
    "mgx_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_vmult_residual": (C.c_int, [vp, vp, vp, vp]),
    "mgx_compute_diagonal": (C.c_int, [vp]),
    "mgx_get_inverse_diagonal": (C.c_int, [vp, C.POINTER(vp)]),
    "mgx_smoother_create": (C.c_int, [vp, C.c_double, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_smoother_destroy": (C.c_int, [vp]),
    "mgx_smoother_get_info": (C.c_int, [vp, C.POINTER(SmootherInfo)]),
    "mgx_smoother_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_smoother_step": (C.c_int, [vp, vp, vp]),
    "mgx_transfer_create": (C.c_int, [vp, vp, C.POINTER(TransferDesc), C.POINTER(vp)]),
    "mgx_transfer_destroy": (C.c_int, [vp]),
    "mgx_prolongate": (C.c_int, [vp, vp, vp, C.c_int, C.c_int]),
    "mgx_restrict_and_add": (C.c_int, [vp, vp, vp, C.c_int]),
    "mgx_solver_create": (C.c_int, [vp, C.POINTER(SolverDesc), C.POINTER(vp)]),
    "mgx_solver_destroy": (C.c_int, [vp]),
    "mgx_solver_solve": (C.c_int, [vp, C.c_int, f64p, f64p]),
    "mgx_solver_solve_hooked": (C.c_int, [vp, C.c_int, f64p, f64p, LEVEL_HOOK_FN, vp]),
    "mgx_solver_solve_cg": (C.c_int, [vp, C.POINTER(C.c_uint), f64p]),
    "mgx_solver_cg_history": (C.c_int, [vp, f64p, C.c_int, C.POINTER(C.c_int)]),
    "mgx_solver_solve_cg_fused": (C.c_int, [vp, C.POINTER(C.c_uint), f64p]),
    "mgx_solver_vmult_with_residual_update": (C.c_int, [vp, vp, vp, C.c_double, f64p]),
    "mgx_vmult_with_cg_update": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, vp, vp, vp, f64p]),
    "mgx_solver_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_solver_do_matvec": (C.c_int, [vp]),
    "mgx_solver_do_matvec_smoother": (C.c_int, [vp]),
    "mgx_solver_get_solution": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_solver_get_vector": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_solver_get_smoother": (C.c_int, [vp, C.c_int, C.POINTER(vp)]),
    "mgx_solver_get_timings": (C.c_int, [vp, f64p]),
    "mgx_solver_enable_timings": (C.c_int, [vp, C.c_int]),
    # mgx_cube.h
    "mgx_cube_create": (C.c_int, [C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_cube_create_numbered": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_cube_create_box": (C.c_int, [C.POINTER(CubeBoxDesc), C.POINTER(vp)]),
    "mgx_cube_create_shell": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_cube_create_shell_ranks": (C.c_int, [C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_cube_level_offset": (C.c_int, [vp]),
    "mgx_cube_cell_nodes": (C.c_int, [vp, C.c_int, f64p]),
    "mgx_cube_entity_multiplicity": (C.POINTER(C.c_uint8), [vp, C.c_int]),
    "mgx_cube_rank": (C.c_int, [vp]),
    "mgx_cube_size": (C.c_int, [vp]),
    "mgx_cube_cells_per_dim3": (None, [vp, C.c_int, C.POINTER(C.c_uint32 * 3), C.POINTER(C.c_uint32 * 3)]),
    "mgx_cube_n_neighbors": (C.c_int, [vp, C.c_int]),
    "mgx_cube_neighbor_rank": (C.c_int, [vp, C.c_int, C.c_int]),
    "mgx_cube_neighbor_count": (C.c_uint32, [vp, C.c_int, C.c_int]),
    "mgx_cube_neighbor_index": (u32p, [vp, C.c_int, C.c_int]),
    "mgx_cube_n_shared": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_shared": (u32p, [vp, C.c_int]),
    "mgx_cube_n_not_owned": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_not_owned": (u32p, [vp, C.c_int]),
    "mgx_cube_weight_shift": (C.POINTER(C.c_uint8), [vp, C.c_int]),
    "mgx_cube_exchange_desc": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(ExchangeDesc), C.POINTER(u32p),
                                        C.POINTER(C.c_int), u32p]),
    "mgx_cube_l2_error_parts": (None, [vp, C.c_int, f64p, f64p, f64p]),
    "mgx_cube_destroy": (C.c_int, [vp]),
    "mgx_cube_n_levels": (C.c_int, [vp]),
    "mgx_cube_degree": (C.c_int, [vp]),
    "mgx_cube_n_cells": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_n_dofs": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_n_constrained": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_cells_per_dim": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_cell_size": (C.c_double, [vp, C.c_int]),
    "mgx_cube_idx27": (u32p, [vp, C.c_int]),
    "mgx_cube_idx27_plain": (u32p, [vp, C.c_int]),
    "mgx_cube_constrained": (u32p, [vp, C.c_int]),
    "mgx_cube_children": (u32p, [vp, C.c_int]),
    "mgx_cube_cell_coords": (u32p, [vp, C.c_int]),
    "mgx_cube_dof_grid": (u32p, [vp, C.c_int]),
    "mgx_cube_shape_values": (f64p, [vp]),
    "mgx_cube_colloc_grad": (f64p, [vp]),
    "mgx_cube_qweights": (f64p, [vp]),
    "mgx_cube_qpoints": (f64p, [vp]),
    "mgx_cube_gll": (f64p, [vp]),
    "mgx_cube_prolong_1d": (f64p, [vp]),
    "mgx_cube_rhs": (f64p, [vp, C.c_int]),
    "mgx_cube_coef_q": (f64p, [vp, C.c_int]),
    "mgx_cube_bc_count": (C.c_uint32, [vp, C.c_int]),
    "mgx_cube_bc_index": (u32p, [vp, C.c_int]),
    "mgx_cube_bc_value": (f64p, [vp, C.c_int]),
    "mgx_cube_operator_desc": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(OperatorDesc)]),
    "mgx_cube_l2_error": (C.c_double, [vp, C.c_int, f64p]),
    "mgx_cube_seeded_vector": (C.c_int, [vp, C.c_int, C.c_uint64, f64p]),
    "mgx_cube_solver_create": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.POINTER(CubeSolver)]),
    "mgx_cube_solver_create_opt": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(CubeSolver)]),
    "mgx_cube_rhs_quadrature": (C.c_int, [vp, C.c_int, f64p]),
    "mgx_compute_residual": (C.c_int, [vp, vp, vp, vp]),
    "mgx_solver_compute_rhs": (C.c_int, [vp, C.c_int, vp]),
    "mgx_cube_solver_destroy": (C.c_int, [C.POINTER(CubeSolver)]),
    "mgx_smoother_set_polynomial_type": (C.c_int, [vp, C.c_int]),
    "mgx_solver_set_polynomial_type": (C.c_int, [vp, C.c_int]),
    "mgx_solver_v_cycle": (C.c_int, [vp]),
    "mgx_solver_reset_smoother": (C.c_int, [vp, C.c_int, C.c_double, C.c_int, C.c_int]),
    "mgx_solver_n_levels": (C.c_int, [vp]),
    "mgx_solver_get_operator": (C.c_int, [vp, C.c_int, C.c_int, C.POINTER(vp)]),
    "mgx_operator_device_indices": (C.c_int, [vp, C.POINTER(u32p), u32p, u32p, C.POINTER(C.c_int)]),
    "mgx_solver_set_agglomeration": (C.c_int, [vp, C.c_int, vp, u32p, C.POINTER(C.c_uint8), C.c_uint32]),
    # include/mgx_dg.h
    "mgx_dg_operator_create": (C.c_int, [vp, C.POINTER(DGOperatorDesc), C.POINTER(vp)]),
    "mgx_dg_operator_destroy": (C.c_int, [vp]),
    "mgx_dg_operator_n_dofs": (C.c_uint64, [vp]),
    "mgx_dg_operator_vector_size": (C.c_uint64, [vp]),
    "mgx_dg_update_ghost_values": (C.c_int, [vp, vp]),
    "mgx_dg_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_dg_vmult_residual": (C.c_int, [vp, vp, vp, vp]),
    "mgx_dg_jacobi_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_dg_vmult_with_chebyshev_update": (C.c_int, [vp, vp, C.c_uint, C.c_double, C.c_double, vp, vp]),
    "mgx_dg_vmult_with_cg_update": (C.c_int, [vp, C.c_double, C.c_double, vp, vp, vp, vp, f64p]),
    "mgx_dg_operator_info": (C.c_int, [vp, f64p, f64p, f64p]),
    "mgx_dg_operator_basis": (C.c_int, [vp, f64p, f64p, f64p]),
    "mgx_dg_solver_create": (C.c_int, [vp, C.POINTER(DGSolverDesc), C.POINTER(vp)]),
    "mgx_dg_solver_destroy": (C.c_int, [vp]),
    "mgx_dg_solver_smoother_info": (C.c_int, [vp, C.POINTER(SmootherInfo)]),
    "mgx_dg_solver_vmult": (C.c_int, [vp, vp, vp]),
    "mgx_dg_solver_solve_cg": (C.c_int, [vp, C.c_double, vp, vp, C.POINTER(C.c_uint), f64p]),
    "mgx_dg_restrict_to_cg": (C.c_int, [vp, vp, vp]),
    "mgx_dg_prolongate_add_cg_to_dg": (C.c_int, [vp, vp, vp]),
    "mgx_dg_vmult_residual_and_restrict_to_cg": (C.c_int, [vp, vp, vp, vp]),
    "mgx_dg_cheby_mesh": (C.c_int, [C.c_int, C.POINTER(C.c_int * 3), C.POINTER(C.c_double * 9)]),
    "mgx_dg_box_neighbours": (C.c_int, [C.POINTER(C.c_int * 3), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
}

_lib = None


class MgxError(RuntimeError):
    def __init__(self, status, message):
        super().__init__("mgx error %d: %s" % (status, message))
        self.status = status


def load():
    """Load libmgx.so; raises if it has not been built (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("multigrid_amd: %s is missing -- build it with "
                          "`python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950); there is no CPU fallback" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status != 0:
        raise MgxError(status, load().mgx_last_error().decode())
