/*
 * mgx_dg.h -- C ABI of the MI355X-native DG (symmetric interior penalty) Laplace operator with the
 * merged Chebyshev update (libmgx.so); BASELINE config 5, first slice: one GPU, affine mesh.
 *
 * Reference interface replaced (paths relative to the reference root):
 *     multigrid::LaplaceOperatorCompactCombine<3,p,Number,type>  common/laplace_operator_dg.h:350-2024
 *     multigrid::JacobiTransformed<3,p,Number,type>              common/laplace_operator_dg.h:2028-2256
 * as driven by matvec_dg_cheby/program.cc:40-200.  Conventions as in mgx.h (status codes, device
 * pointers, the context's stream, no CPU fallback).
 *
 * Vector layout: the reference's DG numbering -- (p+1)^3 contiguous coefficients per cell, x
 * fastest (laplace_operator_dg.h:1141-1146 reads a cell as one contiguous block) -- cells in the
 * order of the neighbour table handed to mgx_dg_operator_create.
 */
#ifndef MGX_DG_H
#define MGX_DG_H

#include "mgx.h"

#ifdef __cplusplus
extern "C" {
#endif

/* `type` template parameter of LaplaceOperatorCompactCombine (laplace_operator_dg.h:343-348) */
#define MGX_DG_HERMITE 0       /* FE_DGQHermite: neighbour access of two node layers only */
#define MGX_DG_GAUSS_LOBATTO 1 /* FE_DGQ */
#define MGX_DG_GAUSS 2         /* FE_DGQArbitraryNodes(QGauss): collocation, no basis change */

#define MGX_DG_BOUNDARY (-1)   /* neighbour entry of a (homogeneous Dirichlet) boundary face */

typedef struct mgx_dg_operator_s *mgx_dg_operator_t;

/* Ghost-cell exchange of a decomposed DG mesh (replaces the MPI_Isend / MPI_Irecv of face data,
 * laplace_operator_dg.h:986-1057, and the send lists built at :607-723).  A rank stores its owned
 * cells first and, behind them, one ghost for every cell of another rank that shares a face with an
 * owned cell; neighbour-table entries >= n_cells address them.  A ghost holds the reference's
 * data_per_face (:565): the whole cell for the two Lagrange bases; for the Hermite-like basis two
 * values per face point, the trace and the normal derivative as the owner computes them from its two
 * node layers (:1015-1039), [face point][2], 2 (p+1)^2 entries.  Ghosts of one rank are contiguous.  Both sides of a pair exchange the same number of cells (true for block
 * decompositions; checked).  An operator application runs the cells without a ghost neighbour while
 * the exchange is in flight on a second stream, the others behind it (the reference waits for the
 * exchange, :1057, before its cell loop); owned cells ordered interior-first make both parts
 * contiguous launches. */
typedef struct
{
  int                    plan_id;
  int                    n_neighbors;
  const int             *neighbor_rank; /* ascending */
  const uint32_t        *count;         /* cells sent to = received from neighbour k */
  const uint32_t *const *send_cells;    /* owned cells to send, in the order the neighbour stores them */
  const uint32_t        *recv_first;    /* first ghost cell (>= n_cells) filled by neighbour k */
} mgx_dg_exchange_desc;

typedef struct
{
  int      degree;             /* 1 .. MGX_MAX_DEGREE */
  int      basis;              /* MGX_DG_* */
  int      number;             /* MGX_F32 (matvec_dg_cheby/program.cc:88) or MGX_F64 */
  uint32_t n_cells;
  /* [n_cells][6] cell behind face 2d+s (direction d, lower s = 0 / upper s = 1 side) or
   * MGX_DG_BOUNDARY; replaces start_indices_on_neighbor / dirichlet_faces
   * (laplace_operator_dg.h:453-459, 480-560).  Host memory, copied. */
  const int32_t *neighbours;
  /* the one cell Jacobian dx/dxi, row-major [real][reference]; the reference asserts a single
   * Jacobian for the whole mesh as well (laplace_operator_dg.h:749-750) */
  double jacobian[9];
  /* decomposed mesh (one rank per GPU; the context carries the communicator, mgx.h): number of
   * ghost cells and their exchange; 0 / NULL on a single rank.  Vectors then hold
   * mgx_dg_operator_vector_size() entries, owned cells first, like deal.II's owned | ghost layout */
  uint32_t                    n_ghost_cells;
  const mgx_dg_exchange_desc *exchange;
} mgx_dg_operator_desc;

/* LaplaceOperatorCompactCombine::reinit (:361-771) + JacobiTransformed::JacobiTransformed
 * (:2031-2038, local_compute_diagonals :2099-2233): 1D basis data, penalty and normal factors,
 * eigenvector basis and the inverse transformed diagonals (one set per combination of Dirichlet
 * faces -- at most 64 -- instead of one per cell). */
int mgx_dg_operator_create(mgx_context_t ctx, const mgx_dg_operator_desc *desc, mgx_dg_operator_t *op);
int mgx_dg_operator_destroy(mgx_dg_operator_t op);

/* LaplaceOperatorCompactCombine::m (:796-800) */
uint64_t mgx_dg_operator_n_dofs(mgx_dg_operator_t op);

/* entries of a vector including the ghost cells (== mgx_dg_operator_n_dofs on a single rank) */
uint64_t mgx_dg_operator_vector_size(mgx_dg_operator_t op);
/* Vector::update_ghost_values (the import of laplace_operator_dg.h:986-1057): fills the ghost cells
 * of vec from their owners.  Every operator application below does this for its source vector
 * (whose ghost part is therefore written although the argument is const). */
int mgx_dg_update_ghost_values(mgx_dg_operator_t op, void *vec);

/* LaplaceOperatorCompactCombine::vmult (:802-806, action 0).  dst must not alias src. */
int mgx_dg_vmult(mgx_dg_operator_t op, void *dst, const void *src);

/* vmult_with_merged_ops<4> (:962-966, epilogue :1812-1818): dst = rhs - A src */
int mgx_dg_vmult_residual(mgx_dg_operator_t op, void *dst, const void *rhs, const void *src);

/* JacobiTransformed::vmult (:2046-2084): dst = T D^-1 T^T src per cell.  dst may alias src. */
int mgx_dg_jacobi_vmult(mgx_dg_operator_t op, void *dst, const void *src);

/* LaplaceOperatorCompactCombine::vmult_with_chebyshev_update (:910-955, epilogue :1839-1860).
 *   iteration_index == 0:  solution = factor2 P^-1 rhs
 *   iteration_index == 1:  new = factor2 P^-1 (rhs - A solution) + (1 + factor1) solution
 *   iteration_index >= 2:  new = ... - factor1 solution_old
 * For iteration_index >= 1 `new` is written over solution_old; the reference then swaps the two
 * vectors (:931) -- with raw pointers that swap is the caller's (the Python binding and
 * tools/matvec_dg_cheby.py do it).  solution and solution_old must not alias. */
int mgx_dg_vmult_with_chebyshev_update(mgx_dg_operator_t op, const void *rhs, unsigned iteration_index,
                                       double factor1, double factor2, void *solution, void *solution_old);

/* LaplaceOperatorCompactCombine::vmult_with_cg_update (:863-908; action 2, epilogue :1827-1838): one iteration of
 * the merged conjugate-gradient loop around the operator,
 *   alpha != 0:  x += alpha p ;  p = beta p + q      alpha == 0:  p = q
 *   q = A p ;  sums = { q.p, r.r, q.r, q.q }  summed over all ranks (host array)
 * The four sums come out of the cell kernel's epilogue (block sums added in a fixed order). */
int mgx_dg_vmult_with_cg_update(mgx_dg_operator_t op, double alpha, double beta, const void *r, void *q, void *p, void *x,
                                double sums[4]);

/* 1D data of the operator for inspection: hermite_derivative_on_face (:408-409), the penalty
 * parameters get_penalty(face 2d) (:789-793) and the 1D generalised eigenvalues (:207-208).
 * Any pointer may be NULL. */
int mgx_dg_operator_info(mgx_dg_operator_t op, double *hermite_derivative_on_face, double penalty[3],
                         double eigenvalues_1d[MGX_MAX_DEGREE + 1]);

/* 1D element data a harness needs for right-hand sides and error norms (what FEEvaluation hands the
 * reference's drivers): shape_values[q*(p+1)+i] = phi_i at Gauss point q of [0,1], the points and
 * weights of the (p+1)-point Gauss rule.  Any pointer may be NULL. */
int mgx_dg_operator_basis(mgx_dg_operator_t op, double *shape_values, double *quadrature_points,
                          double *quadrature_weights);

/* ---- MultigridSolverDG<3,p,Number,double> (common/multigrid_solver_dg.h:55-747): the DG level on top
 * of the FE_Q(p) multigrid hierarchy of the same mesh ---- */
typedef struct mgx_dg_solver_s *mgx_dg_solver_t;
typedef struct
{
  mgx_dg_operator_t matrix_dg;    /* V-cycle number type (multigrid_solver_dg.h:700) */
  mgx_dg_operator_t matrix_dg_dp; /* its fp64 twin for the outer iteration (:703) */
  /* MultigridSolver of the FE_Q(p) hierarchy (include/mgx.h) in the V-cycle number type, built on
   * the same context; the cells of its finest level in the order of the DG operators' cells.  The
   * DG solver re-configures its smoothers as the reference does (:271-291: degree_pre - 1 on the
   * finest FE_Q level, coarse tolerance 2e-3) and runs its V-cycle in place. */
  mgx_solver_t      cfe;
  int               degree_pre;   /* Chebyshev degree of the DG smoother (:300) */
  /* optional, host [n_cells]: a decomposition-independent id of every owned cell (e.g. its
   * lexicographic position in the whole mesh); the start vector of the eigenvalue estimate is
   * ((id (p+1)^3 + local index) mod 11) - mean, deal.II's (global DoF index mod 11) - mean.
   * NULL: the cell's index on this rank.  Decomposed meshes: matrix_dg / matrix_dg_dp carry ghost
   * cells (all device vectors of the solver interface then have mgx_dg_operator_vector_size
   * entries), cfe is the decomposed FE_Q solver of the same partition, not agglomerated. */
  const uint32_t   *cell_global_id;
} mgx_dg_solver_desc;
/* ctor (:58-323), incl. smooth_dg.initialize: eigenvalue estimate with JacobiTransformed */
int mgx_dg_solver_create(mgx_context_t ctx, const mgx_dg_solver_desc *desc, mgx_dg_solver_t *solver);
int mgx_dg_solver_destroy(mgx_dg_solver_t solver);
int mgx_dg_solver_smoother_info(mgx_dg_solver_t solver, mgx_smoother_info *info);
/* MultigridSolverDG::vmult (:429-440): one DG V-cycle = smoother, residual restricted to FE_Q
 * (laplace_operator_dg.h:1798-1819), FE_Q V-cycle, correction embedded back (:1863-1894), smoother;
 * fp64 device vectors in the DG layout */
int mgx_dg_solver_vmult(mgx_dg_solver_t solver, double *dst, const double *src);
/* MultigridSolverDG::solve_cg(tolerance) (:410-424) on a given right-hand side: zero start, at
 * most 100 iterations; iterations = last_step, reduction_rate = (res/res0)^(1/its) */
int mgx_dg_solver_solve_cg(mgx_dg_solver_t solver, double tolerance, const double *rhs, double *solution,
                           unsigned *iterations, double *reduction_rate);
/* the two transfers of the DG level on their own (V-cycle number type): cg = P^T dg (cg is
 * zeroed first; constrained rows stay zero) and dg += P cg */
int mgx_dg_restrict_to_cg(mgx_dg_solver_t solver, void *cg_dst, const void *dg_src);
int mgx_dg_prolongate_add_cg_to_dg(mgx_dg_solver_t solver, void *dg_dst, const void *cg_src);
/* LaplaceOperatorCompactCombine::vmult_residual_and_restrict_to_cg (:852-861; action 1, epilogue :1798-1819) of the
 * solver's V-cycle operator: cg = P^T (rhs - A lhs), the residual changed to the FE_Q basis and added into the FE_Q
 * vector inside the cell kernel (cg zeroed first, summed over the rank interfaces of a decomposed mesh).  The V-cycle
 * runs this form unless the context option "dg_unmerged_restrict" is set. */
int mgx_dg_vmult_residual_and_restrict_to_cg(mgx_dg_solver_t solver, void *cg_dst, const void *rhs, const void *lhs);

/* ---- mesh helpers (stand in for GridGenerator + DoFHandler of the harness) ---- */

/* matvec_dg_cheby/program.cc:55-77: cells per direction and the cell Jacobian of the sheared box
 * after n_cell_steps refinement steps */
int mgx_dg_cheby_mesh(int n_cell_steps, int cells[3], double jacobian[9]);

/* neighbour table of a box of cells[0] x cells[1] x cells[2] cells, all outer faces Dirichlet.
 * ordering 0: lexicographic, x fastest; 1: z-order (space-filling curve, as p4est gives the
 * reference).  cell_ijk (optional) receives the (i, j, k) position of every cell. */
int mgx_dg_box_neighbours(const int cells[3], int ordering, int32_t *neighbours, int32_t *cell_ijk);

#ifdef __cplusplus
}
#endif
#endif
