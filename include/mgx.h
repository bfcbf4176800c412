/*
 * mgx.h -- C ABI of the MI355X-native matrix-free multigrid Laplace path (libmgx.so).
 *
 * The reference has no FFI seam; the de-facto interface of the path are the public members of two
 * C++ class templates:
 *     multigrid::LaplaceOperator<dim,p,number>         common/laplace_operator.h:56-164
 *     multigrid::MultigridSolver<dim,p,Number,Number2> common/multigrid_solver.h:96-782
 * plus the deal.II objects they drive (PreconditionChebyshev, MGTransferMatrixFree, SolverCG).
 * Every entry point below names the reference member (file:line, relative to the reference root)
 * it replaces.  include/multigrid_shim.hpp re-creates the two classes on top of this ABI.
 *
 * Conventions
 *   - plain C types only; every function returns an int status (MGX_OK == 0) and never throws;
 *     mgx_last_error() gives the message of the last failure on the calling thread.
 *   - "device pointer" arguments are HIP device allocations of the level's number type
 *     (float for MGX_F32, double for MGX_F64) holding n_dofs entries laid out exactly like the
 *     locally-owned range of deal.II's LinearAlgebra::distributed::Vector.
 *   - all work is enqueued on the context's HIP stream; results that are returned to the host
 *     (norms, iteration counts) synchronise that stream, everything else is asynchronous
 *     until mgx_sync().
 *   - one host thread per context (the reference is single-threaded per MPI rank,
 *     multigrid_solver.h:153,176).  Contexts do not order their streams against each other: a
 *     vector that one context has allocated, zeroed or written may be handed to an object of
 *     another context only after mgx_sync() on the first.
 *   - there is NO CPU fallback: without a HIP device mgx_context_create() fails.
 */
#ifndef MGX_H
#define MGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGX_OK 0
#define MGX_ERR_INVALID_ARGUMENT (-1)
#define MGX_ERR_NO_DEVICE (-2)
#define MGX_ERR_HIP (-3)
#define MGX_ERR_UNSUPPORTED (-4)
#define MGX_ERR_NOT_CONVERGED (-5)

#define MGX_F32 0
#define MGX_F64 1

#define MGX_INVALID_INDEX 0xFFFFFFFFu /* dealii::numbers::invalid_unsigned_int */
#define MGX_MAX_DEGREE 9              /* poisson_cube/program.cc:69 */

typedef struct mgx_context_s  *mgx_context_t;
typedef struct mgx_operator_s *mgx_operator_t;  /* LaplaceOperator<3,p,number> of one level */
typedef struct mgx_smoother_s *mgx_smoother_t;  /* PreconditionChebyshev<LaplaceOperator,...> */
typedef struct mgx_transfer_s *mgx_transfer_t;  /* one level pair of MGTransferMatrixFree */
typedef struct mgx_solver_s   *mgx_solver_t;    /* MultigridSolver<3,p,Number,double> */

const char *mgx_last_error(void);
const char *mgx_version(void);
/* 1 if the library was built with the round-1 cell-by-cell brick kernels (make crosscheck: the
 * cross-check of the macro-element kernel, context options "cells_form" / "brick_wide_max") */
int mgx_has_cells_form(void);

/* ---- context: device + stream (replaces MPI_InitFinalize / the implicit host execution
 * context, poisson_cube/program.cc:664) ---- */
int mgx_context_create(mgx_context_t *ctx, int device);
/* Options of the context, to be set before the first object is created on it.  The scheduling
 * thresholds (also read from the environment when the context is created: MGX_BRICK_MIN,
 * MGX_OVERLAP_MIN_BRICKS, MGX_RESTRICT_COLOUR_MIN, MGX_CELL_COLOUR_MIN, MGX_FREE_MAX_BRICKS,
 * MGX_FREE_ONE_MAX, MGX_GRAPH_MAX_DOFS; MGX_TRACE) under the names "brick_min", "overlap_min_bricks",
 * "restrict_colour_min", "cell_colour_min", "free_max_bricks", "free_one_max", "graph_max_dofs",
 * "trace"; and selectors of numerically equivalent code paths that exist for tests and A/B timings
 * and are never taken from the environment: "general_kernel", "no_bricks", ("cells_form",
 * "brick_wide_max": cross-check builds only,) "macro_wg_per_cu_x16", "no_diag_table", "no_fused_init", "no_fused_restrict",
 * "no_fused_prolong", "force_fused_transfers", "transfer_v1", "restrict_atomic", "exchange_unfused",
 * "no_graph", "dg_no_overlap", "dg_unmerged_restrict", "no_fused_decomposed", "no_fused_assembly", "no_fused_residual", "no_restrict_scratch",
 * "no_macro_v2", "no_general_bricks" (with the threshold "general_brick_min"), "fused_prolong_min_bricks", "roctx".  "rccl_selftest" lets a one-rank communicator name itself as its own
 * neighbour (emulation of a rank on one GPU; the sums it produces are wrong by construction).
 * Unknown name: MGX_ERR_INVALID_ARGUMENT. */
int mgx_context_set_option(mgx_context_t ctx, const char *name, double value);
int mgx_context_destroy(mgx_context_t ctx);
int mgx_sync(mgx_context_t ctx);
/* free and total bytes of the context's device (hipMemGetInfo) */
int mgx_device_memory_info(mgx_context_t ctx, size_t *free_bytes, size_t *total_bytes);
/* raw HIP stream (hipStream_t) the context enqueues on, for callers that time with HIP events */
void *mgx_context_stream(mgx_context_t ctx);

/* ---- domain decomposition (replaces the MPI layer of deal.II's LinearAlgebra::distributed::Vector:
 * ghost exchange inside every cell_loop, MPI_Allreduce in l2_norm / operator*).
 * Interface DoFs are duplicated on the ranks that share them and kept consistent; after a cell
 * loop the partial sums of the interface DoFs are exchanged and added (one exchange per operator
 * application).  The transport is a pair of callbacks so that the same library runs over RCCL
 * (torch.distributed "nccl", one process per GPU), over gloo (CPU tests) or over MPI (deal.II). */
typedef struct
{
  int   rank, size;
  void *user;
  /* Blocking exchange of one plan: on entry every send buffer is complete (device memory,
   * counts[k] entries of `number` type for neighbour ranks[k]) and the context's stream is idle;
   * on return recv[k] must hold what rank ranks[k] put into its send buffer for this rank.
   * Return 0 on success. */
  int (*exchange)(void *user, int plan_id, int number, int n_neighbors, const int *ranks, const uint32_t *counts,
                  void *const *send, void *const *recv);
  /* in-place sum over all ranks of `count` doubles on the host */
  int (*allreduce_sum)(void *user, double *values, int count);
  /* optional (may be NULL): allocator for the send/recv buffers, so that the transport can own
   * them (e.g. torch CUDA tensors that RCCL sends from / receives into without staging copies);
   * the memory must stay valid for the life of the communicator.  NULL: hipMalloc. */
  void *(*alloc_device)(void *user, size_t bytes);
} mgx_comm_desc;
int mgx_context_set_comm(mgx_context_t ctx, const mgx_comm_desc *comm);

/* Native transport: the exchange becomes one group of ncclSend/ncclRecv and the reductions
 * ncclAllReduce, issued by the library on the context's stream -- ordered with the pack/unpack
 * kernels by the stream, no host synchronisation per exchange (one process per GPU, RCCL over
 * xGMI).  librccl is bound at run time.  Rank 0 obtains an id and the caller distributes its
 * MGX_RCCL_ID_BYTES bytes to all ranks (MPI_Bcast, torch.distributed.broadcast, ...); then every
 * rank calls mgx_context_set_rccl (collective).  mgx_context_use_rccl switches between the native
 * transport and the callbacks of mgx_context_set_comm (if both are set), e.g. to cross-check one
 * against the other. */
#define MGX_RCCL_ID_BYTES 128
int mgx_rccl_unique_id(void *id_out);
int mgx_context_set_rccl(mgx_context_t ctx, int rank, int size, const void *id);
int mgx_context_use_rccl(mgx_context_t ctx, int enable);

/* exchange plan of one level vector layout (host lists, copied).  Only unconstrained DoFs are
 * exchanged: a Dirichlet DoF on a rank interface is an identity row on every rank that holds it and
 * must not appear in index[] / shared[] (mgx_operator_create rejects such a plan). */
typedef struct
{
  int                    plan_id;       /* caller's identifier, passed back to exchange() */
  int                    n_neighbors;
  const int             *neighbor_rank; /* ascending */
  const uint32_t        *count;         /* entries per neighbour */
  const uint32_t *const *index;         /* index[k][i]: local DoF of entry i for neighbour k; both sides
                                           enumerate an interface in the same (global) order */
  const uint32_t        *shared;        /* union of the lists */
  uint32_t               n_shared;
  const uint32_t        *not_owned;     /* shared DoFs owned by a lower rank (skipped in dot products) */
  uint32_t               n_not_owned;
  /* optional caller-owned device buffers (count[k] entries of the operator's number type each),
   * e.g. torch CUDA tensors handed to RCCL; NULL: the library allocates them */
  void *const           *send_buf;
  void *const           *recv_buf;
} mgx_exchange_desc;

/* ---- instrumentation (the reference brackets its phases with LIKWID markers,
 * poisson_cube/program.cc:281-296,347-354; here: HIP events on the context's stream around every
 * launch of the cell-loop kernel of the operators flagged with mgx_operator_set_profiled) ---- */
int mgx_profile_enable(mgx_context_t ctx, int enable);
/* Profiler ranges (roctx, bound at run time; no-ops unless the context option "roctx" is set): the driver's LIKWID
 * regions of the reference -- LIKWID_MARKER_START / STOP("fmg_solver" | "cg_solver" | "matvec" | "matvec_sp"),
 * poisson_cube/program.cc:282-295, 309-321, 348-354, 369-375.  With the option set the library itself brackets the
 * phases of every V-cycle level: vmult_cheby_<level> (laplace_operator.h:732-739), mg_mv_<level>, restrict_<level>,
 * prolongate_<level>, mg_vec_<level>, inhomBC_<level>, coarse_solver_0 (the columns of print_wall_times,
 * multigrid_solver.h:348-371).  rocprofv3 --marker-trace shows them. */
int mgx_range_push(mgx_context_t ctx, const char *name);
int mgx_range_pop(mgx_context_t ctx);
/* Synchronises and returns, for one form of the cell loop, the number of bracketed kernel launches
 * and their summed duration; resets that form's record.  form: 0 plain vmult (16 B/DoF
 * algorithmic: src read + dst write), 1 residual (24 B/DoF), 2 fused Chebyshev iteration
 * (40 B/DoF: x, x_old, b, D^-1 read, x_new written), 3 first Chebyshev step (32 B/DoF),
 * 4 Chebyshev iteration with zero x_old (32 B/DoF), 5 first iteration after a zero guess with
 * x_1 = D^-1 b / theta formed on the fly (24 B/DoF), 6 second such iteration, x_old = x_1 formed
 * on the fly (32 B/DoF), 7 V-cycle residual fused with the restriction to the next coarser level
 * (18 B/DoF).  One bracket spans the colour launches of
 * one application (8 on a structured mesh); `launches` counts the individual kernel launches. */
int mgx_profile_read(mgx_context_t ctx, int form, uint64_t *launches, double *total_ms);

/* ---- device vectors (LinearAlgebra::distributed::Vector<number> storage) ---- */
int mgx_malloc(mgx_context_t ctx, void **dptr, size_t bytes);
int mgx_free(mgx_context_t ctx, void *dptr);
int mgx_upload(mgx_context_t ctx, void *dptr, const void *hptr, size_t bytes);
int mgx_download(mgx_context_t ctx, void *hptr, const void *dptr, size_t bytes);
int mgx_memset_zero(mgx_context_t ctx, void *dptr, size_t bytes); /* Vector::operator=(0) */
int mgx_copy_device(mgx_context_t ctx, void *dst, const void *src, size_t bytes); /* device to device */

/* vector kernels used by the driver; `number` = MGX_F32 / MGX_F64 */
/* dst = src with precision cast: multigrid_solver.h:437,503,507 */
int mgx_copy_cast(mgx_context_t ctx, void *dst, int dst_number, const void *src, int src_number, size_t n);
/* dst += src with precision cast: internal::add_vector multigrid_solver.h:54-67 */
int mgx_add_cast(mgx_context_t ctx, void *dst, int dst_number, const void *src, int src_number, size_t n);
/* x = s*x + a*v : Vector::sadd, multigrid_solver.h:465 */
int mgx_sadd(mgx_context_t ctx, int number, void *x, double s, double a, const void *v, size_t n);
/* x . y and ||x||_2 (Vector::l2_norm multigrid_solver.h:263,444,466; operator* in SolverCG) */
int mgx_dot(mgx_context_t ctx, int number, const void *x, const void *y, size_t n, double *result);
int mgx_l2_norm(mgx_context_t ctx, int number, const void *x, size_t n, double *result);
/* The same reductions with the ownership of the entries stated by the operator the vectors belong to
 * (on a decomposed mesh mgx_dot / mgx_l2_norm find the operator through the vector length; two levels
 * with vectors of one length but different interfaces make that ambiguous and are refused there) */
int mgx_operator_dot(mgx_operator_t op, const void *x, const void *y, double *result);
int mgx_operator_l2_norm(mgx_operator_t op, const void *x, double *result);
/* v[idx[i]] = val[i] (host index/value lists): multigrid_solver.h:257-259,408-409,427-428 */
int mgx_set_entries(mgx_context_t ctx, int number, void *v, const uint32_t *idx_host,
                    const double *val_host, uint32_t count);

/* ---- LaplaceOperator ---- */
typedef struct
{
  int      degree;  /* fe_degree p, 1..9 */
  int      number;  /* MGX_F32 / MGX_F64: template parameter `number` */
  uint32_t n_cells; /* MatrixFree::n_cell_batches() * lanes, real cells only */
  uint32_t n_dofs;  /* locally owned size of the level vector */
  /* LaplaceOperator::get_compressed_dof_indices() (laplace_operator.h:114-118, built by
   * extract_compressed_indices :224-353) de-interleaved to one row of 27 per cell:
   * idx27[27*cell + 9*cz+3*cy+cx] = first DoF of that mesh entity, MGX_INVALID_INDEX where
   * the entity is constrained.  Host pointer, copied. */
  const uint32_t *idx27;
  /* the same table built without constraints (dof-handler slot 1 of matrix_dp,
   * multigrid_solver.h:180-190); needed by the transfers.  Host pointer, copied; may be NULL
   * if the operator is never used in a transfer. */
  const uint32_t *idx27_plain;
  /* MatrixFree::get_constrained_dofs() (laplace_operator.h:592,632,736) */
  const uint32_t *constrained;
  uint32_t        n_constrained;
  /* merged_coefficient of the affine / constant-coefficient branch, [xx,yy,zz,xy,xz,yz]
   * (laplace_operator.h:127, 374-387); the quadrature weight is applied per point (:456-457).
   * Off-diagonal entries (non-Cartesian affine cells, :473-486) select the quadrature-point
   * form of the cell loop. */
  double coef[6];
  /* 1D data of FEEvaluation<3,p,p+1>, row-major (p+1)x(p+1):
   * shape_values[q*(p+1)+i] = GLL-Lagrange basis i at Gauss point q;
   * colloc_grad[q*(p+1)+r] = derivative of the Gauss-point Lagrange basis r at Gauss point q;
   * qweights[q] = Gauss weights on [0,1]. */
  const double *shape_values;
  const double *colloc_grad;
  const double *qweights;
  /* Optional scheduling hint, may be NULL.  If consecutive cells form bricks in Morton order
   * -- 64 cells = 4x4x4 for p <= 4, 8 cells = 2x2x2 (the children of one parent) for p >= 5, which
   * is what p4est/deal.II produce on uniformly refined meshes -- the library runs its atomic-free
   * brick cell loop; brick_colour[k] (one entry per brick, < 32) is a colouring in which bricks
   * that share DoFs differ.  Without the hint a greedy colouring is computed; the brick structure
   * itself is always verified against idx27. */
  const uint8_t *brick_colour;
  /* Optional, may be NULL: a numbering-independent global index per local DoF (the provider's
   * lexicographic grid id).  Used for the start vector of the smoother's eigenvalue estimate
   * (deal.II: "global index mod 11"), so that results do not depend on cell order or on the
   * domain decomposition.  NULL: the local index. */
  const uint32_t *global_index;
  /* Optional, may be NULL: interface exchange plan of a decomposed mesh */
  const mgx_exchange_desc *exchange;
  /* Optional, may be NULL: merged_coefficient of the general branch -- variable coefficient and / or
   * non-affine geometry -- one symmetric tensor per cell and quadrature point with the weight
   * folded in (evaluate_coefficient laplace_operator.h:388-430, applied :493-522).  Layout
   * component-major per cell: coef_q[(cell*6 + c)*(p+1)^3 + q], c over [xx,yy,zz,xy,xz,yz], q
   * lexicographic with x fastest (the reference stores Tensor<1,6> per point; the shim
   * transposes).  Host pointer, copied (in the operator's number type).  When given, coef[] is
   * ignored. */
  const double *coef_q;
} mgx_operator_desc;

/* LaplaceOperator::initialize + evaluate_coefficient (laplace_operator.h:184-220, 357-432) */
int mgx_operator_create(mgx_context_t ctx, const mgx_operator_desc *desc, mgx_operator_t *op);
int mgx_operator_destroy(mgx_operator_t op);
uint32_t mgx_operator_n_dofs(mgx_operator_t op); /* Base::m() */
int mgx_operator_set_profiled(mgx_operator_t op, int profiled);
/* device buffers of neighbour k of the operator's exchange plan */
int mgx_operator_exchange_buffers(mgx_operator_t op, int k, void **send, void **recv, uint32_t *count, int *rank);
/* sums the duplicated interface entries of a level vector over the ranks that share them (in
 * ascending rank order on every rank => bitwise identical copies): Vector::compress(add) */
int mgx_exchange_add(mgx_operator_t op, void *vec);
int mgx_operator_number(mgx_operator_t op);
/* LaplaceOperator::vmult(dst, src) laplace_operator.h:573-601 */
int mgx_vmult(mgx_operator_t op, void *dst, const void *src);
/* LaplaceOperator::compute_residual (laplace_operator.h:804-845) on the device:
 *   dst_i = sum over the cells of  integral( phi_i f )  -  integral( grad phi_i . K grad u )
 * with u = the boundary values in the constrained entries of src (device vector of the operator's number type, zero
 * elsewhere; NULL: homogeneous) read without the constraints (:816-824), K the operator's coefficient (:826-838) and
 * rhs_q[cell][(p+1)^3] = f(x_q) JxW_q at the quadrature points of every cell (device, cell order of the index table,
 * points lexicographic with x fastest; NULL: f = 0) (:839).  Rows of constrained DoFs stay zero; the result is summed
 * over the rank interfaces (:843).  No atomics: the cells are added up in a fixed order. */
int mgx_compute_residual(mgx_operator_t op, void *dst, const void *src, const void *rhs_q);
/* LaplaceOperator::vmult_residual(rhs, lhs, residual) laplace_operator.h:605-634 */
int mgx_vmult_residual(mgx_operator_t op, const void *rhs, const void *lhs, void *residual);
/* LaplaceOperator::compute_diagonal() laplace_operator.h:745-800; the inverse diagonal is kept
 * by the operator (Base::get_matrix_diagonal_inverse()) */
/* LaplaceOperator::vmult_with_cg_update(alpha, beta, r, q, p, x) :638-719, one pass of the cell loop
 * with the vector updates of a PCG step in its before / after hooks:
 *   x += alpha p ;  p = beta p + q ;  q = A p       (alpha == 0:  p = q ;  q = A p)
 *   sums = { q.p, r.r, q.r, q.q }  over all DoFs (summed over the ranks)
 * Constrained rows: the vector updates only; q = 0 there (the reference's loop does not touch them).
 * scratch: device vector of the operator's size and number type that carries partial sums of the
 * brick loop, or NULL (then allocated once and kept by the operator). */
int mgx_vmult_with_cg_update(mgx_operator_t op, double alpha, double beta, const void *r, void *q, void *p, void *x,
                             void *scratch, double sums[4]);
int mgx_compute_diagonal(mgx_operator_t op);
int mgx_get_inverse_diagonal(mgx_operator_t op, const void **dptr);

/* ---- PreconditionChebyshev (deal.II; configured at multigrid_solver.h:269-289) ---- */
/* SmootherType::initialize(matrix, additional_data) + estimate_eigenvalues.
 * degree < 0 == numbers::invalid_unsigned_int (determine from smoothing_range, level 0) */
int mgx_smoother_create(mgx_operator_t op, double smoothing_range, int degree,
                        int eig_cg_n_iterations, mgx_smoother_t *smoother);
int mgx_smoother_destroy(mgx_smoother_t smoother);
typedef struct
{
  double lambda_min, lambda_max, theta, delta;
  int    degree, cg_iterations;
} mgx_smoother_info;
int mgx_smoother_get_info(mgx_smoother_t smoother, mgx_smoother_info *info);
/* AdditionalData::polynomial_type: first_kind (MultigridSolver<dim,p,Number,Number2>,
 * multigrid_solver.h:277-278; the default here) or fourth_kind (the Number == Number2 specialisation,
 * multigrid_solver.h:951-952: delta = lambda_max, first step 4/(3 lambda_max), then
 * factor1 = (2k+1)/(2k+5), factor2 = (8k+12)/(lambda_max (2k+5))).  mgx_smoother_get_info then
 * reports delta = lambda_max.  May be switched at any time between applications. */
#define MGX_CHEBYSHEV_FIRST_KIND 0
#define MGX_CHEBYSHEV_FOURTH_KIND 1
int mgx_smoother_set_polynomial_type(mgx_smoother_t smoother, int polynomial_type);
/* PreconditionChebyshev::vmult (zero start) / ::step (multigrid_solver.h:399,657-659,678) */
int mgx_smoother_vmult(mgx_smoother_t smoother, void *x, const void *b);
int mgx_smoother_step(mgx_smoother_t smoother, void *x, const void *b);

/* ---- MGTransferMatrixFree, one level pair (built at multigrid_solver.h:209-222) ---- */
typedef struct
{
  /* children[8*parent + c] = fine-level cell index of child c (c = x + 2y + 4z) of coarse cell
   * `parent` (deal.II: cell->child(c)); host pointer, copied */
  const uint32_t *children;
  /* prolong_1d[a*(p+1)+i], a in [0,2p]: coarse 1D basis i at the fine patch point a */
  const double *prolong_1d;
  /* optional, may be NULL: weight_shift[27*parent + e] = log2(multiplicity) of patch entity e
   * (deal.II's weights_on_refined, 3^dim per cell), counting the parents of other ranks as well.
   * NULL: computed from the local tables; where that is not possible -- multiplicities that are not
   * powers of two (three cells around an edge of a multi-block mesh), decomposed meshes -- the
   * restriction uses owner weights instead: a shared fine DoF is restricted, with weight 1, by the
   * one parent (of the lowest rank that holds it) owning its entity.  Same R = P^T. */
  const uint8_t *weight_shift;
} mgx_transfer_desc;
int mgx_transfer_create(mgx_operator_t coarse, mgx_operator_t fine, const mgx_transfer_desc *desc,
                        mgx_transfer_t *transfer);
int mgx_transfer_destroy(mgx_transfer_t transfer);
/* prolongate (add==0, multigrid_solver.h:415) / prolongate_and_add (add!=0, :674);
 * with_constraints != 0: the level's Dirichlet entries are treated as zero
 * (transfer.initialize_constraints, :220) */
int mgx_prolongate(mgx_transfer_t transfer, void *fine, const void *coarse, int add, int with_constraints);
/* restrict_and_add (multigrid_solver.h:668) */
int mgx_restrict_and_add(mgx_transfer_t transfer, void *coarse, const void *fine, int with_constraints);

/* ---- MultigridSolver ---- */
typedef struct
{
  int n_levels;      /* maxlevel + 1 (multigrid_solver.h:108-109) */
  int degree_pre;    /* ctor argument degree_pre == degree_post (:126) */
  int n_cycles;      /* ctor argument n_cycles */
  /* per level: the V-cycle-precision operator `matrix` (:740) and the fp64 operator
   * `matrix_dp` (:745).  If the V-cycle number type is fp64, matrix[l] may equal matrix_dp[l]. */
  const mgx_operator_t *matrix;
  const mgx_operator_t *matrix_dp;
  /* per level l >= 1 (entry 0 unused): the level pair (l-1, l) in V-cycle precision with
   * constraints (`transfer`, :691) and in fp64 without (`mg_transfer_no_boundary`, :690).
   * The two may be the same object when the number types coincide. */
  const mgx_transfer_t *transfer;
  const mgx_transfer_t *transfer_dp;
  /* per level: rhs[level] as computed by compute_residual (:261), host fp64, n_dofs entries; rhs or rhs[level] NULL:
   * the level's right-hand side is assembled on the device afterwards (mgx_solver_compute_rhs) */
  const double *const *rhs;
  /* per level: inhomogeneous_bc[level] (:225-253) as index/value lists (host) */
  const uint32_t *const *bc_index;
  const double *const   *bc_value;
  const uint32_t        *bc_count;
} mgx_solver_desc;

/* MultigridSolver ctor from the smoother set-up on (:269-289): computes diagonals, estimates
 * eigenvalues, allocates the level vectors (:709-735) */
int mgx_solver_create(mgx_context_t ctx, const mgx_solver_desc *desc, mgx_solver_t *solver);
int mgx_solver_destroy(mgx_solver_t solver);
/* v_cycle(maxlevel, 1) (multigrid_solver.h:641-681) on the solver's own vectors: defect[maxlevel]
 * in, solution_update[maxlevel] out (mgx_solver_get_vector ids 2, 4) -- for a level put on top of
 * the hierarchy, as MultigridSolverDG does (multigrid_solver_dg.h:605-633) */
int mgx_solver_v_cycle(mgx_solver_t solver);
/* re-creates one level's smoother with other parameters (multigrid_solver_dg.h:271-291 configures
 * its FE_Q hierarchy differently from multigrid_solver.h:269-289); degree < 0: from the tolerance */
int mgx_solver_reset_smoother(mgx_solver_t solver, int level, double smoothing_range, int degree,
                              int eig_cg_n_iterations);
int mgx_solver_n_levels(mgx_solver_t solver);
int mgx_solver_get_operator(mgx_solver_t solver, int level, int fp64, mgx_operator_t *op);
/* device pointer to an operator's compressed index table [n_cells][27] and its sizes */
int mgx_operator_device_indices(mgx_operator_t op, const uint32_t **idx27, uint32_t *n_cells, uint32_t *n_dofs,
                                int *degree);
/* polynomial type of the smoothers above the coarsest level (which keeps the first kind with the
 * degree from its tolerance, multigrid_solver.h:955-959) */
int mgx_solver_set_polynomial_type(mgx_solver_t solver, int polynomial_type);
/* MultigridSolver ctor, multigrid_solver.h:225-261: the right-hand side of a level assembled on the device --
 * mgx_compute_residual of the level's fp64 operator with the solver's boundary values and rhs_q = f(x_q) JxW_q
 * (device, [n_cells][(p+1)^3]; NULL: f = 0) -- into the level's rhs vector (mgx_solver_desc::rhs[level] == NULL) */
int mgx_solver_compute_rhs(mgx_solver_t solver, int level, const double *rhs_q);
/* MultigridSolver::solve(do_analyze) :387-476.  trace (may be NULL) receives for every level
 * l >= 1 the residual norms {start, end} at trace[2*l], trace[2*l+1] when do_analyze != 0
 * (the L2 errors printed next to them need the analytic solution and are computed by the
 * caller from mgx_solver_get_solution, as the reference does on the host, :298-343). */
int mgx_solver_solve(mgx_solver_t solver, int do_analyze, double *reduction_rate, double *trace);
/* The same with the two points of the analysed solve at which the reference evaluates
 * compute_l2_error(level) exposed to the caller: hook(user, level, 0) right after the prolongation
 * of the coarser solution with its boundary values (:420-424, "error start level"), hook(user,
 * level, 1) after the correction has been added (:468-472, "error end level").  The stream is
 * idle during the call; the hook typically calls mgx_solver_get_solution(level, 1) and evaluates
 * the error on the host (mgx_cube_l2_error).  Only called when do_analyze != 0. */
typedef void (*mgx_level_hook)(void *user, int level, int stage);
int mgx_solver_solve_hooked(mgx_solver_t solver, int do_analyze, double *reduction_rate, double *trace,
                            mgx_level_hook hook, void *user);
/* MultigridSolver::solve_cg() :483-493: SolverCG with ReductionControl(1000,1e-16,1e-9) */
int mgx_solver_solve_cg(mgx_solver_t solver, unsigned int *iterations, double *reduction_rate);
/* The residual norms SolverCG handed to its ReductionControl (:486) during the last
 * mgx_solver_solve_cg / mgx_solver_solve_cg_fused: history[0] at the start, history[k] after
 * iteration k.  *count receives their number (iterations + 1); at most `capacity` are written. */
int mgx_solver_cg_history(mgx_solver_t solver, double *history, int capacity, int *count);
/* MultigridSolver::vmult(dst, src) :498-510: one V-cycle; dst/src fp64 device vectors */
int mgx_solver_vmult(mgx_solver_t solver, double *dst, const double *src);
/* Decomposed hierarchies only (no counterpart in the reference, whose MPI ranks keep exchanging on
 * every level): the V-cycle on levels <= `level` runs on `coarse`, an UNDECOMPOSED solver for the
 * same mesh up to that level which every rank creates on a second context of its own (without a
 * communicator).  Per V-cycle the defect of `level` is summed over the ranks into coarse's defect
 * (each DoF by its owner: owned[i] != 0), coarse runs its V-cycle, and every rank reads back the
 * correction of its DoFs through local_to_global[i] (the DoF of coarse's level `level` that local
 * DoF i is).  The finest level always stays decomposed.  The arrays are host memory, copied.
 * `coarse` may have MORE levels than level + 1: its finest level is the seam, i.e. level `level` of `solver` (a hierarchy
 * whose level 0 is level k of the whole mesh, mgx_cube_level_offset: the k coarsest levels then exist on `coarse` only,
 * and mgx_solver_solve starts from coarse's own full multigrid cycle, whose right-hand sides must be those of the same
 * problem).
 * Ownership: `coarse` and its context stay the caller's, but from this call on the coarse context runs on `solver`'s
 * stream (its own stream is destroyed): it must not host other solvers, must be used only through `solver`, and
 * must be destroyed BEFORE solver's context (the Python binding closes them in that order). */
int mgx_solver_set_agglomeration(mgx_solver_t solver, int level, mgx_solver_t coarse, const uint32_t *local_to_global,
                                 const uint8_t *owned, uint32_t n_local);
/* MultigridSolver::vmult_with_residual_update(residual, update, factor) :516-619: the V-cycle
 * as preconditioner with the residual update of the PCG step merged into the two precision casts:
 *   defect = residual + factor update ; V-cycle ; residual += factor update ; update = z
 *   out = { z.residual, z.(factor update) }   (factor == 0: both z.residual)
 * Constrained rows ([n - n_constrained, n), numbered last as the reference assumes :525): identity. */
int mgx_solver_vmult_with_residual_update(mgx_solver_t solver, double *residual, double *update, double factor,
                                          double out[2]);
/* PCG as mgx_solver_solve_cg with the matrix-vector product merged with the vector updates
 * (mgx_vmult_with_cg_update; the CG "fast path with merged vector operations" the reference prepares
 * the two functions above for, multigrid_solver.h:514-515).  Single rank. */
int mgx_solver_solve_cg_fused(mgx_solver_t solver, unsigned int *iterations, double *reduction_rate);
/* MultigridSolver::do_matvec() :624-628 / do_matvec_smoother() :633-637 */
int mgx_solver_do_matvec(mgx_solver_t solver);
int mgx_solver_do_matvec_smoother(mgx_solver_t solver);
/* MultigridSolver::get_solution() :376-382 (boundary values inserted); device fp64 pointer of
 * solution[level] */
int mgx_solver_get_solution(mgx_solver_t solver, int level, int insert_bc, const double **dptr);
/* device pointers of the level vectors, for tests: which = 0 rhs, 1 residual (fp64);
 * 2 defect, 3 t, 4 solution_update (V-cycle precision) */
int mgx_solver_get_vector(mgx_solver_t solver, int level, int which, void **dptr);
int mgx_solver_get_smoother(mgx_solver_t solver, int level, mgx_smoother_t *smoother);
/* MultigridSolver::print_wall_times() :348-371: per level 6 accumulated times
 * {mg_mv, restrict, prolongate, inhomBC, mg_vec, smoother} in seconds (HIP events); resets */
int mgx_solver_get_timings(mgx_solver_t solver, double *timings /* n_levels*6 */);
int mgx_solver_enable_timings(mgx_solver_t solver, int enable);

#ifdef __cplusplus
}
#endif
#endif
