/*
 * mgx_cube.h -- host-side discretisation provider for the poisson_cube problem.
 *
 * In the reference everything in this header is supplied by deal.II (p4est mesh, DoFHandler,
 * MatrixFree, FE_Q shape info) and by the constructor of MultigridSolver; deal.II is not
 * available to this build, so the structured-cube equivalent is provided here and feeds the
 * device ABI of mgx.h.  A deal.II based caller would fill the mgx_* descriptors from
 * LaplaceOperator::get_compressed_dof_indices() etc. instead.
 *
 *   mesh              GridGenerator::subdivided_hyper_cube(tria, n_subdiv, -0.9, 1.0) +
 *                     refine_global(n_refine)            poisson_cube/program.cc:542,570
 *   cell order        forest/Morton order (p4est), children of cell c are 8c..8c+7
 *   DoF numbering     entity-contiguous, lexicographic inside an entity, Dirichlet DoFs last
 *                     (contract of laplace_operator.h:272-340); the order
 *                     of the entities is the provider's choice, see MGX_CUBE_NUMBERING_* below
 *   problem           u = prod sin(3 pi x_d), f = 27 pi^2 u, coefficient 1, Dirichlet on all
 *                     faces                              poisson_cube/program.cc:98-144,266
 */
#ifndef MGX_CUBE_H
#define MGX_CUBE_H

#include "mgx.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mgx_cube_s *mgx_cube_t;

/* Order in which the mesh entities receive their DoF ranges (the compressed index table holds for
 * any order; deal.II's own choice is MatrixFree's renumbering):
 *   BRICK  every entity is numbered by the first cell (Morton order) that contains it, and the
 *          entities numbered by one 4x4x4 (p <= 4) or 2x2x2 (p >= 5) brick of consecutive cells are
 *          grouped by the brick-level entity they lie on: brick interior, then its faces, edges,
 *          corners.  DoFs that the device cell loop completes in the same colour launch then form
 *          long contiguous runs -- no 128-B memory line is shared between launches.
 *   CELL   plain first-touch order cell by cell (one contiguous block per cell),
 *          kept so that tests can show that results do not depend on the choice. */
#define MGX_CUBE_NUMBERING_BRICK 0
#define MGX_CUBE_NUMBERING_CELL 1

int mgx_cube_create(int degree, int n_subdiv, int n_refine, mgx_cube_t *cube); /* BRICK numbering */
int mgx_cube_create_numbered(int degree, int n_subdiv, int n_refine, int numbering, mgx_cube_t *cube);

/* General form: a box of roots[0] x roots[1] x roots[2] cubic coarse cells of size h0 with lower
 * corner (origin,origin,origin), refined n_refine times, distributed over a procs[0] x procs[1] x
 * procs[2] process grid (rank = (pz*procs[1] + py)*procs[0] + px); every rank owns
 * roots[d]/procs[d] coarse cells per direction on every level.  With roots = {1|2} per direction,
 * origin = -1, h0 = 1.9 this is the reference's "doubling" mesh family
 * (GridGenerator::subdivided_hyper_rectangle, poisson_cube/program.cc:509-529), which is also how
 * the problem is scaled over several GPUs: one coarse cube per rank.
 * Interface DoFs are duplicated on the ranks that share them; the lists below
 * drive the exchange (mgx_exchange_desc in mgx.h). */
typedef struct
{
  int    degree, n_refine;
  int    roots[3];
  double origin, h0;
  int    procs[3];
  int    rank;
  int    numbering; /* MGX_CUBE_NUMBERING_* */
  int    geometry;  /* MGX_CUBE_GEOMETRY_*: map of the reference box to physical space */
  int    problem;   /* MGX_CUBE_PROBLEM_*: solution / right-hand side / coefficient */
} mgx_cube_box_desc;
/* Mapped meshes and variable coefficients (poisson_shell, BASELINE config 4).  With a geometry other
 * than CARTESIAN or the SHELL problem every level carries the per-(cell, q) merged coefficient of
 * evaluate_coefficient's general branch (laplace_operator.h:388-430), the cell geometry being the
 * degree-p interpolant of the map (MappingQ, multigrid_solver.h:139):
 *   SHEARED       a constant non-symmetric matrix close to the identity (affine cells, full tensor)
 *   SHELL_SECTOR  one of the six blocks of GridGenerator::hyper_shell(0, 0.5, 1.0, 6)
 *                 (poisson_shell/program.cc:425) as an equiangular cube-sphere sector: curved cells
 *   PROBLEM_SHELL u = sin(2 pi (x+y)), a = 1 + 1e6 prod_e cos^2(2 pi x_e + 0.1 e), f = -div(a grad u)
 *                 (poisson_shell/program.cc:97-137, 157-200, 219-225)
 *   HYPER_SHELL   the whole shell, mgx_cube_create_shell below */
#define MGX_CUBE_GEOMETRY_CARTESIAN 0
#define MGX_CUBE_GEOMETRY_SHEARED 1
#define MGX_CUBE_GEOMETRY_SHELL_SECTOR 2
#define MGX_CUBE_GEOMETRY_HYPER_SHELL 3
#define MGX_CUBE_PROBLEM_CUBE 0
#define MGX_CUBE_PROBLEM_SHELL 1
int mgx_cube_create_box(const mgx_cube_box_desc *desc, mgx_cube_t *cube);
/* The mesh of poisson_shell: GridGenerator::hyper_shell(tria, origin, 0.5, 1.0, n_coarse) with n_coarse = 6
 * (one cell per face of a cube) or 12 (per face of a rhombic dodecahedron), refined n_refine times
 * (poisson_shell/program.cc:425-431); Dirichlet values on both spheres.  The coarse cells ("blocks") are
 * refined uniformly, cells in Morton order block after block (children of cell c: 8c .. 8c+7).  The local
 * coordinates of the blocks are chosen such that every mesh entity is seen in the same lexicographic order
 * by all cells that contain it, so that the compressed index tables of LaplaceOperator hold on the whole
 * shell (laplace_operator.h:272-340); the cell geometry is the degree-p interpolant of the block map
 * (0.5 + 0.5 w) n(u, v), n the normalised bilinear interpolant of the polyhedron face.  problem:
 * MGX_CUBE_PROBLEM_SHELL (poisson_shell) or MGX_CUBE_PROBLEM_CUBE (constant coefficient).
 * The multiplicities of the transfer are not powers of two where three blocks meet: levels carry no
 * weight_shift, mgx_transfer_create derives owner weights. */
int mgx_cube_create_shell(int degree, int n_coarse, int n_refine, int problem, mgx_cube_t *cube);
/* The same distributed over ranks (the reference partitions the refined mesh cell by cell,
 * poisson_shell/program.cc:249,274).  Where n_ranks divides n_coarse, rank r owns the coarse cells
 * [r n_coarse / n_ranks, (r + 1) n_coarse / n_ranks) with everything refined from them.  Otherwise, where the mesh is
 * refined at least once and n_ranks divides 8 n_coarse (8 ranks: 6 of the 48 / 12 of the 96 cells of level 1 each), the
 * cells of LEVEL 1 are dealt out the same way: level i of the object is then level i + 1 of the whole mesh
 * (mgx_cube_level_offset() == 1, mgx_cube_n_levels() == n_refine), and the coarse cells exist only on the undecomposed
 * copy of the coarse levels every rank keeps (mgx_solver_set_agglomeration, mgx_solver_set_coarse_start).  Neither:
 * uneven shares of coarse cells (n_ranks <= n_coarse).  DoFs on faces between cells of different ranks are duplicated
 * and exchanged like the interface DoFs of the block-split cube (mgx_cube_exchange_desc). */
int mgx_cube_create_shell_ranks(int degree, int n_coarse, int n_refine, int problem, int n_ranks, int rank, mgx_cube_t *cube);
/* level of the whole mesh that level 0 of this object is (0 except for the level-1 partition above) */
int mgx_cube_level_offset(mgx_cube_t cube);
/* multi-block meshes: the physical Gauss-Lobatto points of every cell of a level,
 * out[cell][3][(p+1)^3], and the number of cells around each of the 27 entities of every cell
 * (what deal.II's mesh would tell a caller) */
int            mgx_cube_cell_nodes(mgx_cube_t cube, int level, double *out);
const uint8_t *mgx_cube_entity_multiplicity(mgx_cube_t cube, int level);
/* [n_cells][6][(p+1)^3] merged coefficient of a mapped level (NULL on the Cartesian cube) */
const double *mgx_cube_coef_q(mgx_cube_t cube, int level);
int mgx_cube_rank(mgx_cube_t cube);
int mgx_cube_size(mgx_cube_t cube);
void mgx_cube_cells_per_dim3(mgx_cube_t cube, int level, uint32_t local[3], uint32_t global[3]);
/* neighbours of this rank on a level, ascending rank; index lists are ordered by the global grid
 * id of the DoF, so that both sides of an interface enumerate it identically */
int             mgx_cube_n_neighbors(mgx_cube_t cube, int level);
int             mgx_cube_neighbor_rank(mgx_cube_t cube, int level, int k);
uint32_t        mgx_cube_neighbor_count(mgx_cube_t cube, int level, int k);
const uint32_t *mgx_cube_neighbor_index(mgx_cube_t cube, int level, int k);
/* union of the lists, and the subset owned by a lower rank (excluded from this rank's dot products) */
uint32_t        mgx_cube_n_shared(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_shared(mgx_cube_t cube, int level);
uint32_t        mgx_cube_n_not_owned(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_not_owned(mgx_cube_t cube, int level);
/* level >= 1: log2 of the global multiplicity of the 27 patch entities of every parent cell */
const uint8_t  *mgx_cube_weight_shift(mgx_cube_t cube, int level);
int mgx_cube_destroy(mgx_cube_t cube);

int      mgx_cube_n_levels(mgx_cube_t cube);
int      mgx_cube_degree(mgx_cube_t cube);
uint32_t mgx_cube_n_cells(mgx_cube_t cube, int level);
uint32_t mgx_cube_n_dofs(mgx_cube_t cube, int level);
uint32_t mgx_cube_n_constrained(mgx_cube_t cube, int level);
uint32_t mgx_cube_cells_per_dim(mgx_cube_t cube, int level);
double   mgx_cube_cell_size(mgx_cube_t cube, int level);

/* tables (host memory owned by the cube) */
const uint32_t *mgx_cube_idx27(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_idx27_plain(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_constrained(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_children(mgx_cube_t cube, int level); /* level >= 1: [n_cells(level-1)*8] */
const uint32_t *mgx_cube_cell_coords(mgx_cube_t cube, int level);
/* dof -> GLOBAL lexicographic grid id ((gz*Gy+gy)*Gx+gx, G_d = N_d*p+1 of the whole mesh) */
const uint32_t *mgx_cube_dof_grid(mgx_cube_t cube, int level);
const double   *mgx_cube_shape_values(mgx_cube_t cube);
const double   *mgx_cube_colloc_grad(mgx_cube_t cube);
const double   *mgx_cube_qweights(mgx_cube_t cube);
const double   *mgx_cube_qpoints(mgx_cube_t cube);
const double   *mgx_cube_gll(mgx_cube_t cube);
const double   *mgx_cube_prolong_1d(mgx_cube_t cube);

/* MultigridSolver ctor pieces computed on the host (multigrid_solver.h:225-261):
 * inhomogeneous boundary values and rhs = int f phi - int grad u_bc . grad phi */
const double   *mgx_cube_rhs(mgx_cube_t cube, int level); /* assembled at the first call */
/* the integrand of the right-hand side alone: out[cell][(p+1)^3] = f(x_q) JxW_q (laplace_operator.h:839), cells
 * in the order of the index table, points lexicographic -- the input of mgx_compute_residual (include/mgx.h) */
int             mgx_cube_rhs_quadrature(mgx_cube_t cube, int level, double *out);
uint32_t        mgx_cube_bc_count(mgx_cube_t cube, int level);
const uint32_t *mgx_cube_bc_index(mgx_cube_t cube, int level);
const double   *mgx_cube_bc_value(mgx_cube_t cube, int level);

/* fills desc for LaplaceOperator<3,p,number> on `level` (pointers stay owned by the cube) */
int mgx_cube_operator_desc(mgx_cube_t cube, int level, int number, mgx_operator_desc *desc);
/* fills the exchange plan of `level` for a decomposed mesh; the three scratch arrays (at least 26
 * entries each) receive the per-neighbour pointers and must outlive the descriptor's use */
int mgx_cube_exchange_desc(mgx_cube_t cube, int level, int plan_id, mgx_exchange_desc *desc,
                           const uint32_t **index_scratch, int *rank_scratch, uint32_t *count_scratch);

/* MultigridSolver::compute_l2_error (multigrid_solver.h:298-343) for a host copy of
 * solution[level] (boundary values already inserted) */
double mgx_cube_l2_error(mgx_cube_t cube, int level, const double *solution_host);
/* the two local sums (error^2 and volume) of the above, to be added over the ranks */
void mgx_cube_l2_error_parts(mgx_cube_t cube, int level, const double *solution_host, double *err2, double *vol);

/* seeded benchmark vector of SURVEY.md 8d: value depends only on the global grid index, uniform
 * in [-1,1) (splitmix64 of seed+grid id), written in the level's DoF numbering */
int mgx_cube_seeded_vector(mgx_cube_t cube, int level, uint64_t seed, double *out_host);

/* Convenience: everything MultigridSolver's constructor builds, on the device.
 * vcycle_number = MGX_F32 (reference default, program.cc:76) or MGX_F64. */
typedef struct
{
  int             n_levels;
  mgx_operator_t *matrix;     /* [n_levels] V-cycle precision */
  mgx_operator_t *matrix_dp;  /* [n_levels] fp64 (== matrix[l] if vcycle_number is fp64) */
  mgx_transfer_t *transfer;    /* [n_levels], entry 0 NULL */
  mgx_transfer_t *transfer_dp; /* [n_levels], entry 0 NULL (== transfer[l] if fp64) */
  mgx_solver_t    solver;
} mgx_cube_solver;

int mgx_cube_solver_create(mgx_context_t ctx, mgx_cube_t cube, int vcycle_number, int degree_pre, int n_cycles,
                           mgx_cube_solver *out);
/* The same; device_rhs != 0: the right-hand sides of all levels are assembled on the GPU (mgx_solver_compute_rhs,
 * LaplaceOperator::compute_residual laplace_operator.h:804-845) from mgx_cube_rhs_quadrature instead of on the host */
int mgx_cube_solver_create_opt(mgx_context_t ctx, mgx_cube_t cube, int vcycle_number, int degree_pre, int n_cycles,
                               int device_rhs, mgx_cube_solver *out);
int mgx_cube_solver_destroy(mgx_cube_solver *s);

#ifdef __cplusplus
}
#endif
#endif
