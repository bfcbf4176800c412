/*
 * mg_oracle.h -- CPU restatement (plain C + OpenMP) of the reference's matrix-free
 * geometric-multigrid Laplace path for the structured poisson_cube problem.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load this library; the product (libmgx.so) never
 * links, loads or calls anything in oracle/.
 *
 * PARITY STATUS: "parity unpinned by reference tests" -- the reference has no test suite
 * and cannot be compiled here (deal.II >= 9.5 + p4est are absent).  The
 * only known-answer data are the README transcript lines (README.md:135-159), produced with
 * mixed precision / deal.II's own DoF numbering; the oracle is checked against them in
 * tests/test_oracle_readme.py (L2 errors, CG iteration counts, V-cycle reduction rate) and
 * against independent mathematical properties (dense element matrix, symmetry, null space,
 * R = P^T, polynomial reproduction).
 *
 * All reference citations are relative to /root/reference/.
 */
#ifndef MG_ORACLE_H
#define MG_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_INVALID 0xFFFFFFFFu
#define ORC_MAXN 10 /* p <= 9, n = p+1 <= 10 (poisson_cube/program.cc:68-69) */

typedef struct orc_problem orc_problem;

/* Build the whole MultigridSolver state (multigrid_solver.h:100-292) for the cube
 * [-0.9,1]^3 with n_subdiv * 2^n_refine cells per direction (poisson_cube/program.cc:532-545,570),
 * FE_Q(p), Gauss(p+1) quadrature, constant coefficient 1, Dirichlet data and rhs from the
 * manufactured solution prod sin(3 pi x_d) (program.cc:98-144).
 *   degree   : Chebyshev degree for pre- and post-smoothing (multigrid_solver.h:275)
 *   n_cycles : V-cycles per FMG level (multigrid_solver.h:451)
 *   vfloat   : 0 -> V-cycle vectors/operators in fp64, 1 -> fp32 (program.cc:76) */
orc_problem *orc_create(int p, int n_subdiv, int n_refine, int degree, int n_cycles, int vfloat);
/* "doubling" mesh family (poisson_cube/program.cc:509-529): box of sx x sy x sz cubic coarse cells
 * of size 1.9 with lower corner (-1,-1,-1), refined n_refine times */
orc_problem *orc_create_box(int p, int sx, int sy, int sz, int n_refine, int degree, int n_cycles, int vfloat);
/* Mapped meshes and variable coefficients (poisson_shell; laplace_operator.h:388-430, 493-522): the
 * reference box of sx x sy x sz coarse cells of size h0 from (origin,)*3, mapped by `geometry`
 * (isoparametric of degree p, as MappingQ(min(p,10)) multigrid_solver.h:139), problem = solution /
 * rhs / coefficient set */
#define ORC_GEOM_CARTESIAN 0
#define ORC_GEOM_SHEARED 1
#define ORC_GEOM_SHELL_SECTOR 2
#define ORC_GEOM_TABLES 3 /* cell geometry given by the caller's node tables (orc_create_from_mesh) */
#define ORC_PROBLEM_CUBE 0
#define ORC_PROBLEM_SHELL 1
orc_problem *orc_create_mapped(int p, int sx, int sy, int sz, double origin, double h0, int n_refine, int degree,
                               int n_cycles, int vfloat, int geometry, int problem);
/* [n_cells][(p+1)^3][6] merged coefficient of the general branch (NULL on the affine branch) */
const double *orc_coef_q(const orc_problem *P, int level);
void          orc_set_affine_coef(orc_problem *P, int level, const double *coef6);
void orc_cells_per_dim3(const orc_problem *P, int level, int out[3]);
void orc_destroy(orc_problem *P);

int orc_n_levels(const orc_problem *P);
int orc_degree(const orc_problem *P);
uint32_t orc_n_cells(const orc_problem *P, int level);
uint32_t orc_n_dofs(const orc_problem *P, int level);
uint32_t orc_n_constrained(const orc_problem *P, int level);
int orc_cells_per_dim(const orc_problem *P, int level);

/* tables (read-only views) */
const uint32_t *orc_idx27(const orc_problem *P, int level);        /* constrained -> ORC_INVALID */
const uint32_t *orc_idx27_plain(const orc_problem *P, int level);  /* no constraints */
const uint32_t *orc_constrained(const orc_problem *P, int level);  /* constrained dof list */
const uint32_t *orc_cell_coords(const orc_problem *P, int level);  /* [n_cells*3] cx,cy,cz */
const uint32_t *orc_dof_grid(const orc_problem *P, int level);     /* [n_dofs] lexicographic grid id */
const double *orc_shape_values(const orc_problem *P);   /* S[q*n+i]  */
const double *orc_colloc_grad(const orc_problem *P);    /* D[q*n+r]  */
const double *orc_qweights(const orc_problem *P);       /* w[q]      */
const double *orc_qpoints(const orc_problem *P);        /* Gauss points in [0,1] */
const double *orc_gll(const orc_problem *P);            /* GLL nodes in [0,1] */
const double *orc_prolong_1d(const orc_problem *P);     /* P1[a*n+i], a in [0,2p] */
const double *orc_rhs(const orc_problem *P, int level);
const double *orc_inv_diag(const orc_problem *P, int level);
double orc_h(const orc_problem *P, int level);

/* Chebyshev parameters per level (PreconditionChebyshev) */
void orc_set_polynomial_type(orc_problem *P, int fourth_kind);
void orc_reset_smoother(orc_problem *P, int l, double smoothing_range, int degree, int eig_cg_n_iterations);
void orc_cheb_info(const orc_problem *P, int level, double *lambda_min, double *lambda_max,
                   double *theta, double *delta, int *degree, int *cg_its);

/* inhomogeneous boundary map of a level: returns count, fills idx/val if non-NULL */
uint32_t orc_bc(const orc_problem *P, int level, uint32_t *idx, double *val);

/* LaplaceOperator (fp64 instance matrix_dp; 'vf' variants use the V-cycle number type) */
void orc_vmult(const orc_problem *P, int level, double *dst, const double *src);
/* laplace_operator.h:638-719 / multigrid_solver.h:516-619 (out[3]: z.res, z.(factor upd), res.res) */
void orc_vmult_with_cg_update(const orc_problem *P, int level, double alpha, double beta, const double *r, double *q,
                              double *p, double *x, double *sums);
void orc_vmult_with_residual_update(orc_problem *P, double *residual, double *update, double factor, double *out);
void orc_vmult_residual(const orc_problem *P, int level, const double *rhs, const double *lhs,
                        double *res);
/* dense reference: assembles the element matrix with plain gradients (no sum factorisation,
 * no compressed indices) on the lexicographic grid and applies it; for cross-checking */
void orc_vmult_dense_lex(const orc_problem *P, int level, double *dst_lex, const double *src_lex);

/* Chebyshev smoother on V-cycle-precision data held as double at the interface */
void orc_cheb_vmult(orc_problem *P, int level, double *x, const double *b);
void orc_cheb_step(orc_problem *P, int level, double *x, const double *b);

/* MGTransferMatrixFree restated; with_bc: zero coarse Dirichlet entries */
void orc_prolongate(const orc_problem *P, int level, double *fine, const double *coarse, int add,
                    int with_bc);
void orc_restrict_and_add(const orc_problem *P, int level, double *coarse, const double *fine,
                          int with_bc);

/* MultigridSolver::vmult (multigrid_solver.h:498-510): one V-cycle as preconditioner */
void orc_vcycle_apply(orc_problem *P, double *dst, const double *src);

/* MultigridSolver::solve (multigrid_solver.h:387-476).  If trace != NULL it receives, for each
 * level >= 1, 4 doubles {error start, residual start, residual end, error end}
 * (only meaningful with do_analyze).  Returns the V-cycle reduction rate. */
double orc_solve(orc_problem *P, int do_analyze, double *trace);
/* A mesh described by tables instead of the built-in structured box (multi-block meshes such as the
 * hyper_shell of poisson_shell): one entry per level, all arrays copied.  children: [n_cells(level-1)*8]
 * cells of this level (NULL on level 0); dof_gid: run-independent id per DoF; ent_mult: cells around
 * each of the 27 entities of every cell; cell_nodes: [n_cells][3][(p+1)^3] physical Gauss-Lobatto points.
 * The index tables must satisfy the entity-contiguity contract of laplace_operator.h:272-340. */
typedef struct
{
  uint32_t        n_cells, n_dofs, n_constrained;
  const uint32_t *idx27, *idx27_plain, *constrained, *children, *dof_gid;
  const uint8_t  *ent_mult;
  const double   *cell_nodes;
} orc_mesh_level;
orc_problem *orc_create_from_mesh(int p, int n_levels, const orc_mesh_level *mesh, int degree, int n_cycles, int vfloat,
                                  int problem);

/* MultigridSolver::solve_cg (multigrid_solver.h:483-493) */
int orc_solve_cg(orc_problem *P, double *reduction);
/* residual norms of that solve: [0] start, [k] after iteration k; returns their number */
int orc_cg_history(const orc_problem *P, double *out, int capacity);
/* MultigridSolver::compute_l2_error (multigrid_solver.h:298-343) */
double orc_l2_error(orc_problem *P, int level);
const double *orc_solution(orc_problem *P, int level);

/* throughput helpers for bench.py's cpu_baseline leg: run n applications, return seconds */
double orc_time_vmult(orc_problem *P, int level, int n);
double orc_time_vcycle(orc_problem *P, int n);
int orc_num_threads(void);
void orc_set_num_threads(int n); /* n <= 0: back to the default (visible cores, cgroup quota) */

#ifdef __cplusplus
}
#endif
#endif
