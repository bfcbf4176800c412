/*
 * mg_oracle.c -- CPU restatement (plain C + OpenMP) of the reference's matrix-free geometric
 * multigrid Laplace path for poisson_cube.  TEST INFRASTRUCTURE ONLY -- see mg_oracle.h for the
 * usage rule and the parity status ("parity unpinned" by reference tests; pinned against the
 * README transcript and mathematical properties in tests/).
 *
 * Reference files followed (relative to /root/reference/):
 *   common/laplace_operator.h        operator, quadrature-point kernel, residual, diagonal, rhs
 *   common/vector_access_reduced.h   27-entry compressed gather/scatter
 *   common/multigrid_solver.h:54-782 FMG / V-cycle / PCG driver
 *   poisson_cube/program.cc          problem definition (domain, solution, rhs, mesh sizes)
 * deal.II parts (FEEvaluation, PreconditionChebyshev, MGTransferMatrixFree, SolverCG) are not
 * vendored; they are restated from SURVEY.md 8a rows E, R, S, T / Appendix D.
 */
#ifndef _GNU_SOURCE
#  define _GNU_SOURCE /* sched_getaffinity / CPU_COUNT */
#endif
#include "mg_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#ifdef _OPENMP
#  include <omp.h>
#endif
#include <sched.h>

#ifndef M_PI
#  define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------
 * 1D basis: FE_Q(p) = Lagrange polynomials on the p+1 Gauss-Lobatto nodes
 * (poisson_cube/program.cc:175,199); quadrature QGauss<1>(p+1) (multigrid_solver.h:160,186).
 * ------------------------------------------------------------------------------------------ */
typedef struct
{
  int    p, n;
  double gll[ORC_MAXN], gq[ORC_MAXN], gw[ORC_MAXN];
  double S[ORC_MAXN * ORC_MAXN];        /* S[q*n+i] = l_i^GLL(x_q)                     */
  double D[ORC_MAXN * ORC_MAXN];        /* D[q*n+r] = d/dx l_r^Gauss(x_q)              */
  double G[ORC_MAXN * ORC_MAXN];        /* G[q*n+i] = d/dx l_i^GLL(x_q) (dense check)  */
  double P1[2 * ORC_MAXN * ORC_MAXN];   /* P1[a*n+i]: coarse basis i at fine patch pt a */
} orc_basis;

static void legendre(int n, long double x, long double *P, long double *dP)
{
  long double p0 = 1, p1 = x;
  if (n == 0)
    {
      *P  = 1;
      *dP = 0;
      return;
    }
  for (int k = 2; k <= n; ++k)
    {
      long double pk = ((2 * k - 1) * x * p1 - (k - 1) * p0) / k;
      p0             = p1;
      p1             = pk;
    }
  *P  = p1;
  *dP = n * (x * p1 - p0) / (x * x - 1);
}

static long double lagrange(const long double *nodes, int n, int i, long double x)
{
  long double v = 1;
  for (int j = 0; j < n; ++j)
    if (j != i)
      v *= (x - nodes[j]) / (nodes[i] - nodes[j]);
  return v;
}

static long double lagrange_deriv(const long double *nodes, int n, int i, long double x)
{
  long double s = 0;
  for (int k = 0; k < n; ++k)
    if (k != i)
      {
        long double t = 1 / (nodes[i] - nodes[k]);
        for (int j = 0; j < n; ++j)
          if (j != i && j != k)
            t *= (x - nodes[j]) / (nodes[i] - nodes[j]);
        s += t;
      }
  return s;
}

static void basis_init(orc_basis *b, int p)
{
  const int   n = p + 1;
  long double gll[ORC_MAXN], gq[ORC_MAXN], gw[ORC_MAXN];
  b->p = p;
  b->n = n;
  /* Gauss-Legendre: roots of P_n on (-1,1), Newton from Chebyshev guesses */
  for (int i = 0; i < n; ++i)
    {
      long double x = -cosl((long double)M_PI * (i + 0.75L) / (n + 0.5L)), P, dP;
      for (int it = 0; it < 100; ++it)
        {
          legendre(n, x, &P, &dP);
          long double dx = P / dP;
          x -= dx;
          if (fabsl(dx) < 1e-19L)
            break;
        }
      legendre(n, x, &P, &dP);
      gq[i] = (x + 1) / 2;
      gw[i] = 1 / ((1 - x * x) * dP * dP); /* (2/((1-x^2)P'^2))/2 */
    }
  /* Gauss-Lobatto: +-1 and the roots of P_p' ; Newton on q(x) = P_p'(x) using
   * (1-x^2) P'' = 2x P' - p(p+1) P */
  gll[0] = 0;
  gll[p] = 1;
  for (int i = 1; i < p; ++i)
    {
      long double x = -cosl((long double)M_PI * i / p), P, dP;
      for (int it = 0; it < 100; ++it)
        {
          legendre(p, x, &P, &dP);
          long double d2P = (2 * x * dP - (long double)p * (p + 1) * P) / (1 - x * x);
          long double dx  = dP / d2P;
          x -= dx;
          if (fabsl(dx) < 1e-19L)
            break;
        }
      gll[i] = (x + 1) / 2;
    }
  for (int i = 0; i < n; ++i)
    {
      b->gll[i] = (double)gll[i];
      b->gq[i]  = (double)gq[i];
      b->gw[i]  = (double)gw[i];
    }
  for (int q = 0; q < n; ++q)
    for (int i = 0; i < n; ++i)
      {
        b->S[q * n + i] = (double)lagrange(gll, n, i, gq[q]);
        b->G[q * n + i] = (double)lagrange_deriv(gll, n, i, gq[q]);
        b->D[q * n + i] = (double)lagrange_deriv(gq, n, i, gq[q]);
      }
  /* embedding of the parent's basis into the 2 children: fine patch point a = child*p + local */
  for (int a = 0; a <= 2 * p; ++a)
    {
      const int         child = a < p ? 0 : 1;
      const int         loc   = a - child * p;
      const long double xi    = (child + gll[loc]) / 2;
      for (int i = 0; i < n; ++i)
        {
          long double v = lagrange(gll, n, i, xi);
          /* nodes of parent and child coincide at a = 0, p (if p even: also the midpoint), 2p:
           * make those rows exact unit vectors */
          if (fabsl(v) < 1e-18L)
            v = 0;
          if (fabsl(v - 1) < 1e-18L)
            v = 1;
          b->P1[a * n + i] = (double)v;
        }
    }
}

/* extreme eigenvalues of a symmetric tridiagonal matrix by Sturm-sequence bisection */
static int sturm_count(int n, const double *d, const double *e, double x)
{
  int    count = 0;
  double q     = d[0] - x;
  if (q < 0)
    ++count;
  for (int i = 1; i < n; ++i)
    {
      if (q == 0)
        q = 1e-300;
      q = d[i] - x - e[i - 1] * e[i - 1] / q;
      if (q < 0)
        ++count;
    }
  return count; /* number of eigenvalues < x */
}

static void tridiag_extreme_eigs(int n, const double *d, const double *e, double *lo, double *hi)
{
  double gl = d[0], gu = d[0];
  for (int i = 0; i < n; ++i)
    {
      const double r = (i > 0 ? fabs(e[i - 1]) : 0) + (i < n - 1 ? fabs(e[i]) : 0);
      if (d[i] - r < gl)
        gl = d[i] - r;
      if (d[i] + r > gu)
        gu = d[i] + r;
    }
  for (int which = 0; which < 2; ++which)
    {
      const int k = which == 0 ? 1 : n; /* k-th smallest eigenvalue */
      double    a = gl, b = gu;
      for (int it = 0; it < 200; ++it)
        {
          const double m = 0.5 * (a + b);
          if (m == a || m == b)
            break;
          if (sturm_count(n, d, e, m) >= k)
            b = m;
          else
            a = m;
        }
      if (which == 0)
        *lo = 0.5 * (a + b);
      else
        *hi = 0.5 * (a + b);
    }
}

/* ------------------------------------------------------------------------------------------
 * Mesh level: uniform N^3 cells of [-0.9,1]^3 (poisson_cube/program.cc:542), cells in
 * forest/Morton order (p4est order of the reference), DoFs numbered entity by entity in
 * first-touch order so that every mesh entity's DoFs are contiguous and lexicographic
 * (the contract asserted in laplace_operator.h:272-340); Dirichlet DoFs are numbered last
 * (Appendix A: constrained DoFs after all unconstrained ones).
 * ------------------------------------------------------------------------------------------ */
typedef struct
{
  int       level, N;      /* N = cells in x (all directions for the cube) */
  int       Nd[3];         /* cells per direction */
  uint32_t  n_cells, n_dofs, n_constrained;
  uint32_t *idx27, *idx27_plain, *constrained, *cell_coords, *dof_grid;
  int       n_colours;        /* 8 parity classes on the structured box; greedy colouring of a mesh given by tables */
  uint32_t  colour_start[65];
  uint32_t *colour_cells;
  /* mesh given by tables (orc_create_from_mesh): cells around each of the 27 entities of every cell,
   * physical Gauss-Lobatto points of every cell [cell][3][n^3]; NULL on the structured box */
  uint8_t  *ent_mult;
  double   *cell_nodes;
  uint32_t *children_of_parent; /* [n_cells(level-1)*8] -> cell index on this level */
  double    h, coef[6];
  /* general branch (laplace_operator.h:388-430): per (cell, q) merged coefficient with the weight
   * folded in [cell][q][xx,yy,zz,xy,xz,yz]; JxW and the physical quadrature points.  NULL on the
   * Cartesian constant-coefficient mesh (affine branch, one tensor per mesh). */
  double   *coef_q, *jxw, *xq;
} orc_level;

struct orc_problem
{
  int        p, n_subdiv, n_levels, degree, n_cycles, vfloat;
  int        roots[3];  /* coarse cells per direction */
  double     origin;    /* lower corner of the domain (all directions) */
  double     h0;        /* size of a coarse cell */
  int        geometry;  /* ORC_GEOM_* */
  int        problem;   /* ORC_PROBLEM_* */
  orc_basis  basis;
  orc_level *levels;
  void      *Bd, *Bf; /* basis_d / basis_f  */
  void      *vd, *vf; /* vlevel_d[] / vlevel_f[] : V-cycle data in the V-cycle number type */
  void      *cd;      /* cheb_d[] on the double operator (only used when vfloat==0: == vd) */
  /* fp64 outer vectors (multigrid_solver.h:709-719) */
  double **solution, **rhs, **residual;
  /* inhomogeneous boundary values (multigrid_solver.h:225-253) */
  uint32_t *bc_count;
  uint32_t **bc_idx;
  double  **bc_val;
  /* residual norms of the last orc_solve_cg: [0] = start, [k] = after iteration k (what SolverCG
   * hands to its ReductionControl); test infrastructure for comparing iteration histories */
  double cg_history[1001];
  int    cg_history_n;
};

static inline uint32_t morton_compact(uint32_t m)
{ /* gather every third bit */
  uint32_t r = 0;
  for (int b = 0; b < 10; ++b)
    r |= ((m >> (3 * b)) & 1u) << b;
  return r;
}

static void level_init(orc_level *L, int p, const int roots[3], double h0, int level)
{
  const int Nx = roots[0] << level, Ny = roots[1] << level, Nz = roots[2] << level;
  L->level     = level;
  L->N         = Nx;
  L->Nd[0]     = Nx;
  L->Nd[1]     = Ny;
  L->Nd[2]     = Nz;
  L->n_cells   = (uint32_t)Nx * Ny * Nz;
  L->h         = h0 / (1 << level);
  /* merged_coefficient = a * JxW * J^-T J^-1 = h^3 / h^2 (laplace_operator.h:374-387) */
  L->coef[0] = L->coef[1] = L->coef[2] = L->h;
  L->coef[3] = L->coef[4] = L->coef[5] = 0.;
  const uint32_t nc = L->n_cells;
  L->cell_coords    = (uint32_t *)malloc(sizeof(uint32_t) * 3 * (size_t)nc);
  const uint32_t per_root = 1u << (3 * level);
  for (uint32_t c = 0; c < nc; ++c)
    {
      const uint32_t r = c / per_root, m = c % per_root;
      const uint32_t rx = r % roots[0], ry = (r / roots[0]) % roots[1], rz = r / (roots[0] * roots[1]);
      L->cell_coords[3 * (size_t)c + 0] = (rx << level) + morton_compact(m);
      L->cell_coords[3 * (size_t)c + 1] = (ry << level) + morton_compact(m >> 1);
      L->cell_coords[3 * (size_t)c + 2] = (rz << level) + morton_compact(m >> 2);
    }
  /* colour lists */
  L->n_colours    = 8;
  uint32_t cnt[8] = {0};
  for (uint32_t c = 0; c < nc; ++c)
    cnt[(L->cell_coords[3 * (size_t)c] & 1) | ((L->cell_coords[3 * (size_t)c + 1] & 1) << 1) |
        ((L->cell_coords[3 * (size_t)c + 2] & 1) << 2)]++;
  L->colour_start[0] = 0;
  for (int i = 0; i < 8; ++i)
    L->colour_start[i + 1] = L->colour_start[i] + cnt[i];
  L->colour_cells = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)nc);
  uint32_t pos[8];
  for (int i = 0; i < 8; ++i)
    pos[i] = L->colour_start[i];
  for (uint32_t c = 0; c < nc; ++c)
    {
      const int col = (L->cell_coords[3 * (size_t)c] & 1) | ((L->cell_coords[3 * (size_t)c + 1] & 1) << 1) |
                      ((L->cell_coords[3 * (size_t)c + 2] & 1) << 2);
      L->colour_cells[pos[col]++] = c;
    }
  /* entity numbering: entity grid (2Nx+1)(2Ny+1)(2Nz+1), even coordinate = vertex plane, odd =
   * cell interior */
  const size_t Ex = (size_t)(2 * Nx + 1), Ey = (size_t)(2 * Ny + 1), Ez = (size_t)(2 * Nz + 1);
  uint32_t    *ent = (uint32_t *)malloc(sizeof(uint32_t) * Ex * Ey * Ez);
  for (size_t i = 0; i < Ex * Ey * Ez; ++i)
    ent[i] = ORC_INVALID;
  L->idx27       = (uint32_t *)malloc(sizeof(uint32_t) * 27 * (size_t)nc);
  L->idx27_plain = (uint32_t *)malloc(sizeof(uint32_t) * 27 * (size_t)nc);
  uint32_t next = 0;
  for (int pass = 0; pass < 2; ++pass) /* 0: unconstrained entities, 1: Dirichlet boundary */
    {
      for (uint32_t c = 0; c < nc; ++c)
        {
          const uint32_t X = L->cell_coords[3 * (size_t)c], Y = L->cell_coords[3 * (size_t)c + 1],
                         Z = L->cell_coords[3 * (size_t)c + 2];
          for (int cz = 0; cz < 3; ++cz)
            for (int cy = 0; cy < 3; ++cy)
              for (int cx = 0; cx < 3; ++cx)
                {
                  const size_t ex = 2 * X + cx, ey = 2 * Y + cy, ez = 2 * Z + cz;
                  const int    bdry = ex == 0 || ex == Ex - 1 || ey == 0 || ey == Ey - 1 || ez == 0 ||
                                   ez == Ez - 1;
                  if (bdry != pass)
                    continue;
                  const size_t eid = (ez * Ey + ey) * Ex + ex;
                  if (ent[eid] == ORC_INVALID)
                    {
                      ent[eid] = next;
                      next += (uint32_t)((cx == 1 ? p - 1 : 1) * (cy == 1 ? p - 1 : 1) *
                                         (cz == 1 ? p - 1 : 1));
                    }
                }
        }
      if (pass == 0)
        L->n_constrained = next; /* temporarily: number of unconstrained dofs */
    }
  L->n_dofs               = next;
  const uint32_t n_uncons = L->n_constrained;
  L->n_constrained        = L->n_dofs - n_uncons;
  L->constrained          = (uint32_t *)malloc(sizeof(uint32_t) * (L->n_constrained + 1));
  for (uint32_t i = 0; i < L->n_constrained; ++i)
    L->constrained[i] = n_uncons + i;
  /* tables + dof -> lexicographic grid id map */
  const size_t Gx = (size_t)Nx * p + 1, Gy = (size_t)Ny * p + 1;
  L->dof_grid     = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)L->n_dofs);
  for (uint32_t c = 0; c < nc; ++c)
    {
      const uint32_t X = L->cell_coords[3 * (size_t)c], Y = L->cell_coords[3 * (size_t)c + 1],
                     Z = L->cell_coords[3 * (size_t)c + 2];
      for (int cz = 0; cz < 3; ++cz)
        for (int cy = 0; cy < 3; ++cy)
          for (int cx = 0; cx < 3; ++cx)
            {
              const size_t   ex = 2 * X + cx, ey = 2 * Y + cy, ez = 2 * Z + cz;
              const uint32_t base = ent[(ez * Ey + ey) * Ex + ex];
              const int      e    = 9 * cz + 3 * cy + cx;
              L->idx27_plain[27 * (size_t)c + e] = base;
              L->idx27[27 * (size_t)c + e]       = base >= n_uncons ? ORC_INVALID : base;
              const int nx = cx == 1 ? p - 1 : 1, ny = cy == 1 ? p - 1 : 1, nz = cz == 1 ? p - 1 : 1;
              for (int oz = 0; oz < nz; ++oz)
                for (int oy = 0; oy < ny; ++oy)
                  for (int ox = 0; ox < nx; ++ox)
                    {
                      const size_t gx = (size_t)X * p + (cx == 0 ? 0 : (cx == 2 ? p : 1 + ox));
                      const size_t gy = (size_t)Y * p + (cy == 0 ? 0 : (cy == 2 ? p : 1 + oy));
                      const size_t gz = (size_t)Z * p + (cz == 0 ? 0 : (cz == 2 ? p : 1 + oz));
                      L->dof_grid[base + (uint32_t)((oz * ny + oy) * nx + ox)] =
                        (uint32_t)((gz * Gy + gy) * Gx + gx);
                    }
            }
    }
  free(ent);
  L->children_of_parent = NULL;
  if (level > 0)
    {
      const uint32_t npar   = nc / 8;
      L->children_of_parent = (uint32_t *)malloc(sizeof(uint32_t) * 8 * (size_t)npar);
      for (uint32_t pc = 0; pc < npar; ++pc)
        for (uint32_t ch = 0; ch < 8; ++ch)
          L->children_of_parent[8 * (size_t)pc + ch] = 8 * pc + ch; /* Morton order */
    }
}

static void level_free(orc_level *L)
{
  free(L->idx27);
  free(L->idx27_plain);
  free(L->constrained);
  free(L->cell_coords);
  free(L->dof_grid);
  free(L->colour_cells);
  free(L->children_of_parent);
  free(L->coef_q);
  free(L->jxw);
  free(L->xq);
  free(L->ent_mult);
  free(L->cell_nodes);
}

/* number-type instantiations */
#define NUM double
#define SUF d
#include "mg_oracle_num.inc"
#undef NUM
#undef SUF
#define NUM float
#define SUF f
#include "mg_oracle_num.inc"
#undef NUM
#undef SUF

/* V-cycle level data (multigrid_solver.h:725-735, 751) in either number type */
typedef struct
{
  double *defect, *t, *solution_update; /* used when vfloat == 0 */
  float  *defect_f, *t_f, *solution_update_f;
  cheb_d  cheb;
  cheb_f  cheb_f_;
} vlevel;

#define VL(P) ((vlevel *)(P)->vd)
#define BD(P) ((const basis_d *)(P)->Bd)
#define BF(P) ((const basis_f *)(P)->Bf)

/* Problem definitions.
 * ORC_PROBLEM_CUBE : poisson_cube/program.cc:98-104 (solution), :140-144 (rhs), coefficient 1 (:266)
 * ORC_PROBLEM_SHELL: poisson_shell/program.cc:97-137 (u = sin(2 pi (x+y))), :157-200 (coefficient
 *                    1 + 1e6 prod_e cos^2(2 pi x_e + 0.1 e) and its gradient), :219-225 (rhs =
 *                    -(lap u * a + grad a . grad u)) */
static int g_problem = 0; /* set by create_impl for the functions below (one problem per process) */
static double coefficient_a(const double x[3])
{
  if (g_problem != ORC_PROBLEM_SHELL)
    return 1.;
  double prod = 1.;
  for (int e = 0; e < 3; ++e)
    {
      const double cc = cos(2. * M_PI * x[e] + 0.1 * e);
      prod *= cc * cc;
    }
  return 1. + 1.0e6 * prod;
}
static double u_exact(double x, double y, double z)
{
  if (g_problem == ORC_PROBLEM_SHELL)
    return sin(2. * M_PI * (x + y));
  return sin(M_PI * x * 3.) * sin(M_PI * y * 3.) * sin(M_PI * z * 3.);
}
static double f_rhs(double x, double y, double z)
{
  if (g_problem == ORC_PROBLEM_SHELL)
    {
      const double X[3] = {x, y, z};
      const double lap  = -2. * 2. * M_PI * 2. * M_PI * sin(2. * M_PI * (x + y));
      const double gu   = 2. * M_PI * cos(2. * M_PI * (x + y)); /* d/dx = d/dy, d/dz = 0 */
      double       ga[3];
      for (int d = 0; d < 3; ++d)
        {
          double prod = 1.;
          for (int e = 0; e < 3; ++e)
            {
              const double cc = cos(2. * M_PI * X[e] + 0.1 * e);
              prod *= e == d ? -4. * M_PI * cc * sin(2. * M_PI * X[e] + 0.1 * e) : cc * cc;
            }
          ga[d] = 1.0e6 * prod;
        }
      return -(lap * coefficient_a(X) + (ga[0] + ga[1]) * gu);
    }
  return 3. * M_PI * 3. * M_PI * 3. * u_exact(x, y, z);
}

/* Geometry: the reference box [origin, origin + roots h0]^3 mapped to physical space.
 * ORC_GEOM_CARTESIAN: identity.  ORC_GEOM_SHEARED: a constant (non-symmetric) matrix, affine cells
 * with a full merged-coefficient tensor.  ORC_GEOM_SHELL_SECTOR: one of the six blocks of
 * GridGenerator::hyper_shell(0, 0.5, 1.0, 6) (poisson_shell/program.cc:425) as an equiangular
 * cube-sphere sector, s in [0,1]^3 -> (0.5 + 0.5 s_z) (tan a, tan b, 1)/|.| with a, b = (2 s - 1) pi/4. */
static void map_point(const orc_problem *P, const double X[3], double x[3])
{
  if (P->geometry == ORC_GEOM_SHEARED)
    {
      static const double S[3][3] = {{0., 1., 0.5}, {0.3, 0., 1.}, {0.2, 0.4, 0.}};
      for (int d = 0; d < 3; ++d)
        x[d] = X[d] + 0.1 * (S[d][0] * X[0] + S[d][1] * X[1] + S[d][2] * X[2]);
    }
  else if (P->geometry == ORC_GEOM_SHELL_SECTOR)
    {
      double s[3];
      for (int d = 0; d < 3; ++d)
        s[d] = (X[d] - P->origin) / (P->roots[d] * P->h0);
      const double a = (2. * s[0] - 1.) * M_PI / 4., b = (2. * s[1] - 1.) * M_PI / 4.;
      const double v[3] = {tan(a), tan(b), 1.}, r = 0.5 + 0.5 * s[2];
      const double nv   = sqrt(v[0] * v[0] + v[1] * v[1] + 1.);
      for (int d = 0; d < 3; ++d)
        x[d] = r * v[d] / nv;
    }
  else
    for (int d = 0; d < 3; ++d)
      x[d] = X[d];
}

static void grid_to_xyz(const orc_problem *P, const orc_level *L, uint32_t gid, double xyz[3])
{
  const int    p  = P->p;
  const size_t Gx = (size_t)L->Nd[0] * p + 1, Gy = (size_t)L->Nd[1] * p + 1;
  size_t       g[3] = {gid % Gx, (gid / Gx) % Gy, gid / (Gx * Gy)};
  for (int d = 0; d < 3; ++d)
    {
      size_t cell = g[d] / p, loc = g[d] % p;
      if (cell == (size_t)L->Nd[d])
        {
          cell = L->Nd[d] - 1;
          loc  = p;
        }
      xyz[d] = P->origin + L->h * ((double)cell + P->basis.gll[loc]);
    }
  const double X[3] = {xyz[0], xyz[1], xyz[2]};
  map_point(P, X, xyz);
}

/* LaplaceOperator::compute_residual (laplace_operator.h:804-845): dst = int f phi - int C grad
 * u_bc . grad phi ; src holds the Dirichlet values (read through the unconstrained index table,
 * :814,820), the result is scattered through the constrained one. */
static void compute_rhs(const orc_problem *P, const orc_level *L, double *dst, const double *src)
{
  const int      p = P->p, n = p + 1, n3 = n * n * n;
  const basis_d *B = BD(P);
  memset(dst, 0, sizeof(double) * L->n_dofs);
  for (int col = 0; col < L->n_colours; ++col)
    {
      const uint32_t  nc   = L->colour_start[col + 1] - L->colour_start[col];
      const uint32_t *list = L->colour_cells + L->colour_start[col];
#pragma omp parallel
      {
        double *u = (double *)malloc(sizeof(double) * 5 * n3), *t0 = u + n3, *gx = t0 + n3,
               *gy = gx + n3, *gz = gy + n3;
#pragma omp for schedule(static)
        for (uint32_t kk = 0; kk < nc; ++kk)
          {
            const uint32_t c = list[kk];
            gather27_d(p, L->idx27_plain + 27 * (size_t)c, src, u);
            for (int i = 0; i < n3; ++i) /* :823-824 */
              u[i] *= -1.0;
            sweep_d(n, 0, B->S, u, t0, 0);
            sweep_d(n, 1, B->S, t0, u, 0);
            sweep_d(n, 2, B->S, u, t0, 0);
            sweep_d(n, 0, B->D, t0, gx, 0);
            sweep_d(n, 1, B->D, t0, gy, 0);
            sweep_d(n, 2, B->D, t0, gz, 0);
            const double x0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c] : 0.,
                         y0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c + 1] : 0.,
                         z0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c + 2] : 0.;
            if (L->coef_q) /* general branch: full coefficient per q, JxW and x_q from the mapping */
              for (int q = 0; q < n3; ++q)
                {
                  const double *C = L->coef_q + ((size_t)c * n3 + q) * 6, *xq = L->xq + ((size_t)c * n3 + q) * 3;
                  const double  a = gx[q], b = gy[q], cc = gz[q];
                  gx[q] = C[0] * a + C[3] * b + C[4] * cc;
                  gy[q] = C[3] * a + C[1] * b + C[5] * cc;
                  gz[q] = C[4] * a + C[5] * b + C[2] * cc;
                  t0[q] = f_rhs(xq[0], xq[1], xq[2]) * L->jxw[(size_t)c * n3 + q];
                }
            else
            for (int k = 0, q = 0; k < n; ++k)
              for (int j = 0; j < n; ++j)
                for (int i = 0; i < n; ++i, ++q)
                  {
                    const double w = B->w[i] * B->w[j] * B->w[k];
                    const double a = gx[q], b = gy[q], cc = gz[q];
                    gx[q] = (L->coef[0] * a + L->coef[3] * b + L->coef[4] * cc) * w;
                    gy[q] = (L->coef[3] * a + L->coef[1] * b + L->coef[5] * cc) * w;
                    gz[q] = (L->coef[4] * a + L->coef[5] * b + L->coef[2] * cc) * w;
                    /* submit_value(rhs_val) multiplies by JxW = h^3 w_q (:839) */
                    t0[q] = f_rhs(x0 + L->h * P->basis.gq[i], y0 + L->h * P->basis.gq[j],
                                  z0 + L->h * P->basis.gq[k]) *
                            (L->h * L->h * L->h) * w;
                  }
            sweep_d(n, 0, B->Dt, gx, t0, 1);
            sweep_d(n, 1, B->Dt, gy, t0, 1);
            sweep_d(n, 2, B->Dt, gz, t0, 1);
            sweep_d(n, 0, B->St, t0, u, 0);
            sweep_d(n, 1, B->St, u, t0, 0);
            sweep_d(n, 2, B->St, t0, u, 0);
            scatter27_d(p, L->idx27 + 27 * (size_t)c, dst, u);
          }
        free(u);
      }
    }
}

static void set_bc(const orc_problem *P, int level, double *v, int zero)
{
  for (uint32_t i = 0; i < P->bc_count[level]; ++i)
    v[P->bc_idx[level][i]] = zero ? 0. : P->bc_val[level][i];
}

/* Number of threads the host actually grants this process: the smaller of the affinity mask and
 * the cgroup CPU quota (a GPU box hands a 16-CPU share of a much larger machine to one job; one
 * OpenMP thread per *visible* core would oversubscribe it by an order of magnitude). */
static int effective_threads(void)
{
  int n = 1;
#ifdef _OPENMP
  const char *env = getenv("OMP_NUM_THREADS");
  if (env && atoi(env) > 0)
    return atoi(env);
  cpu_set_t set;
  if (sched_getaffinity(0, sizeof(set), &set) == 0)
    n = CPU_COUNT(&set);
  else
    n = omp_get_num_procs();
  FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r");
  if (f)
    {
      char   quota[64];
      double period = 0;
      if (fscanf(f, "%63s %lf", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0)
        {
          const int q = (int)(atof(quota) / period + 0.5);
          if (q >= 1 && q < n)
            n = q;
        }
      fclose(f);
    }
  if (n < 1)
    n = 1;
#endif
  return n;
}

static orc_problem *create_impl(int p, const int roots[3], double origin, double h0, int n_refine, int degree,
                                int n_cycles, int vfloat, int geometry, int problem);

orc_problem *orc_create(int p, int n_subdiv, int n_refine, int degree, int n_cycles, int vfloat)
{
  /* "square" mesh: subdivided_hyper_cube(n_subdiv, -0.9, 1.0) (poisson_cube/program.cc:542) */
  const int roots[3] = {n_subdiv, n_subdiv, n_subdiv};
  if (n_subdiv < 1)
    return NULL;
  return create_impl(p, roots, -0.9, 1.9 / n_subdiv, n_refine, degree, n_cycles, vfloat, ORC_GEOM_CARTESIAN, ORC_PROBLEM_CUBE);
}

orc_problem *orc_create_box(int p, int sx, int sy, int sz, int n_refine, int degree, int n_cycles, int vfloat)
{
  /* "doubling" mesh family: subdivided_hyper_rectangle with cubic coarse cells of size 1.9 from
   * the corner (-1,-1,-1) (poisson_cube/program.cc:509-529) */
  const int roots[3] = {sx, sy, sz};
  if (sx < 1 || sy < 1 || sz < 1)
    return NULL;
  return create_impl(p, roots, -1.0, 1.9, n_refine, degree, n_cycles, vfloat, ORC_GEOM_CARTESIAN, ORC_PROBLEM_CUBE);
}

orc_problem *orc_create_mapped(int p, int sx, int sy, int sz, double origin, double h0, int n_refine, int degree,
                               int n_cycles, int vfloat, int geometry, int problem)
{
  const int roots[3] = {sx, sy, sz};
  if (sx < 1 || sy < 1 || sz < 1 || !(h0 > 0) || geometry < 0 || geometry > 2 || problem < 0 || problem > 1)
    return NULL;
  return create_impl(p, roots, origin, h0, n_refine, degree, n_cycles, vfloat, geometry, problem);
}

const double *orc_coef_q(const orc_problem *P, int l) { return P->levels[l].coef_q; }

/* affine branch with a full tensor (laplace_operator.h:374-387, 473-486): replaces the merged
 * coefficient of a level of a Cartesian problem (tests of the off-diagonal entries) */
void orc_set_affine_coef(orc_problem *P, int l, const double *coef)
{
  for (int i = 0; i < 6; ++i)
    P->levels[l].coef[i] = coef[i];
}

/* evaluate_coefficient, general branch (laplace_operator.h:388-430): per cell and quadrature point
 * coef = a(x_q) JxW J^-T J^-1 with JxW = det J w_q.  The geometry is isoparametric of degree p:
 * the GLL support points of a cell are mapped exactly, Jacobians come from the interpolant. */
static void level_geometry(const orc_problem *P, orc_level *L)
{
  const int        p = P->p, n = p + 1, n3 = n * n * n;
  const orc_basis *b = &P->basis;
  L->coef_q          = (double *)malloc(sizeof(double) * (size_t)L->n_cells * n3 * 6);
  L->jxw             = (double *)malloc(sizeof(double) * (size_t)L->n_cells * n3);
  L->xq              = (double *)malloc(sizeof(double) * (size_t)L->n_cells * n3 * 3);
  const basis_d *B   = BD(P);
#pragma omp parallel
  {
    double *xn = (double *)malloc(sizeof(double) * n3 * 16), *t0 = xn + 3 * n3, *t1 = t0 + n3, *J = t1 + n3; /* J: 9 n3 */
#pragma omp for schedule(static)
    for (uint32_t c = 0; c < L->n_cells; ++c)
      {
        if (L->cell_nodes) /* mesh given by tables: the cell's Gauss-Lobatto points as the caller's mesh places them */
          memcpy(xn, L->cell_nodes + 3 * (size_t)n3 * c, sizeof(double) * 3 * n3);
        else
        for (int k = 0, i3 = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i, ++i3)
              {
                const double X[3] = {P->origin + L->h * (L->cell_coords[3 * (size_t)c] + b->gll[i]),
                                     P->origin + L->h * (L->cell_coords[3 * (size_t)c + 1] + b->gll[j]),
                                     P->origin + L->h * (L->cell_coords[3 * (size_t)c + 2] + b->gll[k])};
                double       x[3];
                map_point(P, X, x);
                for (int d = 0; d < 3; ++d)
                  xn[d * n3 + i3] = x[d];
              }
        for (int d = 0; d < 3; ++d)
          {
            sweep_d(n, 0, B->S, xn + d * n3, t0, 0);
            sweep_d(n, 1, B->S, t0, t1, 0);
            sweep_d(n, 2, B->S, t1, t0, 0); /* t0 = x_d at the quadrature points */
            for (int q = 0; q < n3; ++q)
              L->xq[((size_t)c * n3 + q) * 3 + d] = t0[q];
            for (int e = 0; e < 3; ++e) /* J[d][e] = d x_d / d xi_e */
              sweep_d(n, e, B->D, t0, J + (3 * d + e) * n3, 0);
          }
        for (int k = 0, q = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i, ++q)
              {
                double Jm[3][3], Ji[3][3];
                for (int d = 0; d < 3; ++d)
                  for (int e = 0; e < 3; ++e)
                    Jm[d][e] = J[(3 * d + e) * n3 + q];
                const double det = Jm[0][0] * (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) -
                                   Jm[0][1] * (Jm[1][0] * Jm[2][2] - Jm[1][2] * Jm[2][0]) +
                                   Jm[0][2] * (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]);
                /* Ji = Jm^-1 : d xi_e / d x_d at [e][d] */
                Ji[0][0] = (Jm[1][1] * Jm[2][2] - Jm[1][2] * Jm[2][1]) / det;
                Ji[0][1] = (Jm[0][2] * Jm[2][1] - Jm[0][1] * Jm[2][2]) / det;
                Ji[0][2] = (Jm[0][1] * Jm[1][2] - Jm[0][2] * Jm[1][1]) / det;
                Ji[1][0] = (Jm[1][2] * Jm[2][0] - Jm[1][0] * Jm[2][2]) / det;
                Ji[1][1] = (Jm[0][0] * Jm[2][2] - Jm[0][2] * Jm[2][0]) / det;
                Ji[1][2] = (Jm[0][2] * Jm[1][0] - Jm[0][0] * Jm[1][2]) / det;
                Ji[2][0] = (Jm[1][0] * Jm[2][1] - Jm[1][1] * Jm[2][0]) / det;
                Ji[2][1] = (Jm[0][1] * Jm[2][0] - Jm[0][0] * Jm[2][1]) / det;
                Ji[2][2] = (Jm[0][0] * Jm[1][1] - Jm[0][1] * Jm[1][0]) / det;
                const double  jxw = det * b->gw[i] * b->gw[j] * b->gw[k];
                const double *xq  = L->xq + ((size_t)c * n3 + q) * 3;
                const double  a   = coefficient_a(xq) * jxw;
                double       *C   = L->coef_q + ((size_t)c * n3 + q) * 6;
                static const int pe[6] = {0, 1, 2, 0, 0, 1}, pf[6] = {0, 1, 2, 1, 2, 2};
                for (int m = 0; m < 6; ++m) /* (J^-1 J^-T)[e][f] = sum_d Ji[e][d] Ji[f][d] */
                  C[m] = a * (Ji[pe[m]][0] * Ji[pf[m]][0] + Ji[pe[m]][1] * Ji[pf[m]][1] + Ji[pe[m]][2] * Ji[pf[m]][2]);
                L->jxw[(size_t)c * n3 + q] = jxw;
              }
      }
    free(xn);
  }
}

/* everything MultigridSolver's constructor does once the levels exist: vectors, boundary values, right-hand
 * side, diagonal, smoothers (multigrid_solver.h:168-289) */
static orc_problem *finish_problem(orc_problem *P);

static orc_problem *create_impl(int p, const int roots[3], double origin, double h0, int n_refine, int degree,
                                int n_cycles, int vfloat, int geometry, int problem)
{
  g_problem = problem;
#ifdef _OPENMP
  omp_set_num_threads(effective_threads());
#endif
  if (p < 1 || p > 9 || n_refine < 0)
    return NULL;
  orc_problem *P = (orc_problem *)calloc(1, sizeof(orc_problem));
  P->p           = p;
  P->n_subdiv    = roots[0];
  P->roots[0]    = roots[0];
  P->roots[1]    = roots[1];
  P->roots[2]    = roots[2];
  P->origin      = origin;
  P->h0          = h0;
  P->geometry    = geometry;
  P->problem     = problem;
  P->n_levels    = n_refine + 1;
  P->degree      = degree;
  P->n_cycles    = n_cycles;
  P->vfloat      = vfloat;
  basis_init(&P->basis, p);
  P->Bd = malloc(sizeof(basis_d));
  P->Bf = malloc(sizeof(basis_f));
  basis_init_d((basis_d *)P->Bd, &P->basis);
  basis_init_f((basis_f *)P->Bf, &P->basis);
  P->levels = (orc_level *)calloc(P->n_levels, sizeof(orc_level));
  for (int l = 0; l < P->n_levels; ++l)
    {
      level_init(&P->levels[l], p, P->roots, P->h0, l);
      if (geometry != ORC_GEOM_CARTESIAN || problem != ORC_PROBLEM_CUBE)
        level_geometry(P, &P->levels[l]);
    }
  return finish_problem(P);
}

static orc_problem *finish_problem(orc_problem *P)
{
  const int degree = P->degree, vfloat = P->vfloat;
  P->solution = (double **)calloc(P->n_levels, sizeof(double *));
  P->rhs      = (double **)calloc(P->n_levels, sizeof(double *));
  P->residual = (double **)calloc(P->n_levels, sizeof(double *));
  P->bc_count = (uint32_t *)calloc(P->n_levels, sizeof(uint32_t));
  P->bc_idx   = (uint32_t **)calloc(P->n_levels, sizeof(uint32_t *));
  P->bc_val   = (double **)calloc(P->n_levels, sizeof(double *));
  vlevel *V   = (vlevel *)calloc(P->n_levels, sizeof(vlevel));
  P->vd       = V;
  for (int l = 0; l < P->n_levels; ++l)
    {
      orc_level *L   = &P->levels[l];
      P->solution[l] = (double *)calloc(L->n_dofs, sizeof(double));
      P->rhs[l]      = (double *)calloc(L->n_dofs, sizeof(double));
      P->residual[l] = (double *)calloc(L->n_dofs, sizeof(double));
      /* inhomogeneous boundary values: u at the support points of boundary DoFs, stored only
       * if nonzero (multigrid_solver.h:245-252) */
      P->bc_idx[l] = (uint32_t *)malloc(sizeof(uint32_t) * (L->n_constrained + 1));
      P->bc_val[l] = (double *)malloc(sizeof(double) * (L->n_constrained + 1));
      uint32_t cnt = 0;
      double  *bc_xyz = NULL; /* mesh given by tables: support points of the DoFs from the cells' nodes */
      if (L->cell_nodes)
        {
          const int p = P->p, n = p + 1, n3 = n * n * n;
          bc_xyz      = (double *)calloc(3 * (size_t)L->n_dofs, sizeof(double));
          for (uint32_t c = 0; c < L->n_cells; ++c)
            for (int e = 0; e < 27; ++e)
              {
                const int      cx = e % 3, cy = (e / 3) % 3, cz = e / 9;
                const int      nx = cx == 1 ? p - 1 : 1, ny = cy == 1 ? p - 1 : 1, nz = cz == 1 ? p - 1 : 1;
                const uint32_t base = L->idx27_plain[27 * (size_t)c + e];
                for (int oz = 0; oz < nz; ++oz)
                  for (int oy = 0; oy < ny; ++oy)
                    for (int ox = 0; ox < nx; ++ox)
                      {
                        const int i = cx == 0 ? 0 : (cx == 2 ? p : 1 + ox), j = cy == 0 ? 0 : (cy == 2 ? p : 1 + oy),
                                  k = cz == 0 ? 0 : (cz == 2 ? p : 1 + oz), i3 = (k * n + j) * n + i;
                        const uint32_t dof = base + (uint32_t)((oz * ny + oy) * nx + ox);
                        for (int d = 0; d < 3; ++d)
                          bc_xyz[3 * (size_t)dof + d] = L->cell_nodes[(3 * (size_t)c + d) * n3 + i3];
                      }
              }
        }
      for (uint32_t i = 0; i < L->n_constrained; ++i)
        {
          const uint32_t dof = L->constrained[i];
          double         x[3];
          if (bc_xyz)
            memcpy(x, bc_xyz + 3 * (size_t)dof, sizeof(x));
          else
            grid_to_xyz(P, L, L->dof_grid[dof], x);
          const double v = u_exact(x[0], x[1], x[2]);
          if (v != 0.0)
            {
              P->bc_idx[l][cnt] = dof;
              P->bc_val[l][cnt] = v;
              ++cnt;
            }
        }
      P->bc_count[l] = cnt;
      free(bc_xyz);
      set_bc(P, l, P->solution[l], 0);                      /* multigrid_solver.h:257-259 */
      compute_rhs(P, L, P->rhs[l], P->solution[l]);         /* :261 */
      /* V-cycle vectors and smoother (multigrid_solver.h:168-170, 269-289) */
      if (!vfloat)
        {
          V[l].defect          = (double *)calloc(L->n_dofs, sizeof(double));
          V[l].t               = (double *)calloc(L->n_dofs, sizeof(double));
          V[l].solution_update = (double *)calloc(L->n_dofs, sizeof(double));
          V[l].cheb.inv_diag   = (double *)malloc(sizeof(double) * L->n_dofs);
          V[l].cheb.x_old      = (double *)calloc(L->n_dofs, sizeof(double));
          V[l].cheb.tmp        = (double *)calloc(L->n_dofs, sizeof(double));
          compute_inv_diag_d(P, L, BD(P), V[l].cheb.inv_diag);
          if (l > 0)
            cheb_setup_d(P, L, BD(P), &V[l].cheb, 20., degree, 15);
          else
            cheb_setup_d(P, L, BD(P), &V[l].cheb, 1e-3, -1, (int)L->n_dofs);
        }
      else
        {
          V[l].defect_f          = (float *)calloc(L->n_dofs, sizeof(float));
          V[l].t_f               = (float *)calloc(L->n_dofs, sizeof(float));
          V[l].solution_update_f = (float *)calloc(L->n_dofs, sizeof(float));
          V[l].cheb_f_.inv_diag  = (float *)malloc(sizeof(float) * L->n_dofs);
          V[l].cheb_f_.x_old     = (float *)calloc(L->n_dofs, sizeof(float));
          V[l].cheb_f_.tmp       = (float *)calloc(L->n_dofs, sizeof(float));
          compute_inv_diag_f(P, L, BF(P), V[l].cheb_f_.inv_diag);
          if (l > 0)
            cheb_setup_f(P, L, BF(P), &V[l].cheb_f_, 20., degree, 15);
          else
            cheb_setup_f(P, L, BF(P), &V[l].cheb_f_, 1e-3, -1, (int)L->n_dofs);
          /* keep a double copy of the diagonal for the accessor */
          V[l].cheb.inv_diag = (double *)malloc(sizeof(double) * L->n_dofs);
          for (uint32_t i = 0; i < L->n_dofs; ++i)
            V[l].cheb.inv_diag[i] = V[l].cheb_f_.inv_diag[i];
          V[l].cheb.lambda_min = V[l].cheb_f_.lambda_min;
          V[l].cheb.lambda_max = V[l].cheb_f_.lambda_max;
          V[l].cheb.theta      = V[l].cheb_f_.theta;
          V[l].cheb.delta      = V[l].cheb_f_.delta;
          V[l].cheb.degree     = V[l].cheb_f_.degree;
          V[l].cheb.cg_its     = V[l].cheb_f_.cg_its;
        }
    }
  return P;
}

/* The same solver on a mesh the caller describes by tables -- what deal.II's DoFHandler / MatrixFree would
 * hand over: per level the compressed index tables, the constrained DoFs, the children of every cell of the
 * next coarser level, a run-independent id per DoF (start vector of the eigenvalue estimate), the number of
 * cells around each of the 27 entities of every cell (transfer weights) and the physical Gauss-Lobatto points
 * of every cell, from which the mapping data and the merged coefficient are computed HERE
 * (evaluate_coefficient, laplace_operator.h:388-430).  Used for the multi-block shell of poisson_shell, whose
 * mesh generator lives in the product's provider (mgx_cube_create_shell); all arithmetic stays the oracle's. */
orc_problem *orc_create_from_mesh(int p, int n_levels, const orc_mesh_level *mesh, int degree, int n_cycles, int vfloat,
                                  int problem)
{
  g_problem = problem;
#ifdef _OPENMP
  omp_set_num_threads(effective_threads());
#endif
  if (p < 1 || p > 9 || n_levels < 1 || !mesh)
    return NULL;
  orc_problem *P = (orc_problem *)calloc(1, sizeof(orc_problem));
  P->p           = p;
  P->n_subdiv    = 1;
  P->roots[0] = P->roots[1] = P->roots[2] = 1;
  P->geometry = ORC_GEOM_TABLES;
  P->problem  = problem;
  P->n_levels = n_levels;
  P->degree   = degree;
  P->n_cycles = n_cycles;
  P->vfloat   = vfloat;
  basis_init(&P->basis, p);
  P->Bd = malloc(sizeof(basis_d));
  P->Bf = malloc(sizeof(basis_f));
  basis_init_d((basis_d *)P->Bd, &P->basis);
  basis_init_f((basis_f *)P->Bf, &P->basis);
  P->levels = (orc_level *)calloc(P->n_levels, sizeof(orc_level));
  const int n3 = (p + 1) * (p + 1) * (p + 1);
  for (int l = 0; l < n_levels; ++l)
    {
      orc_level            *L = &P->levels[l];
      const orc_mesh_level *m = &mesh[l];
      L->level         = l;
      L->N             = 0;
      L->n_cells       = m->n_cells;
      L->n_dofs        = m->n_dofs;
      L->n_constrained = m->n_constrained;
      L->h             = 1.;
#define ORC_DUP(dst, src, type, count)                                \
  do                                                                  \
    {                                                                 \
      (dst) = (type *)malloc(sizeof(type) * ((size_t)(count) + 1));   \
      memcpy((dst), (src), sizeof(type) * (size_t)(count));           \
    }                                                                 \
  while (0)
      ORC_DUP(L->idx27, m->idx27, uint32_t, 27 * (size_t)m->n_cells);
      ORC_DUP(L->idx27_plain, m->idx27_plain, uint32_t, 27 * (size_t)m->n_cells);
      ORC_DUP(L->constrained, m->constrained, uint32_t, m->n_constrained);
      ORC_DUP(L->dof_grid, m->dof_gid, uint32_t, m->n_dofs);
      ORC_DUP(L->ent_mult, m->ent_mult, uint8_t, 27 * (size_t)m->n_cells);
      ORC_DUP(L->cell_nodes, m->cell_nodes, double, 3 * (size_t)n3 * m->n_cells);
      if (l > 0)
        ORC_DUP(L->children_of_parent, m->children, uint32_t, 8 * (size_t)mesh[l - 1].n_cells);
#undef ORC_DUP
      /* cell colours for the OpenMP loops: greedy, two cells conflict if they share an entity */
      {
        uint64_t *used   = (uint64_t *)calloc(L->n_dofs, sizeof(uint64_t));
        uint8_t  *colour = (uint8_t *)malloc(L->n_cells);
        uint32_t  cnt[64] = {0};
        int       nco     = 0;
        for (uint32_t c = 0; c < L->n_cells; ++c)
          {
            uint64_t mask = 0;
            for (int e = 0; e < 27; ++e)
              {
                const int inner = (e % 3 == 1) + ((e / 3) % 3 == 1) + (e / 9 == 1);
                if (inner == 3 || (inner > 0 && p == 1))
                  continue;
                mask |= used[L->idx27_plain[27 * (size_t)c + e]];
              }
            int col = 0;
            while (col < 63 && ((mask >> col) & 1u))
              ++col;
            colour[c] = (uint8_t)col;
            cnt[col]++;
            if (col + 1 > nco)
              nco = col + 1;
            for (int e = 0; e < 27; ++e)
              {
                const int inner = (e % 3 == 1) + ((e / 3) % 3 == 1) + (e / 9 == 1);
                if (inner == 3 || (inner > 0 && p == 1))
                  continue;
                used[L->idx27_plain[27 * (size_t)c + e]] |= 1ull << col;
              }
          }
        L->n_colours       = nco;
        L->colour_start[0] = 0;
        for (int i = 0; i < nco; ++i)
          L->colour_start[i + 1] = L->colour_start[i] + cnt[i];
        L->colour_cells = (uint32_t *)malloc(sizeof(uint32_t) * (size_t)L->n_cells);
        uint32_t pos[64];
        for (int i = 0; i < nco; ++i)
          pos[i] = L->colour_start[i];
        for (uint32_t c = 0; c < L->n_cells; ++c)
          L->colour_cells[pos[colour[c]]++] = c;
        free(used);
        free(colour);
      }
      level_geometry(P, L);
    }
  return finish_problem(P);
}


void orc_destroy(orc_problem *P)
{
  if (!P)
    return;
  vlevel *V = VL(P);
  for (int l = 0; l < P->n_levels; ++l)
    {
      level_free(&P->levels[l]);
      free(P->solution[l]);
      free(P->rhs[l]);
      free(P->residual[l]);
      free(P->bc_idx[l]);
      free(P->bc_val[l]);
      free(V[l].defect);
      free(V[l].t);
      free(V[l].solution_update);
      free(V[l].defect_f);
      free(V[l].t_f);
      free(V[l].solution_update_f);
      free(V[l].cheb.inv_diag);
      free(V[l].cheb.x_old);
      free(V[l].cheb.tmp);
      free(V[l].cheb_f_.inv_diag);
      free(V[l].cheb_f_.x_old);
      free(V[l].cheb_f_.tmp);
    }
  free(P->levels);
  free(P->solution);
  free(P->rhs);
  free(P->residual);
  free(P->bc_count);
  free(P->bc_idx);
  free(P->bc_val);
  free(P->vd);
  free(P->Bd);
  free(P->Bf);
  free(P);
}

/* ---- accessors ---- */
int orc_n_levels(const orc_problem *P) { return P->n_levels; }
int orc_degree(const orc_problem *P) { return P->p; }
uint32_t orc_n_cells(const orc_problem *P, int l) { return P->levels[l].n_cells; }
uint32_t orc_n_dofs(const orc_problem *P, int l) { return P->levels[l].n_dofs; }
uint32_t orc_n_constrained(const orc_problem *P, int l) { return P->levels[l].n_constrained; }
int orc_cells_per_dim(const orc_problem *P, int l) { return P->levels[l].N; }
void orc_cells_per_dim3(const orc_problem *P, int l, int out[3])
{
  for (int d = 0; d < 3; ++d)
    out[d] = P->levels[l].Nd[d];
}
const uint32_t *orc_idx27(const orc_problem *P, int l) { return P->levels[l].idx27; }
const uint32_t *orc_idx27_plain(const orc_problem *P, int l) { return P->levels[l].idx27_plain; }
const uint32_t *orc_constrained(const orc_problem *P, int l) { return P->levels[l].constrained; }
const uint32_t *orc_cell_coords(const orc_problem *P, int l) { return P->levels[l].cell_coords; }
const uint32_t *orc_dof_grid(const orc_problem *P, int l) { return P->levels[l].dof_grid; }
const double *orc_shape_values(const orc_problem *P) { return P->basis.S; }
const double *orc_colloc_grad(const orc_problem *P) { return P->basis.D; }
const double *orc_qweights(const orc_problem *P) { return P->basis.gw; }
const double *orc_qpoints(const orc_problem *P) { return P->basis.gq; }
const double *orc_gll(const orc_problem *P) { return P->basis.gll; }
const double *orc_prolong_1d(const orc_problem *P) { return P->basis.P1; }
const double *orc_rhs(const orc_problem *P, int l) { return P->rhs[l]; }
const double *orc_inv_diag(const orc_problem *P, int l) { return VL(P)[l].cheb.inv_diag; }
double orc_h(const orc_problem *P, int l) { return P->levels[l].h; }
const double *orc_solution(orc_problem *P, int l) { return P->solution[l]; }

/* re-creates the smoother of one level with other parameters: MultigridSolverDG configures its FE_Q
 * hierarchy with degree_pre - 1 on the finest level and a coarse tolerance of 2e-3
 * (multigrid_solver_dg.h:271-291); degree < 0: numbers::invalid_unsigned_int */
void orc_reset_smoother(orc_problem *P, int l, double smoothing_range, int degree, int eig_cg_n_iterations)
{
  const orc_level *L = &P->levels[l];
  if (!P->vfloat)
    cheb_setup_d(P, L, BD(P), &VL(P)[l].cheb, smoothing_range, degree, eig_cg_n_iterations);
  else
    {
      cheb_f *F = &VL(P)[l].cheb_f_;
      cheb_d *C = &VL(P)[l].cheb;
      cheb_setup_f(P, L, BF(P), F, smoothing_range, degree, eig_cg_n_iterations);
      C->lambda_min = F->lambda_min;
      C->lambda_max = F->lambda_max;
      C->theta      = F->theta;
      C->delta      = F->delta;
      C->degree     = F->degree;
      C->cg_its     = F->cg_its;
    }
}

/* smoothers of the levels above the coarsest: 0 first_kind (multigrid_solver.h:277-278), 1 fourth_kind
 * (the Number == Number2 specialisation, multigrid_solver.h:951-952) */
void orc_set_polynomial_type(orc_problem *P, int fourth_kind)
{
  for (int l = 1; l < P->n_levels; ++l)
    {
      cheb_d *C = &VL(P)[l].cheb;
      if (P->vfloat)
        {
          cheb_f *F      = &VL(P)[l].cheb_f_;
          F->fourth_kind = fourth_kind;
          F->delta       = fourth_kind ? F->lambda_max : F->lambda_max - F->theta;
          C->delta       = F->delta;
        }
      else
        {
          C->fourth_kind = fourth_kind;
          C->delta       = fourth_kind ? C->lambda_max : C->lambda_max - C->theta;
        }
    }
}

void orc_cheb_info(const orc_problem *P, int l, double *lambda_min, double *lambda_max, double *theta,
                   double *delta, int *degree, int *cg_its)
{
  const cheb_d *C = &VL(P)[l].cheb;
  *lambda_min     = C->lambda_min;
  *lambda_max     = C->lambda_max;
  *theta          = C->theta;
  *delta          = C->delta;
  *degree         = C->degree;
  *cg_its         = C->cg_its;
}

uint32_t orc_bc(const orc_problem *P, int l, uint32_t *idx, double *val)
{
  if (idx)
    memcpy(idx, P->bc_idx[l], sizeof(uint32_t) * P->bc_count[l]);
  if (val)
    memcpy(val, P->bc_val[l], sizeof(double) * P->bc_count[l]);
  return P->bc_count[l];
}

/* ---- operator entry points (fp64 = matrix_dp) ---- */
void orc_vmult(const orc_problem *P, int l, double *dst, const double *src)
{
  vmult_d(P, &P->levels[l], BD(P), dst, src);
}

void orc_vmult_residual(const orc_problem *P, int l, const double *rhs, const double *lhs, double *res)
{
  vmult_residual_d(P, &P->levels[l], BD(P), rhs, lhs, res);
}

/* Independent check: dense element matrix from plain nodal gradients G (no collocation trick, no
 * sum factorisation), applied on the lexicographic grid with explicit Dirichlet identity rows. */
void orc_vmult_dense_lex(const orc_problem *P, int l, double *dst, const double *src)
{
  const orc_level *L = &P->levels[l];
  const orc_basis *b = &P->basis;
  const int        p = P->p, n = p + 1, n3 = n * n * n;
  const size_t     Gx = (size_t)L->Nd[0] * p + 1, Gy = (size_t)L->Nd[1] * p + 1, Gz = (size_t)L->Nd[2] * p + 1;
  double          *A = (double *)calloc((size_t)n3 * n3, sizeof(double));
  /* A_ij = h sum_q w_q grad phi_i . grad phi_j */
  for (int qz = 0; qz < n; ++qz)
    for (int qy = 0; qy < n; ++qy)
      for (int qx = 0; qx < n; ++qx)
        {
          const double w = b->gw[qx] * b->gw[qy] * b->gw[qz] * L->h;
          for (int i = 0; i < n3; ++i)
            {
              const int ix = i % n, iy = (i / n) % n, iz = i / (n * n);
              const double gi[3] = {b->G[qx * n + ix] * b->S[qy * n + iy] * b->S[qz * n + iz],
                                    b->S[qx * n + ix] * b->G[qy * n + iy] * b->S[qz * n + iz],
                                    b->S[qx * n + ix] * b->S[qy * n + iy] * b->G[qz * n + iz]};
              for (int j = 0; j < n3; ++j)
                {
                  const int jx = j % n, jy = (j / n) % n, jz = j / (n * n);
                  const double gj[3] = {b->G[qx * n + jx] * b->S[qy * n + jy] * b->S[qz * n + jz],
                                        b->S[qx * n + jx] * b->G[qy * n + jy] * b->S[qz * n + jz],
                                        b->S[qx * n + jx] * b->S[qy * n + jy] * b->G[qz * n + jz]};
                  A[(size_t)i * n3 + j] += w * (gi[0] * gj[0] + gi[1] * gj[1] + gi[2] * gj[2]);
                }
            }
        }
  memset(dst, 0, sizeof(double) * Gx * Gy * Gz);
  for (int Z = 0; Z < L->Nd[2]; ++Z)
    for (int Y = 0; Y < L->Nd[1]; ++Y)
      for (int X = 0; X < L->Nd[0]; ++X)
        for (int i = 0; i < n3; ++i)
          {
            const size_t gi = (((size_t)Z * p + i / (n * n)) * Gy + ((size_t)Y * p + (i / n) % n)) * Gx +
                              (size_t)X * p + i % n;
            const size_t ix = gi % Gx, iy = (gi / Gx) % Gy, iz = gi / (Gx * Gy);
            if (ix == 0 || ix == Gx - 1 || iy == 0 || iy == Gy - 1 || iz == 0 || iz == Gz - 1)
              continue;
            double s = 0;
            for (int j = 0; j < n3; ++j)
              {
                const size_t jx = (size_t)X * p + j % n, jy = (size_t)Y * p + (j / n) % n,
                             jz = (size_t)Z * p + j / (n * n);
                if (jx == 0 || jx == Gx - 1 || jy == 0 || jy == Gy - 1 || jz == 0 || jz == Gz - 1)
                  continue;
                s += A[(size_t)i * n3 + j] * src[(jz * Gy + jy) * Gx + jx];
              }
            dst[gi] += s;
          }
  for (size_t iz = 0; iz < Gz; ++iz)
    for (size_t iy = 0; iy < Gy; ++iy)
      for (size_t ix = 0; ix < Gx; ++ix)
        if (ix == 0 || ix == Gx - 1 || iy == 0 || iy == Gy - 1 || iz == 0 || iz == Gz - 1)
          dst[(iz * Gy + iy) * Gx + ix] = src[(iz * Gy + iy) * Gx + ix];
  free(A);
}

/* ---- helpers converting the double interface to the V-cycle number type ---- */
static float *to_float(uint32_t n, const double *a)
{
  float *f = (float *)malloc(sizeof(float) * n);
  for (uint32_t i = 0; i < n; ++i)
    f[i] = (float)a[i];
  return f;
}
static void from_float(uint32_t n, double *a, const float *f)
{
  for (uint32_t i = 0; i < n; ++i)
    a[i] = f[i];
}

void orc_cheb_vmult(orc_problem *P, int l, double *x, const double *b)
{
  const orc_level *L = &P->levels[l];
  if (!P->vfloat)
    cheb_vmult_d(P, L, BD(P), &VL(P)[l].cheb, x, b);
  else
    {
      float *bf = to_float(L->n_dofs, b), *xf = (float *)calloc(L->n_dofs, sizeof(float));
      cheb_vmult_f(P, L, BF(P), &VL(P)[l].cheb_f_, xf, bf);
      from_float(L->n_dofs, x, xf);
      free(bf);
      free(xf);
    }
}

void orc_cheb_step(orc_problem *P, int l, double *x, const double *b)
{
  const orc_level *L = &P->levels[l];
  if (!P->vfloat)
    cheb_step_d(P, L, BD(P), &VL(P)[l].cheb, x, b);
  else
    {
      float *bf = to_float(L->n_dofs, b), *xf = to_float(L->n_dofs, x);
      cheb_step_f(P, L, BF(P), &VL(P)[l].cheb_f_, xf, bf);
      from_float(L->n_dofs, x, xf);
      free(bf);
      free(xf);
    }
}

void orc_prolongate(const orc_problem *P, int l, double *fine, const double *coarse, int add, int with_bc)
{
  prolongate_d(P, &P->levels[l - 1], &P->levels[l], BD(P), fine, coarse, add, with_bc);
}

void orc_restrict_and_add(const orc_problem *P, int l, double *coarse, const double *fine, int with_bc)
{
  restrict_and_add_d(P, &P->levels[l - 1], &P->levels[l], BD(P), coarse, fine, with_bc);
}

/* MultigridSolver::v_cycle (multigrid_solver.h:641-681) */
static void v_cycle(orc_problem *P, int level, int my_n_cycles)
{
  vlevel          *V = VL(P);
  const orc_level *L = &P->levels[level];
  if (level == 0) /* :644-651  coarse = smooth[0].vmult (:72-91) */
    {
      if (!P->vfloat)
        cheb_vmult_d(P, L, BD(P), &V[0].cheb, V[0].solution_update, V[0].defect);
      else
        cheb_vmult_f(P, L, BF(P), &V[0].cheb_f_, V[0].solution_update_f, V[0].defect_f);
      return;
    }
  const orc_level *Lc = &P->levels[level - 1];
  for (int c = 0; c < my_n_cycles; ++c)
    {
      if (!P->vfloat)
        {
          if (c == 0) /* :656-659 */
            cheb_vmult_d(P, L, BD(P), &V[level].cheb, V[level].solution_update, V[level].defect);
          else
            cheb_step_d(P, L, BD(P), &V[level].cheb, V[level].solution_update, V[level].defect);
          vmult_residual_d(P, L, BD(P), V[level].defect, V[level].solution_update, V[level].t); /* :663 */
          memset(V[level - 1].defect, 0, sizeof(double) * Lc->n_dofs);                           /* :667 */
          restrict_and_add_d(P, Lc, L, BD(P), V[level - 1].defect, V[level].t, 1);               /* :668 */
          v_cycle(P, level - 1, 1);                                                              /* :671 */
          prolongate_d(P, Lc, L, BD(P), V[level].solution_update, V[level - 1].solution_update, 1, 1); /* :674 */
          cheb_step_d(P, L, BD(P), &V[level].cheb, V[level].solution_update, V[level].defect);   /* :678 */
        }
      else
        {
          if (c == 0)
            cheb_vmult_f(P, L, BF(P), &V[level].cheb_f_, V[level].solution_update_f, V[level].defect_f);
          else
            cheb_step_f(P, L, BF(P), &V[level].cheb_f_, V[level].solution_update_f, V[level].defect_f);
          vmult_residual_f(P, L, BF(P), V[level].defect_f, V[level].solution_update_f, V[level].t_f);
          memset(V[level - 1].defect_f, 0, sizeof(float) * Lc->n_dofs);
          restrict_and_add_f(P, Lc, L, BF(P), V[level - 1].defect_f, V[level].t_f, 1);
          v_cycle(P, level - 1, 1);
          prolongate_f(P, Lc, L, BF(P), V[level].solution_update_f, V[level - 1].solution_update_f, 1, 1);
          cheb_step_f(P, L, BF(P), &V[level].cheb_f_, V[level].solution_update_f, V[level].defect_f);
        }
    }
}

/* defect[level] = v (precision cast, multigrid_solver.h:437, 503) */
static void set_defect(orc_problem *P, int level, const double *v)
{
  vlevel        *V = VL(P);
  const uint32_t n = P->levels[level].n_dofs;
  if (!P->vfloat)
    memcpy(V[level].defect, v, sizeof(double) * n);
  else
    for (uint32_t i = 0; i < n; ++i)
      V[level].defect_f[i] = (float)v[i];
}

void orc_vcycle_apply(orc_problem *P, double *dst, const double *src)
{
  const int      lmax = P->n_levels - 1;
  vlevel        *V    = VL(P);
  const uint32_t n    = P->levels[lmax].n_dofs;
  set_defect(P, lmax, src); /* :503 */
  v_cycle(P, lmax, 1);      /* :505 */
  if (!P->vfloat)           /* :507 */
    memcpy(dst, V[lmax].solution_update, sizeof(double) * n);
  else
    for (uint32_t i = 0; i < n; ++i)
      dst[i] = V[lmax].solution_update_f[i];
}

static double l2_norm(uint32_t n, const double *a) { return sqrt(dot_d(n, a, a)); }

double orc_l2_error(orc_problem *P, int level)
{
  const orc_level *L = &P->levels[level];
  const int        p = P->p, n = p + 1, n3 = n * n * n;
  const basis_d   *B = BD(P);
  set_bc(P, level, P->solution[level], 0); /* :301-302 */
  double err = 0, vol = 0;
#pragma omp parallel reduction(+ : err, vol)
  {
    double *u = (double *)malloc(sizeof(double) * 2 * n3), *t0 = u + n3;
#pragma omp for schedule(static)
    for (uint32_t c = 0; c < L->n_cells; ++c)
      {
        gather27_d(p, L->idx27_plain + 27 * (size_t)c, P->solution[level], u); /* read_dof_values_plain */
        sweep_d(n, 0, B->S, u, t0, 0);
        sweep_d(n, 1, B->S, t0, u, 0);
        sweep_d(n, 2, B->S, u, t0, 0);
        const double x0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c] : 0.,
                     y0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c + 1] : 0.,
                     z0 = L->cell_coords ? P->origin + L->h * L->cell_coords[3 * (size_t)c + 2] : 0.;
        for (int k = 0, q = 0; k < n; ++k)
          for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i, ++q)
              {
                double JxW = B->w[i] * B->w[j] * B->w[k] * L->h * L->h * L->h, d;
                if (L->coef_q) /* mapped mesh: JxW and x_q from the mapping */
                  {
                    const double *xq = L->xq + ((size_t)c * n3 + q) * 3;
                    JxW              = L->jxw[(size_t)c * n3 + q];
                    d                = t0[q] - u_exact(xq[0], xq[1], xq[2]);
                  }
                else
                  d = t0[q] - u_exact(x0 + L->h * P->basis.gq[i], y0 + L->h * P->basis.gq[j], z0 + L->h * P->basis.gq[k]);
                err += d * d * JxW;
                vol += JxW;
              }
      }
    free(u);
  }
  return sqrt(err / vol);
}

/* MultigridSolver::solve (multigrid_solver.h:387-476) */
double orc_solve(orc_problem *P, int do_analyze, double *trace)
{
  vlevel *V              = VL(P);
  double  reduction_rate = 1.;
  const orc_level *L0    = &P->levels[0];
  /* coarse solver invoked twice (:397-400) */
  set_defect(P, 0, P->rhs[0]);
  if (!P->vfloat)
    {
      cheb_vmult_d(P, L0, BD(P), &V[0].cheb, V[0].t, V[0].defect);
      cheb_step_d(P, L0, BD(P), &V[0].cheb, V[0].t, V[0].defect);
      memcpy(P->solution[0], V[0].t, sizeof(double) * L0->n_dofs);
    }
  else
    {
      cheb_vmult_f(P, L0, BF(P), &V[0].cheb_f_, V[0].t_f, V[0].defect_f);
      cheb_step_f(P, L0, BF(P), &V[0].cheb_f_, V[0].t_f, V[0].defect_f);
      from_float(L0->n_dofs, P->solution[0], V[0].t_f);
    }
  for (int level = 1; level < P->n_levels; ++level)
    {
      const orc_level *L = &P->levels[level];
      set_bc(P, level - 1, P->solution[level - 1], 0);                                    /* :408-409 */
      prolongate_d(P, &P->levels[level - 1], L, BD(P), P->solution[level], P->solution[level - 1], 0,
                   0);                                                                    /* :415 */
      double init_residual = 1.;
      if (do_analyze && trace)
        trace[4 * level + 0] = orc_l2_error(P, level);                                    /* :422 */
      set_bc(P, level, P->solution[level], 1);                                            /* :427-428 */
      vmult_residual_d(P, L, BD(P), P->rhs[level], P->solution[level], P->residual[level]); /* :432 */
      set_defect(P, level, P->residual[level]);                                           /* :437 */
      if (do_analyze)
        {
          init_residual = l2_norm(L->n_dofs, P->residual[level]);                         /* :444 */
          if (trace)
            trace[4 * level + 1] = init_residual;
        }
      v_cycle(P, level, P->n_cycles);                                                     /* :451 */
      if (!P->vfloat)                                                                     /* :456 */
        for (uint32_t i = 0; i < L->n_dofs; ++i)
          P->solution[level][i] += V[level].solution_update[i];
      else
        for (uint32_t i = 0; i < L->n_dofs; ++i)
          P->solution[level][i] += V[level].solution_update_f[i];
      if (do_analyze)
        {
          set_bc(P, level, P->solution[level], 1);                                        /* :462-463 */
          vmult_d(P, L, BD(P), P->residual[level], P->solution[level]);                   /* :464 */
          for (uint32_t i = 0; i < L->n_dofs; ++i)                                        /* :465 */
            P->residual[level][i] = P->rhs[level][i] - P->residual[level][i];
          const double res_norm = l2_norm(L->n_dofs, P->residual[level]);                 /* :466 */
          reduction_rate        = pow(res_norm / init_residual, 1. / P->n_cycles);        /* :467 */
          if (trace)
            {
              trace[4 * level + 2] = res_norm;
              trace[4 * level + 3] = orc_l2_error(P, level);                              /* :470 */
            }
        }
    }
  return reduction_rate;
}

/* MultigridSolver::solve_cg (multigrid_solver.h:483-493) with deal.II's SolverCG restated
 * (SURVEY 8a row T): ReductionControl(1000, 1e-16, 1e-9), zero start, V-cycle preconditioner */
/* LaplaceOperator::vmult_with_cg_update (laplace_operator.h:638-719): the vector updates of the
 * before-loop hook on every range (:655-688), the cell loop q = A p (:650-654; constrained rows are
 * not touched and keep the zero the hook wrote), the four sums of the after-loop hook (:689-713) */
void orc_vmult_with_cg_update(const orc_problem *P, int l, double alpha, double beta, const double *r, double *q, double *p,
                              double *x, double *sums)
{
  const orc_level *L = &P->levels[l];
  const uint32_t   n = L->n_dofs;
  for (uint32_t i = 0; i < n; ++i)
    {
      if (alpha == 0.)
        p[i] = q[i];
      else
        {
          x[i] += alpha * p[i];
          p[i] = beta * p[i] + q[i];
        }
      q[i] = 0.;
    }
  vmult_d(P, L, BD(P), q, p);
  for (uint32_t i = 0; i < L->n_constrained; ++i)
    q[L->constrained[i]] = 0.;
  sums[0] = sums[1] = sums[2] = sums[3] = 0.;
  for (uint32_t i = 0; i < n; ++i)
    {
      sums[0] += q[i] * p[i];
      sums[1] += r[i] * r[i];
      sums[2] += q[i] * r[i];
      sums[3] += q[i] * q[i];
    }
}

/* MultigridSolver::vmult_with_residual_update (multigrid_solver.h:516-619).  Constrained rows:
 * identity on the diagonal (:570-577, 598-603); out[2] = residual . residual after the update
 * (not part of the reference's return value; used by the tests of the fused PCG) */
void orc_vmult_with_residual_update(orc_problem *P, double *residual, double *update, double factor, double *out)
{
  const int        lmax = P->n_levels - 1;
  const orc_level *L    = &P->levels[lmax];
  const uint32_t   n    = L->n_dofs;
  unsigned char   *cons = (unsigned char *)calloc(n, 1);
  double          *def = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n);
  for (uint32_t i = 0; i < L->n_constrained; ++i)
    cons[L->constrained[i]] = 1;
  for (uint32_t i = 0; i < n; ++i) /* :527-534 */
    def[i] = factor != 0. ? residual[i] + factor * update[i] : residual[i];
  orc_vcycle_apply(P, z, def); /* :538 */
  out[0] = out[1] = out[2] = 0.;
  for (uint32_t i = 0; i < n; ++i) /* :545-603 */
    {
      const double upd = factor != 0. ? update[i] * factor : 0.;
      const double res = residual[i] + upd;
      const double zi  = cons[i] ? res : z[i];
      residual[i]      = res;
      update[i]        = zi;
      out[0] += zi * res;
      out[1] += factor != 0. ? zi * upd : zi * res;
      out[2] += res * res;
    }
  free(cons);
  free(def);
  free(z);
}

int orc_solve_cg(orc_problem *P, double *reduction)
{
  const int        lmax = P->n_levels - 1;
  const orc_level *L    = &P->levels[lmax];
  const uint32_t   n    = L->n_dofs;
  double          *x    = P->solution[lmax];
  double *r = (double *)malloc(sizeof(double) * n), *z = (double *)malloc(sizeof(double) * n),
         *d = (double *)malloc(sizeof(double) * n), *h = (double *)malloc(sizeof(double) * n);
  memset(x, 0, sizeof(double) * n); /* :488 */
  memcpy(r, P->rhs[lmax], sizeof(double) * n);
  const double res0 = l2_norm(n, r);
  double       res  = res0, rz = 0, rz_old;
  int          it   = 0;
  P->cg_history[0] = res0;
  P->cg_history_n  = 1;
  while (res > 1e-16 && res > 1e-9 * res0 && it < 1000)
    {
      ++it;
      orc_vcycle_apply(P, z, r);
      rz_old = rz;
      rz     = dot_d(n, r, z);
      if (it > 1)
        {
          const double beta = rz / rz_old;
          for (uint32_t i = 0; i < n; ++i)
            d[i] = z[i] + beta * d[i];
        }
      else
        memcpy(d, z, sizeof(double) * n);
      vmult_d(P, L, BD(P), h, d);
      const double alpha = rz / dot_d(n, d, h);
      for (uint32_t i = 0; i < n; ++i)
        {
          x[i] += alpha * d[i];
          r[i] -= alpha * h[i];
        }
      res = l2_norm(n, r);
      P->cg_history[P->cg_history_n++] = res;
    }
  if (reduction)
    *reduction = it > 0 ? pow(res / res0, 1. / it) : 1.; /* :491-492 */
  free(r);
  free(z);
  free(d);
  free(h);
  return it;
}

/* residual norms of the last orc_solve_cg (start + one per iteration); returns their number */
int orc_cg_history(const orc_problem *P, double *out, int capacity)
{
  for (int i = 0; i < P->cg_history_n && i < capacity; ++i)
    out[i] = P->cg_history[i];
  return P->cg_history_n;
}

/* ---- timing helpers for the cpu_baseline leg of bench.py ---- */
static double now_s(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}

double orc_time_vmult(orc_problem *P, int level, int n)
{
  /* MultigridSolver::do_matvec (multigrid_solver.h:624-628) */
  const double t0 = now_s();
  for (int i = 0; i < n; ++i)
    vmult_d(P, &P->levels[level], BD(P), P->residual[level], P->solution[level]);
  return now_s() - t0;
}

double orc_time_vcycle(orc_problem *P, int n)
{
  const int lmax = P->n_levels - 1;
  double   *dst  = (double *)malloc(sizeof(double) * P->levels[lmax].n_dofs);
  const double t0 = now_s();
  for (int i = 0; i < n; ++i)
    orc_vcycle_apply(P, dst, P->rhs[lmax]);
  const double t = now_s() - t0;
  free(dst);
  return t;
}

/* threads of the following calls (bench.py's one-core sample) */
void orc_set_num_threads(int n)
{
#ifdef _OPENMP
  omp_set_num_threads(n > 0 ? n : effective_threads());
#else
  (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}
